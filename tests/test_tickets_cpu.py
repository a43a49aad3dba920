"""The persistent walk kernels' ticket hand-out (struct Tickets in raytracer_amd/csrc/rt_kernels.h), restated:
whatever the queue length and the number of wavefronts, the slots map to every ticket exactly once, grouped
tickets lie `groups` apart, the last part of the queue is handed out singly, and the counters that deal out the
slots (counter k: slots k, k + COUNTERS, ...) hand out every slot exactly once in whatever order wavefronts ask them."""
import numpy as np
import pytest

GRAB_DIV, GRAB_MAX, GRAB_TAIL, COUNTERS = 4, 2, 2, 8  # the defaults of rt_kernels.h


def plan(total, waves):
    tickets = (total + 63) // 64
    share = tickets // (waves * GRAB_DIV)
    per_grab = min(max(share, 1), GRAB_MAX)
    groups = tickets * (16 - GRAB_TAIL) // 16 // per_grab if per_grab > 1 else 0
    singles = tickets - groups * per_grab
    return tickets, per_grab, groups, singles


def tickets_of_slot(slot, per_grab, groups):
    if slot < groups:
        return [slot + k * groups for k in range(per_grab)]
    return [groups * per_grab + (slot - groups)]


@pytest.mark.parametrize("total", [1, 63, 64, 65, 4096, 8294400, 32394792, 1036800 * 6 + 17, 2 ** 31 - 5])
@pytest.mark.parametrize("waves", [4, 1024, 6144, 8192])
def test_every_ticket_is_handed_out_exactly_once(total, waves):
    tickets, per_grab, groups, singles = plan(total, waves)
    assert 1 <= per_grab <= GRAB_MAX and groups * per_grab + singles == tickets
    if tickets > 3_000_000:  # the big queues: check the arithmetic, not every ticket
        slots = np.concatenate([np.arange(0, min(groups, 1000)), np.arange(max(groups - 1000, 0), groups)])
        t = slots[:, None] + np.arange(per_grab)[None, :] * groups
        assert t.max() < groups * per_grab and len(np.unique(t)) == t.size
        assert groups * per_grab + singles - 1 == tickets - 1 and 64 * (tickets - 1) < total <= 64 * tickets
        return
    seen = np.zeros(tickets, np.int32)
    for slot in range(groups + singles):
        for t in tickets_of_slot(slot, per_grab, groups):
            seen[t] += 1
    assert (seen == 1).all()
    assert 64 * (tickets - 1) < total <= 64 * tickets  # the last ticket is the only partial one


def test_small_queues_are_not_grouped_and_large_ones_keep_a_single_ticket_tail():
    assert plan(16200 * 64, 6144)[1] == 1                      # one rank of eight: two or three tickets per wavefront
    tickets, per_grab, groups, singles = plan(32394792, 6144)  # temple 4K shadow queue
    assert per_grab == GRAB_MAX and singles >= tickets // 8 and groups * per_grab >= tickets * 3 // 4


@pytest.mark.parametrize("slots", [0, 1, 7, 8, 9, 1000, 12345])
@pytest.mark.parametrize("waves", [1, 5, 64])
def test_the_counters_hand_out_every_slot_once(slots, waves):
    """Tickets::take: a wavefront asks the counter of its block, takes global slot s * COUNTERS + k for the value s it
    gets from counter k, and moves on to the next counter (for good) when that slot does not exist."""
    rng = np.random.default_rng(slots * 131 + waves)
    counters = [0] * COUNTERS
    tried = [0] * waves
    home = [int(rng.integers(0, 1 << 20)) for _ in range(waves)]  # blockIdx.x
    seen = np.zeros(slots, np.int32)
    live = list(range(waves))
    asked = 0
    while live:
        w = live[int(rng.integers(0, len(live)))]  # any interleaving of the wavefronts
        k = (home[w] + tried[w]) % COUNTERS
        s = counters[k]
        counters[k] += 1
        asked += 1
        g = s * COUNTERS + k
        if g < slots:
            seen[g] += 1
        else:
            tried[w] += 1
            if tried[w] >= COUNTERS:
                live.remove(w)
    assert (seen == 1).all()
    assert asked == slots + waves * COUNTERS  # every wavefront finds every counter empty exactly once
