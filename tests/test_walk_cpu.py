"""The walk kernels' compact DDA state (struct Walk in raytracer_amd/csrc/rt_math.h: cell index + steps left per
axis, optionally cut down to the box of the non-empty cells) against the reference's form (voxelize.c:56-121,
594-611, 727-735: i, j, k, step directions, two-sided range test), both restated in float32 numpy: the same
sequence of cells, the same accumulated t values, the same last cell."""
import numpy as np
import pytest

F = np.float32
FLT_MAX = np.finfo(np.float32).max


def axis_init(lo, cs, idx, o, r):
    p1 = F(lo + F(idx) * cs)
    p2 = F(p1 + cs)
    if r == 0:
        return FLT_MAX, F(0), -1
    t1, t2 = F(F(p1 - o) / r), F(F(p2 - o) / r)
    return F(abs(F(t2 - t1))), (t1 if t1 < t2 else t2), (1 if r > 0 else -1)


def reference_walk(dmin, cs, nv, o, r, ijk, lo, hi, limit=10000):
    """Cells visited by the reference's loop: the start voxel, then steps until an index leaves [lo, hi]."""
    t, dt, step = [None] * 3, [None] * 3, [None] * 3
    for a in range(3):
        dt[a], t[a], step[a] = axis_init(dmin[a], cs[a], ijk[a], o[a], r[a])
    p = list(ijk)
    cells = [(p[0] * nv[1] + p[1]) * nv[2] + p[2]]
    ts = [tuple(t)]
    for _ in range(limit):
        n = [F(t[a] + dt[a]) for a in range(3)]
        if n[0] < n[1]:
            a = 0 if n[0] < n[2] else 2
        else:
            a = 1 if n[1] < n[2] else 2
        p[a] += step[a]; t[a] = n[a]
        if any(p[q] < lo[q] or p[q] > hi[q] for q in range(3)):
            break
        cells.append((p[0] * nv[1] + p[1]) * nv[2] + p[2]); ts.append(tuple(t))
    return cells, ts


def compact_walk(dmin, cs, nv, o, r, ijk, lo, hi, occ_box=None, limit=10000):
    """walk_init + walk_step of rt_math.h."""
    lo, hi = list(lo), list(hi)
    if occ_box is not None:
        for q in range(3):
            lo[q] = max(lo[q], occ_box[0][q]); hi[q] = min(hi[q], occ_box[1][q])
    t, dt, left = [None] * 3, [None] * 3, [None] * 3
    for a in range(3):
        dt[a], t[a], d = axis_init(dmin[a], cs[a], ijk[a], o[a], r[a])
        left[a] = hi[a] - ijk[a] if d > 0 else ijk[a] - lo[a]
    stride = [nv[1] * nv[2], nv[2], 1]
    cell = (ijk[0] * nv[1] + ijk[1]) * nv[2] + ijk[2]
    cells, ts = [cell], [tuple(t)]
    for _ in range(limit):
        n = [F(t[a] + dt[a]) for a in range(3)]
        xy = n[0] < n[1]
        sx, sy = xy and n[0] < n[2], (not xy) and n[1] < n[2]
        a = 0 if sx else 1 if sy else 2
        t[a] = n[a]; left[a] -= 1
        cell += stride[a] if r[a] > 0 else -stride[a]
        if (left[0] | left[1] | left[2]) < 0:
            break
        cells.append(cell); ts.append(tuple(t))
    return cells, ts


def random_case(rng, zero_components):
    nv = [int(x) for x in rng.integers(1, 12, 3)]
    dmin = (rng.standard_normal(3) * 50).astype(F)
    cs = (rng.random(3) * 10 + 0.5).astype(F)
    ijk = [int(rng.integers(0, nv[q])) for q in range(3)]
    o = np.array([dmin[q] + (ijk[q] + rng.random()) * cs[q] for q in range(3)], F)
    r = rng.standard_normal(3)
    for q in range(3):
        if rng.random() < zero_components:
            r[q] = 0.0
    if not r.any():
        r[int(rng.integers(0, 3))] = 1.0
    r = (r / np.linalg.norm(r)).astype(F)
    return nv, dmin, cs, ijk, o, r


@pytest.mark.parametrize("zero_components", [0.0, 0.3])
def test_compact_walk_visits_the_reference_cells(zero_components):
    rng = np.random.default_rng(3)
    for _ in range(1500):
        nv, dmin, cs, ijk, o, r = random_case(rng, zero_components)
        # nearest-hit walk: the whole grid; shadow walk: the box spanned by the start voxel and another voxel
        other = [int(rng.integers(0, nv[q])) for q in range(3)]
        for lo, hi in (([0, 0, 0], [nv[q] - 1 for q in range(3)]),
                       ([min(ijk[q], other[q]) for q in range(3)], [max(ijk[q], other[q]) for q in range(3)])):
            want = reference_walk(dmin, cs, nv, o, r, ijk, lo, hi)
            got = compact_walk(dmin, cs, nv, o, r, ijk, lo, hi)
            assert got[0] == want[0]
            assert np.array_equal(np.array(got[1], F).view(np.uint32), np.array(want[1], F).view(np.uint32))


def test_clamping_to_the_occupied_box_loses_no_occupied_cell():
    rng = np.random.default_rng(4)
    for _ in range(1500):
        nv, dmin, cs, ijk, o, r = random_case(rng, 0.2)
        a = [int(rng.integers(0, nv[q])) for q in range(3)]
        b = [int(rng.integers(a[q], nv[q])) for q in range(3)]
        occupied = lambda c: all(a[q] <= x <= b[q] for q, x in enumerate((c // (nv[1] * nv[2]), (c // nv[2]) % nv[1], c % nv[2])))
        lo, hi = [0, 0, 0], [nv[q] - 1 for q in range(3)]
        want = [c for c in reference_walk(dmin, cs, nv, o, r, ijk, lo, hi)[0] if occupied(c)]
        got = [c for c in compact_walk(dmin, cs, nv, o, r, ijk, lo, hi, (a, b))[0] if occupied(c)]
        assert got == want
