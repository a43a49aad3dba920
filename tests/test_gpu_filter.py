"""The walk kernels' conservative filter and batch culling change nothing but the time.

Option "filter" = 0 makes every listed triangle of every visited cell take the exact Moeller-Trumbore
test (src/intersection.c:58-86), as the reference does; the default path rejects most of them earlier
(bounding sphere of a batch, three Pluecker slabs, distance bounds).  Both must give the same frame,
bit for bit: ids, raw float colour and min/max.
"""
import os

import numpy as np
import pytest

import raytracer_amd as rt

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [
    ("scenes/temple/temple", {}, 640, 360, False),
    ("scenes/temple/temple", {}, 333, 187, False),
    ("scenes/ulica/ulica", {}, 160, 120, False),        # reflections and refractions: incoherent secondary rays
    ("scenes/pokoj/s2", {}, 200, 150, True),            # textured, transparent surfaces
    ("scenes/s2/s2", {}, 160, 120, False),              # planar lights
    ("scenes/s2/s2", {"atr": "scenes/pokoj/s2.atr"}, 120, 90, False),   # planar light and secondary rays
    ("scenes/s3/s3", {}, 320, 240, False),
    ("scenes/s5/s5", {}, 160, 120, False),
    ("scenes/test/s2", {"atr": "scenes/temple/temple.atr"}, 128, 96, False),
]


@pytest.fixture(scope="module")
def renderer():
    os.chdir(ROOT)
    r = rt.Renderer(0)
    yield r
    r.set_option("filter", 1)
    r.close()


def frame(renderer, scene, flt, count=False):
    renderer.set_option("filter", flt)
    renderer.set_option("count", 1 if count else 0)
    try:
        renderer.upload(scene)
        renderer.frame_begin()
        out = renderer.render_host()
        cnt = renderer.counters().asdict() if count else None
    finally:
        renderer.set_option("filter", 1)
        renderer.set_option("count", 0)
    return out, cnt


@pytest.mark.parametrize("prefix,over,w,h,tex", CASES, ids=[c[0].split("/")[1] + ("_glass" if c[1] and "pokoj" in str(c[1]) else "") + "_%dx%d" % (c[2], c[3]) for c in CASES])
def test_filter_on_and_off_give_the_same_frame(renderer, prefix, over, w, h, tex):
    over = {k: os.path.join(ROOT, v) for k, v in over.items()}
    scene = rt.Scene(os.path.join(ROOT, prefix), texture=tex, width=w, height=h, **over)
    on, _ = frame(renderer, scene, 1)
    off, _ = frame(renderer, scene, 0)
    assert np.array_equal(on["ids"], off["ids"])
    assert np.array_equal(on["rgb"].view(np.uint32), off["rgb"].view(np.uint32))
    assert np.array_equal(on["minmax"].view(np.uint32), off["minmax"].view(np.uint32))


@pytest.mark.parametrize("prefix,over,w,h,tex", CASES, ids=[c[0].split("/")[1] + ("_glass" if c[1] and "pokoj" in str(c[1]) else "") + "_%dx%d" % (c[2], c[3]) for c in CASES])
def test_record_level_bundle_test_does_not_change_the_frame(renderer, prefix, over, w, h, tex):
    """Option "record_test": the walk kernels drop, before a batch is staged, the records no ray of the wavefront's bundle
    can pass (1), or test every record of a kept batch (0); -1 chooses by the scene; option "bundle_ends": the shadow walk's
    culling uses that its rays end at their lights.  Same frame, bit for bit, and the same algorithmic counters."""
    over = {k: os.path.join(ROOT, v) for k, v in over.items()}
    scene = rt.Scene(os.path.join(ROOT, prefix), texture=tex, width=w, height=h, **over)
    frames = []
    try:
        for mode, ends in ((1, 0), (0, 0), (-1, -1), (1, 1), (0, 1)):
            renderer.set_option("record_test", mode)
            renderer.set_option("bundle_ends", ends)  # the shadow walk's bundles narrow towards their lights, or are not told so
            frames.append(frame(renderer, scene, 1, count=True))
    finally:
        renderer.set_option("record_test", 1)
        renderer.set_option("bundle_ends", -1)
    (a, ca) = frames[0]
    for b, cb in frames[1:]:
        assert np.array_equal(a["ids"], b["ids"])
        assert np.array_equal(a["rgb"].view(np.uint32), b["rgb"].view(np.uint32))
        assert np.array_equal(a["minmax"].view(np.uint32), b["minmax"].view(np.uint32))
        for k in ("nearest_queries", "shadow_queries", "nearest_steps", "nearest_tests", "shadow_steps", "shadow_tests"):
            assert ca[k] == cb[k], k
    # with the test fewer records reach the ray-by-ray first slab (compared where both walks test the boxes ray by ray
    # first: with bundle_ends; without it the walk that has the record-level test leaves the ray-by-ray box test out)
    # (not with a planar light: its samples make every bundle as wide as the light, the slab test of a wide bundle lets most
    # records through, and short lists skip the box tests)
    if scene.npl == 0:
        assert frames[3][1]["shadow_lane_slots"] <= frames[4][1]["shadow_lane_slots"]


@pytest.mark.parametrize("prefix,over,w,h,tex", CASES, ids=[c[0].split("/")[1] + ("_glass" if c[1] and "pokoj" in str(c[1]) else "") + "_%dx%d" % (c[2], c[3]) for c in CASES])
def test_hits_shaded_by_the_walk_itself_give_the_same_frame(renderer, prefix, over, w, h, tex):
    """Option "fuse_hit": a level that neither spawns rays nor meets a textured triangle shades its hits at the end of the
    nearest-hit walk (1, the default) or in k_hit like every other level (0).  Same frame, bit for bit, same counters."""
    over = {k: os.path.join(ROOT, v) for k, v in over.items()}
    scene = rt.Scene(os.path.join(ROOT, prefix), texture=tex, width=w, height=h, **over)
    try:
        renderer.set_option("fuse_hit", 1)
        a, ca = frame(renderer, scene, 1, count=True)
        renderer.set_option("fuse_hit", 0)
        b, cb = frame(renderer, scene, 1, count=True)
    finally:
        renderer.set_option("fuse_hit", 1)
    assert np.array_equal(a["ids"], b["ids"])
    assert np.array_equal(a["rgb"].view(np.uint32), b["rgb"].view(np.uint32))
    assert np.array_equal(a["minmax"].view(np.uint32), b["minmax"].view(np.uint32))
    for k in ("nearest_queries", "shadow_queries", "nearest_steps", "nearest_tests", "shadow_steps", "shadow_tests"):
        assert ca[k] == cb[k], k


def test_filter_counters(renderer):
    scene = rt.Scene(os.path.join(ROOT, "scenes/temple/temple"), width=320, height=180)
    _, on = frame(renderer, scene, 1, count=True)
    _, off = frame(renderer, scene, 0, count=True)
    # the queries and the algorithmic S, T do not depend on the filter
    for k in ("nearest_queries", "shadow_queries", "nearest_steps", "nearest_tests"):
        assert on[k] == off[k], k
    for f in ("nearest", "shadow"):
        # off: nothing is culled and every executed test ends in the exact one
        assert off[f + "_batches_culled"] == 0
        assert off[f + "_exact"] > 0 and off[f + "_exact"] >= off[f + "_stage3"] >= on[f + "_stage3"]
        # on: batches are culled, few tests reach the exact one, and fewer tests are executed at all
        assert 0 < on[f + "_batches_culled"] < on[f + "_batches"]
        assert on[f + "_exact"] * 10 < off[f + "_exact"]
        assert on[f + "_lane_slots"] < off[f + "_lane_slots"]
        assert on[f + "_stage2"] >= on[f + "_stage3"] >= on[f + "_exact"]


@pytest.mark.parametrize("prefix,over,w,h,tex", [c for c in CASES if "ulica" in c[0] or "pokoj" in str(c) or "s2/s2" in c[0] or c[2] == 333],
                         ids=lambda v: None)
def test_order_of_shading_does_not_change_the_frame(renderer, prefix, over, w, h, tex):
    """Option "order": the shaded hits of a recursion level get their shadow walks and colours in queue order (0), by
    the grid cell of the hit point below the first level (1, the default) or on every level (2).  A node's walks and
    colour are its own: same frame bit for bit, same counters."""
    over = {k: os.path.join(ROOT, v) for k, v in over.items()}
    scene = rt.Scene(os.path.join(ROOT, prefix), texture=tex, width=w, height=h, **over)
    frames = []
    try:
        for mode in (0, 1, 2):
            renderer.set_option("order", mode)
            frames.append(frame(renderer, scene, 1, count=True))
    finally:
        renderer.set_option("order", 1)
    (a, ca) = frames[0]
    assert (a["ids"] >= 0).any()
    for b, cb in frames[1:]:
        assert np.array_equal(a["ids"], b["ids"])
        assert np.array_equal(a["rgb"].view(np.uint32), b["rgb"].view(np.uint32))
        assert np.array_equal(a["minmax"].view(np.uint32), b["minmax"].view(np.uint32))
        for k in ("nearest_queries", "shadow_queries", "nearest_steps", "nearest_tests", "shadow_steps", "shadow_tests"):
            assert ca[k] == cb[k], k
