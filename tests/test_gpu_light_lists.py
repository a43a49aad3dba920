"""Shadow queries answered from the lights' lists (option "light_lists", scenes without transparent triangles) give the
frame the cell walks give, bit for bit: the reference's shadow walk (src/voxelize.c:615-738) reports a blocker iff some
triangle listed in a cell the ray passes blocks it, and the lists hold every triangle that can block
(tests/test_light_lists_cpu.py) while a blocker found in them counts only if the ray's own walk passes one of its cells."""
import os

import numpy as np
import pytest

import raytracer_amd as rt

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [
    ("scenes/temple/temple", {}, 320, 180),
    ("scenes/temple/temple", {}, 960, 540),
    ("scenes/temple/temple", {}, 1001, 333),
    ("scenes/test/s2", {"atr": "scenes/temple/temple.atr"}, 800, 600),
    ("scenes/test/s2", {"atr": "scenes/temple/temple.atr"}, 97, 61),
]


def frame(renderer, scene, lists):
    renderer.set_option("light_lists", lists)
    try:
        renderer.upload(scene)
        renderer.frame_begin()
        renderer.counters()  # (totals since the last call)
        out = renderer.render_host()
        cnt = renderer.counters().asdict()
    finally:
        renderer.set_option("light_lists", 1)
    return out, cnt


@pytest.mark.parametrize("prefix,over,w,h", CASES, ids=["%s_%dx%d" % (c[0].split("/")[1], c[2], c[3]) for c in CASES])
def test_light_lists_on_and_off_give_the_same_frame(prefix, over, w, h):
    os.chdir(ROOT)
    over = {k: os.path.join(ROOT, v) for k, v in over.items()}
    scene = rt.Scene(os.path.join(ROOT, prefix), width=w, height=h, **over)
    r = rt.Renderer(0)
    try:
        on, con = frame(r, scene, 1)
        off, coff = frame(r, scene, 0)
    finally:
        r.close()
    assert np.array_equal(on["ids"], off["ids"])
    assert np.array_equal(on["rgb"].view(np.uint32), off["rgb"].view(np.uint32))
    assert np.array_equal(on["minmax"].view(np.uint32), off["minmax"].view(np.uint32))
    print("%s %dx%d: %d tickets from the lists, %d fell back to the walk" % (prefix, w, h, con["shadow_list_tickets"], con["shadow_list_fallbacks"]))
    assert coff["shadow_list_tickets"] == 0 and coff["shadow_list_fallbacks"] == 0
    assert con["shadow_list_tickets"] > 0
    if w >= 800 and "temple" in prefix:  # (at a low resolution an 8x8 tile of hit points covers more texels than a ticket takes from the lists)
        assert con["shadow_list_tickets"] > 4 * con["shadow_list_fallbacks"], "the lists answer most of the tickets they are tried for"


def test_scenes_the_lists_do_not_apply_to_go_without():
    """A transparent triangle anywhere (ulica, pokoj's glass), or planar lights only: every shadow query takes the walk."""
    os.chdir(ROOT)
    r = rt.Renderer(0)
    try:
        for prefix, over in [("scenes/ulica/ulica", {}), ("scenes/s2/s2", {"atr": os.path.join(ROOT, "scenes/pokoj/s2.atr")})]:
            scene = rt.Scene(os.path.join(ROOT, prefix), width=96, height=64, **over)
            out, cnt = frame(r, scene, 1)
            assert cnt["shadow_list_tickets"] == 0 and cnt["shadow_list_fallbacks"] == 0
    finally:
        r.close()


@pytest.mark.parametrize("seed", list(range(1, 17)))
def test_random_opaque_scenes_with_lights_in_hard_places(tmp_path, seed):
    """Generated scenes without transparent or reflective surfaces (tests/scenegen.py: a room of large triangles, boxes, a
    crowded cluster, slivers), lights in the floor's plane, on a corner of the geometry, inside the cluster: the frame from
    the lists equals the frame from the walks and the CPU oracle's (ids and raw colour bit for bit outside pow())."""
    from oracle import oracle as O
    from conftest import pow_free_pixels
    from tests.scenegen import write_scene
    os.chdir(ROOT)
    prefix = write_scene(str(tmp_path), seed, 256, 192, opaque=True, hard_lights=True)
    scene = rt.Scene(prefix)
    r = rt.Renderer(0)
    try:
        on, con = frame(r, scene, 1)
        off, coff = frame(r, scene, 0)
    finally:
        r.close()
    assert np.array_equal(on["ids"], off["ids"])
    assert np.array_equal(on["rgb"].view(np.uint32), off["rgb"].view(np.uint32))
    assert con["shadow_list_tickets"] + con["shadow_list_fallbacks"] > 0 and coff["shadow_list_tickets"] == 0
    osc = O.OracleScene(prefix)
    o = osc.render(shadow_cache=False, threads=1 if osc.npl else 4)
    assert np.array_equal(on["ids"], o["ids"])
    clean = o["taint"] == 0
    exact = pow_free_pixels(scene, o["ids"]) & clean
    assert np.array_equal(on["rgb"][exact].view(np.uint32), o["rgb"][exact].view(np.uint32))
    fin = clean[..., None] & np.isfinite(o["rgb"])
    rel = np.abs(on["rgb"][fin] - o["rgb"][fin]) / np.maximum(np.abs(o["rgb"][fin]), 1e-30)
    assert rel.max() <= 1e-4
    print("seed %d: %d tickets from the lists, %d walked" % (seed, con["shadow_list_tickets"], con["shadow_list_fallbacks"]))


@pytest.mark.parametrize("w,h", [(960, 540), (777, 431)])
def test_the_lists_own_kernel_and_the_lists_inside_the_walk_kernel_agree(w, h):
    """Option "light_list_kernel": the tickets whose hit points cover few texels are settled by a kernel of their own (a
    tile's hit points for all lights in turn), the others go to the walk kernel as a list of tickets (2: always; 1, the
    default: for shares of a frame large enough); with 0 the walk kernel looks every ticket up itself.  Same frame, same
    number of tickets answered from the lists."""
    os.chdir(ROOT)
    scene = rt.Scene(os.path.join(ROOT, "scenes/temple/temple"), width=w, height=h)
    r = rt.Renderer(0)
    try:
        r.set_option("light_list_kernel", 2)
        a, ca = frame(r, scene, 1)
        r.set_option("light_list_kernel", 0)
        b, cb = frame(r, scene, 1)
    finally:
        r.close()
    assert np.array_equal(a["ids"], b["ids"])
    assert np.array_equal(a["rgb"].view(np.uint32), b["rgb"].view(np.uint32))
    assert ca["shadow_list_tickets"] == cb["shadow_list_tickets"] > 0
    assert ca["shadow_list_fallbacks"] == cb["shadow_list_fallbacks"]


def test_the_lists_are_built_from_the_second_frame_of_a_scene_on():
    """Option "light_lists_after" (default 1; the tests run with 0 from the environment): the first frame of an uploaded scene
    walks (a caller that renders one frame does not wait for the rectangles to be worked out on the host), the second
    builds the lists; an upload of the same scene again changes nothing, another scene starts over.  Same frames."""
    os.chdir(ROOT)
    scene = rt.Scene(os.path.join(ROOT, "scenes/temple/temple"), width=480, height=270)
    other = rt.Scene(os.path.join(ROOT, "scenes/test/s2"), width=480, height=270, atr=os.path.join(ROOT, "scenes/temple/temple.atr"))
    r = rt.Renderer(0)
    try:
        r.set_option("light_lists_after", 1)
        first, c1 = frame(r, scene, 1)
        second, c2 = frame(r, scene, 1)  # (uploads the same scene again)
        third, c3 = frame(r, scene, 1)
        o1, d1 = frame(r, other, 1)
        o2, d2 = frame(r, other, 1)
    finally:
        r.close()
    assert c1["shadow_list_tickets"] == 0 and c2["shadow_list_tickets"] > 0 and c3["shadow_list_tickets"] == c2["shadow_list_tickets"]
    assert d1["shadow_list_tickets"] + d1["shadow_list_fallbacks"] == 0 and d2["shadow_list_tickets"] + d2["shadow_list_fallbacks"] > 0
    for a, b in ((first, second), (first, third), (o1, o2)):
        assert np.array_equal(a["ids"], b["ids"])
        assert np.array_equal(a["rgb"].view(np.uint32), b["rgb"].view(np.uint32))
