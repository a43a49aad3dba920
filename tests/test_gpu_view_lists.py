"""Primary rays answered from the view lists (option "view_lists") give the frame the cell walks give, bit for bit: the
reference's nearest-hit walk (src/voxelize.c:536-613) reports the nearest accepted triangle of the first cell that lists
one before the cell's exit distance; the lists hold every triangle the test can accept for a pixel's ray
(tests/test_view_lists_cpu.py), and the nearest of those counts only if the ray's own walk passes its distance in a cell
that lists it."""
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
    ("scenes/ulica/ulica", {}, 480, 270),
    ("scenes/s2/s2", {"atr": "scenes/pokoj/s2.atr"}, 200, 150),
    ("scenes/test/s2", {"atr": "scenes/temple/temple.atr"}, 800, 600),
    ("scenes/test/s2", {"atr": "scenes/temple/temple.atr"}, 97, 61),
]


def frame(renderer, scene, lists, rank=0, world=1, **options):
    renderer.set_option("view_lists", lists)
    for k, v in options.items():
        renderer.set_option(k, v)
    try:
        renderer.upload(scene)
        renderer.frame_begin()
        renderer.counters()  # (totals since the last call)
        if world == 1:
            out = renderer.render_host()
        else:
            import torch
            dev = torch.device("cuda:0")
            n = renderer.tile_count(rank, world) * 64
            rgb = torch.zeros(n * 3, dtype=torch.float32, device=dev)
            ids = torch.zeros(n, dtype=torch.int32, device=dev)
            mm = torch.zeros(6, dtype=torch.float32, device=dev)
            torch.cuda.synchronize(dev)  # the context enqueues on its own stream, which does not wait for torch's zero fill
            renderer.render(rgb, ids, mm, rank, world)
            renderer.sync()
            out = (rgb.cpu().numpy(), ids.cpu().numpy(), mm.cpu().numpy())
        cnt = renderer.counters().asdict()
    finally:
        renderer.set_option("view_lists", 1)
    return out, cnt


@pytest.mark.parametrize("prefix,over,w,h", CASES, ids=["%s_%dx%d" % (c[0].split("/")[1], c[2], c[3]) for c in CASES])
def test_view_lists_on_and_off_give_the_same_frame(prefix, over, w, h):
    os.chdir(ROOT)
    over = {k: os.path.join(ROOT, v) for k, v in over.items()}
    scene = rt.Scene(os.path.join(ROOT, prefix), width=w, height=h, **over)
    r = rt.Renderer(0)
    try:
        on, con = frame(r, scene, 1)
        off, coff = frame(r, scene, 0)
        coarse, ccoarse = frame(r, scene, 1, view_list_texel_shift=3)
        fine, cfine = frame(r, scene, 1, view_list_texel_shift=0)
        r.set_option("view_list_texel_shift", 1)
    finally:
        r.close()
    for other in (off, coarse, fine):
        assert np.array_equal(on["ids"], other["ids"])
        assert np.array_equal(on["rgb"].view(np.uint32), other["rgb"].view(np.uint32))
        assert np.array_equal(on["minmax"].view(np.uint32), other["minmax"].view(np.uint32))
    tiles = ((w + 7) // 8) * ((h + 7) // 8)
    print("%s %dx%d: %d of %d tickets from the lists, %d fell back to the walk" % (prefix, w, h, con["view_list_tickets"], tiles, con["view_list_fallbacks"]))
    assert coff["view_list_tickets"] == 0 and coff["view_list_fallbacks"] == 0
    assert con["view_list_tickets"] + con["view_list_fallbacks"] <= tiles
    assert con["view_list_tickets"] > 20 * con["view_list_fallbacks"], "the lists answer nearly all tickets"
    assert ccoarse["view_list_tickets"] > 0 and cfine["view_list_tickets"] > 0


@pytest.mark.parametrize("seed", list(range(1, 13)))
def test_random_scenes_with_the_view_lists(tmp_path, seed):
    """Generated scenes (tests/scenegen.py: a room of large triangles, boxes, a crowded cluster, slivers; transparent and
    reflective surfaces; the observer inside the geometry): the frame with the lists equals the frame from the walks and
    the CPU oracle's (ids and raw colour bit for bit outside pow())."""
    from oracle import oracle as O
    from conftest import pow_free_pixels
    from tests.scenegen import write_scene
    os.chdir(ROOT)
    prefix = write_scene(str(tmp_path), seed, 256, 192, opaque=seed % 3 == 0, hard_lights=seed % 2 == 0)
    scene = rt.Scene(prefix)
    r = rt.Renderer(0)
    try:
        on, con = frame(r, scene, 1)
        off, coff = frame(r, scene, 0)
    finally:
        r.close()
    assert np.array_equal(on["ids"], off["ids"])
    assert np.array_equal(on["rgb"].view(np.uint32), off["rgb"].view(np.uint32))
    assert con["view_list_tickets"] + con["view_list_fallbacks"] > 0 and coff["view_list_tickets"] == 0
    osc = O.OracleScene(prefix)
    o = osc.render(shadow_cache=False, threads=1 if osc.npl else 4)
    assert np.array_equal(on["ids"], o["ids"])
    clean = o["taint"] == 0
    exact = pow_free_pixels(scene, o["ids"]) & clean
    assert np.array_equal(on["rgb"][exact].view(np.uint32), o["rgb"][exact].view(np.uint32))
    print("seed %d: %d tickets from the lists, %d walked" % (seed, con["view_list_tickets"], con["view_list_fallbacks"]))


def test_a_share_of_the_tiles_and_a_moving_camera():
    """Tiles dealt out round-robin to ranks look their texels up by the tile's place on the screen; the lists are built
    anew for every frame's camera."""
    os.chdir(ROOT)
    scene = rt.Scene(os.path.join(ROOT, "scenes/temple/temple"), width=640, height=360)
    r = rt.Renderer(0)
    try:
        for rank, world in [(0, 2), (1, 2), (2, 3)]:
            a, ca = frame(r, scene, 1, rank, world)
            b, cb = frame(r, scene, 0, rank, world)
            for x, y in zip(a, b):
                assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
            assert ca["view_list_tickets"] > 0
        cam = scene.camera
        for k in range(3):
            for q in range(3):  # the observer and the screen move together
                cam.ob[q] += 0.37; cam.ul[q] += 0.37; cam.ur[q] += 0.37; cam.bl[q] += 0.37
            on, con = frame(r, scene, 1)
            off, coff = frame(r, scene, 0)
            assert np.array_equal(on["ids"], off["ids"])
            assert np.array_equal(on["rgb"].view(np.uint32), off["rgb"].view(np.uint32))
            assert con["view_list_tickets"] > 0
    finally:
        r.close()


def test_a_frame_whose_lists_do_not_fit_their_buffer_walks():
    """The references of a frame's lists are counted before they are written; with room for fewer (option
    "view_list_capacity", there for this test) nothing is written and every ticket walks: same frame."""
    os.chdir(ROOT)
    scene = rt.Scene(os.path.join(ROOT, "scenes/temple/temple"), width=640, height=360)
    r = rt.Renderer(0)
    try:
        full, cfull = frame(r, scene, 1)
        small, csmall = frame(r, scene, 1, view_list_capacity=1000)
        again, cagain = frame(r, scene, 1, view_list_capacity=0)
    finally:
        r.close()
    for other in (small, again):
        assert np.array_equal(full["ids"], other["ids"])
        assert np.array_equal(full["rgb"].view(np.uint32), other["rgb"].view(np.uint32))
    assert cfull["view_list_tickets"] > 0 and cagain["view_list_tickets"] == cfull["view_list_tickets"]
    assert csmall["view_list_tickets"] == 0 and csmall["view_list_fallbacks"] == 0


@pytest.mark.parametrize("name,prefix,over", [("temple", "scenes/temple/temple", {}), ("ulica", "scenes/ulica/ulica", {})])
def test_cameras_anywhere_in_the_scene_looking_anywhere(name, prefix, over):
    """Sixteen cameras at random places inside the scene's box (also inside its geometry, on its surfaces' planes, a hair
    from a vertex), looking in random directions with random fields of view: the frame with the lists is the frame from
    the walks."""
    os.chdir(ROOT)
    w, h = 240, 136
    scene = rt.Scene(os.path.join(ROOT, prefix), width=w, height=h, **over)
    V = scene.vertices().reshape(-1, 3).astype(np.float64)
    lo, hi = V.min(0), V.max(0)
    rng = np.random.default_rng(7)
    r = rt.Renderer(0)
    tickets = 0
    try:
        for k in range(16):
            if k % 4 == 3:  # a hair from a vertex
                ob = V[rng.integers(0, len(V))] + rng.standard_normal(3) * 1e-4 * (hi - lo).max()
            else:
                ob = lo + rng.random(3) * (hi - lo)
            fwd = rng.standard_normal(3); fwd /= np.linalg.norm(fwd)
            right = np.cross(fwd, rng.standard_normal(3)); right /= np.linalg.norm(right)
            up = np.cross(right, fwd)
            sd = (hi - lo).max() * 10.0 ** rng.uniform(-3, -1)
            hw = sd * rng.uniform(0.3, 2.0); hh = hw * h / w
            c = ob + fwd * sd
            cam = scene.camera
            for q in range(3):
                cam.ob[q] = ob[q]
                cam.ul[q] = c[q] - right[q] * hw + up[q] * hh
                cam.ur[q] = c[q] + right[q] * hw + up[q] * hh
                cam.bl[q] = c[q] - right[q] * hw - up[q] * hh
            on, con = frame(r, scene, 1)
            off, coff = frame(r, scene, 0)
            assert np.array_equal(on["ids"], off["ids"]), "camera %d" % k
            assert np.array_equal(on["rgb"].view(np.uint32), off["rgb"].view(np.uint32)), "camera %d" % k
            tickets += con["view_list_tickets"]
            print("%s camera %d: %d tickets from the lists, %d walked, %d pixels hit" % (name, k, con["view_list_tickets"], con["view_list_fallbacks"], int((on["ids"] >= 0).sum())))
    finally:
        r.close()
    assert tickets > 0


@pytest.mark.parametrize("prefix,w,h", [("scenes/temple/temple", 640, 360), ("scenes/ulica/ulica", 320, 180)])
def test_a_ticket_the_walks_do_not_confirm_takes_the_walk(prefix, w, h):
    """No scene here makes a ray's walk disagree with the lists' answer, so the way back to the walk is exercised by option
    "view_list_test_fallback": every other ticket is treated as unconfirmed after its lookup and walks from its untouched
    state.  Same frame; half of the tickets counted as walked."""
    os.chdir(ROOT)
    scene = rt.Scene(os.path.join(ROOT, prefix), width=w, height=h)
    r = rt.Renderer(0)
    try:
        ref, cref = frame(r, scene, 0)
        fb, cfb = frame(r, scene, 1, view_list_test_fallback=1)
        r.set_option("view_list_test_fallback", 0)
    finally:
        r.close()
    assert np.array_equal(ref["ids"], fb["ids"])
    assert np.array_equal(ref["rgb"].view(np.uint32), fb["rgb"].view(np.uint32))
    assert np.array_equal(ref["minmax"].view(np.uint32), fb["minmax"].view(np.uint32))
    tiles = ((w + 7) // 8) * ((h + 7) // 8)
    assert cfb["view_list_tickets"] + cfb["view_list_fallbacks"] == tiles
    assert abs(cfb["view_list_fallbacks"] - tiles // 2) <= tiles // 8 + 1


def test_one_context_through_scenes_of_growing_and_shrinking_size():
    """The lists' buffers are kept from frame to frame and grown by what needs more room (texels, references, triangles):
    a small scene at a large resolution, then a scene of many more triangles at the same resolution, a lower one, the small
    scene again: the frame with the lists is the frame from the walks every time."""
    os.chdir(ROOT)
    r = rt.Renderer(0)
    try:
        for prefix, over, w, h in [("scenes/test/s2", {"atr": "scenes/temple/temple.atr"}, 1600, 900), ("scenes/s3/s3", {}, 1600, 900),
                                    ("scenes/temple/temple", {}, 1600, 900), ("scenes/temple/temple", {}, 200, 120),
                                    ("scenes/ulica/ulica", {}, 320, 180), ("scenes/test/s2", {"atr": "scenes/temple/temple.atr"}, 2000, 1100)]:
            over = {k: os.path.join(ROOT, v) for k, v in over.items()}
            scene = rt.Scene(os.path.join(ROOT, prefix), width=w, height=h, **over)
            on, con = frame(r, scene, 1)
            off, coff = frame(r, scene, 0)
            assert np.array_equal(on["ids"], off["ids"]), prefix
            assert np.array_equal(on["rgb"].view(np.uint32), off["rgb"].view(np.uint32)), prefix
            assert con["view_list_tickets"] > 0 and coff["view_list_tickets"] == 0
    finally:
        r.close()
