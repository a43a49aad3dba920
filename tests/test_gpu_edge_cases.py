"""Edge cases of the GPU trace path, all through the C ABI and all against the CPU oracle or the
reference's own digests: ragged image sizes, empty light sets, the three voxelisation modes, chunked
queues, idempotence, error codes, and the C command line end to end."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd.api import rgba_to_u8
from oracle import oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KNOWN = json.load(open(os.path.join(ROOT, "tests", "golden", "known_answers.json")))


@pytest.fixture(scope="module")
def renderer():
    os.chdir(ROOT)
    r = rt.Renderer(0)
    yield r
    r.close()


def both(renderer, prefix, w, h, **over):
    scene = rt.Scene(prefix, width=w, height=h, **over)
    renderer.upload(scene)
    renderer.frame_begin()
    g = renderer.render_host()
    o = O.OracleScene(prefix, width=w, height=h, **over).render(shadow_cache=False)
    return scene, g, o


def assert_same(g, o):
    assert np.array_equal(g["ids"], o["ids"])
    assert np.array_equal(g["rgb"].view(np.uint32), o["rgb"].view(np.uint32)) or np.allclose(g["rgb"], o["rgb"], rtol=1e-5, atol=0)
    assert np.allclose(g["minmax"], o["minmax"], rtol=1e-6)


@pytest.mark.parametrize("w,h", [(1, 1), (7, 3), (9, 17), (64, 8), (65, 9)])
def test_ragged_image_sizes(renderer, w, h):
    _, g, o = both(renderer, os.path.join(ROOT, "scenes/s3/s3"), w, h)
    assert_same(g, o)


def test_rays_that_never_enter_the_grid_leave_pixels_untouched(renderer):
    """scenes/test: 31 % of the primary rays miss the domain; those pixels stay 0 and out of min/max."""
    over = {"atr": os.path.join(ROOT, "scenes/temple/temple.atr")}
    _, g, o = both(renderer, os.path.join(ROOT, "scenes/test/s2"), 200, 150, **over)
    assert_same(g, o)
    assert (g["ids"] < 0).sum() > 0.5 * g["ids"].size


def test_no_lights(renderer, tmp_path):
    """A missing .lgt is only a warning in the reference: nl = 0, ambient and secondary rays only."""
    _, g, o = both(renderer, os.path.join(ROOT, "scenes/s3/s3"), 80, 60, lgt=str(tmp_path / "none.lgt"))
    assert_same(g, o)


@pytest.mark.parametrize("cfg_text", ["voxmode FIXED voxparams 5 4 6\n", "voxmode MODIFIED_DEFAULT voxparams 0.5 2 1.5\n",
                                      "gamma 1.8 distmod 3.5 voxmode DEFAULT voxparams 1 1 1\n",
                                      "voxmode FIXED voxparams 1 1 1\n",      # one cell lists all 1034 triangles: 65 batches, 3 header chunks
                                      "voxmode FIXED voxparams 64 40 64\n"])  # 163 840 cells: occupancy map too big for LDS, short lists
def test_voxelisation_modes_and_config_keys(renderer, tmp_path, cfg_text):
    cfg = tmp_path / "r.cfg"
    cfg.write_text(cfg_text)
    prefix = os.path.join(ROOT, "scenes/pokoj/s2")
    scene, g, o = both(renderer, prefix, 96, 72, cfg=str(cfg))
    assert_same(g, o)
    og = O.OracleScene(prefix, width=96, height=72, cfg=str(cfg))
    og.render(shadow_cache=False, rows=(0, 1))
    gg, ogr = renderer.grid(), og.grid()
    assert np.array_equal(gg["nv"], ogr["nv"]) and np.array_equal(gg["refs"], ogr["refs"])


def test_invalid_voxel_coefficients_are_rejected_like_the_reference(renderer, tmp_path):
    cfg = tmp_path / "bad.cfg"
    cfg.write_text("voxmode FIXED voxparams 4 0 4\n")
    scene = rt.Scene(os.path.join(ROOT, "scenes/s3/s3"), width=32, height=24, cfg=str(cfg))
    renderer.upload(scene)
    with pytest.raises(rt.RtgpuError) as e:
        renderer.frame_begin()
    assert e.value.code == 7  # E_INVALID_PARAM_VALUE


def test_chunked_queues_give_the_same_frame(renderer):
    """A tiny queue budget cuts the frame into many chunks of tiles (secondary rays included)."""
    prefix = os.path.join(ROOT, "scenes/pokoj/s2")
    scene = rt.Scene(prefix, width=200, height=150)
    renderer.upload(scene)
    renderer.frame_begin()
    whole = renderer.render_host()
    renderer.set_option("chunk_bytes", 2 << 20)
    try:
        renderer.frame_begin()
        parts = renderer.render_host()
    finally:
        renderer.set_option("chunk_bytes", 8 << 30)
    assert np.array_equal(whole["ids"], parts["ids"])
    assert np.array_equal(whole["rgb"].view(np.uint32), parts["rgb"].view(np.uint32))
    assert np.array_equal(whole["minmax"], parts["minmax"])


def test_frames_are_idempotent_and_independent_of_the_persistent_grid_size(renderer):
    prefix = os.path.join(ROOT, "scenes/s5/s5")
    scene = rt.Scene(prefix, width=160, height=120)
    renderer.upload(scene)
    renderer.frame_begin()
    a = renderer.render_host()
    renderer.frame_begin()  # the reference would grow the domain again here; the GPU library does not
    b = renderer.render_host()
    renderer.set_option("waves_per_cu", 4)
    try:
        c = renderer.render_host()
    finally:
        renderer.set_option("waves_per_cu", 0)  # 0 = the default
    for other in (b, c):
        assert np.array_equal(a["ids"], other["ids"])
        assert np.array_equal(a["rgb"].view(np.uint32), other["rgb"].view(np.uint32))


def test_planar_lights_refuse_more_than_one_rank_in_one_call(renderer):
    import torch
    scene = rt.Scene(os.path.join(ROOT, "scenes/s2/s2"), width=32, height=24)
    renderer.upload(scene)
    renderer.frame_begin()
    n = renderer.tile_count(0, 2) * 64
    rgb = torch.zeros(n * 3, device="cuda"); ids = torch.zeros(n, dtype=torch.int32, device="cuda")
    mm = torch.zeros(6, device="cuda")
    torch.cuda.synchronize()  # the context enqueues on its own stream, which does not wait for torch's
    with pytest.raises(rt.RtgpuError) as e:
        renderer.render(rgb, ids, mm, 0, 2)
    assert e.value.code == 102


@pytest.mark.parametrize("over,world", [({}, 2), ({}, 3), ({"atr": "scenes/pokoj/s2.atr"}, 2)], ids=["s2_w2", "s2_w3", "s2_glass_w2"])
def test_planar_lights_on_several_ranks_draw_rand_in_the_reference_order(renderer, over, world):
    """trace / exchange of hit counts / shade with `world` contexts on one GPU: the reassembled frame equals the
    single-rank frame bit for bit (which equals the reference: test_gpu_parity), secondary rays included."""
    import torch
    from raytracer_amd import dist as D
    over = {k: os.path.join(ROOT, v) for k, v in over.items()}
    w, h = 120, 90
    scene = rt.Scene(os.path.join(ROOT, "scenes/s2/s2"), width=w, height=h, **over)
    renderer.upload(scene)
    renderer.frame_begin()
    whole = renderer.render_host()
    ntiles = D.tiles_of(w, h)
    ranks = [rt.Renderer(0) for _ in range(world)]
    try:
        counts = []
        for k, r in enumerate(ranks):
            r.upload(scene); r.frame_begin(); r.render_trace(k, world)
            c = torch.zeros(r.tile_count(k, world) * 64, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()  # the contexts enqueue on their own streams, which do not wait for torch's
            r.hit_counts(c); r.sync()
            counts.append(c)
        per_rank = D.tile_count(ntiles, 0, world) * 64
        every = torch.zeros(world * per_rank, dtype=torch.int32, device="cuda")
        for k, c in enumerate(counts):
            every[k * per_rank:k * per_rank + c.numel()] = c
        total = int(every.sum().item())  # (.item() waits for torch's stream: `every` is complete)
        rgbs, idss, mms = [], [], []
        for k, r in enumerate(ranks):
            n = r.tile_count(k, world) * 64
            rgb = torch.zeros(n * 3, device="cuda"); ids = torch.zeros(n, dtype=torch.int32, device="cuda"); mm = torch.zeros(6, device="cuda")
            torch.cuda.synchronize()
            r.render_shade(rgb, ids, mm, every, per_rank, total); r.sync()
            rgbs.append(rgb.cpu().numpy().reshape(-1, 3)); idss.append(ids.cpu().numpy()); mms.append(mm.cpu().numpy())
        rgb = D.untile_reference(np.concatenate(rgbs), world, w, h)
        ids = D.untile_reference(np.concatenate(idss), world, w, h)
        assert np.array_equal(ids, whole["ids"])
        assert np.array_equal(rgb.view(np.uint32), whole["rgb"].view(np.uint32))
        mm = np.stack(mms)
        assert np.array_equal(np.concatenate([mm[:, :3].min(0), mm[:, 3:].max(0)]), whole["minmax"])
    finally:
        for r in ranks:
            r.close()


@pytest.mark.parametrize("name", ["s3", "temple", "ulica", "test"])
def test_native_resolution_bmp_matches_reference_digest(renderer, name, tmp_path):
    """Whole pipeline incl. tone map and BMP writer: the file must be byte-identical to the reference's."""
    k = KNOWN[name]
    over = {a: os.path.join(ROOT, b) for a, b in k["overrides"].items()}
    out = str(tmp_path / "o.bmp")
    scene = rt.Scene(os.path.join(ROOT, k["prefix"]), **over)
    renderer.upload(scene)
    renderer.frame_begin()
    g = renderer.render_host()
    rgba = renderer.tonemap_host(g["rgb"], g["minmax"], scene.gamma)
    rt.save_bmp24(rgba, out)
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == k["bmp_sha256"]


def test_temple_4k_bmp_is_byte_identical_to_the_reference(renderer, tmp_path):
    """The frame the headline number is quoted on, every ray of it, tone map and BMP writer included: SHA-256 of
    the file the reference binary (built from /root/reference/src) writes for `-s scenes/temple/temple` with
    the camera line set to 3840 2160."""
    out = str(tmp_path / "o.bmp")
    scene = rt.Scene(os.path.join(ROOT, "scenes/temple/temple"), width=3840, height=2160)
    renderer.upload(scene)
    renderer.frame_begin()
    g = renderer.render_host()
    rgba = renderer.tonemap_host(g["rgb"], g["minmax"], scene.gamma)
    rt.save_bmp24(rgba, out)
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == "47a5b58a51f53a8ff5cc73fc010f1738942488e14a0019eb62aa63bb5cb70f7a"


def test_ulica_384x216_against_the_reference_frame(renderer):
    """Depth-5 reflections and refractions, 64 lights: ids bit-exact, 8-bit image within 1/255 per channel.
    The raw colours differ from the reference's in a few hundred pixels because of its per-triangle shadow
    cache (src/voxelize.c:647-655), which depends on the order pixels are rendered in and is not reproduced;
    they are bit-identical to the cache-less CPU oracle (checked at this size when the fixture was made)."""
    z = np.load(os.path.join(ROOT, "tests/golden/ulica_384x216.npz"))
    scene = rt.Scene(os.path.join(ROOT, "scenes/ulica/ulica"), width=384, height=216)
    renderer.upload(scene)
    renderer.frame_begin()
    g = renderer.render_host()
    assert np.array_equal(g["ids"], z["ids"])
    assert np.array_equal(g["minmax"], z["minmax"])
    img = rgba_to_u8(renderer.tonemap_host(g["rgb"], g["minmax"], scene.gamma)).astype(int)
    d = np.abs(img - z["img"].astype(int))
    assert d.max() <= 1
    assert (d.max(-1) > 0).mean() < 1e-3


@pytest.mark.parametrize("w,h,bpp", [(400, 300, 24), (65, 9, 24), (7, 3, 24), (66, 5, 24), (65, 9, 32)])
def test_bmp_rows_packed_on_the_device_equal_the_host_writer(renderer, tmp_path, w, h, bpp):
    """rtgpu_bmp_rows + rtgpu_bmp_header against rtBitmapSave (padding of odd widths, bottom-up rows, B G R order);
    at s3's native size the file also has the reference's digest."""
    import torch
    scene = rt.Scene(os.path.join(ROOT, "scenes/s3/s3"), width=w, height=h)
    renderer.upload(scene)
    renderer.frame_begin()
    g = renderer.render_host()
    rgba = renderer.tonemap_host(g["rgb"], g["minmax"], scene.gamma)
    data = renderer.bmp_file(torch.from_numpy(rgba.view(np.int32).reshape(-1)).cuda(), bpp)
    if bpp == 24:
        out = str(tmp_path / "o.bmp")
        rt.save_bmp24(rgba, out)
        assert data == open(out, "rb").read()
        if (w, h) == (400, 300):
            assert hashlib.sha256(data).hexdigest() == KNOWN["s3"]["bmp_sha256"]
    else:
        row = 4 * w
        assert len(data) == 54 + row * h and data[28] == 32
        px = np.frombuffer(data[54:], np.uint8).reshape(h, w, 4)[::-1]
        assert np.array_equal(px[..., 2], (rgba >> 24) & 0xff) and np.array_equal(px[..., 0], (rgba >> 8) & 0xff) and not px[..., 3].any()


def test_command_line_end_to_end(tmp_path):
    """The C host (loaders -> rtVisualizedSceneRaytrace -> ToBitmap -> rtBitmapSave) as a user runs it."""
    k = KNOWN["s3"]
    out, ids = str(tmp_path / "o.bmp"), str(tmp_path / "ids.bin")
    exe = os.path.join(ROOT, "raytracer_amd", "raytrace")
    p = subprocess.run([exe, "-s", os.path.join(ROOT, k["prefix"]), "-o", out, "-I", ids], cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert hashlib.sha256(open(ids, "rb").read()).hexdigest() == k["ids_sha256"]
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == k["bmp_sha256"]


def test_command_line_resolution_override_and_textured_config(tmp_path):
    exe = os.path.join(ROOT, "raytracer_amd", "raytrace")
    out = str(tmp_path / "o.bmp")
    p = subprocess.run([exe, "-s", os.path.join(ROOT, "scenes/pokoj/s2"), "-C", os.path.join(ROOT, "scenes/cegly.cfg"),
                        "-W", "160", "-H", "120", "-o", out], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    z = np.load(os.path.join(ROOT, "tests/golden/pokoj_cegly.npz"))
    data = np.frombuffer(open(out, "rb").read()[54:], np.uint8).reshape(120, 160 * 3)[::-1].reshape(120, 160, 3)[:, :, ::-1]
    # same image as the oracle's (textured, cegly.cfg gamma 2.0) outside the reference's undefined pixels
    osc = O.OracleScene(os.path.join(ROOT, "scenes/pokoj/s2"), texture=True, width=160, height=120, cfg=os.path.join(ROOT, "scenes/cegly.cfg"))
    o = osc.render(shadow_cache=False)
    want = O.rgba_to_u8(O.tonemap(o["rgb"], o["minmax"], osc.gamma)).astype(int)
    clean = o["taint"] == 0
    assert np.abs(data.astype(int) - want)[clean].max() <= 1
    assert np.array_equal(o["ids"], z["ids"])


@pytest.mark.parametrize("depth", [1, 2])
@pytest.mark.parametrize("name", ["s3", "s2", "temple"])
def test_command_line_devices_1_issues_the_collectives_and_writes_the_reference_bmp(tmp_path, name, depth):
    """`raytrace --devices 1`: the one-process N-GPU host (host/multigpu.c) with a communicator of one rank: colour range
    all-reduce (MIN and MAX), RGBA8 tile send/receive, and for s2's planar light the all-gather of the hit counts
    between the two halves of the frame.  The BMP is the reference's, with one frame in flight and with two (what the
    host does with N > 1: a second set of contexts, streams, buffers and communicators)."""
    k = KNOWN[name]
    out = str(tmp_path / "o.bmp")
    exe = os.path.join(ROOT, "raytracer_amd", "raytrace")
    p = subprocess.run([exe, "-s", os.path.join(ROOT, k["prefix"]), "-o", out, "--devices", "1", "--frames", "5"], cwd=str(tmp_path),
                       env=dict(os.environ, RTGPU_FRAMES_IN_FLIGHT=str(depth)), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    assert "ray-tracing on 1 device(s) done" in p.stdout
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == k["bmp_sha256"]


@pytest.mark.parametrize("name,ndev,depth", [("s3", 3, 1), ("s3", 2, 2), ("s2", 3, 1), ("temple", 4, 2), ("ulica", 3, 2)])
def test_command_line_devices_n_rehearsed_on_one_gpu(tmp_path, name, ndev, depth):
    """`raytrace --devices N` with RTGPU_LOOPBACK=1: N ranks (contexts, streams, one host thread each) on the one GPU of
    the box, the collectives replaced by copies on that GPU.  The image is the reference's (s3, s2 with its planar light
    and the hit-count exchange, temple; ulica at 192x108: the single-device command line's), and the per-device calls of
    a frame run side by side: rank 1's call has started before rank 0's has returned, in every frame, although
    rtgpu_render waits for its device once per chunk in scenes with secondary rays (s3, ulica)."""
    import re
    exe = os.path.join(ROOT, "raytracer_amd", "raytrace")
    out = str(tmp_path / "o.bmp")
    if name == "ulica":
        args = ["-s", os.path.join(ROOT, "scenes/ulica/ulica"), "-W", "192", "-H", "108"]
        one = str(tmp_path / "one.bmp")
        p = subprocess.run([exe] + args + ["-o", one], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-1500:]
        want = hashlib.sha256(open(one, "rb").read()).hexdigest()
    else:
        args = ["-s", os.path.join(ROOT, KNOWN[name]["prefix"])]
        want = KNOWN[name]["bmp_sha256"]
    frames = 5
    p = subprocess.run([exe] + args + ["-o", out, "--devices", str(ndev), "--frames", str(frames)], cwd=str(tmp_path),
                       env=dict(os.environ, RTGPU_LOOPBACK="1", RTGPU_TRACE="1", RTGPU_FRAMES_IN_FLIGHT=str(depth)),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == want
    calls = {}
    for f, r, job, a, b in re.findall(r"trace: frame (\d+) rank (\d+) (.*?): started ([0-9.]+) ms, returned ([0-9.]+) ms", p.stderr):
        if "shade" not in job:
            calls[(int(f), int(r))] = (float(a), float(b))
    assert len(calls) == frames * ndev
    for f in range(frames):
        assert calls[(f, 1)][0] < calls[(f, 0)][1], (f, calls[(f, 0)], calls[(f, 1)])


def test_command_line_devices_beyond_the_node_is_an_error(tmp_path):
    exe = os.path.join(ROOT, "raytracer_amd", "raytrace")
    p = subprocess.run([exe, "-s", os.path.join(ROOT, "scenes/s3/s3"), "-o", str(tmp_path / "o.bmp"), "--devices", "64"], cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 1 and "64 devices asked for" in p.stderr
    assert not os.path.exists(str(tmp_path / "o.bmp"))


def _ulica_768(renderer):
    z = np.load(os.path.join(ROOT, "tests/golden/ulica_768x432.npz"))
    scene = rt.Scene(os.path.join(ROOT, "scenes/ulica/ulica"), width=768, height=432)
    renderer.upload(scene)
    return z, scene


def test_ulica_768x432_in_several_chunks_against_the_reference_frame(renderer):
    """64 lights, secondary rays to depth 5, the frame cut into a first small chunk with worst-case queues and several
    chunks with queues sized from what the scene produces: primary-hit ids bit-exact against the unmodified reference,
    8-bit image within 1/255 (its shadow cache is not reproduced: a handful of channel values differ by one)."""
    z, scene = _ulica_768(renderer)
    renderer.set_option("chunk_bytes", 256 << 20)
    try:
        renderer.frame_begin()
        g = renderer.render_host()
    finally:
        renderer.set_option("chunk_bytes", 8 << 30)
    assert np.array_equal(g["ids"], z["ids"])
    assert np.array_equal(g["minmax"], z["minmax"])
    img = rgba_to_u8(renderer.tonemap_host(g["rgb"], g["minmax"], scene.gamma)).astype(int)
    d = np.abs(img - z["img"].astype(int))
    print("ulica 768x432: 8-bit max diff %d, channel values off by one %d" % (d.max(), int((d > 0).sum())))
    assert d.max() <= 1
    assert (d.max(-1) > 0).mean() < 1e-3


def test_chunk_size_and_queue_sizing_do_not_change_the_frame(renderer):
    """The same ulica frame as one chunk with worst-case queues, as many chunks with queues sized from the scene, and
    with a margin so small that chunks overflow their queues and are rendered again: ids, raw colours and colour range
    are bit-identical, and so is a second frame (the sizing learnt from the first one is reused)."""
    z, scene = _ulica_768(renderer)
    frames, chunks = [], []
    settings = [dict(queue_sizing=0, chunk_bytes=8 << 30), dict(queue_sizing=1, chunk_bytes=256 << 20),
                dict(queue_sizing=1, chunk_bytes=192 << 20, queue_margin=300), dict(queue_sizing=1, chunk_bytes=64 << 20)]
    try:
        for st in settings:
            for k, v in st.items():
                renderer.set_option(k, v)
            for rep in range(2):
                renderer.frame_begin()
                frames.append(renderer.render_host())
                cnt = renderer.counters()
                chunks.append((int(cnt.chunks), int(cnt.chunks_rerendered)))
    finally:
        for k, v in dict(queue_sizing=1, chunk_bytes=8 << 30, queue_margin=1500).items():
            renderer.set_option(k, v)
    print("chunks (rendered again) per frame:", chunks)
    assert chunks[0] == (1, 0) and chunks[1] == (1, 0)      # worst-case queues, 8 GiB: one chunk
    assert chunks[2][0] >= 3 and chunks[3][0] >= 3          # several chunks; the second frame needs no first small chunk
    assert chunks[4][1] >= 1 or chunks[5][1] >= 1           # the thin margin makes some chunk overflow its queues
    assert chunks[6][0] > chunks[2][0]
    a = frames[0]
    assert np.array_equal(a["ids"], z["ids"])
    for b in frames[1:]:
        assert np.array_equal(a["ids"], b["ids"])
        assert np.array_equal(a["rgb"].view(np.uint32), b["rgb"].view(np.uint32))
        assert np.array_equal(a["minmax"].view(np.uint32), b["minmax"].view(np.uint32))


def test_short_forms_of_the_double_precision_idioms_are_exact_for_every_float():
    """normalize3 and dist3 compute (float)(1.0 / sqrt((double)s)) and (float)sqrt((double)s) (src/vectormath.h:45-48,
    60-66) by shorter instruction sequences; the library compares both forms on the device for all 2^32 inputs."""
    import ctypes as C
    r = rt.Renderer(0)
    try:
        L = rt.gpu_lib()
        L.rtgpu_math_selfcheck.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        bad = (C.c_uint64 * 2)(1, 1)
        assert L.rtgpu_math_selfcheck(r._h, bad) == 0
        assert (bad[0], bad[1]) == (0, 0)
    finally:
        r.close()
