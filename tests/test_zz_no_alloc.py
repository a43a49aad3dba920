"""rtgpu_render_static, the entry point that allocates nothing, against rtgpu_render: same frame bit for bit, on calls
in a row that share the static scratch array (the later ones with smaller inputs), and a clear refusal beyond its size."""
import os

import numpy as np
import pytest

import raytracer_amd as rt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def frames(r, scene, world=1, rank=0):
    """The rank's tile-major share through both entry points: ((rgb, ids, minmax) of render, of render_static)."""
    import torch
    r.upload(scene)
    r.frame_begin()
    n = r.tile_count(rank, world) * 64
    out = []
    for call in (r.render, r.render_static):
        rgb = torch.full((n * 3,), 7.0, dtype=torch.float32, device="cuda")
        ids = torch.full((n,), -7, dtype=torch.int32, device="cuda")
        mm = torch.zeros(6, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()  # the context enqueues on its own stream, which does not wait for torch's fills
        call(rgb, ids, mm, rank, world)
        r.sync()
        out.append((rgb.cpu().numpy(), ids.cpu().numpy(), mm.cpu().numpy()))
    return out


def same(a, b):
    assert np.array_equal(a[1], b[1]), "primary-hit ids differ"
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), "raw colours differ"
    assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32)), "colour range differs"


def test_static_render_equals_render_on_calls_in_a_row():
    r = rt.Renderer(0)
    try:
        # 1. temple 640x360: 230 400 pixels x 124 B of queues = 27 MiB of the 64 MiB array
        a, b = frames(r, rt.Scene(os.path.join(ROOT, "scenes/temple/temple"), width=640, height=360))
        same(a, b)
        assert (a[1] >= 0).sum() > 50000
        # 2. a smaller input with everything the queues can hold: ulica 64x48, 64 lights, secondary rays to depth 5
        a, b = frames(r, rt.Scene(os.path.join(ROOT, "scenes/ulica/ulica"), width=64, height=48))
        same(a, b)
        # 3. smaller again, one rank's share of three, ragged size
        a, b = frames(r, rt.Scene(os.path.join(ROOT, "scenes/s3/s3"), width=97, height=61), world=3, rank=1)
        same(a, b)
    finally:
        r.close()


def test_static_render_refuses_what_does_not_fit_and_leaves_the_context_usable():
    import torch
    r = rt.Renderer(0)
    try:
        assert int(rt.gpu_lib().rtgpu_static_scratch_bytes()) == 64 << 20
        scene = rt.Scene(os.path.join(ROOT, "scenes/temple/temple"), width=1920, height=1080)  # 245 MiB of queues
        r.upload(scene)
        r.frame_begin()
        n = r.tile_count() * 64
        rgb = torch.zeros(n * 3, dtype=torch.float32, device="cuda"); ids = torch.zeros(n, dtype=torch.int32, device="cuda")
        mm = torch.zeros(6, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        with pytest.raises(rt.RtgpuError) as e:
            r.render_static(rgb, ids, mm)
        assert e.value.code == 102 and "static scratch array holds 67108864" in str(e.value)
        # planar lights are refused too
        r.upload(rt.Scene(os.path.join(ROOT, "scenes/s2/s2"), width=32, height=24))
        r.frame_begin()
        with pytest.raises(rt.RtgpuError) as e:
            r.render_static(rgb, ids, mm)
        assert e.value.code == 102 and "planar lights" in str(e.value)
        # the refusals changed nothing: the same context renders, statically and not, and the two agree
        a, b = frames(r, rt.Scene(os.path.join(ROOT, "scenes/s3/s3"), width=96, height=72))
        same(a, b)
    finally:
        r.close()


def test_static_render_allocates_no_device_memory():
    """The property in the entry point's name: free device memory is the same before and after a call (the first call of a
    process loads nothing either: the kernels' code object came with the library)."""
    import torch
    r = rt.Renderer(0)
    try:
        scene = rt.Scene(os.path.join(ROOT, "scenes/ulica/ulica"), width=64, height=48)
        r.upload(scene)
        r.frame_begin()
        n = r.tile_count() * 64
        rgb = torch.zeros(n * 3, dtype=torch.float32, device="cuda"); ids = torch.zeros(n, dtype=torch.int32, device="cuda")
        mm = torch.zeros(6, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        r.render_static(rgb, ids, mm); r.sync()
        for _ in range(3):
            free0 = torch.cuda.mem_get_info()[0]
            r.render_static(rgb, ids, mm); r.sync()
            assert torch.cuda.mem_get_info()[0] == free0
        # rtgpu_render on the same context does allocate (its queues), which is what the comparison is worth
        free0 = torch.cuda.mem_get_info()[0]
        r.render(rgb, ids, mm); r.sync()
        assert torch.cuda.mem_get_info()[0] < free0
    finally:
        r.close()


def test_static_render_from_two_contexts_on_two_streams_is_ordered_by_the_library():
    """Two contexts of one GPU share the array.  Calls that follow each other on the host, each context on its own stream
    and nothing waiting in between, do not overlap on the device: every frame is the one rtgpu_render gives."""
    import torch
    ra, rb = rt.Renderer(0), rt.Renderer(0)
    try:
        sa = rt.Scene(os.path.join(ROOT, "scenes/temple/temple"), width=640, height=360)
        sb = rt.Scene(os.path.join(ROOT, "scenes/pokoj/s2"), width=96, height=72)
        want_a, _ = frames(ra, sa)
        want_b, _ = frames(rb, sb)
        bufs = []
        for r in (ra, rb):
            n = r.tile_count() * 64
            bufs.append([(torch.zeros(n * 3, dtype=torch.float32, device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda"),
                          torch.zeros(6, dtype=torch.float32, device="cuda")) for _ in range(3)])
        torch.cuda.synchronize()
        for k in range(3):  # a, b, a, b, a, b: enqueued back to back
            ra.render_static(*bufs[0][k])
            rb.render_static(*bufs[1][k])
        ra.sync(); rb.sync()
        for k in range(3):
            same(want_a, tuple(t.cpu().numpy() for t in bufs[0][k]))
            same(want_b, tuple(t.cpu().numpy() for t in bufs[1][k]))
    finally:
        ra.close(); rb.close()
