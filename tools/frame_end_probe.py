#!/usr/bin/env python3
"""Two frames in flight on one GPU: what the end of a frame (range exchange, tone map, gather, untile) does to the
throughput.  The frame of dist.FrameRenderer on one rank with pieces of the N > 1 path added one by one:
tools/frame_end_probe.py [scene w h].  Variants: plain; scale (the two small element-wise kernels of exchange_minmax);
reduce (those and the all-reduce of a group of one); copy (the gather's copy of the RGBA8 tiles); all (reduce + copy: what
RTGPU_BENCH_FORCE_COLLECTIVES=1 runs)."""
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from raytracer_amd.dist import FramePipeline  # noqa: E402
from tools.diag import SCENES  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "temple"
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
scene = rt.Scene(os.path.join(ROOT, SCENES[name]), width=w, height=h)
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
NCCL = os.environ.get("RT_NO_NCCL") != "1"  # (RT_NO_NCCL=1: no process group, for the variants that need none)
if NCCL:
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group("nccl", device_id=dev)
# RT_DUMMY_STREAMS=k: k streams made (and kept) before the frames' own, which moves every later stream to another of the
# hardware queues HIP shares among a process's streams
dummies = [torch.cuda.Stream(dev) for _ in range(int(os.environ.get("RT_DUMMY_STREAMS", "0")))]
# RT_PREALLOC_MB=m: m MiB of device memory taken (and kept) before the contexts allocate theirs
held = torch.empty(int(os.environ.get("RT_PREALLOC_MB", "0")) << 20, dtype=torch.uint8, device=dev)
FRAMES = int(os.environ.get("AB_FRAMES", "40"))


def frame_of(slot, variant):
    r = slot.r
    sign = torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32, device=dev)
    whole = torch.empty(slot.npix_local, dtype=torch.int32, device=dev)

    def frame():
        r.frame_begin()
        r.render(slot.rgb, slot.ids, slot.minmax, 0, 1)
        mm = slot.minmax
        if variant in ("scale", "reduce", "all"):
            both = mm * sign
            if variant != "scale":
                dist.all_reduce(both, op=dist.ReduceOp.MAX)
            mm = both * sign
        r.tonemap(slot.rgb, slot.npix_local, mm, scene.gamma, slot.rgba)
        src = slot.rgba[:slot.npix_local]
        if variant in ("copy", "all"):
            whole.copy_(src)
            src = whole
        r.untile(src, 1, 4, slot.image)
        return slot.image
    return frame


for variant in os.environ.get("RT_VARIANTS", "plain,scale,reduce,copy,all,plain").split(","):
    pipe = FramePipeline(lambda: rt.Renderer(0), scene, 0, 1, 2)
    for s in pipe.slots:
        s.frame = frame_of(s, variant)
    best = None
    for rep in range(4):
        for _ in range(4):
            pipe.frame()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(FRAMES):
            pipe.frame()
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) * 1e3 / FRAMES
        best = ms if best is None or ms < best else best
    pipe.close()
    print("%-7s %.3f ms per frame" % (variant, best), flush=True)
if NCCL:
    dist.destroy_process_group()
