#!/usr/bin/env python3
"""Time of one rank's share of a frame without the exchange: tools/share_time.py [world] [scene w h] [waves_per_cu].
Renders the tiles of rank 0 of `world` (frame_begin + render + local tone map) and prints ms per frame."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from tools.diag import SCENES  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
name = sys.argv[2] if len(sys.argv) > 2 else "temple"
w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (3840, 2160)
scene = rt.Scene(os.path.join(ROOT, SCENES[name]), width=w, height=h)
r = rt.Renderer(0, timing=1)
if len(sys.argv) > 5:
    r.set_option("waves_per_cu", int(sys.argv[5]))
for kv in os.environ.get("RT_OPTIONS", "").split(","):  # e.g. RT_OPTIONS=split=0,pipeline=0
    if "=" in kv:
        r.set_option(kv.split("=")[0], int(kv.split("=")[1]))
r.upload(scene)
dev = torch.device("cuda:0")
r.set_stream(torch.cuda.current_stream(dev))
for wd in sorted({1, world}):
    r.frame_begin()
    n = r.tile_count(0, wd) * 64
    rgb = torch.empty(n * 3, dtype=torch.float32, device=dev)
    ids = torch.empty(n, dtype=torch.int32, device=dev)
    mm = torch.empty(6, dtype=torch.float32, device=dev)
    rgba = torch.empty(n, dtype=torch.int32, device=dev)
    for rep in range(3):
        r.timings()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(10):
            r.frame_begin()
            r.render(rgb, ids, mm, 0, wd)
            r.tonemap(rgb, n, mm, scene.gamma, rgba)
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) * 100
    t = {k: round(v / 10, 3) for k, v in r.timings().asdict().items() if k.endswith("_ms")}
    print("world %d: %.3f ms per frame for rank 0's share; kernels %s" % (wd, ms, t), flush=True)
