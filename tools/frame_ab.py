#!/usr/bin/env python3
"""Whole-frame time (set-up of the next frame overlapped, as bench.py runs it) under several option settings.
usage: tools/frame_ab.py scene w h 'opt=val,opt=val' ['...' ...]     ('' = the defaults)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import raytracer_amd as rt  # noqa: E402
from tools.diag import SCENES  # noqa: E402

scene, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sc = rt.Scene(os.path.join(ROOT, SCENES[scene]), width=w, height=h)
dev = torch.device("cuda:0")
for spec in sys.argv[4:] or [""]:
    opts = dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in spec.split(",") if kv)
    r = rt.Renderer(0)
    for k, v in opts.items():
        r.set_option(k, v)
    r.upload(sc)
    r.frame_begin()
    n = r.tile_count(0, 1) * 64
    rgb = torch.zeros(n * 3, dtype=torch.float32, device=dev); ids = torch.zeros(n, dtype=torch.int32, device=dev)
    mm = torch.zeros(6, dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)
    best = 1e9
    for rep in range(5):
        for _ in range(3):
            r.frame_begin(); r.render(rgb, ids, mm)
        r.sync()
        t0 = time.perf_counter()
        frames = int(os.environ.get("AB_FRAMES", "20"))
        for _ in range(frames):
            r.frame_begin(); r.render(rgb, ids, mm)
        r.sync()
        best = min(best, (time.perf_counter() - t0) / frames * 1e3)
    print(spec or "defaults", "%.3f ms per frame" % best, flush=True)
    r.close()
