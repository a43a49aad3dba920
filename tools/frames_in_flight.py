#!/usr/bin/env python3
"""Throughput of one rank's share of a frame with several frames in flight on one GPU:
tools/frames_in_flight.py [world] [scene w h].  Each frame in flight has its own context (queues, grid buffers) and
stream; frames are dealt to the contexts in turn.  Mode "free": the streams run as the hardware schedules them;
mode "stagger": a frame's trace half starts when the previous frame's trace half has ended, so that the tail of one
frame's shadow walk meets the bulk of the next frame's kernels."""
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
dev = torch.device("cuda:0")
FRAMES = 24


def run(wd, k, mode, wpc=0):
    ctxs, streams, bufs, evs = [], [], [], []
    for i in range(k):
        r = rt.Renderer(0)
        if wpc:
            r.set_option("waves_per_cu", wpc)
        for kv in os.environ.get("RT_OPTIONS", "").split(","):  # e.g. RT_OPTIONS=view_lists=0
            if "=" in kv:
                r.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        r.upload(scene)
        s = torch.cuda.Stream(dev)
        r.set_stream(s)
        r.frame_begin()
        n = r.tile_count(0, wd) * 64
        bufs.append((torch.empty(n * 3, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.int32, device=dev),
                     torch.empty(6, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.int32, device=dev), n))
        ctxs.append(r); streams.append(s); evs.append(torch.cuda.Event())
    best = None
    for rep in range(3):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for f in range(FRAMES):
            i = f % k
            r, s = ctxs[i], streams[i]
            rgb, ids, mm, rgba, n = bufs[i]
            if mode == "stagger" and f > 0 and k > 1:
                s.wait_event(evs[(f - 1) % k])
            if f < k or os.environ.get("RT_SKIP_SETUP") != "1":  # (RT_SKIP_SETUP=1: the frames after the first reuse its grid and lists: what the set-up costs)
                r.frame_begin()
            if mode == "stagger":
                r.render_trace(0, wd)
                evs[i].record(s)
                r.render_shade(rgb, ids, mm)
            else:
                r.render(rgb, ids, mm, 0, wd)
            r.tonemap(rgb, n, mm, scene.gamma, rgba)
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) * 1e3 / FRAMES
        best = ms if best is None or ms < best else best
    for r in ctxs:
        r.close()
    print("world %d, %d frame(s) in flight, %s, waves per CU %d: %.3f ms per frame" % (wd, k, mode, wpc, best), flush=True)


for wd in [int(x) for x in os.environ.get("RT_WORLDS", "1,%d" % world).split(",")]:
    for k in [int(x) for x in os.environ.get("RT_DEPTHS", "1,2,3").split(",")]:
        for mode in os.environ.get("RT_MODES", "free,stagger").split(","):
            if k == 1 and mode == "stagger":
                continue
            for wpc in [int(x) for x in os.environ.get("RT_WPC", "0").split(",")]:
                run(wd, k, mode, wpc)
