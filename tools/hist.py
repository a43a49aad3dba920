#!/usr/bin/env python3
"""Of the batches whose records are tested: how many records had a lane pass the first slab, and how many (lane, record)
pairs that makes.  Needs a library built with -DRT_HIST (RTGPU_LIB).  usage: tools/hist.py [scene w h]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt
from tools.diag import SCENES

scene = sys.argv[1] if len(sys.argv) > 1 else "temple"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
h = int(sys.argv[3]) if len(sys.argv) > 3 else 2160
sc = rt.Scene(os.path.join(ROOT, SCENES[scene]), width=w, height=h)
r = rt.Renderer(0)
r.upload(sc)
r.set_option("count", 1)  # the per-mode statistics come from the counting variants of the kernels
r.frame_begin(); r.render_host()
out = (C.c_ulonglong * 92)()
L = rt.gpu_lib()
L.rtgpu_debug_hist.argtypes = [C.c_void_p, C.c_void_p]
L.rtgpu_debug_hist(r._h, out)
v = list(out)
for k, name in enumerate(("nearest", "shadow")):
    b, p = v[18 * k:18 * k + 9], v[18 * k + 9:18 * k + 18]
    left, p[0] = p[0], 0
    tb, tp = sum(b) or 1, sum(p) or 1
    print(name, "batches tested", tb, "pairs", tp, "records left by the bundle's slab test: %.2f per batch" % (left / tb))
    for i in range(9):
        print("  %d records passed by some lane: %5.1f %% of batches, %5.1f %% of pairs, %5.1f pairs per batch" % (i, 100.0 * b[i] / tb, 100.0 * p[i] / tp, p[i] / max(b[i], 1)))
    c = v[36 + 4 * k:40 + 4 * k]
    print("  headers looked at %d, passed the bundle test %d (%.1f %%), kept by the ray-by-ray test %d (%.1f %% of headers, %.1f %% of candidates)"
          % (c[0], c[1], 100.0 * c[1] / max(c[0], 1), c[2], 100.0 * c[2] / max(c[0], 1), 100.0 * c[2] / max(c[1], 1)))
    print("  rays that may meet the box of a kept batch: %.2f on average" % (c[3] / max(c[2], 1)))
    m = v[44 + 12 * k:56 + 12 * k]
    for i, mode in enumerate(("cell", "lane", "ray")):
        vis, wv, ln, rays = m[4 * i:4 * i + 4]
        print("  %s mode: %d visits (%.1f rays per visit), %d wave-level record tests (%.1f per visit), lane use %.3f"
              % (mode, vis, rays / max(vis, 1), wv, wv / max(vis, 1), ln / max(64 * wv, 1)))
    q = v[68 + 12 * k:80 + 12 * k]
    print("  header chunks staged %d: %.1f %% without a candidate, %.1f %% without a batch kept by the box tests, %.1f %% without a batch left by the record-level test"
          % (q[0], 100.0 * q[1] / max(q[0], 1), 100.0 * q[2] / max(q[0], 1), 100.0 * q[3] / max(q[0], 1)))
    for i, cls in enumerate(("1 chunk", "2 chunks", "3-4 chunks", "5+ chunks")):
        print("  lists of %s: %d cell-mode visits, %d headers" % (cls, q[4 + 2 * i], q[5 + 2 * i]))
