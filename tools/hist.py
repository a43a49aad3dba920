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
r.frame_begin(); r.render_host()
out = (C.c_ulonglong * 36)()
L = rt.gpu_lib()
L.rtgpu_debug_hist.argtypes = [C.c_void_p, C.c_void_p]
L.rtgpu_debug_hist(r._h, out)
v = list(out)
for k, name in enumerate(("nearest", "shadow")):
    b, p = v[18 * k:18 * k + 9], v[18 * k + 9:18 * k + 18]
    tb, tp = sum(b) or 1, sum(p) or 1
    print(name, "batches tested", tb, "pairs", tp)
    for i in range(9):
        print("  %d records passed by some lane: %5.1f %% of batches, %5.1f %% of pairs, %5.1f pairs per batch" % (i, 100.0 * b[i] / tb, 100.0 * p[i] / tp, p[i] / max(b[i], 1)))
