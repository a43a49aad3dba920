#!/usr/bin/env python3
"""Share of the walk kernels' wave time per phase, from a library built with tools/phases.sh.
usage: python tools/phases.py [scene w h]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("RTGPU_LIB", os.path.join(ROOT, "tools/_variants/librtgpu_phases.so"))
sys.path.insert(0, ROOT)
import raytracer_amd as rt
from tools.diag import SCENES

NAMES = ["ray fetch, set-up, result", "group and mode of a visit", "header staging, batch_cos", "batch culling",
         "record staging", "triangle tests", "stepping to next cell", "lane / ray mode"]


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "temple"
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
    h = int(sys.argv[3]) if len(sys.argv) > 3 else 2160
    sc = rt.Scene(os.path.join(ROOT, SCENES[scene]), width=w, height=h)
    r = rt.Renderer(0, timing=1)
    r.upload(sc)
    L = rt.gpu_lib()
    L.rtgpu_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int]
    out = (C.c_int64 * 16)()
    r.frame_begin(); r.render_host()
    L.rtgpu_phase_cycles(r._h, out, 1)
    r.frame_begin(); r.render_host()
    t = r.timings().asdict()
    L.rtgpu_phase_cycles(r._h, out, 1)
    for k, name in enumerate(("nearest", "shadow")):
        v = list(out[8 * k:8 * k + 8])
        tot = sum(v) or 1
        print("%s walk: %.3f ms (instrumented build), %.3e wave-cycles" % (name, t[name + "_ms"], tot))
        for n, x in zip(NAMES, v):
            print("  %-28s %5.1f %%" % (n, 100.0 * x / tot))


if __name__ == "__main__":
    main()
