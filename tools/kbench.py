#!/usr/bin/env python3
"""Kernel A/B helper: per-kernel-family device time of one frame for each library given on the command line.
usage: tools/kbench.py scene width height lib1.so [lib2.so ...]   (each library runs in its own process)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(scene, w, h, reps):
    sys.path.insert(0, ROOT)
    import raytracer_amd as rt
    from tools.diag import SCENES
    sc = rt.Scene(os.path.join(ROOT, SCENES[scene]), width=w, height=h)
    r = rt.Renderer(0, timing=1)
    if os.environ.get("RT_WAVES_PER_CU"):  # size of the persistent grids, for libraries built with another RT_WAVES
        r.set_option("waves_per_cu", int(os.environ["RT_WAVES_PER_CU"]))
    r.upload(sc)
    best = None
    for _ in range(reps):
        r.frame_begin()
        r.render_host()
        t = r.timings().asdict()
        if best is None or t["shadow_ms"] + t["nearest_ms"] < best["shadow_ms"] + best["nearest_ms"]:
            best = t
    print(json.dumps({k: round(v, 3) for k, v in best.items()}))


if __name__ == "__main__":
    if sys.argv[1] == "--one":
        one(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), 3)
    else:
        scene, w, h = sys.argv[1:4]
        for lib in sys.argv[4:]:
            env = dict(os.environ, RTGPU_LIB=os.path.abspath(lib))
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", scene, w, h], env=env, capture_output=True, text=True)
            print(os.path.basename(lib), out.stdout.strip() or out.stderr[-400:], flush=True)
