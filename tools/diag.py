#!/usr/bin/env python3
"""Prints the ray statistics of one frame (counting kernel variants) as JSON: rays, S/T, SIMD efficiency.
usage: tools/diag.py [scene] [width height]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# (the measuring tools, which import this module for its scene table, want a scene's light lists from its first frame on)
os.environ.setdefault("RTGPU_LIGHT_LISTS_AFTER", "0")
import raytracer_amd as rt  # noqa: E402

SCENES = {"temple": "scenes/temple/temple", "ulica": "scenes/ulica/ulica", "s3": "scenes/s3/s3", "s5": "scenes/s5/s5",
          "pokoj": "scenes/pokoj/s2", "s2": "scenes/s2/s2"}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "temple"
    w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (None, None)
    scene = rt.Scene(os.path.join(ROOT, SCENES[name]), width=w, height=h)
    r = rt.Renderer(0, count=1, timing=1)
    r.upload(scene)
    r.frame_begin()
    t0 = time.time()
    r.render_host()
    c = r.counters().asdict()
    t = r.timings().asdict()
    c["nearest_simd_eff"] = c["nearest_lane_tests"] / max(c["nearest_lane_slots"], 1)
    c["shadow_simd_eff"] = c["shadow_lane_tests"] / max(c["shadow_lane_slots"], 1)
    c["rays"] = c["nearest_queries"] + c["shadow_queries"]
    print(json.dumps({"scene": name, "width": scene.width, "height": scene.height, "counters": c, "timings_ms": t,
                      "wall_s": time.time() - t0}))


if __name__ == "__main__":
    main()
