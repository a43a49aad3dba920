import os, sys, json
sys.path.insert(0, os.getcwd())
import raytracer_amd as rt
sc = rt.Scene("scenes/ulica/ulica", width=1920, height=1080)
for opts in ({}, {"order": 2}, {"order": 2, "order_parts": 4}, {"record_test": 0}, {"bundle_ends": 0}):
    r = rt.Renderer(0, timing=1)
    for k, v in opts.items(): r.set_option(k, v)
    r.upload(sc)
    best = None
    for _ in range(3):
        r.frame_begin(); r.render_host(); t = r.timings().asdict()
        if best is None or t["shadow_ms"] < best["shadow_ms"]: best = t
    print(opts, {k: round(v, 2) for k, v in best.items() if k in ("nearest_ms", "shadow_ms", "hit_ms", "accumulate_ms")}, flush=True)
    r.close()
