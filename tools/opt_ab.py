#!/usr/bin/env python3
"""Per-kernel device time of one frame under several option settings, each in a fresh context.
usage: tools/opt_ab.py scene w h 'opt=val,opt=val' ['...' ...]     ('' = the defaults)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt  # noqa: E402
from tools.diag import SCENES  # noqa: E402

scene, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sc = rt.Scene(os.path.join(ROOT, SCENES[scene]), width=w, height=h)
for spec in sys.argv[4:] or [""]:
    opts = dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in spec.split(",") if kv)
    r = rt.Renderer(0, timing=1)
    for k, v in opts.items():
        r.set_option(k, v)
    r.upload(sc)
    best = None
    for _ in range(int(os.environ.get("AB_FRAMES", "20"))):
        r.frame_begin(); r.render_host(); t = r.timings().asdict()
        if best is None or t["shadow_ms"] + t["nearest_ms"] < best["shadow_ms"] + best["nearest_ms"]:
            best = t
    r.counters()
    r.frame_begin(); r.render_host()
    cn = r.counters().asdict()
    print(spec or "defaults", {k: round(v, 3) for k, v in best.items() if k.endswith("_ms")},
          {k: cn[k] for k in ("shadow_list_tickets", "shadow_list_fallbacks", "view_list_tickets", "view_list_fallbacks")}, flush=True)
    r.close()
