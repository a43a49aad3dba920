#!/usr/bin/env python3
"""Many generated scenes (tests/scenegen.py), each rendered with the light and view lists and with the walks alone: the
frames must be equal bit for bit.  usage: tools/soak.py [first_seed [count [width height]]]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RTGPU_LIGHT_LISTS_AFTER", "0")
import raytracer_amd as rt  # noqa: E402
from tests.scenegen import write_scene  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 64
w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (320, 240)
r = rt.Renderer(0)
bad = 0
tot = {"shadow_list_tickets": 0, "shadow_list_fallbacks": 0, "view_list_tickets": 0, "view_list_fallbacks": 0}
with tempfile.TemporaryDirectory() as tmp:
    for seed in range(first, first + count):
        d = os.path.join(tmp, "s%d" % seed)
        os.makedirs(d)
        prefix = write_scene(d, seed, w, h, opaque=seed % 2 == 0, hard_lights=seed % 3 != 0)
        scene = rt.Scene(prefix)
        frames = []
        for lists in (1, 0):
            r.set_option("light_lists", lists); r.set_option("view_lists", lists)
            r.upload(scene); r.frame_begin(); r.counters()
            frames.append(r.render_host())
            if lists:
                c = r.counters().asdict()
                for k in tot:
                    tot[k] += c[k]
        same = np.array_equal(frames[0]["ids"], frames[1]["ids"]) and np.array_equal(frames[0]["rgb"].view(np.uint32), frames[1]["rgb"].view(np.uint32))
        if not same:
            bad += 1
            print("seed %d: frames differ" % seed, flush=True)
r.close()
print("%d scenes from seed %d at %dx%d: %d differ; %s" % (count, first, w, h, bad, tot), flush=True)
sys.exit(1 if bad else 0)
