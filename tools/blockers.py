#!/usr/bin/env python3
"""How coherent are the blockers of the shadow rays of one 8x8 tile?  Needs a library built with -DRT_HIST (the shadow
walk then records which triangle blocked a ray).  usage: RTGPU_LIB=... tools/blockers.py [scene w h]"""
import ctypes as C
import os
import sys

import numpy as np

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
r.frame_begin()
g = r.render_host()
n = r.tile_count() * 64
ns = sc.nl
ts = np.zeros((ns, n), np.float32)
L = rt.gpu_lib()
L.rtgpu_debug_ts.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int]
assert L.rtgpu_debug_ts(r._h, ts.ctypes.data, n, ns) == 0
ts = ts.reshape(ns, n // 64, 64)
tw = (w + 7) // 8
# ids tile-major, to know which lanes had a shadow query at all
ids = g["ids"]
pad = np.full(((h + 7) // 8 * 8, tw * 8), -1, np.int32); pad[:h, :w] = ids
hit = pad.reshape(-1, 8, tw, 8).swapaxes(1, 2).reshape(-1, 64) >= 0
for s in range(ns):
    t = ts[s]
    q = hit                               # lanes with a query
    walk_blocked = (t <= -2) & q          # blocked by a triangle found in the walk
    early = (t == -1) & q                 # back-face early-out (or a blocker found in lane / ray mode)
    lit = (t >= 0) & q
    tiles_q = q.any(1)
    all_dark = tiles_q & ~(lit.any(1))
    blk = np.where(walk_blocked, (-t - 2).astype(np.int64), -1)
    # the most frequent blocker of a tile and how many of its walk-blocked lanes it accounts for
    same = 0
    for i in np.nonzero(walk_blocked.any(1))[0][::37]:
        v = blk[i][blk[i] >= 0]
        same += np.bincount(v).max() / len(v)
    cnt = len(np.nonzero(walk_blocked.any(1))[0][::37])
    print("light %d: queries %d, lit %.3f, back-face %.3f, blocked in the walk %.3f; tiles with queries %d, of them without a lit lane %.3f; "
          "share of a tile's blocked lanes that its most frequent blocker accounts for %.3f"
          % (s, q.sum(), lit.sum() / q.sum(), early.sum() / q.sum(), walk_blocked.sum() / q.sum(), tiles_q.sum(), all_dark.sum() / max(tiles_q.sum(), 1), same / max(cnt, 1)))

# With a -DRT_TICKETS build as well: what the tickets without a lit lane cost (shader clocks), per light 0 and 1
L.rtgpu_ticket_log.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
buf = np.zeros((1 << 18, 12), np.uint32)
got = C.c_int(0)
if L.rtgpu_ticket_log(r._h, 1, buf.ctypes.data, buf.shape[0], C.byref(got)) == 0 and got.value > 0:
    ntick = n // 64
    for s in range(min(ns, (1 << 18) // ntick)):
        cyc = buf[s * ntick:(s + 1) * ntick, 0].astype(np.int64)
        t = ts[s]
        q = hit
        lit = ((t >= 0) & q).any(1)
        has = q.any(1)
        walked = ((t <= -2) & q).any(1) | lit
        dark = has & ~lit
        one = np.zeros(ntick, bool)   # dark, and one triangle accounts for every lane blocked in the walk
        for i in np.nonzero(dark)[0]:
            v = t[i][(t[i] <= -2) & q[i]]
            one[i] = len(v) > 0 and (v == v[0]).all()
        tot = cyc.sum()
        print("light %d: clocks in tickets: without queries %.3f, dark %.3f (of them with a single blocker %.3f), with a lit lane %.3f; mean clocks dark %.0f, lit %.0f"
              % (s, cyc[~has].sum() / tot, cyc[dark].sum() / tot, cyc[one].sum() / tot, cyc[lit].sum() / tot, cyc[dark].mean(), cyc[lit].mean()))
