"""CPU estimate of how many batches of a cell's list survive different bounding volumes.

For a sample of primary and shadow rays of a scene (one ray per 8x8 tile of the full-size image) and every
cell a ray's segment crosses, counts the batches (groups of B consecutive records of the cell's Morton-ordered
list, as k_grid_sort orders them) whose bound the ray meets: bounding sphere, axis-aligned box, and the ideal
(some triangle of the batch has its own box met).  Geometry only: no rounding margins.

usage: python tools/cull_sim.py [scene [w h [batch]]]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt
from oracle import oracle as O


def morton10(x):
    x = x & 0x3ff
    x = (x | (x << 16)) & 0x30000ff
    x = (x | (x << 8)) & 0x300f00f
    x = (x | (x << 4)) & 0x30c30c3
    x = (x | (x << 2)) & 0x9249249
    return x


def seg_box(o, d, tmax, lo, hi):
    """o, d [n,3], tmax [n]; lo, hi [m,3] -> bool [n,m]: segment o + t d, t in [0,tmax], meets the box."""
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
    t0 = (lo[None] - o[:, None]) * inv[:, None]
    t1 = (hi[None] - o[:, None]) * inv[:, None]
    tn = np.nanmax(np.minimum(t0, t1), axis=2)
    tf = np.nanmin(np.maximum(t0, t1), axis=2)
    return (tn <= tf) & (tf >= 0) & (tn <= tmax[:, None])


def seg_sphere(o, d, tmax, c, r):
    oc = c[None] - o[:, None]
    tc = np.einsum("nmk,nk->nm", oc, d)
    tcl = np.clip(tc, 0, tmax[:, None])
    p = oc - tcl[..., None] * d[:, None]
    return np.einsum("nmk,nmk->nm", p, p) <= r[None] ** 2


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "temple"
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
    h = int(sys.argv[3]) if len(sys.argv) > 3 else 2160
    B = int(sys.argv[4]) if len(sys.argv) > 4 else 16
    prefix = os.path.join(ROOT, "scenes/%s/%s" % (name, name))
    os.chdir(ROOT)
    tw, th = w // 8, h // 8
    sc = rt.Scene(prefix, width=tw, height=th)
    osc = O.OracleScene(prefix, width=tw, height=th)
    fr = osc.render(shadow_cache=False)
    g = osc.grid()
    V = osc.vertices()[0].astype(np.float64).reshape(-1, 3, 3)
    nv, cs, dmin = g["nv"], g["cs"].astype(np.float64), g["dmin"].astype(np.float64)
    start, refs = g["cell_start"], g["refs"]
    cam = sc.camera
    ob, ul, bl, ur = [np.array(list(x), np.float64) for x in (cam.ob, cam.ul, cam.bl, cam.ur)]
    ys, xs = np.mgrid[0:th, 0:tw]
    P = ul + (ur - ul) * ((xs[..., None] + 0.5) / tw) + (bl - ul) * ((ys[..., None] + 0.5) / th)
    D = P - ob
    D /= np.linalg.norm(D, axis=-1, keepdims=True)
    ids = fr["ids"]
    hit = ids >= 0
    D = D[hit]; T = V[ids[hit]]
    n = np.cross(T[:, 1] - T[:, 0], T[:, 2] - T[:, 0])
    dist = np.einsum("nk,nk->n", T[:, 0] - ob, n) / np.einsum("nk,nk->n", D, n)
    hp = ob + dist[:, None] * D
    lights = np.array([[sc.desc.lights[i].p[k] for k in range(3)] for i in range(sc.nl)], np.float64)
    rays = [("primary", np.repeat(ob[None], len(D), 0), D, dist)]
    so, sd, st = [], [], []
    for L in lights:
        v = L - hp
        ln = np.linalg.norm(v, axis=1)
        facing = np.einsum("nk,nk->n", v, n) * np.sign(-np.einsum("nk,nk->n", D, n)) > 0
        so.append(hp[facing]); sd.append((v / ln[:, None])[facing]); st.append(ln[facing])
    rays.append(("shadow", np.concatenate(so), np.concatenate(sd), np.concatenate(st)))

    # per cell: Morton order, batches, bounds
    cells = [c for c in range(len(start) - 1) if start[c + 1] > start[c]]
    cell_ijk = np.array([[c // (nv[2] * nv[1]), (c // nv[2]) % nv[1], c % nv[2]] for c in cells])
    cell_lo = dmin + cell_ijk * cs
    cell_hi = cell_lo + cs
    big = 0.5 * cs.max()
    batches = []
    for ci, c in enumerate(cells):
        t = refs[start[c]:start[c + 1]]
        tv = V[t]
        if 32 < len(t) <= 4096:
            ext = (tv.max(1) - tv.min(1)).max(1)
            rel = np.clip((tv.mean(1) - cell_lo[ci]) / cs * 1024, 0, 1023).astype(np.int64)
            code = (morton10(rel[:, 0]) << 2) | (morton10(rel[:, 1]) << 1) | morton10(rel[:, 2])
            key = ((ext <= big).astype(np.int64) << 30 | code) << 32 | np.arange(len(t))
            tv = tv[np.argsort(key)]
        nb = (len(t) + B - 1) // B
        lo = np.array([tv[b * B:(b + 1) * B].min((0, 1)) for b in range(nb)])
        hi = np.array([tv[b * B:(b + 1) * B].max((0, 1)) for b in range(nb)])
        ctr = 0.5 * (lo + hi)
        rad = np.array([np.linalg.norm(tv[b * B:(b + 1) * B] - ctr[b], axis=2).max() for b in range(nb)])
        cnt = np.array([len(tv[b * B:(b + 1) * B]) for b in range(nb)])
        tlo, thi = tv.min(1), tv.max(1)
        batches.append((lo, hi, ctr, rad, cnt, tlo, thi))

    for kind, o, d, tmax in rays:
        tot = dict(rays=len(o), visits=0, batches=0, listed=0, sphere=0, box=0, both=0, ideal_batches=0, ideal_tris=0,
                   sphere_recs=0, box_recs=0)
        for s0 in range(0, len(o), 20000):
            oo, dd, tt = o[s0:s0 + 20000], d[s0:s0 + 20000], tmax[s0:s0 + 20000]
            crossing = seg_box(oo, dd, tt, cell_lo, cell_hi)
            for ci in np.nonzero(crossing.any(0))[0]:
                r = np.nonzero(crossing[:, ci])[0]
                lo, hi, ctr, rad, cnt, tlo, thi = batches[ci]
                sp = seg_sphere(oo[r], dd[r], tt[r], ctr, rad)
                bx = seg_box(oo[r], dd[r], tt[r], lo, hi)
                tb = seg_box(oo[r], dd[r], tt[r], tlo, thi)
                idl = np.add.reduceat(tb, np.arange(0, len(tlo), B), axis=1) > 0
                tot["visits"] += len(r); tot["batches"] += len(r) * len(cnt); tot["listed"] += len(r) * cnt.sum()
                tot["sphere"] += sp.sum(); tot["box"] += bx.sum(); tot["both"] += (sp & bx).sum()
                tot["sphere_recs"] += (sp * cnt[None]).sum(); tot["box_recs"] += (bx * cnt[None]).sum()
                tot["ideal_batches"] += idl.sum(); tot["ideal_tris"] += tb.sum()
        print(kind, "B =", B)
        for k, v in tot.items():
            print("  %-14s %12d  per ray %8.2f" % (k, v, v / tot["rays"]))
        print("  survive: sphere %.3f  box %.3f  both %.3f  ideal %.3f" % tuple(tot[k] / tot["batches"] for k in ("sphere", "box", "both", "ideal_batches")))


if __name__ == "__main__":
    main()
