"""Random scenes in the reference's file formats (.brs .atr .lgt .cam .cfg), for parity tests beyond the bundled
scenes: a room of large triangles, boxes, a dense cluster of small triangles (one crowded grid cell), slivers,
glass panes and mirrors facing each other, lights inside the room, the observer outside or inside the grid."""
import os

import numpy as np


def _quad(tris, a, b, c, d, sid):
    tris.append((a, b, c, sid))
    tris.append((a, c, d, sid))


def _box(tris, lo, hi, sid):
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    p = [np.array(v, float) for v in ((x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0),
                                       (x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1))]
    for q in ((0, 1, 2, 3), (4, 5, 6, 7), (0, 1, 5, 4), (2, 3, 7, 6), (1, 2, 6, 5), (0, 3, 7, 4)):
        _quad(tris, p[q[0]], p[q[1]], p[q[2]], p[q[3]], sid)


def write_scene(dirpath, seed, width=64, height=48, opaque=False, hard_lights=False):
    """Writes <dirpath>/r.brs .atr .lgt .cam (and for some seeds .cfg); returns the prefix <dirpath>/r.
    opaque: no surface is transparent or reflective (the same geometry, lights and camera as without).
    hard_lights: the first lights are moved to where shadow rays are hardest for a table of directions around the light:
    into the floor's plane (every ray from a floor point lies in that plane), onto a corner of the room's geometry,
    into the crowded cluster."""
    rng = np.random.default_rng(seed)
    ext = float(rng.choice([3.0, 12.0, 60.0]))  # half extent of the room
    centre = rng.uniform(-1.0, 1.0, 3) * float(rng.choice([0.0, 10.0, 500.0]))
    tris = []
    # room: floor, back wall, one side wall (large triangles that span many cells)
    lo, hi = centre - ext, centre + ext
    _quad(tris, np.array([lo[0], lo[1], lo[2]]), np.array([hi[0], lo[1], lo[2]]), np.array([hi[0], lo[1], hi[2]]),
          np.array([lo[0], lo[1], hi[2]]), 0)
    _quad(tris, np.array([lo[0], lo[1], hi[2]]), np.array([hi[0], lo[1], hi[2]]), np.array([hi[0], hi[1], hi[2]]),
          np.array([lo[0], hi[1], hi[2]]), 1)
    _quad(tris, np.array([lo[0], lo[1], lo[2]]), np.array([lo[0], lo[1], hi[2]]), np.array([lo[0], hi[1], hi[2]]),
          np.array([lo[0], hi[1], lo[2]]), 4)
    # boxes standing on the floor
    for _ in range(int(rng.integers(2, 6))):
        c = centre + rng.uniform(-0.7, 0.7, 3) * ext
        s = rng.uniform(0.05, 0.3, 3) * ext
        _box(tris, (c[0] - s[0], lo[1], c[2] - s[2]), (c[0] + s[0], lo[1] + 2 * s[1], c[2] + s[2]), int(rng.integers(0, 7)))
    # a dense cluster of small triangles: hundreds of them in one or two grid cells
    cc = centre + rng.uniform(-0.5, 0.5, 3) * ext
    for _ in range(int(rng.integers(150, 700))):
        p = cc + rng.standard_normal(3) * 0.04 * ext
        e1, e2 = rng.standard_normal(3) * 0.02 * ext, rng.standard_normal(3) * 0.02 * ext
        tris.append((p, p + e1, p + e2, int(rng.integers(0, 7))))
    # slivers: long and thin, some nearly axis-parallel
    for _ in range(int(rng.integers(5, 30))):
        p = centre + rng.uniform(-0.8, 0.8, 3) * ext
        d = rng.standard_normal(3)
        if rng.random() < 0.5:
            d[int(rng.integers(0, 3))] = 0.0
        d = d / max(np.linalg.norm(d), 1e-9) * ext * rng.uniform(0.2, 1.0)
        w = rng.standard_normal(3) * ext * 10.0 ** rng.uniform(-4, -2)
        tris.append((p, p + d, p + 0.5 * d + w, int(rng.integers(0, 7))))
    # glass panes and mirrors, some facing each other
    for _ in range(int(rng.integers(2, 6))):
        c = centre + rng.uniform(-0.6, 0.6, 3) * ext
        u = rng.standard_normal(3); u /= np.linalg.norm(u)
        v = np.cross(u, rng.standard_normal(3)); v /= max(np.linalg.norm(v), 1e-9)
        su, sv = rng.uniform(0.1, 0.5, 2) * ext
        sid = int(rng.choice([2, 3, 5]))
        _quad(tris, c - u * su - v * sv, c + u * su - v * sv, c + u * su + v * sv, c - u * su + v * sv, sid)
        if rng.random() < 0.5:  # a parallel partner a little further along the normal
            n = np.cross(u, v) * ext * rng.uniform(0.05, 0.3)
            _quad(tris, c + n - u * su - v * sv, c + n + u * su - v * sv, c + n + u * su + v * sv, c + n - u * su + v * sv,
                  int(rng.choice([2, 3, 5])))
    order = rng.permutation(len(tris))  # list order is the tie-break and the order of the transparency product
    tris = [tris[i] for i in order]
    fmt = (lambda x: "%.6f" % x) if seed % 2 == 0 else (lambda x: "%.9g" % np.float32(x))
    prefix = os.path.join(dirpath, "r")
    with open(prefix + ".brs", "w") as f:
        f.write("// random scene, seed %d\n%d\n" % (seed, 3 * len(tris)))
        for t in tris:
            for p in t[:3]:
                f.write("%s %s %s\n" % tuple(fmt(x) for x in p))
        f.write("%d\n" % len(tris))
        for i in range(len(tris)):
            f.write("%d %d %d\n" % (3 * i, 3 * i + 1, 3 * i + 2))
        sid = [t[3] for t in tris]
        for i in range(0, len(sid), 16):
            f.write(" ".join(str(s) for s in sid[i:i + 16]) + "\n")
    # kd ks g ka R G B kt eta kr
    surf = [(0.8, 0.0, 0, 0.0, 200, 180, 160, 0.0, 1.0, 0.0), (0.6, 0.4, 20, 0.02, 90, 120, 200, 0.0, 1.0, 0.0),
            (0.3, 0.3, 30, 0.0, 220, 220, 220, 0.0, 1.0, 0.6), (0.1, 0.2, 40, 0.0, 180, 255, 200, 0.7, 1.4, 0.1),
            (0.5, 0.0, 0, 0.1, 1, 0.5, 0.25, 0.0, 1.0, 0.0), (0.2, 0.0, 0, 0.0, 255, 200, 120, 0.4, 1.1, 0.0),
            (0.4, 0.5, 8, 0.0, 40, 40, 60, 0.0, 1.0, 0.2)]
    with open(prefix + ".atr", "w") as f:
        f.write("%d\n" % len(surf))
        for s in surf:
            vals = [x * (rng.uniform(0.8, 1.0) if k in (0, 1, 7, 9) else 1.0) for k, x in enumerate(s)]
            if opaque:
                vals[7] = 0.0; vals[9] = 0.0
            f.write(" ".join("%g" % x for x in vals) + "\n")
    with open(prefix + ".lgt", "w") as f:
        nl = int(rng.integers(1, 5))
        f.write("%d\n" % nl)
        for li in range(nl):
            p = centre + rng.uniform(-0.9, 0.9, 3) * ext
            p[1] = centre[1] + rng.uniform(0.2, 0.95) * ext
            if hard_lights and li == 0:
                p[1] = lo[1]                      # in the floor's plane
            elif hard_lights and li == 1:
                p = np.array([lo[0], lo[1], hi[2]])  # the corner where floor, back wall and side wall meet
            elif hard_lights and li == 2:
                p = cc.copy()                     # inside the crowded cluster
            f.write("%s %s %s %g %g %g %g\n" % (fmt(p[0]), fmt(p[1]), fmt(p[2]), ext * ext * rng.uniform(50, 500),
                                                 rng.uniform(0, 1), rng.uniform(0, 1), rng.uniform(0, 1)))
    if seed % 6 == 0:  # a planar light under the ceiling: 16 rand() samples per shaded hit, drawn in the reference's pixel order
        a = centre + np.array([rng.uniform(-0.5, 0.2), rng.uniform(0.6, 0.9), rng.uniform(-0.5, 0.2)]) * ext
        with open(prefix + ".pnr", "w") as f:
            f.write("1\n%g %g %g %g\n" % (ext * ext * rng.uniform(50, 300), rng.uniform(0, 1), rng.uniform(0, 1), rng.uniform(0, 1)))
            for p in (a, a + np.array([0.3 * ext, 0.0, 0.0]), a + np.array([0.0, 0.0, 0.3 * ext])):
                f.write("%s %s %s\n" % tuple(fmt(x) for x in p))
    # the observer: outside the room looking in, or inside it
    inside = rng.random() < 0.4
    ob = centre + (rng.uniform(-0.5, 0.5, 3) * ext if inside else np.array([rng.uniform(-0.5, 0.5), rng.uniform(0.1, 0.8), -rng.uniform(1.5, 3.0)]) * ext)
    target = centre + rng.uniform(-0.3, 0.3, 3) * ext
    fw = target - ob; fw /= max(np.linalg.norm(fw), 1e-9)
    right = np.cross(np.array([0.0, 1.0, 0.0]), fw); right /= max(np.linalg.norm(right), 1e-9)
    up = np.cross(fw, right)
    dist, hw = 0.1 * ext, 0.1 * ext * rng.uniform(0.5, 1.2)
    hh = hw * height / width
    mid = ob + fw * dist
    with open(prefix + ".cam", "w") as f:
        for p in (ob, mid - right * hw + up * hh, mid - right * hw - up * hh, mid + right * hw + up * hh):
            f.write("%s %s %s\n" % tuple(fmt(x) for x in p))
        f.write("%d %d\n" % (width, height))
    mode = seed % 4
    if mode == 1:
        open(prefix + ".cfg", "w").write("voxmode FIXED voxparams %d %d %d\n" % tuple(int(x) for x in rng.integers(1, 24, 3)))
    elif mode == 2:
        open(prefix + ".cfg", "w").write("gamma 2.0 voxmode MODIFIED_DEFAULT voxparams %g %g %g\n" % tuple(rng.uniform(0.4, 2.5, 3)))
    return prefix
