"""The light lists' completeness, on the CPU: every triangle the reference's test (src/intersection.c:58-86, restated in
binary32 in test_filter_cpu.py) accepts as a blocker of a shadow ray (src/raytrace.c:206-216: origin at the hit point,
direction normalised towards the light, src/vectormath.h:120-128) is listed in the texel the kernel looks the ray up in.
The rectangles come from the library's own host code (rtgpu_light_list_pairs, rtgpu_light_list_rects: no GPU needed);
the texel of a ray is worked out here with the kernel's binary32 expressions."""
import ctypes as C
import os

import numpy as np
import pytest

import raytracer_amd as rt
from test_filter_cpu import F, aim_anywhere, aim_edges, aim_inside, exact_test

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    L = rt.gpu_lib()
    L.rtgpu_light_list_pairs.restype = C.c_int
    L.rtgpu_light_list_pairs.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    L.rtgpu_light_list_rects.restype = C.c_int
    L.rtgpu_light_list_rects.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_float, C.c_int64, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return L


def shadow_ray(org, light):
    """vectormath.h:120-128 and dist3: direction and dmax in binary32."""
    rr = (light - org).astype(F)
    s2 = rr[:, 0] * rr[:, 0] + rr[:, 1] * rr[:, 1] + rr[:, 2] * rr[:, 2]
    with np.errstate(all="ignore"):
        inv = (1.0 / np.sqrt(s2.astype(np.float64))).astype(F)
    return rr * inv[:, None], np.sqrt(s2.astype(np.float64)).astype(F)


FACE_EXT = 1.25  # the faces of the cube overlap: a face's coordinates run over [-1.25, 1.25] (rt_lightlists.h)


def face_coords(org, light, axis=None):
    """The kernel's texel coordinates of a ray on the face of axis `axis` (None: the dominant axis of fl(a - P)) and of the
    sign d has there: (p, q) = the other two components in ascending order over that component's magnitude, binary32.
    on: the ray can be looked up on that face (both coordinates within FACE_EXT)."""
    d = (org - light).astype(F)
    ad = np.abs(d)
    ax = np.where((ad[:, 0] >= ad[:, 1]) & (ad[:, 0] >= ad[:, 2]), 0, np.where(ad[:, 1] >= ad[:, 2], 1, 2))
    if axis is not None:
        ax = np.full(len(d), axis)
    rows = np.arange(len(d))
    w = ad[rows, ax]
    o0 = np.where(ax == 0, 1, 0); o1 = np.where(ax == 2, 1, 2)
    with np.errstate(all="ignore"):
        p = (d[rows, o0] / w).astype(F); q = (d[rows, o1] / w).astype(F)
        on = (np.abs(p) <= F(FACE_EXT)) & (np.abs(q) <= F(FACE_EXT))
    face = ax * 2 + (d[rows, ax] < 0)
    return face, p, q, on


def make_light_cases(rng, n, scale, offset, aim, light_mode):
    """Triangles of size ~scale around `offset`, a light, and an origin on the line from the light through a point of the
    triangle's plane chosen by `aim`, beyond that point."""
    a = (rng.standard_normal((n, 3)) * scale + offset).astype(F)
    ab = (rng.standard_normal((n, 3)) * scale).astype(F)
    ac = (rng.standard_normal((n, 3)) * scale).astype(F)
    sliver = rng.random(n) < 0.15
    ac = np.where(sliver[:, None], (ab * (rng.random((n, 1)) * 2 + 0.1).astype(F) + ac * F(1e-3)).astype(F), ac)
    j = (a + ab).astype(F); k = (a + ac).astype(F)
    ab = (j - a).astype(F); ac = (k - a).astype(F)  # the edges as k_preprocess forms them
    u, v = aim(rng, n)
    A64, AB64, AC64 = a.astype(np.float64), ab.astype(np.float64), ac.astype(np.float64)
    target = A64 + u[:, None] * AB64 + v[:, None] * AC64
    nrm = np.cross(AB64, AC64)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-300)
    dirn = rng.standard_normal((n, 3)); dirn /= np.linalg.norm(dirn, axis=1, keepdims=True)
    dist = scale * 10.0 ** rng.uniform(-2, 1.5, n)
    if light_mode == "near_plane":  # the light a hair off the triangle's plane: every ray to it grazes the triangle
        inplane = dirn - nrm * np.einsum("ij,ij->i", dirn, nrm)[:, None]
        inplane /= np.maximum(np.linalg.norm(inplane, axis=1, keepdims=True), 1e-300)
        hgt = scale * 10.0 ** rng.uniform(-9, -2, n) * rng.choice([-1.0, 1.0], n)
        light = target + inplane * dist[:, None] + nrm * hgt[:, None]
    elif light_mode == "near_triangle":
        light = target + dirn * (scale * 10.0 ** rng.uniform(-7, -1, n))[:, None]
    else:
        light = target + dirn * dist[:, None]
    light = light.astype(F)
    back = target - light.astype(np.float64)
    bl = np.maximum(np.linalg.norm(back, axis=1, keepdims=True), 1e-300)
    beyond = scale * 10.0 ** rng.uniform(-4, 1.5, n)
    o = (target + back / bl * beyond[:, None]).astype(F)
    return a, j, k, ab, ac, o, light


@pytest.mark.parametrize("scale,offset", [(1.0, 0.0), (0.01, 50.0), (20.0, 300.0), (1e-3, 0.0), (5.0, -1000.0)])
@pytest.mark.parametrize("aim", [aim_inside, aim_edges, aim_anywhere], ids=["inside", "edges", "anywhere"])
@pytest.mark.parametrize("light_mode", ["generic", "near_plane", "near_triangle"])
def test_every_blocker_lies_in_its_triangles_rectangle(scale, offset, aim, light_mode):
    rng = np.random.default_rng((hash((scale, offset, aim.__name__, light_mode)) & 0xffff) + 11)
    n = 120000
    a, j, k, ab, ac, o, light = make_light_cases(rng, n, scale, offset, aim, light_mode)
    r, dmax = shadow_ray(o, light)
    acc, _, d, _, _, _ = exact_test(a, ab, ac, o, r)
    with np.errstate(all="ignore"):
        blocker = acc & (d > F(0.00001)) & (d < dmax)
    if aim is not aim_anywhere and light_mode == "generic" and scale >= 0.01:  # (tiny triangles fail the test's |det| >= 1e-6)
        assert blocker.mean() > 0.2
    # rectangles for origins up to olim: in chunks of similar |o|, so that the tolerance is the one the ray's own norm gives
    onorm = np.sqrt((o.astype(np.float64) ** 2).sum(1))
    order = np.argsort(onorm)
    verts = np.concatenate([a, j, k], 1).astype(F)
    out = np.zeros((n, 6, 5), np.float64)
    L = _lib()
    for chunk in np.array_split(order, 24):
        vv = np.ascontiguousarray(verts[chunk]); ll = np.ascontiguousarray(light[chunk]); oo = np.zeros((len(chunk), 6, 5), np.float64)
        olim = float(onorm[chunk].max()) * 1.0002 + 1e-30
        assert L.rtgpu_light_list_pairs(len(chunk), vv.ctypes.data, ll.ctypes.data, C.c_float(olim), oo.ctypes.data) == 0
        out[chunk] = oo
    rows = np.arange(n)
    looked_up = 0
    for axis in (None, 0, 1, 2):  # the ray's own face, and every face it can be looked up on from a neighbour's side
        face, p, q, on = face_coords(o, light, axis)
        rc = out[rows, face]
        inside = (rc[:, 0] > 0) & (p >= rc[:, 1]) & (p <= rc[:, 2]) & (q >= rc[:, 3]) & (q <= rc[:, 4])
        if axis is None:
            assert on[np.isfinite(p) & np.isfinite(q)].all()  # (a ray whose origin is the light has no direction)
        missed = blocker & on & ~inside
        looked_up += int((blocker & on).sum())
        assert not missed.any(), "%d blockers outside their triangle's rectangle on axis %s, e.g. case %d" % (missed.sum(), axis, np.flatnonzero(missed)[0])
        if axis is None and aim is aim_anywhere and light_mode == "generic":
            # the rectangles are not everything: most rays that miss are outside
            assert inside[~acc].mean() < 0.6
    assert looked_up > 1.15 * blocker.sum() or blocker.sum() == 0  # (some rays do lie in the overlap of two faces)


def _scene_tables(name, R=256):
    import json
    k = json.load(open(os.path.join(ROOT, "tests/golden/known_answers.json")))[name]
    sc = rt.Scene(os.path.join(ROOT, k["prefix"]), **{key: os.path.join(ROOT, val) for key, val in k.get("overrides", {}).items()})
    V = np.ascontiguousarray(sc.vertices().reshape(-1, 9).astype(F))
    lights = np.ascontiguousarray(np.array([[sc.desc.lights[i].p[c] for c in range(3)] for i in range(sc.nl)], F))
    L = _lib()
    n_rects = C.c_int64(0); refs = C.c_int64(0); olim = C.c_float(0)
    faces = np.zeros((sc.nl * 12, 8), F)
    assert L.rtgpu_light_list_rects(len(V), V.ctypes.data, sc.nl, lights.ctypes.data, R, C.c_float(0.0), 0, None, faces.ctypes.data,
                                    C.byref(olim), C.byref(n_rects), C.byref(refs)) == 0
    rects = np.zeros((n_rects.value, 3), np.uint32)
    assert L.rtgpu_light_list_rects(len(V), V.ctypes.data, sc.nl, lights.ctypes.data, R, C.c_float(0.0), n_rects.value, rects.ctypes.data,
                                    faces.ctypes.data, C.byref(olim), C.byref(n_rects), C.byref(refs)) == 0
    return V, lights, faces, rects, olim.value, refs.value


@pytest.mark.parametrize("name", ["test", "temple", "s3"])
def test_a_scenes_table_lists_every_blocker_of_rays_between_its_surfaces_and_lights(name):
    R = 256
    V, lights, faces, rects, olim, refs = _scene_tables(name, R)
    nt = len(V)
    assert rects[:, 2].max() < nt and refs == ((((rects[:, 1] >> 8) & 255) - (rects[:, 1] & 255) + 1).astype(np.int64) *
                                               (((rects[:, 1] >> 24) & 255) - ((rects[:, 1] >> 16) & 255) + 1)).sum()
    rng = np.random.default_rng(5)
    a, ab, ac = V[:, 0:3], (V[:, 3:6] - V[:, 0:3]).astype(F), (V[:, 6:9] - V[:, 0:3]).astype(F)
    n = 1500
    # origins: on the scene's surfaces, half of them within a hair of an edge of a triangle as the light sees it
    ti = rng.integers(0, nt, n)
    u, v = aim_inside(rng, n)
    ue, ve = aim_edges(rng, n)
    edge = rng.random(n) < 0.5
    u = np.where(edge, ue, u); v = np.where(edge, ve, v)
    li = rng.integers(0, len(lights), n)
    P = lights[li]
    target = a[ti].astype(np.float64) + u[:, None] * ab[ti].astype(np.float64) + v[:, None] * ac[ti].astype(np.float64)
    # ... continued away from the light by a random stretch (a receiver behind the triangle), or the point itself
    back = target - P.astype(np.float64)
    stretch = np.where(rng.random(n) < 0.3, 0.0, 10.0 ** rng.uniform(-3, 0.5, n))
    o = (target + back * stretch[:, None]).astype(F)
    keep = np.sqrt((o.astype(np.float64) ** 2).sum(1)) * 1.0001 <= olim
    assert keep.mean() > 0.5
    o, P, li = o[keep], P[keep], li[keep]
    n = len(o)
    r, dmax = shadow_ray(o, P)
    face, p, q, on = face_coords(o, P)
    # window 0 of the ray's face if the ray lies in it, else window 1 (the whole face)
    f0 = (li * 6 + face) * 2
    W0 = faces[f0]
    in0 = (W0[:, 7].view(np.uint32) != 0) & (p >= W0[:, 0]) & (p <= W0[:, 4]) & (q >= W0[:, 1]) & (q <= W0[:, 5])
    fi = np.where(in0, f0, f0 + 1)
    Fp = faces[fi]
    inwin = (Fp[:, 7].view(np.uint32) != 0) & (p >= Fp[:, 0]) & (p <= Fp[:, 4]) & (q >= Fp[:, 1]) & (q <= Fp[:, 5])
    assert inwin.mean() > 0.95, "nearly all rays between the scene's surfaces and its lights can be looked up"
    assert in0.mean() > 0.5
    with np.errstate(all="ignore"):
        tu = np.clip(((p - Fp[:, 0]).astype(F) * Fp[:, 2]).astype(F), 0, R - 1).astype(np.int64)
        tv = np.clip(((q - Fp[:, 1]).astype(F) * Fp[:, 3]).astype(F), 0, R - 1).astype(np.int64)
    a0 = (rects[:, 1] & 255).astype(np.int64); a1 = ((rects[:, 1] >> 8) & 255).astype(np.int64)
    b0 = ((rects[:, 1] >> 16) & 255).astype(np.int64); b1 = ((rects[:, 1] >> 24) & 255).astype(np.int64)
    by_face = {f: np.flatnonzero(rects[:, 0] == f) for f in np.unique(fi)}
    checked = blockers = listed_total = 0
    for x in np.flatnonzero(inwin):
        oo = np.repeat(o[x:x + 1], nt, 0); rr = np.repeat(r[x:x + 1], nt, 0)
        acc, _, d, _, _, _ = exact_test(a, ab, ac, oo, rr)
        with np.errstate(all="ignore"):
            blk = np.flatnonzero(acc & (d > F(0.00001)) & (d < dmax[x]))
        sel = by_face[fi[x]]
        cover = sel[(a0[sel] <= tu[x]) & (a1[sel] >= tu[x]) & (b0[sel] <= tv[x]) & (b1[sel] >= tv[x])]
        listed = set(rects[cover, 2].tolist())
        missing = [int(t) for t in blk if int(t) not in listed]
        assert not missing, "ray %d: blockers %s are not in its texel's list" % (x, missing)
        checked += 1; blockers += len(blk); listed_total += len(listed)
    assert checked > 300 and blockers > 100
    print("%s: %d rays, %d blockers, %.1f triangles listed per texel looked at, %d references in the table" % (name, checked, blockers, listed_total / checked, refs))
