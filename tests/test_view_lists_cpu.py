"""The view lists' completeness, on the CPU: every triangle the reference's test (src/intersection.c:58-86, restated in
binary32 in test_filter_cpu.py) accepts for the primary ray of a pixel (src/vectormath.h:108-118 as the kernels evaluate it:
binary32, no FMA, the reciprocal square root through double) has that pixel in its rectangle.  The rectangles come from the
library's own code for them (rtgpu_view_list_rects: the function the device runs, compiled for the host; no GPU needed)."""
import ctypes as C
import os

import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd.api import Camera
from test_filter_cpu import F, aim_anywhere, aim_edges, aim_inside, exact_test

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    L = rt.gpu_lib()
    L.rtgpu_view_list_rects.restype = C.c_int
    L.rtgpu_view_list_rects.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    return L


def cameras(ob, ul, bl, ur, sw, sh):
    """n x (ob, ul, bl, ur, sw, sh) in the layout of rtgpu_camera."""
    n = len(ob)
    cam = np.zeros(n, dtype=[("f", F, 12), ("i", np.int32, 2)])
    cam["f"][:, 0:3] = ob; cam["f"][:, 3:6] = ul; cam["f"][:, 6:9] = bl; cam["f"][:, 9:12] = ur
    cam["i"][:, 0] = sw; cam["i"][:, 1] = sh
    assert cam.itemsize == C.sizeof(Camera)
    return cam


def primary_rays(ob, ul, bl, ur, sw, sh, px, py):
    """k_nearest's primary_dir: every operation in binary32, in the order written there."""
    w_inv = (F(1.0) / sw.astype(F)).astype(F); h_inv = (F(1.0) / sh.astype(F)).astype(F)
    xc = (px.astype(F) * w_inv).astype(F); yc = (py.astype(F) * h_inv).astype(F)
    d = ((xc[:, None] * (ur - ul).astype(F)).astype(F) + (yc[:, None] * (bl - ul).astype(F)).astype(F)).astype(F)
    d = ((d + ul).astype(F) - ob).astype(F)
    s = ((d[:, 0] * d[:, 0]).astype(F) + (d[:, 1] * d[:, 1]).astype(F)).astype(F)
    s = (s + (d[:, 2] * d[:, 2]).astype(F)).astype(F)
    with np.errstate(all="ignore"):
        inv = (1.0 / np.sqrt(s.astype(np.float64))).astype(F)
    return (d * inv[:, None]).astype(F)


def rects_of(V, cam):
    out = np.zeros((len(V), 5), np.int32)
    V = np.ascontiguousarray(V, F)
    assert _lib().rtgpu_view_list_rects(len(V), V.ctypes.data, cam.ctypes.data, len(cam), out.ctypes.data) == 0
    return out


def make_view_cases(rng, n, scale, offset, aim, mode):
    """Triangles of size ~scale around `offset`, and for each a camera one of whose pixels looks at a point of the
    triangle's plane chosen by `aim`.  mode "near_plane": the observer a hair off the triangle's plane (every ray to the
    triangle grazes it); "close": the observer close to the triangle against its size; "far": far away."""
    ab = (rng.standard_normal((n, 3)) * scale).astype(F)
    ac = (rng.standard_normal((n, 3)) * scale).astype(F)
    sliver = rng.random(n) < 0.15
    ac = np.where(sliver[:, None], (ab * (rng.random((n, 1)) * 2 + 0.1).astype(F) + ac * F(1e-3)).astype(F), ac)
    u, v = aim(rng, n)
    target = rng.standard_normal((n, 3)) * scale + offset
    a = (target - u[:, None] * ab.astype(np.float64) - v[:, None] * ac.astype(np.float64)).astype(F)
    j = (a + ab).astype(F); k = (a + ac).astype(F)
    ab = (j - a).astype(F); ac = (k - a).astype(F)  # the edges as k_preprocess forms them
    target = a.astype(np.float64) + u[:, None] * ab.astype(np.float64) + v[:, None] * ac.astype(np.float64)
    nrm = np.cross(ab.astype(np.float64), ac.astype(np.float64))
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-300)
    dirn = rng.standard_normal((n, 3)); dirn /= np.linalg.norm(dirn, axis=1, keepdims=True)
    if mode == "near_plane":
        inplane = dirn - nrm * np.einsum("ij,ij->i", dirn, nrm)[:, None]
        inplane /= np.maximum(np.linalg.norm(inplane, axis=1, keepdims=True), 1e-300)
        hgt = scale * 10.0 ** rng.uniform(-9, -2, n) * rng.choice([-1.0, 1.0], n)
        dist = scale * 10.0 ** rng.uniform(-1, 1.5, n)
        ob = target + inplane * dist[:, None] + nrm * hgt[:, None]
    else:
        dist = scale * 10.0 ** (rng.uniform(-2, 0.5, n) if mode == "close" else rng.uniform(0.5, 3, n))
        ob = target + dirn * dist[:, None]
    ob = ob.astype(F)
    # a screen in front of the observer, the view direction off the direction to the target by up to ~30 degrees
    to_t = target - ob.astype(np.float64)
    to_t /= np.maximum(np.linalg.norm(to_t, axis=1, keepdims=True), 1e-300)
    fwd = to_t + rng.standard_normal((n, 3)) * rng.uniform(0, 0.35, (n, 1))
    fwd /= np.linalg.norm(fwd, axis=1, keepdims=True)
    right = np.cross(fwd, rng.standard_normal((n, 3))); right /= np.linalg.norm(right, axis=1, keepdims=True)
    up = np.cross(right, fwd)
    sd = scale * 10.0 ** rng.uniform(-2, 1, n)  # the screen's distance
    hw = sd * rng.uniform(0.5, 1.5, n); hh = hw * rng.uniform(0.4, 1.0, n)
    centre = ob.astype(np.float64) + fwd * sd[:, None]
    ul = (centre - right * hw[:, None] + up * hh[:, None]).astype(F)
    ur = (centre + right * hw[:, None] + up * hh[:, None]).astype(F)
    bl = (centre - right * hw[:, None] - up * hh[:, None]).astype(F)
    sw = rng.choice([64, 97, 640, 1001, 3840, 7680], n).astype(np.int32)
    sh = np.maximum((sw * rng.uniform(0.4, 1.0, n)).astype(np.int32), 8)
    # the pixel whose ray passes nearest the target: the target's direction in screen coordinates
    x00 = ul.astype(np.float64) - ob.astype(np.float64)
    U = ur.astype(np.float64) - ul.astype(np.float64); Vv = bl.astype(np.float64) - ul.astype(np.float64)
    # solve x00 + s U + t V = lambda to_t for (s, t, lambda)
    Mx = np.stack([U, Vv, -to_t], axis=2)
    with np.errstate(all="ignore"):
        sol = np.linalg.solve(Mx, -x00[:, :, None])[:, :, 0]
    px = np.rint(sol[:, 0] * sw).astype(np.int64); py = np.rint(sol[:, 1] * sh).astype(np.int64)
    return a, ab, ac, ob, ul, bl, ur, sw, sh, px, py


SPREAD = [(dx, dy) for dx in (-2, -1, 0, 1, 2) for dy in (-2, -1, 0, 1, 2)]


@pytest.mark.parametrize("scale,offset", [(1.0, 0.0), (0.01, 50.0), (20.0, 300.0), (5.0, -1000.0)])
@pytest.mark.parametrize("aim", [aim_inside, aim_edges, aim_anywhere], ids=["inside", "edges", "anywhere"])
@pytest.mark.parametrize("mode", ["far", "close", "near_plane"])
def test_every_accepted_pixel_lies_in_its_triangles_rectangle(scale, offset, aim, mode):
    import zlib
    rng = np.random.default_rng(zlib.crc32(repr((scale, offset, aim.__name__, mode)).encode()))
    n = 40000
    a, ab, ac, ob, ul, bl, ur, sw, sh, px, py = make_view_cases(rng, n, scale, offset, aim, mode)
    V = np.concatenate([a, (a + ab).astype(F), (a + ac).astype(F)], axis=1)
    # (k_preprocess forms the edges from the uploaded vertices: they must come out as ab, ac again)
    assert np.array_equal((V[:, 3:6] - V[:, 0:3]).astype(F), ab) and np.array_equal((V[:, 6:9] - V[:, 0:3]).astype(F), ac)
    rect = rects_of(V, cameras(ob, ul, bl, ur, sw, sh))
    accepted = 0
    for dx, dy in SPREAD:
        qx, qy = px + dx, py + dy
        onscreen = (qx >= 0) & (qx < sw) & (qy >= 0) & (qy < sh)
        r = primary_rays(ob, ul, bl, ur, sw, sh, qx, qy)
        acc = exact_test(a, ab, ac, ob, r)[0] & onscreen
        inside = (rect[:, 0] == 1) & (qx >= rect[:, 1]) & (qx <= rect[:, 2]) & (qy >= rect[:, 3]) & (qy <= rect[:, 4])
        bad = acc & ~inside
        assert not bad.any(), "case %d: pixel (%d, %d) accepted outside the rectangle %s" % (
            np.flatnonzero(bad)[0], qx[np.flatnonzero(bad)[0]], qy[np.flatnonzero(bad)[0]], rect[np.flatnonzero(bad)[0]])
        accepted += int(acc.sum())
    assert accepted > (n // 50 if aim is aim_anywhere else n // 4), "the cases do aim at their triangles"
    # and the rectangles are no larger than they need be by much: most triangles seen from afar keep to a small part of the screen
    if mode == "far" and aim is aim_inside:
        area = (rect[:, 2] - rect[:, 1] + 1).astype(np.float64) * (rect[:, 4] - rect[:, 3] + 1) / (sw.astype(np.float64) * sh)
        assert np.median(area[rect[:, 0] == 1]) < 0.25


@pytest.mark.parametrize("name,w,h", [("test", 96, 54), ("temple", 96, 54), ("s3", 64, 48)])
def test_a_scenes_rectangles_hold_every_pixel_a_triangle_is_accepted_for(name, w, h):
    """Every pixel of a small frame against every triangle of a bundled scene."""
    import json
    k = json.load(open(os.path.join(ROOT, "tests/golden/known_answers.json")))[name]
    sc = rt.Scene(os.path.join(ROOT, k["prefix"]), width=w, height=h,
                  **{key: os.path.join(ROOT, val) for key, val in k.get("overrides", {}).items()})
    V = np.ascontiguousarray(sc.vertices().reshape(-1, 9).astype(F))
    cam = sc.camera
    ob = np.array([cam.ob[:]], F); ul = np.array([cam.ul[:]], F); bl = np.array([cam.bl[:]], F); ur = np.array([cam.ur[:]], F)
    rect = rects_of(V, cameras(ob, ul, bl, ur, np.array([w], np.int32), np.array([h], np.int32)))
    yy, xx = np.mgrid[0:h, 0:w]
    px = xx.ravel().astype(np.int64); py = yy.ravel().astype(np.int64)
    npx = len(px)
    rep = lambda v: np.repeat(v, npx, axis=0)
    r = primary_rays(rep(ob), rep(ul), rep(bl), rep(ur), np.full(npx, w, np.int32), np.full(npx, h, np.int32), px, py)
    a, ab, ac = V[:, 0:3], (V[:, 3:6] - V[:, 0:3]).astype(F), (V[:, 6:9] - V[:, 0:3]).astype(F)
    o = rep(ob)
    hits = 0
    step = max(1, 1500000 // npx)
    for t0 in range(0, len(V), step):
        t1 = min(t0 + step, len(V))
        m = t1 - t0
        tile = lambda v: np.repeat(v[t0:t1], npx, axis=0)
        acc = exact_test(tile(a), tile(ab), tile(ac), np.tile(o, (m, 1)), np.tile(r, (m, 1)))[0].reshape(m, npx)
        R = rect[t0:t1]
        inside = ((R[:, 0:1] == 1) & (px[None, :] >= R[:, 1:2]) & (px[None, :] <= R[:, 2:3]) & (py[None, :] >= R[:, 3:4]) & (py[None, :] <= R[:, 4:5]))
        assert not (acc & ~inside).any()
        hits += int(acc.sum())
    assert hits > (npx // 2 if name != "test" else 0)  # (the camera of scenes/test looks past most of its few triangles)
    # the rectangles are tight enough to be of use: on average a pixel is in far fewer rectangles than there are triangles
    area = ((rect[:, 2] - rect[:, 1] + 1).astype(np.float64) * (rect[:, 4] - rect[:, 3] + 1))[rect[:, 0] == 1].sum()
    print("%s: %.1f rectangles per pixel, %.2f accepted triangles per pixel, %d triangles" % (name, area / npx, hits / npx, len(V)))
    assert area / npx < 0.2 * len(V) + 50
