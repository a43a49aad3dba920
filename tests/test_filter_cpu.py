"""The walk kernels' staged triangle filter, restated in numpy and checked against the exact test.

The HIP kernels put a conservative filter (raytracer_amd/csrc/rt_math.h: pl_stage1, pl_stage2,
pl_surely_inside, pl_sdist; rt_kernels.h: nearest_needs_exact, shadow_verdict, box_may_hit) in front
of the reference's Moeller-Trumbore test (src/intersection.c:58-86).  The filter must never decide
differently from that test.  Here both are evaluated in float32 on millions of rays, most of them aimed
at the places where rounding decides (triangle edges and vertices, grazing incidence, slivers, origins
far from the triangle), and every implication the kernels rely on is asserted:

    exact test accepts              =>  no slab stage rejects, the batch's bounding box is not culled
    "surely inside" + sign sure     =>  exact det/u/v tests accept
    distance verdicts               =>  agree with the exact d

float32 FMA is emulated as float32(float64(a) * float64(b) + float64(c)) (exact product, one extra
rounding of the sum in rare ties: a last-bit effect, orders of magnitude below the margins under test).
"""
import numpy as np
import pytest

F = np.float32
EPS = 5.9604644775390625e-8


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)


def cross32(a, b):  # vectormath.h:87-92, float32, no FMA
    return np.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1], a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2],
                     a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], 1)


def dot32(a, b):  # vectormath.h:95-97
    return a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1] + a[:, 2] * b[:, 2]


def exact_test(a, ab, ac, o, r):
    """src/intersection.c:58-86 without the early returns: accept mask and d, u, v, det."""
    with np.errstate(all="ignore"):
        p = cross32(r, ac)
        det = dot32(ab, p)
        ok_det = ~((det > F(-0.000001)) & (det < F(0.000001)))
        inv = F(1.0) / det
        tv = o - a
        u = dot32(tv, p) * inv
        ok_u = ~((u < 0) | (u > 1))
        q = cross32(tv, ab)
        v = dot32(r, q) * inv
        ok_v = ~((v < 0) | (u + v > 1))
        d = dot32(ac, q) * inv
        ok_d = ~(d < 0)
    return ok_det & ok_u & ok_v & ok_d, ok_det & ok_u & ok_v, d, u, v, det


def preprocess(a, ab, ac):
    """k_preprocess's filter record: relabelled (longest edge = ik), double arithmetic, rounded once."""
    A, AB, AC = a.astype(np.float64), ab.astype(np.float64), ac.astype(np.float64)
    B, C, BC = A + AB, A + AC, AC - AB
    lab, lac, lbc = (np.linalg.norm(x, axis=1) for x in (AB, AC, BC))
    keep = (lac >= lab) & (lac >= lbc)
    swap = ~keep & (lab >= lbc)
    a2 = np.where((keep | swap)[:, None], A, B)
    ab2 = np.where(keep[:, None], AB, np.where(swap[:, None], AC, -AB))
    ac2 = np.where(keep[:, None], AC, np.where(swap[:, None], AB, BC))
    N, Q, P = np.cross(ac2, ab2), np.cross(ac2, a2), np.cross(a2, ab2)
    L = np.maximum(lab, np.maximum(lac, lbc))
    amax = np.maximum(np.linalg.norm(A, axis=1), np.maximum(np.linalg.norm(B, axis=1), np.linalg.norm(C, axis=1)))
    k1 = (144.0 * EPS * L * 1.001).astype(F)
    k2 = ((144.0 * EPS * L * amax + 56.0 * EPS * L * L) * 1.001 + 1e-30).astype(F)
    an = np.einsum("ij,ij->i", a2, N).astype(F)
    verts = np.stack([A, B, C], 1)
    return dict(N=N.astype(F), ac=ac2.astype(F), Q=Q.astype(F), ab=ab2.astype(F), P=P.astype(F), an=an, k1=k1, k2=k2,
                L=L, verts=verts)


def ray_side(o, r):
    """pl_ray of rt_math.h: the moment in double, rounded once; the bound of |o| and o.r in float32."""
    O, R = o.astype(np.float64), r.astype(np.float64)
    m = np.cross(O, R).astype(F)
    with np.errstate(all="ignore"):
        s = fma(o[:, 0], o[:, 0], fma(o[:, 1], o[:, 1], o[:, 2] * o[:, 2]))
        onorm = fma(np.sqrt(s).astype(F), np.full_like(s, F(1.0001)), np.full_like(s, F(1e-18)))
        odr = fma(o[:, 0], r[:, 0], fma(o[:, 1], r[:, 1], o[:, 2] * r[:, 2]))
    return m, onorm, odr


def stages(T, o, r, m, kc):
    N, ac, Q, ab, P = T["N"], T["ac"], T["Q"], T["ab"], T["P"]
    det = fma(r[:, 0], N[:, 0], fma(r[:, 1], N[:, 1], r[:, 2] * N[:, 2]))
    rq = fma(r[:, 0], Q[:, 0], fma(r[:, 1], Q[:, 1], r[:, 2] * Q[:, 2]))
    tp = fma(m[:, 0], ac[:, 0], fma(m[:, 1], ac[:, 1], fma(m[:, 2], ac[:, 2], -rq)))
    x1 = fma(np.full_like(tp, 2), tp, -det)
    Tt = fma(np.abs(det), np.full_like(det, F(1.00001)), kc)
    p1 = ~(np.abs(x1) > Tt)
    rp = fma(r[:, 0], P[:, 0], fma(r[:, 1], P[:, 1], r[:, 2] * P[:, 2]))
    nv = fma(m[:, 0], ab[:, 0], fma(m[:, 1], ab[:, 1], fma(m[:, 2], ab[:, 2], rp)))
    x2 = fma(np.full_like(nv, -2), nv, -det)
    x3 = fma(np.full_like(nv, 2), tp - nv, -det)
    p2 = ~(np.maximum(np.abs(x2), np.abs(x3)) > Tt)
    ad = np.abs(det)
    w = np.maximum(np.abs(x1), np.maximum(np.abs(x2), np.abs(x3)))
    inside = (w <= fma(ad, np.full_like(ad, F(0.99999)), -kc)) & (ad - kc > F(1.01e-6))
    sure = ad - kc > F(1.01e-6)
    dp = fma(-o[:, 0], N[:, 0], fma(-o[:, 1], N[:, 1], fma(-o[:, 2], N[:, 2], T["an"])))
    sd = np.where(det < 0, -dp, dp)
    return p1, p2, inside, sure, sd, ad


def batch_header(T):
    """k_grid_bounds for a batch of one triangle, float32 like the kernel."""
    verts = T["verts"].astype(F)
    lo, hi = verts.min(1), verts.max(1)
    c = (F(0.5) * (lo + hi)).astype(F)
    r2 = ((verts - c[:, None, :]) ** 2).sum(2).max(1).astype(F)
    cn = np.sqrt((c * c).sum(1)).astype(F)
    radius = ((np.sqrt(r2) * F(1.001) + F(0.00001) * cn) * F(1.0001) + F(1e-30)).astype(F)
    N = T["N"]
    with np.errstate(all="ignore"):
        nn = np.sqrt(N[:, 0] * N[:, 0] + N[:, 1] * N[:, 1] + N[:, 2] * N[:, 2]).astype(F)
        L = T["k1"] * F(1.0 / (144.0 * EPS))
        gb = (F(32.0 * EPS) * L * L / nn * F(1.0001)).astype(F)
        ga = (F(0.9) * gb * L * F(1.0001)).astype(F)
        nrm = (N * (F(1.0) / nn)[:, None]).astype(F)
        # the header's normal: three signed 10-bit fractions (pack_normal), the zero vector for a degenerate triangle
        ok = np.abs(nrm).sum(1) < 2
        nq = np.where(ok[:, None], np.rint(np.where(ok[:, None], nrm, 0) * F(511.0)), 0).astype(F)
    bad = ~((gb >= 0) & (ga >= 0))
    radius = np.where(bad, F(np.inf), radius)
    h = ((F(0.5) * (hi - lo) * F(1.001) + F(0.00001) * cn[:, None]) * F(1.0001) + F(1e-30)).astype(F)
    h = np.where(bad[:, None], F(np.inf), h)
    return dict(c=c, radius=radius, h=h, ga=ga, gb=gb, nq=nq)


def batch_cos(H, axis, rho):
    """batch_cos of rt_kernels.h for batches of one triangle: bound of 1 / |cos(ray, normal)| over a bundle."""
    n = H["nq"]
    with np.errstate(all="ignore"):
        s = (axis * F(1.0 / 511.0)).astype(F)
        c0 = np.abs(s[:, 0] * n[:, 0] + s[:, 1] * n[:, 1] + s[:, 2] * n[:, 2])
        c0 = np.where(c0 >= 0, c0, F(0))
        cosb = c0 * F(0.9999) - F(0.0018) - rho
        least = np.maximum(F(0.001), F(0.25) * H["gb"])
        return np.where(cosb > least, (F(1.0) / cosb).astype(F) * F(1.01), F(np.inf)).astype(F)


def _box_ray(c, r, m, odr):
    ex = fma(-c[:, 2], r[:, 1], fma(c[:, 1], r[:, 2], -m[:, 0]))
    ey = fma(-c[:, 0], r[:, 2], fma(c[:, 2], r[:, 0], -m[:, 1]))
    ez = fma(-c[:, 1], r[:, 0], fma(c[:, 0], r[:, 1], -m[:, 2]))
    tc = fma(c[:, 0], r[:, 0], fma(c[:, 1], r[:, 1], fma(c[:, 2], r[:, 2], -odr)))
    return ex, ey, ez, tc


def _box_separated(e, r, hw, W, delta, tmax):
    ex, ey, ez, tc = e
    hx, hy, hz = hw
    ax, ay, az = np.abs(r[:, 0]), np.abs(r[:, 1]), np.abs(r[:, 2])
    sx = fma(hy, az, fma(hz, ay, delta))
    sy = fma(hx, az, fma(hz, ax, delta))
    sz = fma(hx, ay, fma(hy, ax, delta))
    hr = fma(hx, ax, fma(hy, ay, fma(hz, az, W)))
    return (np.abs(ex) > sx) | (np.abs(ey) > sy) | (np.abs(ez) > sz) | (tc + hr < 0) | (tc - hr > tmax)


def bundle_box_may_hit(H, inv_cos, axis, ml, odl, onorm, tmax, ro, rho, re=None, T=None, endless=False):
    """box_test_bundle of rt_kernels.h, operation by operation in float32: the leader's ray (axis, ml, odl) for every
    ray within ro, rho of it; re, T: for rays that end, the spread of the end points and the leader's last parameter
    (None: rays without an end); endless: hits beyond a ray's end count too (transparent triangles in the shadow walk).  Returns (may hit, W for a single ray of the bundle, delta, spread = the spread of the
    bundle's lines at the batch for the record-level test)."""
    c, h = H["c"], H["h"]
    k = lambda x: np.full_like(onorm, F(x))
    if re is None:
        re, T = np.full_like(onorm, F(np.inf)), np.full_like(onorm, F(np.inf))
    with np.errstate(all="ignore"):
        e = _box_ray(c, axis, ml, odl)
        ex, ey, ez, tc = e
        d2 = fma(ex, ex, fma(ey, ey, ez * ez))
        on5 = F(0.00001) * onorm
        tv = fma(np.sqrt(fma(tc, tc, d2)), k(1.001), on5 + H["radius"])
        tv = fma(ro, k(1.001), tv)
        delta = fma(k(0.00001), tv, on5 + on5)
        w_ray = fma(fma(H["gb"], tv, H["ga"]), inv_cos, delta)
        reach = (h[:, 0] + h[:, 1] + h[:, 2]) + (w_ray + w_ray)
        widen = fma(rho, tv + reach, ro)
        far = np.maximum(ro, re)
        s0 = np.maximum(tc - reach - far, F(0)) / fma(k(2), far, T) * F(0.9998)
        s1 = np.maximum(tc + reach + far, F(0)) / fma(k(-2), far, T) * F(1.0002)
        if not endless:
            s1 = np.minimum(s1, F(1))
        sn = np.minimum(s0, s1)
        near_side = fma(np.abs(F(1) - sn), ro, sn * re)
        far_side = fma(np.abs(F(1) - s1), ro, s1 * re)
        narrowing = np.maximum(near_side, far_side) * F(1.0001) + w_ray
        widen = np.where(F(4) * far <= T, np.fmin(widen, narrowing), widen)
        W = w_ray + widen * F(1.001)
        te = T - tc
        tb = fma(np.sqrt(fma(te, te, d2)), k(1.001), on5 + H["radius"])
        spread = np.fmin(fma(tv, rho, F(1.000001) * ro), fma(tb, rho, F(1.000001) * re))
        out = _box_separated(e, axis, (h[:, 0] + W, h[:, 1] + W, h[:, 2] + W), W, delta, tmax)
    return ~out, w_ray, delta, spread


def bundle_slabs_pass(T, kc_max, spread, axis, ml, rho):
    """bundle_slabs_pass of rt_kernels.h, operation by operation in float32: may any ray of the bundle pass the record's three slabs?"""
    N, ac, Q, ab, P = T["N"], T["ac"], T["Q"], T["ab"], T["P"]
    k = lambda x: np.full_like(spread, F(x))
    with np.errstate(all="ignore"):
        det = fma(axis[:, 0], N[:, 0], fma(axis[:, 1], N[:, 1], axis[:, 2] * N[:, 2]))
        rq = fma(axis[:, 0], Q[:, 0], fma(axis[:, 1], Q[:, 1], axis[:, 2] * Q[:, 2]))
        tp = fma(ml[:, 0], ac[:, 0], fma(ml[:, 1], ac[:, 1], fma(ml[:, 2], ac[:, 2], -rq)))
        rp = fma(axis[:, 0], P[:, 0], fma(axis[:, 1], P[:, 1], axis[:, 2] * P[:, 2]))
        nv = fma(ml[:, 0], ab[:, 0], fma(ml[:, 1], ab[:, 1], fma(ml[:, 2], ab[:, 2], rp)))
        x1 = fma(k(2), tp, -det)
        x2 = fma(k(-2), nv, -det)
        x3 = fma(k(2), tp - nv, -det)
        nN = np.sqrt(fma(N[:, 0], N[:, 0], fma(N[:, 1], N[:, 1], N[:, 2] * N[:, 2]))) * F(1.001)
        L = np.sqrt(fma(ac[:, 0], ac[:, 0], fma(ac[:, 1], ac[:, 1], ac[:, 2] * ac[:, 2]))) * F(1.002)
        rn = rho * nN
        E2 = (F(2.0) * L) * spread * F(1.001)
        Tt = fma(np.abs(det) + rn, k(1.00001), rn + (kc_max + kc_max))
        return ~((np.maximum(np.abs(x1), np.abs(x2)) > Tt + E2) | (np.abs(x3) > Tt + (E2 + E2)))


def ray_box_may_hit(H, w_ray, delta, r, m, odr, tmax):
    """box_test_ray of rt_kernels.h: one ray of a bundle against the box widened by the bundle's W and delta."""
    c, h = H["c"], H["h"]
    with np.errstate(all="ignore"):
        out = _box_separated(_box_ray(c, r, m, odr), r, (h[:, 0] + w_ray, h[:, 1] + w_ray, h[:, 2] + w_ray), w_ray, delta, tmax)
    return ~out


def box_may_hit(H, inv_cos, r, m, odr, onorm, tmax):
    """Both levels for a bundle of this one ray (ro = 0, rho = 1e-7, as make_bundle gives)."""
    zero, tiny = np.zeros_like(onorm), np.full_like(onorm, F(1e-7))
    b, w_ray, delta, _ = bundle_box_may_hit(H, inv_cos, r, m, odr, onorm, tmax, zero, tiny)
    return b & ray_box_may_hit(H, w_ray, delta, r, m, odr, tmax)


def make_cases(rng, n, scale, offset, aim):
    """Triangles of size ~scale around `offset`; rays from origins up to 30 scales away, aimed at a point
    of the triangle's plane chosen by `aim` (barycentric coordinates), then normalised like the kernels do."""
    a = (rng.standard_normal((n, 3)) * scale + offset).astype(F)
    shape = rng.random(n)
    ab = (rng.standard_normal((n, 3)) * scale).astype(F)
    ac = (rng.standard_normal((n, 3)) * scale).astype(F)
    sliver = shape < 0.15  # nearly collinear edges
    ac = np.where(sliver[:, None], (ab * (rng.random((n, 1)) * 2 + 0.1).astype(F) + ac * F(1e-3)).astype(F), ac)
    u, v = aim(rng, n)
    target = a.astype(np.float64) + u[:, None] * ab.astype(np.float64) + v[:, None] * ac.astype(np.float64)
    dist = scale * 10.0 ** rng.uniform(-2, 1.5, n)
    dirn = rng.standard_normal((n, 3))
    grazing = rng.random(n) < 0.2  # direction nearly in the triangle's plane
    nrm = np.cross(ab.astype(np.float64), ac.astype(np.float64))
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-300)
    dirn = np.where(grazing[:, None], dirn - nrm * np.einsum("ij,ij->i", dirn, nrm)[:, None] * (1 - 1e-3 * rng.random((n, 1))), dirn)
    dirn /= np.linalg.norm(dirn, axis=1, keepdims=True)
    o = (target - dirn * dist[:, None]).astype(F)
    behind = rng.random(n) < 0.1
    dirn = np.where(behind[:, None], -dirn, dirn)
    # normalise the way vectormath.h:60-66 does: float sum of squares, double rsqrt rounded to float
    rr = (target - o.astype(np.float64)).astype(F) * np.where(behind, F(-1), F(1))[:, None]
    s = rr[:, 0] * rr[:, 0] + rr[:, 1] * rr[:, 1] + rr[:, 2] * rr[:, 2]
    inv = (1.0 / np.sqrt(s.astype(np.float64))).astype(F)
    r = rr * inv[:, None]
    return a, ab, ac, o, r


def aim_inside(rng, n):
    u = rng.random(n); v = rng.random(n) * (1 - u)
    return u, v


def aim_edges(rng, n):
    """On or within a few ulps of an edge or a vertex."""
    u = rng.random(n); v = rng.random(n) * (1 - u)
    which = rng.integers(0, 6, n)
    tiny = (rng.standard_normal(n) * 10.0 ** rng.uniform(-9, -4, n))
    u = np.where(which == 0, tiny, u); v = np.where(which == 1, tiny, v)
    v = np.where(which == 2, 1 - u + tiny, v)
    u = np.where(which == 3, tiny, u); v = np.where(which == 3, tiny, v)
    u = np.where(which == 4, 1 + tiny, u); v = np.where(which == 4, tiny, v)
    u = np.where(which == 5, tiny, u); v = np.where(which == 5, 1 + tiny, v)
    return u, v


def aim_anywhere(rng, n):
    return rng.uniform(-2, 3, n), rng.uniform(-2, 3, n)


@pytest.mark.parametrize("scale,offset", [(1.0, 0.0), (0.01, 50.0), (20.0, 300.0), (1e-3, 0.0), (5.0, -1000.0)])
@pytest.mark.parametrize("aim", [aim_inside, aim_edges, aim_anywhere], ids=["inside", "edges", "anywhere"])
def test_filter_never_contradicts_the_exact_test(scale, offset, aim):
    rng = np.random.default_rng(hash((scale, offset, aim.__name__)) & 0xffff)
    n = 300000
    a, ab, ac, o, r = make_cases(rng, n, scale, offset, aim)
    acc, acc_uv, d, u, v, det = exact_test(a, ab, ac, o, r)
    T = preprocess(a, ab, ac)
    m, onorm, odr = ray_side(o, r)
    kc = fma(T["k1"], onorm, T["k2"])
    p1, p2, inside, sure, sd, ad = stages(T, o, r, m, kc)
    lmax = F(T["L"].max() * 1.001)
    e = lmax * kc
    # 1. no slab stage rejects what the exact test accepts
    assert not (acc & ~p1).any(), "first slab rejected an accepted hit"
    assert not (acc & ~p2).any(), "second slab pair rejected an accepted hit"
    # 2. "surely inside" proves the exact det, u, v tests
    assert not (inside & ~acc_uv).any(), "surely-inside verdict on a triangle the exact test rejects"
    # 3. distance verdicts of the nearest walk: never skip an accepted hit that is nearer than best
    best = (np.abs(d) * rng.uniform(0.5, 1.5, n).astype(F)).astype(F)
    behind = sure & (sd + e < 0)
    farther = sure & (sd - e > 0) & (sd - e > best * (ad + kc) * F(1.00001))
    assert not (behind & acc).any(), "skipped as behind the origin, but the exact test accepts"
    assert not (farther & acc & (d < best)).any(), "skipped as farther than best, but the exact d is nearer"
    # 4. distance verdicts of the shadow walk
    dmax = (np.abs(d) * rng.uniform(0.5, 1.5, n).astype(F)).astype(F)
    blocker = acc & (d > F(0.00001)) & (d < dmax)
    out = sure & ((sd + e < F(0.0000099) * (ad - kc)) | (sd - e > dmax * (ad + kc) * F(1.00001)))
    assert not (out & blocker).any(), "opaque triangle declared harmless, but it blocks"
    sure_block = inside & (sd - e > F(0.0000101) * (ad + kc)) & (sd + e < dmax * (ad - kc) * F(0.99999))
    assert not (sure_block & ~blocker).any(), "declared a blocker, but the exact test disagrees"
    sure_transparent = inside & (sd - e > 0)
    assert not (sure_transparent & ~acc).any(), "declared an accepted hit, but the exact test rejects (d < 0)"
    # 5. the bounding box of the triangle (as k_grid_bounds builds it), widened by the visit's fuzz, is never
    #    culled for an accepted hit; the bundle is the ray and a second direction up to 2 degrees away
    H = batch_header(T)
    other = (r + (rng.standard_normal((n, 3)) * 10.0 ** rng.uniform(-6, -1.5, (n, 1))).astype(F)).astype(F)
    slab_stats = []
    for axis in (r, other):
        dlt = r - axis
        rho = (np.sqrt(dot32(dlt, dlt)) * F(1.001) + F(1e-7)).astype(F)
        inv_cos = batch_cos(H, axis, rho)
        assert box_may_hit(H, inv_cos, r, m, odr, onorm, np.full(n, np.inf, F))[acc].all(), "box culled an accepted hit"
        assert box_may_hit(H, inv_cos, r, m, odr, onorm, best)[acc & (d < best)].all(), "box culled a hit nearer than best"
        assert box_may_hit(H, inv_cos, r, m, odr, onorm, dmax)[blocker].all(), "box culled a blocker"
        # the bundle variant: the leader's ray (direction axis, origin up to a few triangle sizes away) stands for this ray
        ol = (o + (rng.standard_normal((n, 3)) * scale * 10.0 ** rng.uniform(-4, 0.5, (n, 1))).astype(F)).astype(F)
        ml, onl, odl = ray_side(ol, axis)
        dorg = o - ol
        ro = (np.sqrt(dot32(dorg, dorg)) * F(1.001)).astype(F)
        onb = np.maximum(onorm, onl)
        for bound, which, what in ((np.full(n, np.inf, F), acc, "an accepted hit"), (best, acc & (d < best), "a nearer hit"),
                                   (dmax, blocker, "a blocker")):
            tb = np.maximum(bound, bound[::-1])  # the bundle's bound is the largest of its rays'
            may, w_ray, delta, spread = bundle_box_may_hit(H, inv_cos, axis, ml, odl, onb, tb, ro, rho)
            assert may[which].all(), "bundle test culled " + what
            # ... and with the bundle's W and delta this ray's own test must not cull it either
            assert ray_box_may_hit(H, w_ray, delta, r, m, odr, bound)[which].all(), "ray test with the bundle's margins culled " + what
            if bound is dmax:
                # rays that end (shadow rays): this ray at o + dmax r, the leader, with a range of its own, at ol + Tl axis
                with np.errstate(all="ignore"):
                    Tl = (dmax * rng.uniform(0.7, 1.4, n).astype(F)).astype(F)
                    de = (o + dmax[:, None] * r) - (ol + Tl[:, None] * axis)
                    re = (np.sqrt(dot32(de, de)) * F(1.001) + F(2e-6) * (onb + np.maximum(dmax, Tl))).astype(F)
                may, w_ray, delta, spread_e = bundle_box_may_hit(H, inv_cos, axis, ml, odl, onb, np.maximum(tb, Tl), ro, rho, re, Tl)
                assert may[which].all(), "bundle test of ending rays culled " + what
                kc_max = fma(T["k1"], onb, np.abs(T["k2"]))
                assert bundle_slabs_pass(T, kc_max, spread_e, axis, ml, rho)[acc].all(), "slab test of ending rays dropped a triangle one of them hits"
        # the record-level test of the bundle: a triangle that this ray of the bundle hits is never dropped for the bundle
        kc_max = fma(T["k1"], onb, np.abs(T["k2"]))
        slab = bundle_slabs_pass(T, kc_max, spread, axis, ml, rho)
        assert slab[acc].all(), "the bundle's slab test dropped a triangle one of its rays hits"
        slab_stats.append((slab[~acc].mean(), float(np.median(rho))))
    fuzz = batch_cos(H, r, np.full(n, F(1e-7), F))
    # the filter is not vacuous: it rejects most misses and proves most clear hits
    if aim is aim_anywhere:
        assert slab_stats[0][0] < 0.9, "the bundle's slab test keeps everything: vacuous"
        assert (~p1 | ~p2)[~acc].mean() > 0.5
        assert (~box_may_hit(H, fuzz, r, m, odr, onorm, np.full(n, np.inf, F)))[~acc].mean() > 0.2
    if aim is aim_inside and scale == 1.0 and offset == 0.0:  # (small triangles far from the origin leave little to prove)
        assert inside[acc].mean() > 0.5


@pytest.mark.parametrize("scale,offset", [(1.0, 0.0), (0.01, 50.0), (20.0, 300.0), (5.0, -1000.0)])
def test_box_culling_with_axis_parallel_rays(scale, offset):
    """Rays along one coordinate axis or inside a coordinate plane: the separating directions r x e_i of the box
    test vanish there and the comparison must not be decided by rounding."""
    rng = np.random.default_rng(11)
    n = 200000
    a, ab, ac, o, r = make_cases(rng, n, scale, offset, aim_edges)
    # rebuild the rays: same targets (o + dist r), directions snapped to an axis (two zero components) or a plane (one)
    dist = np.linalg.norm((a - o).astype(np.float64), axis=1)
    target = o.astype(np.float64) + r.astype(np.float64) * dist[:, None]
    keep = rng.integers(0, 3, n)
    planes = rng.random(n) < 0.5
    dirn = rng.standard_normal((n, 3))
    for q in range(3):
        zero = np.where(planes, keep == q, keep != q)
        dirn[:, q] = np.where(zero, 0.0, dirn[:, q])
    dirn /= np.linalg.norm(dirn, axis=1, keepdims=True)
    o = (target - dirn * dist[:, None]).astype(F)
    rr = dirn.astype(F)
    s_ = rr[:, 0] * rr[:, 0] + rr[:, 1] * rr[:, 1] + rr[:, 2] * rr[:, 2]
    r = rr * (1.0 / np.sqrt(s_.astype(np.float64))).astype(F)[:, None]
    acc, acc_uv, d, u, v, det = exact_test(a, ab, ac, o, r)
    assert acc.mean() > 0.1
    T = preprocess(a, ab, ac)
    m, onorm, odr = ray_side(o, r)
    H = batch_header(T)
    inv_cos = batch_cos(H, r, np.full(n, F(1e-7), F))
    dmax = (np.abs(d) * rng.uniform(0.5, 1.5, n).astype(F)).astype(F)
    assert box_may_hit(H, inv_cos, r, m, odr, onorm, np.full(n, np.inf, F))[acc].all()
    assert box_may_hit(H, inv_cos, r, m, odr, onorm, dmax)[acc & (d < dmax)].all()
    assert (~box_may_hit(H, inv_cos, r, m, odr, onorm, np.full(n, np.inf, F)))[~acc].mean() > 0.05


def test_nan_rays_and_triangles_are_never_rejected():
    rng = np.random.default_rng(5)
    n = 1000
    a, ab, ac, o, r = make_cases(rng, n, 1.0, 0.0, aim_anywhere)
    r[::2] = np.nan
    a[1::2, 0] = np.nan
    T = preprocess(a, ab, ac)
    m, onorm, odr = ray_side(o, r)
    with np.errstate(all="ignore"):
        kc = fma(T["k1"], onorm, T["k2"])
        p1, p2, inside, sure, sd, ad = stages(T, o, r, m, kc)
    assert p1.all() and p2.all() and not inside.any() and not sure.any()


@pytest.mark.parametrize("scale,offset", [(1.0, 0.0), (0.05, 20.0), (10.0, -200.0)])
def test_bundles_of_rays_that_end_at_one_point(scale, offset):
    """Shadow rays of a point light: two rays from neighbouring origins to the same point (the light, beyond, on or
    before the triangle) as a bundle that narrows towards it.  A blocker of the second ray (an accepted hit inside
    (1e-5, dmax)) is never culled by the leader's box test nor dropped by the record-level test."""
    rng = np.random.default_rng(int(scale * 1000) + 7)
    n = 300000
    a, ab, ac, o, r0 = make_cases(rng, n, scale, offset, aim_edges)
    # the light: on the first ray, at 0.3 .. 3 times the distance of the aimed-at point
    d0 = np.linalg.norm((a - o).astype(np.float64), axis=1) + 1e-30
    light = (o.astype(np.float64) + r0.astype(np.float64) * (d0 * 10.0 ** rng.uniform(-0.5, 0.5, n))[:, None]).astype(F)

    def shadow_ray(org):
        rr = (light - org).astype(F)
        s2 = rr[:, 0] * rr[:, 0] + rr[:, 1] * rr[:, 1] + rr[:, 2] * rr[:, 2]
        inv = (1.0 / np.sqrt(s2.astype(np.float64))).astype(F)
        return rr * inv[:, None], np.sqrt(s2.astype(np.float64)).astype(F)   # direction, dmax as dist3 gives it

    r, dmax = shadow_ray(o)
    ol = (o + (rng.standard_normal((n, 3)) * scale * 10.0 ** rng.uniform(-3, 0, (n, 1))).astype(F)).astype(F)
    axis, Tl = shadow_ray(ol)
    acc, acc_uv, d, u, v, det = exact_test(a, ab, ac, o, r)
    blocker = acc & (d > F(0.00001)) & (d < dmax)
    assert blocker.mean() > 0.05
    T = preprocess(a, ab, ac)
    m, onorm, odr = ray_side(o, r)
    ml, onl, odl = ray_side(ol, axis)
    H = batch_header(T)
    dlt, dorg = r - axis, o - ol
    rho = (np.sqrt(dot32(dlt, dlt)) * F(1.001) + F(1e-7)).astype(F)
    ro = (np.sqrt(dot32(dorg, dorg)) * F(1.001)).astype(F)
    onb = np.maximum(onorm, onl)
    inv_cos = batch_cos(H, axis, rho)
    de = (o + dmax[:, None] * r) - (ol + Tl[:, None] * axis)
    re = (np.sqrt(dot32(de, de)) * F(1.001) + F(2e-6) * (onb + np.maximum(dmax, Tl))).astype(F)
    assert np.median(re / np.maximum(ro, F(1e-30))) < 0.2   # the bundles do narrow (re is the float slack of the end points)
    may, w_ray, delta, spread = bundle_box_may_hit(H, inv_cos, axis, ml, odl, onb, np.maximum(dmax, Tl), ro, rho, re, Tl)
    assert may[blocker].all(), "the narrowing bundle's box test culled a blocker"
    assert ray_box_may_hit(H, w_ray, delta, r, m, odr, dmax)[blocker].all()
    kc_max = fma(T["k1"], onb, np.abs(T["k2"]))
    assert bundle_slabs_pass(T, kc_max, spread, axis, ml, rho)[acc].all(), "the record-level test dropped a triangle the second ray hits"
    # a transparent triangle attenuates wherever the ray meets it, beyond the light too
    may_e, w_e, delta_e, _ = bundle_box_may_hit(H, inv_cos, axis, ml, odl, onb, np.full(n, np.inf, F), ro, rho, re, Tl, endless=True)
    assert may_e[acc].all(), "the bundle's box test culled an accepted hit beyond or before the light"
    assert ray_box_may_hit(H, w_e, delta_e, r, m, odr, np.full(n, np.inf, F))[acc].all()
    # and the narrowing model culls more than the widening one
    may_wide, _, _, _ = bundle_box_may_hit(H, inv_cos, axis, ml, odl, onb, np.maximum(dmax, Tl), ro, rho)
    assert may.sum() <= may_wide.sum()


def fan_slabs_pass(T, kc_max, lo, hi, light, smax):
    """fan_slabs_pass of rt_kernels.h, operation by operation in float32: may any shadow ray of the point light `light`
    that starts in the box lo..hi pass the record's three slabs?"""
    N, ac, Q, ab, P = T["N"], T["ac"], T["Q"], T["ab"], T["P"]
    k = lambda x: np.full(len(N), F(x))
    with np.errstate(all="ignore"):
        c = (F(0.5) * (lo + hi)).astype(F)
        h = fma(np.maximum(hi - c, c - lo), np.full_like(c, F(1.000001)), np.full_like(c, F(1e-30)))
        rc = (light - c).astype(F)
        mc = np.stack([fma(c[:, 1], rc[:, 2], -(c[:, 2] * rc[:, 1])), fma(c[:, 2], rc[:, 0], -(c[:, 0] * rc[:, 2])),
                       fma(c[:, 0], rc[:, 1], -(c[:, 1] * rc[:, 0]))], 1)
        det = fma(rc[:, 0], N[:, 0], fma(rc[:, 1], N[:, 1], rc[:, 2] * N[:, 2]))
        rq = fma(rc[:, 0], Q[:, 0], fma(rc[:, 1], Q[:, 1], rc[:, 2] * Q[:, 2]))
        tp = fma(mc[:, 0], ac[:, 0], fma(mc[:, 1], ac[:, 1], fma(mc[:, 2], ac[:, 2], -rq)))
        rp = fma(rc[:, 0], P[:, 0], fma(rc[:, 1], P[:, 1], rc[:, 2] * P[:, 2]))
        nv = fma(mc[:, 0], ab[:, 0], fma(mc[:, 1], ab[:, 1], fma(mc[:, 2], ab[:, 2], rp)))
        x1, x2, x3 = fma(k(2), tp, -det), fma(k(-2), nv, -det), fma(k(2), tp - nv, -det)
        Lx, Ly, Lz = light[:, 0], light[:, 1], light[:, 2]
        gt = np.stack([fma(Ly, ac[:, 2], fma(-Lz, ac[:, 1], Q[:, 0])), fma(Lz, ac[:, 0], fma(-Lx, ac[:, 2], Q[:, 1])),
                       fma(Lx, ac[:, 1], fma(-Ly, ac[:, 0], Q[:, 2]))], 1)
        gv = np.stack([fma(Ly, ab[:, 2], fma(-Lz, ab[:, 1], -P[:, 0])), fma(Lz, ab[:, 0], fma(-Lx, ab[:, 2], -P[:, 1])),
                       fma(Lx, ab[:, 1], fma(-Ly, ab[:, 0], -P[:, 2]))], 1)
        two = np.full_like(gt, F(2))
        g1 = fma(two, gt, N)
        g2 = fma(-two, gv, N)
        g3 = fma(-two, gv, g1)
        over = lambda v: fma(np.abs(v[:, 0]), h[:, 0], fma(np.abs(v[:, 1]), h[:, 1], np.abs(v[:, 2]) * h[:, 2]))
        lnorm = np.abs(Lx) + np.abs(Ly) + np.abs(Lz)
        e = F(0.06) * fma(T["k1"], lnorm, np.abs(T["k2"])) * (h[:, 0] + h[:, 1] + h[:, 2])
        Tt = fma(np.abs(det) + over(N), k(1.00001), F(3.0) * kc_max * smax)
        s1, s2, s3 = fma(over(g1), k(1.001), e + e), fma(over(g2), k(1.001), e + e), fma(over(g3), k(1.001), F(4.0) * e)
        return ~((np.abs(x1) - s1 > Tt) | (np.abs(x2) - s2 > Tt) | (np.abs(x3) - s3 > Tt))


@pytest.mark.parametrize("scale,offset", [(1.0, 0.0), (0.05, 20.0), (10.0, -200.0), (1.0, 3000.0)])
@pytest.mark.parametrize("box", ["point", "patch", "flat", "wide"])
def test_fan_of_a_point_lights_shadow_rays(scale, offset, box):
    """The record-level test for the shadow rays of one point light (fan_slabs_pass): whatever box of hit points holds
    the ray's origin (the origin alone, a small patch around it, a patch that is flat in one axis, a box as large as the
    scene), a triangle the exact test accepts for the ray is never dropped.  Rays aimed at edges and vertices, lights
    before, on and beyond the triangle, coordinates far from the origin (where the gradients cancel)."""
    rng = np.random.default_rng(int(scale * 1000) + int(offset) + len(box))
    n = 300000
    a, ab, ac, o, r0 = make_cases(rng, n, scale, offset, aim_edges)
    d0 = np.linalg.norm((a - o).astype(np.float64), axis=1) + 1e-30
    light = (o.astype(np.float64) + r0.astype(np.float64) * (d0 * 10.0 ** rng.uniform(-0.5, 0.5, n))[:, None]).astype(F)
    rr = (light - o).astype(F)
    s2 = rr[:, 0] * rr[:, 0] + rr[:, 1] * rr[:, 1] + rr[:, 2] * rr[:, 2]
    r = rr * (1.0 / np.sqrt(s2.astype(np.float64))).astype(F)[:, None]
    dmax = np.sqrt(s2.astype(np.float64)).astype(F)
    acc, acc_uv, d, u, v, det = exact_test(a, ab, ac, o, r)
    assert acc.mean() > 0.05
    size = {"point": 0.0, "patch": 10.0 ** rng.uniform(-3, -1, (n, 1)), "flat": 10.0 ** rng.uniform(-3, -1, (n, 1)), "wide": 3.0}[box]
    other = (o + (rng.standard_normal((n, 3)) * scale * size).astype(F)).astype(F)
    if box == "flat":
        ax = rng.integers(0, 3, n)
        other[np.arange(n), ax] = o[np.arange(n), ax]
    lo, hi = np.minimum(o, other), np.maximum(o, other)
    T = preprocess(a, ab, ac)
    _, on1, _ = ray_side(o, r)
    _, on2, _ = ray_side(other, r)
    kc_max = fma(T["k1"], np.maximum(on1, on2), np.abs(T["k2"]))
    ro = (light - other).astype(np.float64)
    smax = (np.maximum(dmax, np.sqrt((ro * ro).sum(1)).astype(F)) * F(1.00001)).astype(F)
    keep = fan_slabs_pass(T, kc_max, lo, hi, light, smax)
    assert keep[acc].all(), "the fan's record-level test dropped a triangle a ray of the fan hits"
    if box in ("point", "patch") and offset == 0.0:
        # and it does drop records: the same fans against triangles somewhere else
        T2 = preprocess(np.roll(a, 1, 0), np.roll(ab, 1, 0), np.roll(ac, 1, 0))
        acc2 = exact_test(np.roll(a, 1, 0), np.roll(ab, 1, 0), np.roll(ac, 1, 0), o, r)[0]
        keep2 = fan_slabs_pass(T2, fma(T2["k1"], np.maximum(on1, on2), np.abs(T2["k2"])), lo, hi, light, smax)
        assert keep2[acc2].all()
        assert (~keep2).mean() > 0.3


def fan_plane_behind(T, kc_max, lo, hi, light, smax):
    """fan_plane_behind of rt_kernels.h, operation by operation in float32: does the record's plane surely lie behind every
    shadow ray of the point light `light` that starts in the box lo..hi?"""
    N, ac, an = T["N"], T["ac"], T["an"]
    with np.errstate(all="ignore"):
        c = (F(0.5) * (lo + hi)).astype(F)
        h = fma(np.maximum(hi - c, c - lo), np.full_like(c, F(1.000001)), np.full_like(c, F(1e-30)))
        rc = (light - c).astype(F)
        det = fma(rc[:, 0], N[:, 0], fma(rc[:, 1], N[:, 1], rc[:, 2] * N[:, 2]))
        dp = fma(-c[:, 0], N[:, 0], fma(-c[:, 1], N[:, 1], fma(-c[:, 2], N[:, 2], an)))
        ob = (fma(np.abs(N[:, 0]), h[:, 0], fma(np.abs(N[:, 1]), h[:, 1], np.abs(N[:, 2]) * h[:, 2])) * F(1.001)).astype(F)
        L = (np.sqrt(fma(ac[:, 0], ac[:, 0], fma(ac[:, 1], ac[:, 1], ac[:, 2] * ac[:, 2]))) * F(1.002)).astype(F)
        sdp = np.where(np.signbit(det), -dp, dp).astype(F)
        return (np.abs(det) - ob > F(1.5) * kc_max * smax) & (sdp + ob < F(-2.0) * L * kc_max)


@pytest.mark.parametrize("scale,offset", [(1.0, 0.0), (0.05, 20.0), (10.0, -200.0), (1.0, 3000.0)])
@pytest.mark.parametrize("box", ["point", "patch", "flat", "wide"])
def test_plane_behind_the_fan_never_drops_a_hit(scale, offset, box):
    """The other record-level test of the light lists (fan_plane_behind): a triangle the exact test accepts for a ray of the
    fan, with whatever distance d >= 0, is never declared to lie behind the fan's rays; and triangles that do lie behind
    (the same fans turned round) mostly are."""
    rng = np.random.default_rng(int(scale * 1000) + int(offset) + 3 * len(box))
    n = 300000
    a, ab, ac, o, r0 = make_cases(rng, n, scale, offset, aim_edges)
    d0 = np.linalg.norm((a - o).astype(np.float64), axis=1) + 1e-30
    # the light on the ray: before, on (within a hair) and beyond the triangle
    stretch = np.where(rng.random(n) < 0.3, 1.0 + rng.standard_normal(n) * 1e-6, 10.0 ** rng.uniform(-0.5, 0.5, n))
    light = (o.astype(np.float64) + r0.astype(np.float64) * (d0 * stretch)[:, None]).astype(F)
    rr = (light - o).astype(F)
    s2 = rr[:, 0] * rr[:, 0] + rr[:, 1] * rr[:, 1] + rr[:, 2] * rr[:, 2]
    with np.errstate(all="ignore"):
        r = rr * (1.0 / np.sqrt(s2.astype(np.float64))).astype(F)[:, None]
    dmax = np.sqrt(s2.astype(np.float64)).astype(F)
    acc = exact_test(a, ab, ac, o, r)[0]
    assert acc.mean() > 0.05
    size = {"point": 0.0, "patch": 10.0 ** rng.uniform(-3, -1, (n, 1)), "flat": 10.0 ** rng.uniform(-3, -1, (n, 1)), "wide": 3.0}[box]
    other = (o + (rng.standard_normal((n, 3)) * scale * size).astype(F)).astype(F)
    if box == "flat":
        ax = rng.integers(0, 3, n)
        other[np.arange(n), ax] = o[np.arange(n), ax]
    lo, hi = np.minimum(o, other), np.maximum(o, other)
    T = preprocess(a, ab, ac)
    _, on1, _ = ray_side(o, r)
    _, on2, _ = ray_side(other, r)
    kc_max = fma(T["k1"], np.maximum(on1, on2), np.abs(T["k2"]))
    ro = (light - other).astype(np.float64)
    smax = (np.maximum(dmax, np.sqrt((ro * ro).sum(1)).astype(F)) * F(1.00001)).astype(F)
    behind = fan_plane_behind(T, kc_max, lo, hi, light, smax)
    assert not behind[acc].any(), "a triangle a ray of the fan hits was declared to lie behind the fan"
    if box in ("point", "patch") and offset == 0.0:
        # the rays turned round (from the origin away from the triangle, towards a light on the other side): the plane is behind
        light2 = (o.astype(np.float64) - r0.astype(np.float64) * (d0 * 2.0)[:, None]).astype(F)
        ro2 = (light2 - other).astype(np.float64)
        smax2 = (np.sqrt(np.maximum((ro2 * ro2).sum(1), ((light2 - o).astype(np.float64) ** 2).sum(1))).astype(F) * F(1.00001)).astype(F)
        assert fan_plane_behind(T, kc_max, lo, hi, light2, smax2).mean() > 0.5


def apex_slabs_pass(T, kc_max, o, axis, rho):
    """apex_slabs_pass of rt_view_kernels.h, operation by operation in float32: may any ray that starts at o with a direction
    within rho of `axis` pass the record's three slabs with a hit in front of o?"""
    N, ac, Q, ab, P, an = T["N"], T["ac"], T["Q"], T["ab"], T["P"], T["an"]
    with np.errstate(all="ignore"):
        w1 = np.stack([fma(ac[:, 1], o[:, 2], fma(-ac[:, 2], o[:, 1], -Q[:, 0])), fma(ac[:, 2], o[:, 0], fma(-ac[:, 0], o[:, 2], -Q[:, 1])),
                       fma(ac[:, 0], o[:, 1], fma(-ac[:, 1], o[:, 0], -Q[:, 2]))], 1)
        w2 = np.stack([fma(ab[:, 1], o[:, 2], fma(-ab[:, 2], o[:, 1], P[:, 0])), fma(ab[:, 2], o[:, 0], fma(-ab[:, 0], o[:, 2], P[:, 1])),
                       fma(ab[:, 0], o[:, 1], fma(-ab[:, 1], o[:, 0], P[:, 2]))], 1)
        two = np.full_like(w1, F(2))
        g1 = fma(two, w1, -N)
        g2 = fma(-two, w2, -N)
        g3 = fma(-two, w2, g1)
        along = lambda v: fma(axis[:, 0], v[:, 0], fma(axis[:, 1], v[:, 1], axis[:, 2] * v[:, 2]))
        norm = lambda v: (np.sqrt(fma(v[:, 0], v[:, 0], fma(v[:, 1], v[:, 1], v[:, 2] * v[:, 2]))).astype(F) * F(1.001)).astype(F)
        det = along(N)
        rn = (rho * norm(N)).astype(F)
        Tt = fma(np.abs(det) + rn, np.full_like(det, F(1.00001)), kc_max + kc_max)
        out = ((np.abs(along(g1)) - rho * norm(g1) > Tt) | (np.abs(along(g2)) - rho * norm(g2) > Tt) | (np.abs(along(g3)) - rho * norm(g3) > Tt))
        dp = fma(-o[:, 0], N[:, 0], fma(-o[:, 1], N[:, 1], fma(-o[:, 2], N[:, 2], an)))
        sdp = np.where(det < 0, -dp, dp)
        behind = (np.abs(det) - rn > F(1.5) * kc_max) & (sdp < F(-2.0) * norm(ac) * kc_max)
    return ~(out | behind)


@pytest.mark.parametrize("scale,offset", [(1.0, 0.0), (0.05, 20.0), (10.0, -200.0), (1.0, 3000.0)])
@pytest.mark.parametrize("cone", ["ray", "tile", "wide"])
def test_rays_from_one_point_in_a_cone(scale, offset, cone):
    """The record-level test for the primary rays of one tile (apex_slabs_pass): whatever cone around an axis holds the ray
    (the ray alone, a cone as narrow as a tile of pixels, a wide one), a triangle the exact test accepts for the ray is
    never dropped; and triangles elsewhere are dropped.  Rays aimed at edges and vertices, observers near and far,
    coordinates far from the origin."""
    rng = np.random.default_rng(int(scale * 1000) + int(offset) + 7 * len(cone))
    n = 300000
    a, ab, ac, o, r = make_cases(rng, n, scale, offset, aim_edges)
    acc = exact_test(a, ab, ac, o, r)[0]
    assert acc.mean() > 0.05
    spread = {"ray": 0.0, "tile": 10.0 ** rng.uniform(-4, -2, (n, 1)), "wide": 10.0 ** rng.uniform(-1.5, 0, (n, 1))}[cone]
    ax = r.astype(np.float64) + rng.standard_normal((n, 3)) * spread
    ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    axis = ax.astype(F)
    dr = (r - axis).astype(F)
    with np.errstate(all="ignore"):
        rho = fma(np.sqrt(dr[:, 0] * dr[:, 0] + dr[:, 1] * dr[:, 1] + dr[:, 2] * dr[:, 2]).astype(F), np.full(n, F(1.001)), np.full(n, F(1e-7)))
    rho = (rho * rng.uniform(1.0, 2.0, n).astype(F)).astype(F)  # (the cone of a tile is as wide as its widest ray)
    T = preprocess(a, ab, ac)
    _, onorm, _ = ray_side(o, r)
    kc_max = fma(T["k1"], onorm, np.abs(T["k2"]))
    keep = apex_slabs_pass(T, kc_max, o, axis, rho)
    assert keep[acc].all(), "the cone's record-level test dropped a triangle a ray of the cone hits"
    if cone in ("ray", "tile") and offset == 0.0:
        a2, ab2, ac2 = np.roll(a, 1, 0), np.roll(ab, 1, 0), np.roll(ac, 1, 0)
        T2 = preprocess(a2, ab2, ac2)
        acc2 = exact_test(a2, ab2, ac2, o, r)[0]
        keep2 = apex_slabs_pass(T2, fma(T2["k1"], onorm, np.abs(T2["k2"])), o, axis, rho)
        assert keep2[acc2].all()
        assert (~keep2).mean() > 0.3
