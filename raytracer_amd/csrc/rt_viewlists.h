// View lists: the nearest-hit queries of the PRIMARY rays answered from a table over the screen instead of a walk
// through the grid (voxelize.c:536-613).  The screen is cut into texels of 8 x 8 pixels (one tile of the kernels) or a
// power of two times that, and a texel lists every triangle that the exact test (intersection.c:58-86) can accept for
// the ray of one of its pixels.  Built per frame on the device (the camera may move from frame to frame): view_rect gives
// a triangle's rectangle of pixels, in double precision; k_view_rects, k_view_pieces and k_view_scatter make the lists;
// k_nearest's VL instance reads them.
// The same idea as the light lists (rt_lightlists.h) with the observer in the light's place.
//
// What has to hold.  The ray of pixel (px, py) is o = ob, r = normalize(fl(dir)) with dir = ul - ob + (px / w) (ur - ul) +
// (py / h) (bl - ul) evaluated in binary32 (vectormath.h:108-118).  Claim: if tri_test accepts triangle T for that ray,
// (px, py) lies in T's rectangle.  Then the nearest of a texel's accepted triangles is the nearest accepted triangle of
// the whole scene, and the reference's walk reports it iff the cell in which the walk passes that distance lists it
// (k_nearest checks that with the ray's own walk).
//
// Derivation.  x(px, py) = x00 + px U + py V, x00 = ul - ob, U = (ur - ul) / w, V = (bl - ul) / h in exact arithmetic on the
// camera's floats; fl(dir) = x + e with |e_k| <= 7.1 eps M_k, M_k = |ur_k - ul_k| + |bl_k - ul_k| + |ul_k| + |ob_k| (two
// products of at most 4.02 eps each, three sums), so |e| <= 21 eps M, M the largest M_k; r is, component by component,
// within 2 eps of c fl(dir) for a common factor c = 1 / |fl(dir)|.  With the triangle's vertices V0 = i, V1 = i + ij,
// V2 = i + ik as the test sees them, w_k = V_k - ob and the edge functions E_0(x) = x.(w_1 x w_2), E_1 = x.(w_2 x w_0),
// E_2 = x.(w_0 x w_1), whose sum is D(x) = x.(ij x ik): tri_test's numerators are, exactly on its float inputs, -E_1(r),
// -E_2(r) and -E_0(r) (the ray starts at the apex itself), det* = -D(r).  If tri_test accepts,
// sigma E_k(r) >= -(31.5 eps L (|o| + A) + 17.8 eps L^2), sigma = sign(D(r)) (rt_lightlists.h; tol below is more than
// twice that), hence, divided by c <= 1 / (|x| - |e|):
//     sigma E_k(x) >= -T_k,   T_k = 1.0001 xmax (tol + 4.2 eps |G_k|) + 21 eps M |G_k| + (double rounding),
// xmax the largest |x| over the screen (at a corner), G_k the edge function's vector: three half planes in (px, py).  An
// accepted hit has d >= 0: with the observer clearly off the triangle's plane (the test of rt_lightlists.h with the
// observer as the light) sigma is the sign of D on the triangle itself; otherwise both signs are taken.  The screen's
// rectangle of pixel centres is clipped by the three half planes; the bounding box of what is left, widened by 1e-6
// pixels, is the rectangle.  tests/test_view_lists_cpu.py aims rays at edges, vertices and grazing triangles and checks
// the claim against the binary32 restatement of the ray and of tri_test.
#pragma once
#include <cmath>
#include <cstdint>

#include "rt_types.h"

namespace rt {

struct ViewRect { int32_t a0, a1, b0, b1; };  // pixel columns a0..a1, rows b0..b1

namespace vl_detail {
struct V3 { double x, y, z; };
__host__ __device__ inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__host__ __device__ inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__host__ __device__ inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__host__ __device__ inline double len(V3 a) { return sqrt(dot(a, a)); }
__host__ __device__ inline double max2(double a, double b) { return a > b ? a : b; }
struct Poly { int n; double p[12], q[12]; };
// keeps the part of the polygon where a p + b q + c >= 0
__host__ __device__ inline void clip(Poly &P, double a, double b, double c) {
  Poly O; O.n = 0;
  for (int e = 0; e < P.n; e++) {
    const int f = e + 1 < P.n ? e + 1 : 0;
    const double de = a * P.p[e] + b * P.q[e] + c, df = a * P.p[f] + b * P.q[f] + c;
    if (de >= 0.0) { O.p[O.n] = P.p[e]; O.q[O.n] = P.q[e]; O.n++; }
    if ((de >= 0.0) != (df >= 0.0)) {
      const double t = de / (de - df);
      O.p[O.n] = P.p[e] + (P.p[f] - P.p[e]) * t; O.q[O.n] = P.q[e] + (P.q[f] - P.q[e]) * t; O.n++;
    }
  }
  P = O;
}
}  // namespace vl_detail

// The rectangle of pixels of one triangle (v: i, j, k as uploaded) for the camera; false: none (no pixel's ray can be
// accepted for it).
__host__ __device__ inline bool view_rect(const float *v, const CameraD &cam, ViewRect &out) {
  using namespace vl_detail;
  const double eps = 5.9604644775390625e-8;  // 2^-24
  const V3 ob = {cam.ob[0], cam.ob[1], cam.ob[2]}, ul = {cam.ul[0], cam.ul[1], cam.ul[2]};
  const V3 ur = {cam.ur[0], cam.ur[1], cam.ur[2]}, bl = {cam.bl[0], cam.bl[1], cam.bl[2]};
  const double w = cam.sw, h = cam.sh;
  const V3 x00 = sub(ul, ob), du = sub(ur, ul), dv = sub(bl, ul);
  const V3 U = {du.x / w, du.y / w, du.z / w}, V = {dv.x / h, dv.y / h, dv.z / h};
  const double M = max2(fabs(du.x) + fabs(dv.x) + fabs(ul.x) + fabs(ob.x),
                        max2(fabs(du.y) + fabs(dv.y) + fabs(ul.y) + fabs(ob.y), fabs(du.z) + fabs(dv.z) + fabs(ul.z) + fabs(ob.z)));
  // the largest |x| over the screen: at a corner
  const V3 c1 = {x00.x + du.x, x00.y + du.y, x00.z + du.z}, c2 = {x00.x + dv.x, x00.y + dv.y, x00.z + dv.z};
  const V3 c3 = {c1.x + dv.x, c1.y + dv.y, c1.z + dv.z};
  const double xmax = max2(max2(len(x00), len(c1)), max2(len(c2), len(c3))) * 1.0001 + 21.0 * eps * M;
  const double olim = len(ob) * 1.0002 + 1e-30;
  const double smax = 2.0 * olim;
  const double delta = 4.0 * eps * smax;
  const float ijf[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]}, ikf[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};  // binary32, as k_preprocess
  const V3 V0 = {v[0], v[1], v[2]}, ij = {ijf[0], ijf[1], ijf[2]}, ik = {ikf[0], ikf[1], ikf[2]};
  const V3 V1 = {V0.x + ij.x, V0.y + ij.y, V0.z + ij.z}, V2 = {V0.x + ik.x, V0.y + ik.y, V0.z + ik.z};
  const V3 w0 = sub(V0, ob), w1 = sub(V1, ob), w2 = sub(V2, ob);
  const V3 G[3] = {cross(w1, w2), cross(w2, w0), cross(w0, w1)};
  const V3 Nn = cross(ij, ik);
  const double L = max2(len(ij), max2(len(ik), len(sub(ik, ij))));
  const double A = max2(len(V0), max2(len(V1), len(V2)));
  const double tol = (72.0 * eps * L * (olim + A) + 28.0 * eps * L * L) * 1.01;
  const double nN = len(Nn);
  const double DT = dot(w0, Nn);  // D on the triangle's own directions; |DT| = |N| (the observer's distance from the plane)
  if (!(A < 1e15 && M < 1e15)) {  // (numbers whose products overflow binary32, or a NaN: no claim is made, every pixel)
    out.a0 = 0; out.a1 = cam.sw - 1; out.b0 = 0; out.b1 = cam.sh - 1;
    return true;
  }
  bool one_sign = false;
  if (nN > 0.0) {
    const double hd = fabs(DT) / nN;
    one_sign = hd - delta > 12.0 * eps * smax && nN * (hd - delta) > 2.001 * eps * L * L * (8.9 * smax + 10.5 * (olim + A));
  }
  double u0 = 1e30, u1 = -1e30, v0 = 1e30, v1 = -1e30;
  for (int pass = 0; pass < 2; pass++) {
    double sigma;
    if (one_sign) { if (pass) break; sigma = DT > 0.0 ? 1.0 : -1.0; } else sigma = pass ? -1.0 : 1.0;
    Poly poly; poly.n = 4;
    const double m = 0.01;  // (pixel centres have whole coordinates 0 .. w - 1, 0 .. h - 1)
    poly.p[0] = -m; poly.q[0] = -m; poly.p[1] = w - 1.0 + m; poly.q[1] = -m;
    poly.p[2] = w - 1.0 + m; poly.q[2] = h - 1.0 + m; poly.p[3] = -m; poly.q[3] = h - 1.0 + m;
    for (int k = 0; k < 3 && poly.n > 0; k++) {
      const double gl = len(G[k]);
      const double T = 1.0001 * xmax * (tol + 4.2 * eps * gl) + 21.0 * eps * M * gl + 1e-12 * (gl * xmax + 1e-300);
      clip(poly, sigma * dot(G[k], U), sigma * dot(G[k], V), sigma * dot(G[k], x00) + T);
    }
    for (int e = 0; e < poly.n; e++) {
      u0 = u0 < poly.p[e] ? u0 : poly.p[e]; u1 = u1 > poly.p[e] ? u1 : poly.p[e];
      v0 = v0 < poly.q[e] ? v0 : poly.q[e]; v1 = v1 > poly.q[e] ? v1 : poly.q[e];
    }
  }
  if (!(u0 <= u1)) return false;
  const double wdn = 1e-6 + 1e-9 * (w + h);
  u0 -= wdn; u1 += wdn; v0 -= wdn; v1 += wdn;
  // whole pixels inside [u0, u1] x [v0, v1]
  const double a0 = ceil(u0), a1 = floor(u1), b0 = ceil(v0), b1 = floor(v1);
  if (a1 < 0.0 || a0 > w - 1.0 || b1 < 0.0 || b0 > h - 1.0 || a0 > a1 || b0 > b1) return false;
  out.a0 = (int32_t)(a0 < 0.0 ? 0.0 : a0); out.a1 = (int32_t)(a1 > w - 1.0 ? w - 1.0 : a1);
  out.b0 = (int32_t)(b0 < 0.0 ? 0.0 : b0); out.b1 = (int32_t)(b1 > h - 1.0 ? h - 1.0 : b1);
  return true;
}

}  // namespace rt
