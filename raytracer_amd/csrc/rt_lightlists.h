// Light lists: for a scene without transparent triangles, the any-hit queries towards one point light can be answered
// from a table around the light instead of a walk through the grid (voxelize.c:615-738).  The six faces of a cube around
// the light are cut into texels, and a texel lists every triangle that the exact test (intersection.c:58-86) can accept
// for a shadow ray whose direction, seen from the light, falls into the texel.  This file is the host side: for every
// (light, triangle, face) the rectangle of texels, worked out in double precision.  The device fills the lists from the
// rectangles (k_ll_count, k_ll_fill) and k_shadow reads them.
//
// What has to hold.  The shadow ray of a hit point a towards the light P is o = a, r = fl((P - a) / |P - a|)
// (vectormath.h:120-128), and the kernel picks the texel from d = fl(a - P): a face whose axis w has d_w != 0 of the
// face's sign, with the face coordinates (p, q) = (d_u / |d_w|, d_v / |d_w|) in binary32, both within kFaceExt (the faces
// overlap: a ray's own face is the one of d's dominant axis, where |p|, |q| <= 1, and a tile of hit points across a seam
// is looked up on one face).  Claim: if tri_test accepts triangle T for (o, r) with |o| <= olim and a distance d in
// (1e-5, dmax), then (p, q) lies in one of T's rectangles on that face.  Then a ray none of whose texel's triangles blocks
// it is not blocked by any triangle at all, whatever cells its walk would have visited.
//
// Derivation.  With the triangle's vertices V0 = i, V1 = i + ij, V2 = i + ik as the test sees them (ij, ik rounded to
// binary32) and w_k = V_k - P, the edge functions of a direction x are E_0(x) = x.(w_1 x w_2), E_1 = x.(w_2 x w_0),
// E_2 = x.(w_0 x w_1); their sum is D(x) = x.(ij x ik), and E_k / D are the barycentric coordinates of the point where
// the line through P along x meets the triangle's plane.  tri_test's numerators are, exactly on its float inputs,
// tp* = r.(ik x (o - i)) = -E_1(r) + r.(ik x (a - P)), and likewise vp* and det* - tp* - vp* with E_2, E_0; det* = -D(r).
// r is, component by component, within 2 eps of -c fl(a - P) for a common factor c, and fl(a - P) within eps of a - P, so
// the last term is at most 3.1 eps L s (L the longest edge, s = |a - P| <= olim + |P|; twice that for the third numerator).
// If tri_test accepts, then sign(det*) X* >= -(31.5 eps L (|o| + A) + 17.8 eps L^2) for each of the three numerators X*
// (from its rounding bounds |det - det*| <= 8.9 eps L^2, |X - X*| <= 10.5 eps L (|o| + A), rt_math.h; A the largest vertex
// norm; when det's sign is wrong, |det*| <= 8.9 eps L^2 and the conditions u >= 0, v >= 0, u + v <= 1 bound all three from
// below with the other sign).  tol = 72 eps L (olim + A) + 28 eps L^2, half the filter's kc, is more than twice that.
// Dividing by c |d_w| (= |x| up to 1e-6 for the face point x = d / |d_w|, |x| <= sqrt(1 + 2 kFaceExt^2) < 2.0315):
//     sigma E_k(x) >= -T_k,   T_k = 2.0315 (tol + 6.4 eps L s) + 4.1 eps |w_a x w_b| + (double rounding),  sigma = sign(D(d))
// on the face plane x = (p, q, +-1): three half planes in (p, q).  A blocker has its plane point between a and the light
// unless the light lies within rounding of the plane: if |ij x ik| (h - delta) > 2.001 eps L^2 (8.9 s + 10.5 (olim + A))
// and h - delta > 12 eps s (h the light's distance from the plane, delta = 4 eps s what the ray's line misses the light
// by), a hit beyond the light would have d >= dmax; then sigma is the sign of D on the triangle itself, w_0.(ij x ik).
// Otherwise both signs are taken.  The face square, a little enlarged, is clipped by the three half planes; the bounding
// box of what is left, widened by 2e-5 for the binary32 evaluation of (p, q) and of the texel index, is the rectangle.
// tests/test_light_lists_cpu.py aims rays at edges, vertices and grazing triangles and checks the claim against the
// binary32 restatement of tri_test.
//
// Windows.  A face's texels cover a window [p0, p1] x [q0, q1] of the face, not the whole face: the part where the
// rectangles are (a light far outside the scene sees it under a small angle).  A few very large triangles (a ground
// plane) would blow the window up, so it is the box of the 1st to 99th percentile of the rectangles' edges unless that is
// nearly everything.  A ray outside its face's window is not answered from the lists (its ticket takes the walk).  The
// table has room for a second window per face, the whole face, for those rays (kWholeFaceWindow); it is left empty:
// seen from a far light the scene falls into a few of its texels, and a ticket that looks there goes through most of
// the scene's triangles (temple 3840x2160: the shadow kernel 1.56 -> 1.75 ms with it).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace rt {

struct LLRect { uint32_t face; uint32_t rect; int32_t tri; };  // face = (light * 6 + face) * 2 + window; rect = a0 | a1 << 8 | b0 << 16 | b1 << 24 (texel columns a, rows b)
struct LLFaceH { float p0, q0, sp, sq, p1, q1; uint32_t base, on; };  // texel of (p, q): column floor((p - p0) sp), row floor((q - q0) sq); base: index of the face's first texel

struct LightListsHost {
  int R = 0;
  float olim = 0.0f;
  std::vector<LLRect> rects;
  std::vector<LLFaceH> faces;  // nl * 6 * 2: per light and face window 0 and window 1
  uint64_t refs = 0;           // sum of the rectangles' areas
};

constexpr bool kWholeFaceWindow = false;  // window 1 of every face (see above)
constexpr double kFaceExt = 1.25;  // a face's coordinates run over [-kFaceExt, kFaceExt]: the faces overlap, so that hit points on both sides of a seam can be looked up on one face

namespace ll_detail {
struct D3 { double x, y, z; };
inline D3 sub(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline D3 cross(D3 a, D3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline double len(D3 a) { return std::sqrt(dot(a, a)); }
inline double comp(D3 a, int k) { return k == 0 ? a.x : k == 1 ? a.y : a.z; }
struct Poly { int n; double p[12], q[12]; };
// keeps the part of the polygon where a p + b q + c >= 0
inline void clip(Poly &P, double a, double b, double c) {
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
struct URect { int tri; double u0, u1, v0, v1; };
}  // namespace ll_detail

// The rectangle of one triangle (v: i, j, k as uploaded) on face f of the cube around the light P, for rays with |o| <= olim
// (already widened by the kernel's factor); false: none.
inline bool face_rect(const float *v, ll_detail::D3 P, double olim, int f, ll_detail::URect &out) {
  using namespace ll_detail;
  const double eps = 5.9604644775390625e-8;  // 2^-24
  const int ax = f >> 1, o0 = ax == 0 ? 1 : 0, o1 = ax == 2 ? 1 : 2;
  const double sg = (f & 1) ? -1.0 : 1.0;
  const double smax = olim + len(P);
  const double delta = 4.0 * eps * smax;
  const float ijf[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]}, ikf[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};  // binary32, as k_preprocess
  const D3 V0 = {v[0], v[1], v[2]}, ij = {ijf[0], ijf[1], ijf[2]}, ik = {ikf[0], ikf[1], ikf[2]};
  const D3 V1 = {V0.x + ij.x, V0.y + ij.y, V0.z + ij.z}, V2 = {V0.x + ik.x, V0.y + ik.y, V0.z + ik.z};
  const D3 w0 = sub(V0, P), w1 = sub(V1, P), w2 = sub(V2, P);
  const D3 G[3] = {cross(w1, w2), cross(w2, w0), cross(w0, w1)};
  const D3 Nn = cross(ij, ik);
  const double L = std::max(len(ij), std::max(len(ik), len(sub(ik, ij))));
  const double A = std::max(len(V0), std::max(len(V1), len(V2)));
  const double tol = (72.0 * eps * L * (olim + A) + 28.0 * eps * L * L) * 1.01;
  const double eta = 6.4 * eps * L * smax;
  const double nN = len(Nn);
  const double DT = dot(w0, Nn);  // D on the triangle's own directions; |DT| = |N| h
  bool one_sign = false;
  if (nN > 0.0) {
    const double h = std::fabs(DT) / nN;
    one_sign = h - delta > 12.0 * eps * smax && nN * (h - delta) > 2.001 * eps * L * L * (8.9 * smax + 10.5 * (olim + A));
  }
  double u0 = 1e30, u1 = -1e30, v0 = 1e30, v1 = -1e30;
  for (int pass = 0; pass < 2; pass++) {
    double sigma;
    if (one_sign) { if (pass) break; sigma = DT > 0.0 ? 1.0 : -1.0; } else sigma = pass ? -1.0 : 1.0;
    const double m = kFaceExt + 1e-4;
    Poly poly; poly.n = 4;
    poly.p[0] = -m; poly.q[0] = -m; poly.p[1] = m; poly.q[1] = -m;
    poly.p[2] = m; poly.q[2] = m; poly.p[3] = -m; poly.q[3] = m;
    for (int k = 0; k < 3 && poly.n > 0; k++) {
      const double gl = len(G[k]);
      const double T = 2.0315 * (tol + eta) + 4.1 * eps * gl + 1e-12 * (gl + 1.0);  // (|x| <= sqrt(1 + 2 kFaceExt^2) on the face)
      clip(poly, sigma * comp(G[k], o0), sigma * comp(G[k], o1), sigma * comp(G[k], ax) * sg + T);
    }
    for (int e = 0; e < poly.n; e++) {
      u0 = std::min(u0, poly.p[e]); u1 = std::max(u1, poly.p[e]);
      v0 = std::min(v0, poly.q[e]); v1 = std::max(v1, poly.q[e]);
    }
  }
  if (u0 > u1) return false;
  const double wdn = 2e-5;
  u0 -= wdn; u1 += wdn; v0 -= wdn; v1 += wdn;
  if (u1 < -kFaceExt || u0 > kFaceExt || v1 < -kFaceExt || v0 > kFaceExt) return false;
  out.tri = 0; out.u0 = std::max(u0, -kFaceExt); out.u1 = std::min(u1, kFaceExt); out.v0 = std::max(v0, -kFaceExt); out.v1 = std::min(v1, kFaceExt);
  return true;
}

// verts: nt x 9 floats (i, j, k as uploaded); lights: nl x 3 floats.  olim <= 0: chosen from the scene (1.5 times the
// 90th percentile of the vertex norms).
inline void build_light_rects(int nt, const float *verts, int nl, const float *lights, int R, float olim_in, LightListsHost &out) {
  using namespace ll_detail;
  out.R = R;
  out.rects.clear();
  out.faces.assign((size_t)nl * 12, LLFaceH{0, 0, 0, 0, 0, 0, 0u, 0u});
  out.refs = 0;
  if (nt <= 0 || nl <= 0) { out.olim = 0.0f; return; }
  double olim = olim_in;
  if (!(olim > 0.0)) {
    std::vector<double> norms((size_t)nt * 3);
    for (int t = 0; t < nt; t++)
      for (int v = 0; v < 3; v++) {
        const float *p = verts + 9 * (size_t)t + 3 * v;
        norms[(size_t)t * 3 + v] = std::sqrt((double)p[0] * p[0] + (double)p[1] * p[1] + (double)p[2] * p[2]);
      }
    const size_t k = norms.size() * 9 / 10;
    std::nth_element(norms.begin(), norms.begin() + k, norms.end());
    olim = 1.5 * norms[k] + 1e-3;
  }
  out.olim = (float)olim;
  olim = (double)out.olim * 1.0002;  // (the kernel compares its upper bound of |o| with the float)
  std::vector<URect> ur;
  std::vector<double> edges;
  for (int li = 0; li < nl; li++) {
    const D3 P = {lights[3 * li], lights[3 * li + 1], lights[3 * li + 2]};
    for (int f = 0; f < 6; f++) {
      ur.clear();
      for (int t = 0; t < nt; t++) {
        URect r;
        if (face_rect(verts + 9 * (size_t)t, P, olim, f, r)) { r.tri = t; ur.push_back(r); }
      }
      if (ur.empty()) continue;
      // the fine window
      double win[4];
      for (int side = 0; side < 4; side++) {
        edges.resize(ur.size());
        for (size_t i = 0; i < ur.size(); i++) edges[i] = side == 0 ? ur[i].u0 : side == 1 ? ur[i].u1 : side == 2 ? ur[i].v0 : ur[i].v1;
        const bool low = side == 0 || side == 2;
        size_t k = low ? edges.size() / 100 : edges.size() - 1 - edges.size() / 100;
        std::nth_element(edges.begin(), edges.begin() + k, edges.end());
        win[side] = edges[k];
      }
      {
        double lo_u = 1e30, hi_u = -1e30, lo_v = 1e30, hi_v = -1e30;
        for (const URect &r : ur) { lo_u = std::min(lo_u, r.u0); hi_u = std::max(hi_u, r.u1); lo_v = std::min(lo_v, r.v0); hi_v = std::max(hi_v, r.v1); }
        // nearly everything anyway: take everything
        if (win[1] - win[0] > 0.7 * (hi_u - lo_u)) { win[0] = lo_u; win[1] = hi_u; }
        if (win[3] - win[2] > 0.7 * (hi_v - lo_v)) { win[2] = lo_v; win[3] = hi_v; }
      }
      // window 0: where most rectangles are; window 1: the whole face, for the rays outside window 0 (left out when
      // window 0 is nearly the whole face already: then window 0 is the whole face)
      const bool whole = win[1] - win[0] > 1.8 * kFaceExt && win[3] - win[2] > 1.8 * kFaceExt;
      for (int k = 0; k < (kWholeFaceWindow ? 2 : 1); k++) {
        if (k == 1 && whole) break;
        double w[4] = {win[0], win[1], win[2], win[3]};
        if (k == 1 || whole) { w[0] = -kFaceExt; w[1] = kFaceExt; w[2] = -kFaceExt; w[3] = kFaceExt; }
        const uint32_t fi = (uint32_t)((li * 6 + f) * 2 + k);
        LLFaceH &F = out.faces[fi];
        const double wu = std::max(w[1] - w[0], 1e-4), wv = std::max(w[3] - w[2], 1e-4);
        F.p0 = (float)(w[0] - 1e-6); F.q0 = (float)(w[2] - 1e-6);
        F.p1 = (float)(w[0] + wu + 1e-6); F.q1 = (float)(w[2] + wv + 1e-6);
        F.sp = (float)(R / ((double)F.p1 - (double)F.p0) * 0.99999); F.sq = (float)(R / ((double)F.q1 - (double)F.q0) * 0.99999);
        F.on = 1u;
        const double p0 = F.p0, q0 = F.q0, p1 = F.p1, q1 = F.q1, sp = F.sp, sq = F.sq;
        auto texel = [R](double x, double x0, double sc) {
          const double t = std::floor((x - x0) * sc);
          return (int)std::min(std::max(t, 0.0), (double)(R - 1));
        };
        for (const URect &r : ur) {
          if (r.u1 < p0 || r.u0 > p1 || r.v1 < q0 || r.v0 > q1) continue;
          const int a0 = texel(std::max(r.u0, p0), p0, sp), a1 = texel(std::min(r.u1, p1), p0, sp);
          const int b0 = texel(std::max(r.v0, q0), q0, sq), b1 = texel(std::min(r.v1, q1), q0, sq);
          out.rects.push_back(LLRect{fi, (uint32_t)a0 | (uint32_t)a1 << 8 | (uint32_t)b0 << 16 | (uint32_t)b1 << 24, r.tri});
          out.refs += (uint64_t)(a1 - a0 + 1) * (uint64_t)(b1 - b0 + 1);
        }
      }
    }
  }
  uint32_t base = 0;
  for (LLFaceH &F : out.faces) { F.base = base; if (F.on) base += (uint32_t)R * (uint32_t)R; }
}

}  // namespace rt
