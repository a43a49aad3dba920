// Device arithmetic of the trace path.
//
// The reference is strict IEEE binary32 with no fused multiply-add, except for a handful of spots
// where C's promotion rules make it compute in double; a flipped last bit in any geometric predicate
// changes which triangle a ray hits.  Every helper here therefore spells out the reference's
// operation order (file:line under /root/reference/src) and the file is compiled with
// -ffp-contract=off, correctly rounded division/sqrt and denormals preserved.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <limits.h>
#include "rt_types.h"

#define RT_D __device__ __forceinline__

namespace rt {

RT_D f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_D f3 ld3(const float *p) { return mk3(p[0], p[1], p[2]); }
RT_D void st3(float *p, f3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
RT_D f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_D f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_D f3 mul3(f3 a, float t) { return mk3(a.x * t, a.y * t, a.z * t); }
RT_D f3 neg3(f3 a) { return mk3(-a.x, -a.y, -a.z); }

// vectormath.h:95-97: (a0*b0 + a1*b1) + a2*b2
RT_D float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// vectormath.h:87-92
RT_D f3 cross3(f3 a, f3 b) {
  return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// vectormath.h:60-66: `1.0f/sqrt(float)` promotes to double: sqrt and the division are binary64, the
// quotient is rounded to float once.  rsqrt_exact is that chain as written; rsqrt_ref gives the same float for every
// input with a fifth of the double-precision instructions: y = 1/sqrt(s) from the hardware's estimate and two Newton
// steps is within 2^-47 of the true value, and so is the chain's double result (two correctly rounded operations), so
// both round to the same float unless a midpoint between two floats lies within 2^-46 of y: the low 29 bits of y's
// mantissa then lie within 2^6 of 2^28.  Those inputs (a window of 2^11, four in a million) and everything outside
// the plain normal range take the chain itself.  rtgpu_math_selfcheck compares the two for all 2^32 inputs.
RT_D float rsqrt_exact(float s) { return (float)(1.0 / sqrt((double)s)); }
#ifndef RT_FAST_F64
#define RT_FAST_F64 1
#endif
RT_D float rsqrt_ref(float s) {
#if RT_FAST_F64
  if (s > 1e-30f && s < 1e30f) {  // (a NaN fails both)
    const double x = (double)s;
    double y = __builtin_amdgcn_rsq(x);
#pragma unroll
    for (int it = 0; it < 2; it++) {
      const double g = x * y, h = 0.5 * y;
      y = __builtin_fma(y, __builtin_fma(-g, h, 0.5), y);
    }
    const uint32_t low = (uint32_t)(unsigned long long)__double_as_longlong(y) & 0x1fffffffu;
    if (low - (0x10000000u - 1024u) > 2048u) return (float)y;
  }
#endif
  return rsqrt_exact(s);
}
RT_D f3 normalize3(f3 v) {
  float s = v.x * v.x + v.y * v.y + v.z * v.z;
  float inv = rsqrt_ref(s);
  return mk3(v.x * inv, v.y * inv, v.z * inv);
}
// vectormath.h:122-127: normalised b - a
RT_D f3 ray3(f3 a, f3 b) { return normalize3(sub3(b, a)); }
// vectormath.h:45-48: sqrt evaluated in double, result rounded to float.  For a float argument that is the correctly
// rounded binary32 square root: rounding a square root first to 53 and then to 24 bits cannot differ from rounding it
// to 24 bits at once (53 >= 2 x 24 + 2), and the compiler's sqrtf is the correctly rounded one here (also compared for
// all 2^32 inputs by rtgpu_math_selfcheck).
RT_D float sqrt_exact(float s) { return (float)sqrt((double)s); }
RT_D float sqrt_ref(float s) {
#if RT_FAST_F64
  return __builtin_sqrtf(s);
#else
  return sqrt_exact(s);
#endif
}
RT_D float dist3(f3 a, f3 b) {
  float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
  return sqrt_ref(dx * dx + dy * dy + dz * dz);
}
// vectormath.h:131-136: o + d*r
RT_D f3 raypoint3(f3 o, f3 r, float d) { return mk3(o.x + d * r.x, o.y + d * r.y, o.z + d * r.z); }
// vectormath.h:144-159: normalize((2*n)*ndl - l)
RT_D f3 reflect3(f3 n, f3 l, float ndl) {
  return normalize3(mk3(2.0f * n.x * ndl - l.x, 2.0f * n.y * ndl - l.y, 2.0f * n.z * ndl - l.z));
}
// vectormath.h:162-169: the sqrt is binary64 and so is the subtraction it feeds; f is rounded once.
// Total internal reflection gives sqrt(<0) = NaN and an all-NaN direction, exactly as on the CPU.
RT_D f3 refract3(f3 n, f3 l, float eta) {
  float ndl = dot3(n, l);
  float arg = 1.0f - (eta * eta) * (1.0f - ndl * ndl);
  float f = (float)((double)(eta * ndl) - sqrt((double)arg));
  return normalize3(mk3(f * n.x - eta * l.x, f * n.y - eta * l.y, f * n.z - eta * l.z));
}

RT_D float absf_ref(float x) { return x >= 0.0f ? x : -x; }  // common.h:27
// common.h:35 MIN(a,b,c) with its exact comparison structure (matters when a NaN is involved)
RT_D float min3_ref(float a, float b, float c) {
  float ab = a < b ? a : b;
  return c < ab ? c : ab;
}
RT_D int min3i(int a, int b, int c) { int ab = a < b ? a : b; return c < ab ? c : ab; }
RT_D int max3i(int a, int b, int c) { int ab = a > b ? a : b; return c > ab ? c : ab; }

// float -> int32 the way the reference binary does it (x86-64 cvttss2si): truncation, and INT_MIN for
// NaN or out-of-range input.  The GPU conversion saturates and maps NaN to 0, which would put a NaN
// point INSIDE voxel 0.
RT_D int f2i_x86(float x) {
  return (x > -2147483904.0f && x < 2147483648.0f) ? (int)x : INT_MIN;
}

// voxelize.h:32-43
RT_D bool voxel_of(const GridP &g, f3 p, int &i, int &j, int &k) {
  i = f2i_x86((p.x - g.dmin[0]) / g.cs[0]);
  j = f2i_x86((p.y - g.dmin[1]) / g.cs[1]);
  k = f2i_x86((p.z - g.dmin[2]) / g.cs[2]);
  if (i < 0 || i >= g.nv[0]) return false;
  if (j < 0 || j >= g.nv[1]) return false;
  if (k < 0 || k >= g.nv[2]) return false;
  return true;
}
// voxelize.h:47-49
RT_D int cell_index(const GridP &g, int i, int j, int k) { return (i * g.nv[1] + j) * g.nv[2] + k; }

// voxelize.c:417-471
RT_D bool startup_voxel(const GridP &g, f3 o, f3 r, int &i, int &j, int &k) {
  if (voxel_of(g, o, i, j, k)) return true;
  float d1 = FLT_MAX, d2 = FLT_MAX;
  const float oo[3] = {o.x, o.y, o.z}, rr[3] = {r.x, r.y, r.z};
#pragma unroll
  for (int a = 0; a < 3; a++) {
    if (rr[a] != 0.0f) {
      float d = (g.dmin[a] - oo[a]) / rr[a];
      if (d > 0.0f) {
        if (d < d1) { d2 = d1; d1 = d; } else if (d < d2) d2 = d;
      }
      d = (g.dmax[a] - oo[a]) / rr[a];
      if (d > 0.0f) {
        if (d < d1) { d2 = d1; d1 = d; } else if (d < d2) d2 = d;
      }
    }
  }
  if (voxel_of(g, raypoint3(o, r, d1), i, j, k)) return true;
  if (voxel_of(g, raypoint3(o, r, d2), i, j, k)) return true;
  return false;
}

// 3D-DDA state (voxelize.c:56-121).  tx/ty/tz are ACCUMULATED sums exactly like the reference's.
struct Dda {
  float tx, ty, tz, dtx, dty, dtz;
  int i, j, k;
  int di, dj, dk;
};

RT_D void dda_axis(float lo, float cs, int idx, float o, float r, float &dt, float &t, int &step) {
  float p1 = lo + idx * cs;
  float p2 = p1 + cs;
  if (r == 0.0f) {
    dt = FLT_MAX;
    t = 0.0f;
  } else {
    float t1 = (p1 - o) / r;
    float t2 = (p2 - o) / r;
    dt = absf_ref(t2 - t1);
    t = t1 < t2 ? t1 : t2;
  }
  step = r > 0.0f ? 1 : -1;
}
RT_D void dda_init(const GridP &g, f3 o, f3 r, int i, int j, int k, Dda &w) {
  w.i = i; w.j = j; w.k = k;
  dda_axis(g.dmin[0], g.cs[0], i, o.x, r.x, w.dtx, w.tx, w.di);
  dda_axis(g.dmin[1], g.cs[1], j, o.y, r.y, w.dty, w.ty, w.dj);
  dda_axis(g.dmin[2], g.cs[2], k, o.z, r.z, w.dtz, w.tz, w.dk);
}
// voxelize.c:594-606 (ties go to the later axis)
RT_D void dda_step(Dda &w) {
  // written with selects: the walk kernels run this in per-lane loops, where branches cost exec-mask bookkeeping
  const float xn = w.tx + w.dtx, yn = w.ty + w.dty, zn = w.tz + w.dtz;
  const bool xy = xn < yn;
  const bool sx = xy & (xn < zn), sy = !xy & (yn < zn);
  const bool sz = !(sx | sy);
  w.i += sx ? w.di : 0; w.tx = sx ? xn : w.tx;
  w.j += sy ? w.dj : 0; w.ty = sy ? yn : w.ty;
  w.k += sz ? w.dk : 0; w.tz = sz ? zn : w.tz;
}

// The walk kernels' form of the same state: the voxel as its cell index, and per axis the number of steps left
// before the walk leaves its index range (the grid for a nearest-hit walk, the box spanned by the two end voxels
// for a shadow walk, voxelize.c:607-611 and :727-735).  An axis only ever moves in the direction of its ray
// component, so "steps left >= 0" is the reference's two-sided range test.  10 registers instead of 19, which is
// what keeps the stepping loops free of spills.
struct Walk {
  float tx, ty, tz, dtx, dty, dtz;
  int cell;
  int nx, ny, nz;
};
// clamp: the range is cut down to the box of the non-empty cells; a walk that leaves that box meets no triangle
// any more (each axis moves one way only), so ending it there changes no result, only the number of steps.
// occ_box: that box, lo[3] then hi[3] (g.occ_box, or the kernel's copy of it in LDS: one trip to memory less per ticket).
RT_D void walk_init(const GridP &g, f3 o, f3 r, int i, int j, int k, const int lo_[3], const int hi_[3], bool clamp, const int32_t *occ_box, Walk &w) {
  int lo[3] = {lo_[0], lo_[1], lo_[2]}, hi[3] = {hi_[0], hi_[1], hi_[2]};
  if (clamp) {
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int ol = occ_box[q], oh = occ_box[3 + q];
      lo[q] = lo[q] > ol ? lo[q] : ol;
      hi[q] = hi[q] < oh ? hi[q] : oh;
    }
  }
  int di, dj, dk;
  dda_axis(g.dmin[0], g.cs[0], i, o.x, r.x, w.dtx, w.tx, di);
  dda_axis(g.dmin[1], g.cs[1], j, o.y, r.y, w.dty, w.ty, dj);
  dda_axis(g.dmin[2], g.cs[2], k, o.z, r.z, w.dtz, w.tz, dk);
  w.cell = cell_index(g, i, j, k);
  w.nx = di > 0 ? hi[0] - i : i - lo[0];
  w.ny = dj > 0 ? hi[1] - j : j - lo[1];
  w.nz = dk > 0 ? hi[2] - k : k - lo[2];
}
// One step (voxelize.c:594-606, ties go to the later axis); false = the walk has left its range (cell is then invalid).
RT_D bool walk_step(const GridP &g, f3 r, Walk &w) {
  const float xn = w.tx + w.dtx, yn = w.ty + w.dty, zn = w.tz + w.dtz;
  const bool xy = xn < yn;
  const bool sx = xy & (xn < zn), sy = !xy & (yn < zn);
  const int stride_x = g.nv[1] * g.nv[2], stride_y = g.nv[2];
  w.tx = sx ? xn : w.tx;
  w.ty = sy ? yn : w.ty;
  w.tz = (sx | sy) ? w.tz : zn;
  w.nx -= sx ? 1 : 0;
  w.ny -= sy ? 1 : 0;
  w.nz -= (sx | sy) ? 0 : 1;
  const int along = sx ? stride_x : sy ? stride_y : 1;
  const bool forward = sx ? r.x > 0.0f : sy ? r.y > 0.0f : r.z > 0.0f;  // step = r > 0 ? 1 : -1 (voxelize.c:66-118)
  w.cell += forward ? along : -along;
  return (w.nx | w.ny | w.nz) >= 0;
}
RT_D void cell_to_voxel(const GridP &g, int cell, int &i, int &j, int &k) {
  k = cell % g.nv[2];
  j = (cell / g.nv[2]) % g.nv[1];
  i = cell / (g.nv[2] * g.nv[1]);
}

// Möller–Trumbore exactly as intersection.c:58-86 evaluates it.  full (optional, statistics): the test ran to its
// last line, the division that gives d (45 flops; a test that ends at the determinant costs 14).
RT_D bool tri_test(f3 a, f3 ab, f3 ac, f3 o, f3 r, float &d, float &u, float &v, bool *full = nullptr) {
  if (full) *full = false;
  f3 p = cross3(r, ac);
  float det = dot3(ab, p);
  if (det > -0.000001f && det < 0.000001f) return false;
  float inv = 1.0f / det;
  f3 tv = sub3(o, a);
  u = dot3(tv, p) * inv;
  if (u < 0.0f || u > 1.0f) return false;
  f3 q = cross3(tv, ab);
  v = dot3(r, q) * inv;
  if (v < 0.0f || u + v > 1.0f) return false;
  if (full) *full = true;
  d = dot3(ac, q) * inv;
  if (d < 0.0f) return false;
  return true;
}

// ------------------------------------------------------------------------------------------------
// Staged conservative filter in front of tri_test.  It never decides differently from tri_test: a stage
// either proves what tri_test would answer or hands the triangle on; what no stage can prove goes to
// tri_test itself.  All stages are branch-free FMA chains on per-triangle constants (k_preprocess) and the
// ray's Pluecker coordinates (r, m = o x r).
//
// Geometry.  With det = ij.(r x ik), tp = (o-i).(r x ik), vp = r.((o-i) x ij), dp = ik.((o-i) x ij), tri_test
// accepts iff |det| >= 1e-6, u = tp/det in [0,1], v = vp/det >= 0, u+v <= 1, d = dp/det >= 0.  By the cyclic
// identity of the triple product
//     det = r.N            N = ik x ij
//     tp  = m.ik - r.Q     Q = ik x i
//     vp  = -(m.ij + r.P)  P = i x ij
//     dp  = i.N - o.N
// and (u, v, 1-u-v) all lie in [0,1] exactly when the ray pierces the triangle, whichever vertex is taken
// as i: the filter may label the vertices as it likes.  k_preprocess labels them so that ik is the longest
// edge, which makes the first slab, u in [0,1], the narrowest of the three.
//     x in [0,1]  <=>  |2 x det - det| <= |det|      (no division, no sign case)
//
// Rounding.  eps = 2^-24, L = longest edge, A = largest vertex norm, |r| <= 1+1e-6, x* = exact value on
// the float inputs.  tri_test's route (no FMA):   |det-det*| <= 8.9 eps L^2,  |tp-tp*|, |vp-vp*| <= 10.5 eps L (|o|+A),
// |dp-dp*| <= 10.5 eps L^2 (|o|+A).  This route (inputs rounded once from double, FMA chains):
// |det'-det*| <= 4.1 eps L^2, |tp'-tp*|, |vp'-vp*| <= 7.2 eps L (|o|+A), |dp'-dp*| <= 5.1 eps L^2 (|o|+A).
// (Standard dot-product bounds gamma_n sum|x_i y_i| with sum|x_i y_i| <= |x||y|.)  If tri_test accepts then
// for X in {tp, vp, det-tp-vp}:  |2X'-det'| <= |det'| (1 + 7 eps) + 71 eps L (|o|+A) + 28 eps L^2.
// k_preprocess stores k1 = 144 eps L, k2 = 144 eps L A + 56 eps L^2 (twice the bound); kc = k1 |o|' + k2.
//   reject         some |2X'-det'| > 1.00001 |det'| + kc          => tri_test rejects
//   surely inside  all  |2X'-det'| <= 0.99999 |det'| - kc, |det'| - kc > 1.01e-6
//                                                                 => tri_test's det, u, v tests pass
//   distance       d = dp/det against a bound c with E = lmax kc >= 9 x (|dp'-dp| bound) and |det'-det| <= kc/4:
//                  s dp' - E > c (|det'| + kc) => d > c,   s dp' + E < c (|det'| - kc) => d < c,  s = sign det'
// Anything a NaN touches compares false everywhere and ends in tri_test, like every uncertain case.
// A lane that must not take part passes kc = -inf: anything but a NaN is > -inf, rejected by the first slab
// (a NaN ray is never blocked and never finds a hit nearer than a number, so it has nothing to lose there).
struct PlTest {
  float det, tp, x1, x2, x3, T;
};
// first slab: 12 instructions
RT_D bool pl_stage1(f3 N, f3 ac, f3 Q, f3 r, f3 m, float kc, PlTest &t) {
  t.det = __builtin_fmaf(r.x, N.x, __builtin_fmaf(r.y, N.y, r.z * N.z));
  const float rq = __builtin_fmaf(r.x, Q.x, __builtin_fmaf(r.y, Q.y, r.z * Q.z));
  t.tp = __builtin_fmaf(m.x, ac.x, __builtin_fmaf(m.y, ac.y, __builtin_fmaf(m.z, ac.z, -rq)));
  t.x1 = __builtin_fmaf(2.0f, t.tp, -t.det);
  t.T = __builtin_fmaf(__builtin_fabsf(t.det), 1.00001f, kc);
  return !(__builtin_fabsf(t.x1) > t.T);
}
// the other two slabs: 11 instructions
RT_D bool pl_stage2(f3 ab, f3 P, f3 r, f3 m, PlTest &t) {
  const float rp = __builtin_fmaf(r.x, P.x, __builtin_fmaf(r.y, P.y, r.z * P.z));
  const float nv = __builtin_fmaf(m.x, ab.x, __builtin_fmaf(m.y, ab.y, __builtin_fmaf(m.z, ab.z, rp)));  // -vp
  t.x2 = __builtin_fmaf(-2.0f, nv, -t.det);
  t.x3 = __builtin_fmaf(2.0f, t.tp - nv, -t.det);
  return !(__builtin_fmaxf(__builtin_fabsf(t.x2), __builtin_fabsf(t.x3)) > t.T);
}
RT_D bool pl_surely_inside(const PlTest &t, float kc) {
  const float ad = __builtin_fabsf(t.det);
  const float w = __builtin_fmaxf(__builtin_fabsf(t.x1), __builtin_fmaxf(__builtin_fabsf(t.x2), __builtin_fabsf(t.x3)));
  return (w <= __builtin_fmaf(ad, 0.99999f, -kc)) & (ad - kc > 1.01e-6f);
}
// sign of det' is the sign of tri_test's det, and that one passes the 1e-6 test
RT_D bool pl_sign_sure(const PlTest &t, float kc) { return __builtin_fabsf(t.det) - kc > 1.01e-6f; }
// s * dp': the numerator of d with the sign of det' folded in
RT_D float pl_sdist(f3 N, float an, f3 o, const PlTest &t) {
  const float dp = __builtin_fmaf(-o.x, N.x, __builtin_fmaf(-o.y, N.y, __builtin_fmaf(-o.z, N.z, an)));
  return __int_as_float(__float_as_int(dp) ^ (__float_as_int(t.det) & (int)0x80000000));
}
// Per-ray side of the filter: m = o x r (double, rounded once), an upper bound of |o|, and o.r.
// The bound and o.r in binary32: the sum of squares and its root are off by less than 3 eps, far inside the factor
// 1.0001 (an overflow gives +inf: nothing is culled; squares that underflow, |o| < 1e-18, are covered by the addend);
// o.r is only used by the box tests, whose delta = 1e-5 (2 |o| + ...) is far above its 3 eps |o|.
RT_D void pl_ray(f3 o, f3 r, f3 &m, float &onorm, float &odr) {
  const double ox = o.x, oy = o.y, oz = o.z, rx = r.x, ry = r.y, rz = r.z;
  m = mk3((float)(oy * rz - oz * ry), (float)(oz * rx - ox * rz), (float)(ox * ry - oy * rx));
#if RT_FAST_F64
  onorm = __builtin_fmaf(__builtin_sqrtf(__builtin_fmaf(o.x, o.x, __builtin_fmaf(o.y, o.y, o.z * o.z))), 1.0001f, 1e-18f);
  odr = __builtin_fmaf(o.x, r.x, __builtin_fmaf(o.y, r.y, o.z * r.z));
#else
  onorm = (float)sqrt(ox * ox + oy * oy + oz * oz) * 1.0001f;
  odr = (float)(ox * rx + oy * ry + oz * rz);
#endif
}

// float -> uint32_t as the reference binary converts the tone-mapped channel (64-bit cvttss2si, low
// half kept): NaN and out-of-range give 0, negatives wrap.
RT_D uint32_t f2u_x86(float x) {
  if (!(x > -9223373136366403584.0f && x < 9223372036854775808.0f)) return 0u;
  return (uint32_t)(unsigned long long)(long long)x;
}

}  // namespace rt
