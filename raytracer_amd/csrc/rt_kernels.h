// HIP kernels of the trace path (gfx950, wave64).  Included once by rtgpu.hip.
//
// Frame set-up          k_preprocess, k_tri_boxes -> k_grid_offsets -> k_grid_fill -> k_grid_sort -> k_grid_bounds
// Pixel loop (a chunk)  k_chunk_init -> [k_nearest -> k_hit] per recursion level (level 0 computes its primary rays)
//                       [-> k_subtree_hits, k_tree_starts, k_assign_ranks with planar lights]
//                       -> [k_order_count -> k_order_scan -> k_order_place below level 0 ->] k_shadow per level
//                       -> k_accumulate per level, deepest first (level 0 folds the colour range of its blocks) -> k_minmax_final
// After the exchange    k_tonemap, k_untile, k_bmp_rows
#pragma once
#include "rt_math.h"
#include "rt_texture.h"
#include "rt_viewlists.h"

#ifndef RT_DENSE
#define RT_DENSE 7  // a batch with at least this many records left by the bundle's test has all its records tested in fixed groups of four
#endif
#ifndef RT_SHADOW_RAYCULL
// Does the shadow walk test the candidates' boxes ray by ray before the record-level test of the bundle?  Where the
// rays of a ticket start on one tile the bundle's own box test lets little through that a single ray would not meet,
// and the record-level test drops the rest (temple 4K 4.20 -> 4.12 ms without); where they start scattered and meet at
// the light (the kernels built with ENDS) it pays (ulica 1080p 116.5 against 118.9 ms without).
#define RT_SHADOW_RAYCULL(RECTEST, ENDS) ((ENDS) || !(RECTEST))
#endif
#ifndef RT_FLAT
constexpr float kLLFaceExt = 1.25f;  // (rt_lightlists.h: kFaceExt)
#ifndef RT_LL_TEXELS
#define RT_LL_TEXELS 4  // a ticket takes its lists from the finest level at which its hit points cover at most this many texels
#endif
#ifndef RT_WAVES_LL
#define RT_WAVES_LL 6  // waves per SIMD of the shadow walk's instance with the light lists
#endif
#define RT_FLAT 64  // lists of up to this many records are culled record by record, 64 at a time, without the batch level (stream_cell; 0: never)
#endif
#ifndef RT_WAVES_ACC
#define RT_WAVES_ACC 4  // k_accumulate: 128 instead of 131 registers buy a fourth wave per SIMD (0.27 -> 0.23 ms); five spill (0.51 ms)
#endif
#ifndef RT_WAVES_ACC_PLAIN
#define RT_WAVES_ACC_PLAIN 8  // k_accumulate without the specular term's pow (scenes without a surface with ks > 0): 62 registers
#endif
#ifndef RT_WAVES_HIT
#define RT_WAVES_HIT 4  // k_hit: five waves per SIMD spill (0.14 -> 0.28 ms)
#endif
#ifndef RT_WAVES
#define RT_WAVES 6  // waves per SIMD the shadow walk is compiled for (80 VGPRs; with 8 and 64 VGPRs it spills and runs 6 % slower)
#endif
#ifndef RT_WAVES_ENDS
#define RT_WAVES_ENDS 7  // ... and its instances for rays that end at their lights (scenes with secondary rays): ulica 1080p 105 -> 101 ms with seven (72 VGPRs), where temple's instance loses 6 %
#endif
#ifndef RT_WAVES_VIEW
#define RT_WAVES_VIEW 5  // ... and its instance that answers the primary rays from the view lists
#endif
#ifndef RT_WAVES_NEAREST
#define RT_WAVES_NEAREST 5  // ... and the nearest-hit walk: it keeps more per ray (the best hit), 96 VGPRs hold it without spills (4 % faster than 6 waves)
#endif

namespace rt {

RT_D int lane_id() { return threadIdx.x & 63; }
RT_D unsigned long long lanes_below(int lane) { return (1ull << lane) - 1ull; }

// A cell's list reaches the lanes of a wavefront through the wavefront's private LDS slice: batches of kBatch records
// are copied there with coalesced 16-byte direct-to-LDS loads, then every lane reads the same record back (LDS
// broadcast).  LDS operations of one wavefront execute in issue order, so no barrier is needed between the copy and
// the reads, nor before the next batch overwrites the slice (stream_cell).
using v4f = float __attribute__((ext_vector_type(4)));
using v2f = float __attribute__((ext_vector_type(2)));
constexpr int kRecVecs = kRecFloats / 4;
constexpr int kStageLoads = (kBatch * kRecVecs + 63) / 64;

// 16 bytes per lane from global memory straight into LDS (global_load_lds_dwordx4): the LDS address is
// the wave-uniform `lds_base` plus 16 x lane.
RT_D void glds16(const float4 *src, float *lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                   (__attribute__((address_space(3))) void *)lds_base, 16, 0, 0);
}

// True if the predicate holds in any active lane.  The result goes through an opaque scalar register so
// that the branch on it stays a scalar branch (otherwise it is merged into the per-lane predicate and
// the common path pays for the exec-mask bookkeeping of the rare one).
RT_D bool wave_any(bool p) {
  unsigned long long m = __ballot(p);
  asm volatile("" : "+s"(m));
  return m != 0;
}

// The lanes in which the predicate holds, through an opaque scalar register like wave_any.
RT_D unsigned long long wave_mask(bool p) {
  unsigned long long m = __ballot(p);
  asm volatile("" : "+s"(m));
  return m;
}

RT_D float lane_bcast(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
RT_D f3 lane_bcast3(f3 v, int l) { return mk3(lane_bcast(v.x, l), lane_bcast(v.y, l), lane_bcast(v.z, l)); }

// Minima / maxima across lanes with DPP moves (no LDS traffic).  row_min16: every 16th lane (15, 31, 47, 63)
// ends up with the minimum of its row of 16 lanes; wave_max: the maximum of all 64 lanes, in every lane.
template <int CTRL, int ROW_MASK, bool MAX>
RT_D float dpp_minmax(float x) {
  const float y = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), CTRL, ROW_MASK, 0xf, false));
  return MAX ? fmaxf(x, y) : fminf(x, y);
}
template <bool MAX, int WIDTH = 16>
RT_D float row_minmax16(float x) {
  x = dpp_minmax<0x111, 0xf, MAX>(x);  // row_shr:1
  x = dpp_minmax<0x112, 0xf, MAX>(x);  // row_shr:2
  x = dpp_minmax<0x114, 0xf, MAX>(x);  // row_shr:4
  if (WIDTH == 16) x = dpp_minmax<0x118, 0xf, MAX>(x);  // row_shr:8
  return x;  // WIDTH 8: lanes 7 and 15 of a row hold the result of their half
}
RT_D float wave_max(float x);
// The lane whose walk has made the least progress (smallest entry parameter of its current cell) among the lanes
// of mask m: the rays of a wavefront cross cell boundaries at slightly different steps, and letting the stragglers
// go first brings them into the cells where the others wait, so that a cell is visited by one large group instead
// of several small ones.
RT_D int least_advanced(unsigned long long m, bool alive, float tx, float ty, float tz);
RT_D float wave_max(float x) {
  x = row_minmax16<true>(x);
  x = dpp_minmax<0x142, 0xa, true>(x);  // row_bcast:15 into rows 1 and 3
  x = dpp_minmax<0x143, 0xc, true>(x);  // row_bcast:31 into rows 2 and 3
  return lane_bcast(x, 63);
}

RT_D int least_advanced(unsigned long long m, bool alive, float tx, float ty, float tz) {
  const float key = alive ? __builtin_fmaxf(tx, __builtin_fmaxf(ty, tz)) : INFINITY;
  const float least = -wave_max(-key);
  const unsigned long long who = __ballot(alive && key == least);
  return __ffsll(who != 0 ? who : m) - 1;  // (a NaN key equals nothing: then the first lane of the mask)
}

// The lane a visit's bundle is described about: the spreads about a ray in the middle of the group are about half of
// those about a ray at its rim (temple 4K: shadow walk 4.37 -> 4.21 ms).  tile: the wavefront's rays are the pixels of an
// 8x8 tile or the shadow rays of their hit points (lane = 8 y + x): the lane of the group nearest the tile's middle.
// Otherwise the rays come in queue order (secondary rays, nodes ordered by the cell of their hit point), in which
// neighbours lie close together: the lane of the group nearest the middle of the wavefront.
RT_D int central_lane(bool in, int lane, bool tile, int fallback) {
  if (!tile) {  // (wave-uniform) the set bit of the group's mask nearest the middle, of two equally near the lower: scalar bit arithmetic
    const unsigned long long m = __ballot(in);
    const uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
    const int below = lo != 0u ? 31 - __clz((int)lo) : -1000, above = hi != 0u ? 31 + __ffs((int)hi) : 1000;
    return m == 0ull ? fallback : (63 - 2 * below <= 2 * above - 63 ? below : above);
  }
  asm volatile("" : "+v"(lane));  // (the key is worked out per visit: as a loop invariant it would be spilled and reloaded)
  const int x2 = 2 * (lane & 7) - 7, y2 = 2 * (lane >> 3) - 7, l2 = 2 * lane - 63;
  const int off = tile ? x2 * x2 + y2 * y2 : (l2 < 0 ? -l2 : l2);
  const float key = in ? (float)(off * 64 + lane) : INFINITY;  // (an integer below 2^13: exact)
  const float least = -wave_max(-key);
  return least < INFINITY ? ((int)least & 63) : fallback;
}

// The rays a wavefront walks through one cell, as a bundle around the leader's ray (direction axis, moment
// m = o x axis, odr = o.axis): every ray of the group has |r - axis| <= rho, |o - o_leader| <= ro, |o| <= onorm
// and a parameter range that ends before tmax.  Rays that end somewhere (shadow rays, at their lights; tend = the
// ray's own last parameter) also have end points o + tend r within re of the leader's, whose range is [0, T]: the
// bundle of a point light's shadow rays narrows towards the light.  Rays without an end: re = T = infinity.
// The shadow rays of ONE point light, as a fan: the ray from a hit point a to the light L has the direction (L - a) / s,
// s = |L - a|, and the moment a x (L - a) / s = (a x L) / s, so s times each of the filter's quantities is an AFFINE
// function of a: s det = (L - a).N, s tp = a.(L x ac + Q) - L.Q, s nv = a.(L x ab - P) + L.P (nv = -vp).  Over the box
// c +- h of the group's hit points such a function stays within sum_k |gradient_k| h_k of its value at c, with the
// gradients -N, Gt = L x ac + Q, Gv = L x ab - P known exactly: no Lipschitz bound with the longest edge, the fan's
// narrowing towards the light and a flat patch of hit points are accounted for by themselves (fan_slabs_pass).
// c +- h: the box of the hit points; L: the light; smax >= every ray's s.  on = false: no fan (rays of several lights,
// a planar light's samples, a NaN among the hit points, the nearest-hit walk).
struct Fan { bool on; f3 c, h, rc, mc, L; float smax, hsum, lnorm; };  // c +- h: the box; rc = L - c, mc = c x rc; hsum = h.x + h.y + h.z, lnorm >= |L|
struct Bundle { f3 axis; float rho; f3 m; float odr, onorm, ro, tmax, re, T; Fan fan; };
RT_D Fan no_fan() { Fan f; f.on = false; f.c = f.h = f.rc = f.mc = f.L = mk3(0.0f, 0.0f, 0.0f); f.smax = f.hsum = f.lnorm = 0.0f; return f; }
RT_D Fan make_fan(bool in, f3 a, f3 L, float smax) {
  Fan f;
  f.on = __ballot(in && (a.x != a.x || a.y != a.y || a.z != a.z)) == 0;  // (a NaN would be dropped by the maxima)
  const f3 lo = mk3(-wave_max(in ? -a.x : -INFINITY), -wave_max(in ? -a.y : -INFINITY), -wave_max(in ? -a.z : -INFINITY));
  const f3 hi = mk3(wave_max(in ? a.x : -INFINITY), wave_max(in ? a.y : -INFINITY), wave_max(in ? a.z : -INFINITY));
  const f3 c = mk3(0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z));
  // half extents that surely reach both ends from the rounded centre
  const f3 h = mk3(__builtin_fmaf(__builtin_fmaxf(hi.x - c.x, c.x - lo.x), 1.000001f, 1e-30f),
                   __builtin_fmaf(__builtin_fmaxf(hi.y - c.y, c.y - lo.y), 1.000001f, 1e-30f),
                   __builtin_fmaf(__builtin_fmaxf(hi.z - c.z, c.z - lo.z), 1.000001f, 1e-30f));
  const f3 rc = sub3(L, c);
  const f3 mc = mk3(__builtin_fmaf(c.y, rc.z, -(c.z * rc.y)), __builtin_fmaf(c.z, rc.x, -(c.x * rc.z)), __builtin_fmaf(c.x, rc.y, -(c.y * rc.x)));
  // (wave-uniform values that vector instructions produced: once per visit, back into scalar registers, so that the
  // record test reads them as scalar operands instead of holding fifteen vector registers)
  f.c = lane_bcast3(c, 0); f.h = lane_bcast3(h, 0); f.rc = lane_bcast3(rc, 0); f.mc = lane_bcast3(mc, 0); f.L = lane_bcast3(L, 0);
  f.smax = lane_bcast(smax, 0);
  f.hsum = lane_bcast(h.x + h.y + h.z, 0);
  f.lnorm = lane_bcast(__builtin_fabsf(L.x) + __builtin_fabsf(L.y) + __builtin_fabsf(L.z), 0);
  return f;
}
// TICKET: onorm_all, tmax_all are wave-uniform upper bounds of |o| and of the range's end over ALL the ticket's rays, worked
// out once per ticket (they enter margins and the opaque range test only); else they are worked out here, for the group.
// one_end: every ray of the ticket ends at the same point (the shadow rays of one point light): the end points o + tend r
// differ from it by rounding only.
template <bool ENDS, bool TICKET = false>
RT_D Bundle make_bundle(bool in, f3 o, f3 r, f3 pm, float odr, float onorm, float tmax, float tend, int leader,
                        float onorm_all = -1.0f, float tmax_all = -1.0f, bool one_end = false) {
  Bundle b;
  b.axis = lane_bcast3(r, leader);
  b.m = lane_bcast3(pm, leader);
  b.odr = lane_bcast(odr, leader);
  const f3 dr = sub3(r, b.axis), dorg = sub3(o, lane_bcast3(o, leader));
  // a NaN would be dropped by the max: it becomes +inf (nothing is culled for a group with such a ray)
  float rho = in ? __builtin_amdgcn_sqrtf(dot3(dr, dr)) * 1.001f + 1e-7f : 0.0f;
  float ro = in ? __builtin_amdgcn_sqrtf(dot3(dorg, dorg)) * 1.001f : 0.0f;
  if (!(rho >= 0.0f)) rho = INFINITY;
  if (!(ro >= 0.0f)) ro = INFINITY;
  b.rho = wave_max(rho);
  b.ro = wave_max(ro);
  if (TICKET) {
    b.onorm = onorm_all;
    b.tmax = tmax_all;
  } else {
    float on = in ? onorm : 0.0f;
    float tm = in ? tmax : -INFINITY;
    if (!(on >= 0.0f)) on = INFINITY;
    if (tm != tm) tm = INFINITY;
    b.onorm = wave_max(on);
    b.tmax = wave_max(tm);
  }
  b.re = INFINITY; b.T = INFINITY;
  b.fan = no_fan();
  if (ENDS) {  // (a walk kernel's rays all end, or none does; decided at compile time, the bundle's numbers stay in scalar registers)
    // end points in float: off the line's own point at tend by at most 3 eps (|o| + tend), covered by 2e-6 (onorm + tend)
    float re_all;
    if (TICKET && one_end) {
      re_all = 2e-6f * (b.onorm + b.tmax);  // (tend <= tmax_all: the rays' ranges end where they meet)
    } else {
      const f3 e = mk3(__builtin_fmaf(tend, r.x, o.x), __builtin_fmaf(tend, r.y, o.y), __builtin_fmaf(tend, r.z, o.z));
      const f3 de = sub3(e, lane_bcast3(e, leader));
      float re = in ? __builtin_fmaf(__builtin_amdgcn_sqrtf(dot3(de, de)), 1.001f, 2e-6f * (onorm + tend)) : 0.0f;
      if (!(re >= 0.0f)) re = INFINITY;
      re_all = wave_max(re);
    }
    b.re = lane_bcast(re_all, 0);  // (a wave-uniform value that vector arithmetic produced: back into a scalar register)
    b.T = lane_bcast(tend, leader);
  }
  return b;
}

// For a batch of a cell's list and the rays of one bundle: an upper bound of 1 / |cos(ray, normal)| over the batch's
// triangles.  |cos| >= |axis.n| - rho for the unit normal n, and the header's 10-bit normal is within 0.0017 of n.  A
// triangle the bundle may be coplanar with (bound below 1e-3, or below a quarter of the batch's gb) makes the batch's
// value infinite: that batch is not culled in this visit.  Two lanes per batch work it out, four normals each (lane b and
// lane b + 32), before lane b tests the batch's box.
RT_D uint32_t pack_normal(float x, float y, float z) {  // (a NaN or infinite component: the zero vector, no bound)
  if (!(__builtin_fabsf(x) + __builtin_fabsf(y) + __builtin_fabsf(z) < 2.0f)) return 0u;
  const int qx = (int)rintf(x * 511.0f), qy = (int)rintf(y * 511.0f), qz = (int)rintf(z * 511.0f);
  return (uint32_t)(qx & 0x3ff) | (uint32_t)(qy & 0x3ff) << 10 | (uint32_t)(qz & 0x3ff) << 20;
}
// Smallest |axis.n| over four of the eight packed normals of a batch's header (the ones from `from`; records: how many
// of the batch's eight are there), and the bound for the batch from the smallest over all eight.
RT_D float batch_cos_part(const float *H, int from, int records, const Bundle &bu) {
  const float q = 1.0f / 511.0f;
  const float sx = bu.axis.x * q, sy = bu.axis.y * q, sz = bu.axis.z * q;
  const v4f a = *reinterpret_cast<const v4f *>(H + 12 + from);
  const float words[4] = {a.x, a.y, a.z, a.w};
  float c0 = INFINITY;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t w = __float_as_uint(words[j]);
    const float nx = (float)((int)(w << 22) >> 22), ny = (float)((int)(w << 12) >> 22), nz = (float)((int)(w << 2) >> 22);
    float c = __builtin_fabsf(sx * nx + sy * ny + sz * nz);
    if (!(c >= 0.0f)) c = 0.0f;  // a NaN must not be dropped by the min
    if (from + j < records) c0 = __builtin_fminf(c0, c);
  }
  return c0;
}
RT_D float batch_inv_cos(float c0, float gb, const Bundle &bu) {
  const float cosb = c0 * 0.9999f - 0.0018f - bu.rho;
  const float least = __builtin_fmaxf(0.001f, 0.25f * gb);  // gb / 4: below it det may change its sign by rounding
  return cosb > least ? __builtin_amdgcn_rcpf(cosb) * 1.01f : INFINITY;
}

// Can the exact test report a hit of the ray (r, m = o x r, odr = o.r, |o| <= onorm) with a triangle of the
// batch, with a ray parameter in [0, tmax]?  false only if surely not.  c, radius, h, ga, gb come from the
// batch header, inv_cos from batch_cos for a bundle that contains this ray.
//
// Geometry: a hit point lies on the ray's line and in a triangle, and all triangles lie in the box c +- h.  A
// line meets a box only if no direction r x e_i (e_i the coordinate axes) separates them:
// |((c - o) x r)_x| <= h_y |r_z| + h_z |r_y| and cyclically; and the parameters of the box's points lie within
// h.|r| of the centre's parameter tc = (c - o).r.
// Rounding.  The exact test (intersection.c:58-86) evaluates p = r x ik, det = ij.p, tv = o - i, u det = tv.p,
// q = tv x ij, v det = r.q, d det = ik.q in binary32 without FMA.  With L the longest edge and eps = 2^-24, a
// cross product is off by at most 2.83 eps |a||b| and a three-term dot product by 3 eps |a||b| plus what its
// operands carry, so the numerators of u and v are off by e1 <= 6.83 eps |tv| L, that of d by e3 <= 6.83 eps |tv| L^2
// and det by e2 <= 5.9 eps L^2.  As long as |det| > e2 (no sign flip), a triangle the test accepts has true
// barycentric coordinates within eta = (e1 + 1.01 e2) / |det| of [0,1]: the point P where the ray's line meets
// the plane is within 2 eta L of the triangle, and the reported d within (e3 + (|tv| + 2 L) e2) / |det| of P's
// parameter.  With |det| = |n| |cos| (n = ij x ik, the angle between ray and normal) both together are at most
//     (26.4 eps |tv| + 23.7 eps L) L^2 / (|n| |cos|)  <=  (gb tv + ga) / |cos|,   gb = 32 eps L^2 / |n|,  ga = 0.9 gb L
// (batch maxima of ga, gb from k_grid_bounds; tv = |o - c| + radius >= |o - i| for every triangle of the batch;
// batch_cos makes 1 / |cos| infinite unless |cos| > gb / 4, which implies |det| > 8 eps L^2 > e2).  On 21 million
// rays aimed at edges, vertices, slivers and grazing incidence the observed excess stays below 0.14 of this bound.
// Every half extent grows by that much, W.  The extents come inflated by 0.1% + 1e-5 |c| from k_grid_bounds, and
// delta = 1e-5 (2 |o| + tv) >= 1e-5 (|o| + |c|), far above the rounding of (c - o) x r and tc, is added to W and to
// every comparison (the separating direction vanishes for an axis-parallel ray: 0 <= 0 must not be decided by
// rounding).  A NaN compares false: not culled.
// Two levels.  box_test_bundle: the ray is a bundle's leader and the answer is for every ray of the bundle at once
// (onorm, tmax: the bundle's maxima).  A point of such a ray at parameter t is within ro + |t| rho of the leader's
// point at t, and |t| <= tv + |h| + 2 W where the ray meets the widened box, so the extents grow by that much more
// (or, for rays that end, by what the narrowing bundle allows at the box: see there); tv by ro.  It hands back W before that growth and delta: tv and onorm there are upper bounds of every ray's own,
// so these two serve any single ray of the bundle.  box_test_ray: one ray against the box widened by them (the
// walk kernels keep c, -+delta, h + W, W in their LDS copy of the header; delta's sign bit marks a batch with a
// transparent triangle).
struct BoxRay { float ex, ey, ez, tc; };
RT_D BoxRay box_ray(f3 c, f3 r, f3 m, float odr) {
  BoxRay b;
  b.ex = __builtin_fmaf(-c.z, r.y, __builtin_fmaf(c.y, r.z, -m.x));
  b.ey = __builtin_fmaf(-c.x, r.z, __builtin_fmaf(c.z, r.x, -m.y));
  b.ez = __builtin_fmaf(-c.y, r.x, __builtin_fmaf(c.x, r.y, -m.z));
  b.tc = __builtin_fmaf(c.x, r.x, __builtin_fmaf(c.y, r.y, __builtin_fmaf(c.z, r.z, -odr)));
  return b;
}
// hw = h + W.  true = separated: no point of the ray with a parameter in [0, tmax] lies in the widened box
RT_D bool box_separated(const BoxRay &b, f3 r, f3 hw, float W, float delta, float tmax) {
  const float ax = __builtin_fabsf(r.x), ay = __builtin_fabsf(r.y), az = __builtin_fabsf(r.z);
  const float sx = __builtin_fmaf(hw.y, az, __builtin_fmaf(hw.z, ay, delta));
  const float sy = __builtin_fmaf(hw.x, az, __builtin_fmaf(hw.z, ax, delta));
  const float sz = __builtin_fmaf(hw.x, ay, __builtin_fmaf(hw.y, ax, delta));
  const float hr = __builtin_fmaf(hw.x, ax, __builtin_fmaf(hw.y, ay, __builtin_fmaf(hw.z, az, W)));
  return (__builtin_fabsf(b.ex) > sx) | (__builtin_fabsf(b.ey) > sy) | (__builtin_fabsf(b.ez) > sz) | (b.tc + hr < 0.0f) | (b.tc - hr > tmax);
}
RT_D bool box_test_bundle(f3 c, float radius, f3 h, bool transparent, float inv_cos, float ga, float gb, float slack, const Bundle &bu,
                          bool transparent_counts_anywhere, bool ends, float &w_ray, float &delta, float &spread) {
  const BoxRay b = box_ray(c, bu.axis, bu.m, bu.odr);
  const float d2 = __builtin_fmaf(b.ex, b.ex, __builtin_fmaf(b.ey, b.ey, b.ez * b.ez));
  const float on5 = 0.00001f * bu.onorm;
  float tv = __builtin_fmaf(__builtin_amdgcn_sqrtf(__builtin_fmaf(b.tc, b.tc, d2)), 1.001f, on5 + radius);  // |o - c|^2 = d2 + tc^2
  tv = __builtin_fmaf(bu.ro, 1.001f, tv);  // >= |o - v| for every ray origin o of the bundle and every vertex v of the batch
  delta = __builtin_fmaf(0.00001f, tv, on5 + on5);
  w_ray = __builtin_fmaf(__builtin_fmaf(gb, tv, ga), inv_cos, delta) + slack;
  const float reach = (h.x + h.y + h.z) + (w_ray + w_ray);  // >= the box's extent along any direction, margins included
  float widen = __builtin_fmaf(bu.rho, tv + reach, bu.ro);
  // shadow walk: a transparent triangle attenuates wherever it is hit (d >= 0); an opaque one matters only before tmax
  const bool endless = transparent_counts_anywhere & transparent;
  const float tmax = endless ? INFINITY : bu.tmax;
  spread = __builtin_fmaf(tv, bu.rho, 1.000001f * bu.ro);  // the spread of the bundle's lines at the batch, for the record-level test (bundle_slabs_pass)
  if (ends) {
    // Rays that end: the point of a ray at the fraction s >= 0 of its range is within w(s) = |1 - s| ro + s re of the
    // leader's point at the same fraction, whose parameter s T lies within reach + w(s) of tc.  With far = max(ro, re)
    // (>= w(s) up to the end) that puts s between (tc - reach - far) / (T + 2 far) and (tc + reach + far) / (T - 2 far),
    // and w is convex: its largest value over the box is the one at the box's near or far side.  A hit counts up to the
    // ray's end (s <= 1), a transparent triangle's in the shadow walk beyond it too.
    // (A hit the exact test reports inside the range may lie up to w_ray outside it: that moves s by w_ray / T,
    // which adds at most w_ray (ro + re) / T <= w_ray.)  Only for 4 far <= T; a NaN leaves the other bound in force.
    const float far = __builtin_fmaxf(bu.ro, bu.re);
    const float s0 = __builtin_fmaxf(b.tc - reach - far, 0.0f) / __builtin_fmaf(2.0f, far, bu.T) * 0.9998f;
    float s1 = __builtin_fmaxf(b.tc + reach + far, 0.0f) / __builtin_fmaf(-2.0f, far, bu.T) * 1.0002f;
    if (!endless) s1 = __builtin_fminf(s1, 1.0f);
    const float sn = __builtin_fminf(s0, s1);
    const float near_side = __builtin_fmaf(__builtin_fabsf(1.0f - sn), bu.ro, sn * bu.re);
    const float far_side = __builtin_fmaf(__builtin_fabsf(1.0f - s1), bu.ro, s1 * bu.re);
    const float narrowing = __builtin_fmaxf(near_side, far_side) * 1.0001f + w_ray;
    if (4.0f * far <= bu.T) widen = __builtin_fminf(widen, narrowing);
    // ... and the lines' spread with the leader's end point e_L as the point the moments are taken about:
    // tb >= |e_L - v| like tv >= |o - v|, |e_L - c|^2 = d2 + (T - tc)^2
    const float te = bu.T - b.tc;
    const float tb = __builtin_fmaf(__builtin_amdgcn_sqrtf(__builtin_fmaf(te, te, d2)), 1.001f, on5 + radius);
    spread = __builtin_fminf(spread, __builtin_fmaf(tb, bu.rho, 1.000001f * bu.re));
  }
  const float W = w_ray + widen * 1.001f;
  return !box_separated(b, bu.axis, mk3(h.x + W, h.y + W, h.z + W), W, delta, tmax);
}
RT_D bool box_test_ray(f3 c, float delta_and_flag, f3 hw, float W, f3 r, f3 m, float odr, float tmax_opaque, bool transparent_counts_anywhere) {
  const bool transparent = __float_as_int(delta_and_flag) < 0;
  const float tmax = (transparent_counts_anywhere & transparent) ? INFINITY : tmax_opaque;
  return !box_separated(box_ray(c, r, m, odr), r, hw, W, __builtin_fabsf(delta_and_flag), tmax);
}

// Where a walk kernel's wavefronts spend their time: a diagnostic that exists only in builds with -DRT_PHASES
// (tools/phases.sh).  end<K>() adds the shader clock since the previous mark to phase K; the sums over all
// wavefronts land in g_phase (nearest walk 0-7, shadow walk 8-15; rtgpu_phase_cycles reads them).  Phases:
// 0 ray fetch, set-up and result write, 1 choosing the group and mode of a visit, 2 header staging and
// batch_cos, 3 batch culling, 4 record staging, 5 triangle tests (3 and 4 are counted in 5 unless RT_PHASES=2),
// 6 stepping to the next occupied cell, 7 lane and ray mode.
__device__ unsigned long long g_phase[16];
#ifdef RT_HIST
__device__ unsigned long long g_hist[2][2][9];  // [walk][0 = batches, 1 = (lane, record) pairs][records of the batch some lane passed]
__device__ unsigned long long g_mode[2][3][4];  // [walk][cell, lane, ray mode][visits, wave-level record tests, lane-level record tests, rays in the group]
__device__ unsigned long long g_cand[2][4];     // [walk][headers looked at, passed the bundle's box test, kept by the ray-by-ray test, cell visits]
__device__ unsigned long long g_chunk[2][12];   // [walk][header chunks staged, without a candidate, without a kept batch, without a batch left by the record-level test,
                                                //        visits and headers of lists of 1, 2, 3-4, 5+ chunks]
#endif
// Per-ticket log of the walk kernels, only in builds with -DRT_TICKETS (tools/tickets.py): shader clocks from
// taking the ticket to its last result, cell visits, batches that reached the ray-by-ray test, records tested.
#ifdef RT_TICKETS
constexpr int kTicketLog = 1 << 18, kTicketWords = 12;
__device__ uint32_t g_ticket_log[2][kTicketLog][kTicketWords];
struct TicketLog {
  unsigned long long t0; uint32_t visits, batches, tests;
  RT_D void begin() { t0 = __builtin_amdgcn_s_memtime(); visits = batches = tests = 0; }
  // a wavefront's life: g_phase[which * 8 + 0..2] += clocks inside tickets, clocks from entry to exit, 1; [3] = the longest life
  static RT_D void life(int which, int lane, unsigned long long busy, unsigned long long born) {
    const unsigned long long all = __builtin_amdgcn_s_memtime() - born;
    if (lane == 0) {
      atomicAdd(&g_phase[which * 8], busy); atomicAdd(&g_phase[which * 8 + 1], all); atomicAdd(&g_phase[which * 8 + 2], 1ull);
      atomicMax(&g_phase[which * 8 + 3], all);
    }
  }
  template <typename PH>
  RT_D void end(int which, uint32_t base, int lane, PH &ph) {
    const unsigned long long t = __builtin_amdgcn_s_memtime() - t0;
    if (lane == 0 && (base >> 6) < (uint32_t)kTicketLog) {
      uint32_t *e = g_ticket_log[which][base >> 6];
      e[0] = (uint32_t)t; e[1] = visits; e[2] = batches; e[3] = tests;
#ifdef RT_PHASES
      for (int i = 0; i < 8; i++) { e[4 + i] = (uint32_t)ph.acc[i]; ph.acc[i] = 0; }  // with the phase clocks in: per ticket
#endif
    }
  }
};
#define RT_TICKET(x) x
#else
#define RT_TICKET(x)
#endif
#ifdef RT_PHASES
struct Phases {
  unsigned long long t0, acc[8];
  RT_D void begin() {
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = 0;
  }
  template <int K, bool PER_BATCH = false> RT_D void end() {
    if (PER_BATCH && RT_PHASES < 2) return;  // -DRT_PHASES=2 also separates phases 3, 4, 5 (one mark per batch: heavier)
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    acc[K] += t - t0;
    t0 = t;
  }
  RT_D void flush(int base, int lane) {
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 8; i++) atomicAdd(&g_phase[base + i], acc[i]);
    }
  }
};
#else
struct Phases {
  RT_D void begin() {}
  template <int K, bool PER_BATCH = false> RT_D void end() {}
  RT_D void flush(int, int) {}
};
#endif

RT_D void stage_batch(const TriRec *__restrict__ recs, uint32_t c0, int n, float *rs, int lane) {
  const int nvec = n * kRecVecs;
  const float4 *src = reinterpret_cast<const float4 *>(recs + c0);
  // global -> LDS without passing through registers: lane L's 16 bytes land at base + 16 L
  static_assert(kStageLoads <= 5, "staging is written out for up to five loads per lane");
  if (lane < nvec) glds16(src + lane, rs);
  if (kStageLoads > 1 && lane + 64 < nvec) glds16(src + lane + 64, rs + 256);
  if (kStageLoads > 2 && lane + 128 < nvec) glds16(src + lane + 128, rs + 512);
  if (kStageLoads > 3 && lane + 192 < nvec) glds16(src + lane + 192, rs + 768);
  if (kStageLoads > 4 && lane + 256 < nvec) glds16(src + lane + 256, rs + 1024);
}
RT_D void wait_staged() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// Field access of a record in LDS or in memory (float index: n 0-2, k1 3, ik 4-6, q 7-9, id 10, k2 11,
// ij 12-14, p 15-17, an 18, kt 19).  The first slab needs n, ik, q: two 16-byte reads and one 8-byte read.
struct RecA { f3 n, ac, q; };
RT_D RecA rec_a(const float *R) {
  const v4f x = *reinterpret_cast<const v4f *>(R);
  const v4f y = *reinterpret_cast<const v4f *>(R + 4);
  const v2f z = *reinterpret_cast<const v2f *>(R + 8);
  RecA a; a.n = mk3(x.x, x.y, x.z); a.ac = mk3(y.x, y.y, y.z); a.q = mk3(y.w, z.x, z.y);
  return a;
}

struct RecB { f3 ab, p; float an, kt; };
RT_D RecB rec_b(const float *R) {
  const v4f x = *reinterpret_cast<const v4f *>(R + 12);
  const v4f y = *reinterpret_cast<const v4f *>(R + 16);
  RecB b; b.ab = mk3(x.x, x.y, x.z); b.p = mk3(x.w, y.x, y.y); b.an = y.z; b.kt = y.w;
  return b;
}

// The three slabs of the filter (pl_stage1, pl_stage2) for a whole bundle at once: can ANY ray of the bundle pass them for
// this record?  false only if surely none can, and then tri_test rejects the triangle for every ray of the bundle.
// For one ray, with the exact values on the float inputs marked *: det* = r.N, tp* = (o - i).(r x ik), vp* = r.((o - i) x ij),
// X1* = 2 tp* - det*, X2* = 2 vp* - det*, X3* = 2 (tp* + vp*) - det*, and tri_test accepts only if for each of them
// |X*| <= |det*| (1 + 7 eps) + kc, kc = k1 |o| + k2 (rt_math.h: the bound proven for the computed values, 71 eps L (|o| + A)
// + 28 eps L^2, plus what the computed values are off by, at most 31 eps L (|o| + A) + 5 eps L^2, stays below k1 |o| + k2 = 144
// eps L (|o| + A) + 56 eps L^2).  A ray of the bundle has r = axis + dr, |dr| <= rho, o = o_L + do, |do| <= ro, so against
// the leader's values  |det* - det_L*| <= rho |N|,  |tp* - tp_L*| = |do.(r x ik) + (o_L - i).(dr x ik)| <= E = (1.000001 ro + tv rho) L
// (tv >= |o_L - i| from box_test_bundle; L = |ik|, the longest edge).  The moment of a line can be taken about any of its
// points: for rays that end, about the end points e = o + tend r, which are within re of the leader's: E = (1.000001 re +
// tb rho) L as well, tb >= |e_L - i|; spread = the smaller of the two brackets, from box_test_bundle.  The same for vp
// with |ij| <= L, hence
// |X1* - X1_L*|, |X2* - X2_L*| <= 2 E + rho |N| and |X3* - X3_L*| <= 4 E + rho |N|.  The leader's computed X', det' are
// within kc_L / 4 of X_L*, det_L*.  So no ray of the bundle can satisfy the acceptance conditions if for one of the three
//     |X'| > (|det'| + rho |N|) 1.00001 + D + 2 kc_max,      kc_max = k1 onorm_max + k2 (+ slack: infinite when the filter is off)
// (2 kc_max covers kc / 4 + kc / 4 (1 + 7 eps) + kc of the ray itself); E and the norms are rounded up by 0.1 %.  A NaN compares false.
RT_D bool bundle_slabs_pass(const RecA &A, f3 ab, f3 P, float kc_max, float spread, const Bundle &bu) {
  const float det = __builtin_fmaf(bu.axis.x, A.n.x, __builtin_fmaf(bu.axis.y, A.n.y, bu.axis.z * A.n.z));
  const float rq = __builtin_fmaf(bu.axis.x, A.q.x, __builtin_fmaf(bu.axis.y, A.q.y, bu.axis.z * A.q.z));
  const float tp = __builtin_fmaf(bu.m.x, A.ac.x, __builtin_fmaf(bu.m.y, A.ac.y, __builtin_fmaf(bu.m.z, A.ac.z, -rq)));
  const float rp = __builtin_fmaf(bu.axis.x, P.x, __builtin_fmaf(bu.axis.y, P.y, bu.axis.z * P.z));
  const float nv = __builtin_fmaf(bu.m.x, ab.x, __builtin_fmaf(bu.m.y, ab.y, __builtin_fmaf(bu.m.z, ab.z, rp)));  // -vp
  const float x1 = __builtin_fmaf(2.0f, tp, -det);
  const float x2 = __builtin_fmaf(-2.0f, nv, -det);
  const float x3 = __builtin_fmaf(2.0f, tp - nv, -det);
  const float nN = __builtin_amdgcn_sqrtf(__builtin_fmaf(A.n.x, A.n.x, __builtin_fmaf(A.n.y, A.n.y, A.n.z * A.n.z))) * 1.001f;
  const float L = __builtin_amdgcn_sqrtf(__builtin_fmaf(A.ac.x, A.ac.x, __builtin_fmaf(A.ac.y, A.ac.y, A.ac.z * A.ac.z))) * 1.002f;  // (|ij| <= |ik| up to rounding)
  const float rn = bu.rho * nN;
  const float E2 = (2.0f * L) * spread * 1.001f;  // 2 E
  const float T = __builtin_fmaf(__builtin_fabsf(det) + rn, 1.00001f, rn + (kc_max + kc_max));
  return !((__builtin_fmaxf(__builtin_fabsf(x1), __builtin_fabsf(x2)) > T + E2) | (__builtin_fabsf(x3) > T + (E2 + E2)));
}

// Can ANY ray of the fan pass the three slabs of this record?  false only if surely none can.  tri_test accepts a ray
// only if for each slab |X*| <= |det*| (1 + 7 eps) + kc with the exact values on the ray's float inputs (bundle_slabs_pass);
// times s, and with the ray's float direction within 2 eps of (L - a) / s (which moves X*, det* by less than kc / 50):
// |X(a)| <= 1.00001 |det(a)| + 1.1 kc s for the affine X, det above.  With |X(a)| >= |X(c)| - spread_X, |det(a)| <=
// |det(c)| + spread_det, s <= smax, kc <= kc_max, no ray of the fan passes a slab with
//     |X(c)| - spread_X > 1.00001 (|det(c)| + spread_det) + 3 kc_max smax.
// Rounding: rc = L - c, mc = c x rc and the dot products in binary32 are off by less than 0.1 kc_max smax per quantity
// (|c| <= onorm_max; kc = 144 eps L (|o| + A) + 56 eps L^2 with L the longest edge = |ac|), the X(c) by less than 0.5:
// inside the 3.  The gradients L x ac + Q, L x ab - P cancel where light and triangle lie far from the origin of the
// coordinates: each is off by at most 4 eps |ac| (|L| + A) <= 0.028 (k1 |L| + k2), times the box: e below covers twice
// that.  A NaN compares false: passes.
RT_D bool fan_slabs_pass(const RecA &A, f3 ab, f3 P, float kc_max, float k1max, float k2max, const Fan &f) {
  const f3 h = f.h, rc = f.rc, mc = f.mc;
  const float det = __builtin_fmaf(rc.x, A.n.x, __builtin_fmaf(rc.y, A.n.y, rc.z * A.n.z));
  const float rq = __builtin_fmaf(rc.x, A.q.x, __builtin_fmaf(rc.y, A.q.y, rc.z * A.q.z));
  const float tp = __builtin_fmaf(mc.x, A.ac.x, __builtin_fmaf(mc.y, A.ac.y, __builtin_fmaf(mc.z, A.ac.z, -rq)));
  const float rp = __builtin_fmaf(rc.x, P.x, __builtin_fmaf(rc.y, P.y, rc.z * P.z));
  const float nv = __builtin_fmaf(mc.x, ab.x, __builtin_fmaf(mc.y, ab.y, __builtin_fmaf(mc.z, ab.z, rp)));
  const float x1 = __builtin_fmaf(2.0f, tp, -det), x2 = __builtin_fmaf(-2.0f, nv, -det), x3 = __builtin_fmaf(2.0f, tp - nv, -det);
  // gradients with respect to the hit point
  const f3 gt = mk3(__builtin_fmaf(f.L.y, A.ac.z, __builtin_fmaf(-f.L.z, A.ac.y, A.q.x)), __builtin_fmaf(f.L.z, A.ac.x, __builtin_fmaf(-f.L.x, A.ac.z, A.q.y)),
                    __builtin_fmaf(f.L.x, A.ac.y, __builtin_fmaf(-f.L.y, A.ac.x, A.q.z)));
  const f3 gv = mk3(__builtin_fmaf(f.L.y, ab.z, __builtin_fmaf(-f.L.z, ab.y, -P.x)), __builtin_fmaf(f.L.z, ab.x, __builtin_fmaf(-f.L.x, ab.z, -P.y)),
                    __builtin_fmaf(f.L.x, ab.y, __builtin_fmaf(-f.L.y, ab.x, -P.z)));
  const f3 g1 = mk3(__builtin_fmaf(2.0f, gt.x, A.n.x), __builtin_fmaf(2.0f, gt.y, A.n.y), __builtin_fmaf(2.0f, gt.z, A.n.z));
  const f3 g2 = mk3(__builtin_fmaf(-2.0f, gv.x, A.n.x), __builtin_fmaf(-2.0f, gv.y, A.n.y), __builtin_fmaf(-2.0f, gv.z, A.n.z));
  const f3 g3 = mk3(__builtin_fmaf(-2.0f, gv.x, g1.x), __builtin_fmaf(-2.0f, gv.y, g1.y), __builtin_fmaf(-2.0f, gv.z, g1.z));
  auto over_box = [&](f3 v) { return __builtin_fmaf(__builtin_fabsf(v.x), h.x, __builtin_fmaf(__builtin_fabsf(v.y), h.y, __builtin_fabsf(v.z) * h.z)); };
  const float e = 0.06f * __builtin_fmaf(k1max, f.lnorm, k2max) * f.hsum;
  const float T = __builtin_fmaf(__builtin_fabsf(det) + over_box(A.n), 1.00001f, 3.0f * kc_max * f.smax);
  const float s1 = __builtin_fmaf(over_box(g1), 1.001f, e + e), s2 = __builtin_fmaf(over_box(g2), 1.001f, e + e), s3 = __builtin_fmaf(over_box(g3), 1.001f, 4.0f * e);
  return !((__builtin_fabsf(x1) - s1 > T) | (__builtin_fabsf(x2) - s2 > T) | (__builtin_fabsf(x3) - s3 > T));
}

// Does the triangle's plane lie behind every ray of the fan (no ray can reach it: tri_test's d is negative)?  true only
// if surely so.  With dp* = i.N - o.N and det* = r.N (rt_math.h) d = dp* / det*, and tri_test accepts only if its own d is
// not negative, which with its rounding (|dp - dp*| <= 10.5 eps L^2 (|o| + A), |det - det*| <= 8.9 eps L^2) means
// sign(det*) dp* >= -10.5 eps L^2 (|o| + A) once |det*| > 8.9 eps L^2.  For the fan's ray from a: s det* = (L - a).N up to
// 4 eps L^2 s, and dp* = an - a.N, both affine in a with gradient -N: over the box c +- h they stay within over_box(N) of
// their values at c.  So no ray of the fan is accepted if |det(c)| - over_box(N) > 1.5 kc_max smax (the sign is the same for
// all of them and det* is large enough) and sign(det(c)) dp(c) + over_box(N) < -2 |ac| kc_max (|ac| the longest edge; kc >=
// 144 eps L (|o| + A) + 56 eps L^2 leaves more than ten times the rounding of either side, binary32 evaluation included).
RT_D bool fan_plane_behind(const RecA &A, float an, float kc_max, const Fan &f) {
  const float det = __builtin_fmaf(f.rc.x, A.n.x, __builtin_fmaf(f.rc.y, A.n.y, f.rc.z * A.n.z));
  const float dp = __builtin_fmaf(-f.c.x, A.n.x, __builtin_fmaf(-f.c.y, A.n.y, __builtin_fmaf(-f.c.z, A.n.z, an)));
  const float ob = __builtin_fmaf(__builtin_fabsf(A.n.x), f.h.x, __builtin_fmaf(__builtin_fabsf(A.n.y), f.h.y, __builtin_fabsf(A.n.z) * f.h.z)) * 1.001f;
  const float L = __builtin_amdgcn_sqrtf(__builtin_fmaf(A.ac.x, A.ac.x, __builtin_fmaf(A.ac.y, A.ac.y, A.ac.z * A.ac.z))) * 1.002f;
  const float sdp = __int_as_float(__float_as_int(dp) ^ (__float_as_int(det) & (int)0x80000000));
  return (__builtin_fabsf(det) - ob > 1.5f * kc_max * f.smax) & (sdp + ob < -2.0f * L * kc_max);
}

// Walks the batches of one cell's list.  The batch headers are staged in LDS kHdrChunk at a time; for each
// batch the bundle's leader ray, widened to contain every ray of the bundle, is tested against the batch's box first (one
// lane per batch); for the batches that remain cull(centre, radius, half extents, transparent, inv_cos, ga, gb) decides
// ray by ray, for the whole wavefront, whether any of its rays can meet the box; only then are the batch's records staged and tested:
//   batch(k1, k2)      once per surviving batch, with the batch's filter constants
//   f(record, index, first)  the test of one record, for all lanes (first: pass1 has not seen the record, for the
//                      statistics); true (wave-uniform) if some lane passed the first slab and the rest of the test ran, and only then
//   more()             (wave-uniform) is asked after every batch: false = stop.
//   pass1(record)      the side-effect-free first slab on its own: batches in which most records survive run it for four
//                      records at a time before f; also for the statistics of diagnostic builds.
// RECTEST: with the record-level test of the bundle (bundle_slabs_pass) between the ray-by-ray test of the boxes and the
// staging; without it every record of a kept batch is tested.  It pays where bundles are narrow against the triangles.
// ENDS: the bundle's rays end (Bundle.re, .T are set) and the bundle test uses that.
// RAYCULL: with the ray-by-ray test of the candidates' boxes (cull).
constexpr int kSliceFloatsC = kHdrChunk * kHdrFloats + 2 * kBatch * kRecFloats;  // LDS floats per wavefront (kSliceFloats)
template <bool RECTEST, bool ENDS, bool RAYCULL, typename C, typename T, typename B, typename P, typename F, typename G>
RT_D void stream_cell(const GridP &g, uint32_t hbeg, uint32_t beg, uint32_t cnt, float *slice, int lane, const Bundle &bundle_,
                      bool transparent_counts_anywhere, Phases &ph, C &&cull, T &&tally, B &&batch,
                      P &&pass1, F &&f, G &&more) {
  ph.template end<1>();
  asm volatile("" : "+v"(lane));  // (per-lane addresses below are worked out per visit instead of being kept, spilled, for the whole kernel)
  float *hs = slice;                             // kHdrChunk headers
  float *rs = slice + kHdrChunk * kHdrFloats;    // two buffers of kBatch records
  const TriRec *__restrict__ recs = g.cell_tris;
  const float4 *hsrc = reinterpret_cast<const float4 *>(g.cell_hdrs + hbeg);
  const uint32_t nb = (cnt + kBatch - 1) / kBatch;
#if RT_FLAT
  // Short lists (up to RT_FLAT records: nine of ten cell visits in a scene of many small cells) without the batch level:
  // 64 records at a time, one lane per record reads its record from memory and asks whether any ray of the bundle can
  // pass its three slabs (bundle_slabs_pass, with the distance bound and the lines' spread box_test_bundle works out for
  // the record's batch); the records that are left go to LDS in list order and are tested there.  One memory round trip
  // and one pass of the wavefront instead of header staging, grazing bound, two box tests and the record-level pass.
  if (RECTEST && ENDS && cnt <= (uint32_t)RT_FLAT) {  // (in the walks whose rays start on one tile short lists are rare, and the pass costs their kernels registers)
    Bundle bundle = bundle_;
    asm volatile("" : "+s"(bundle.axis.x), "+s"(bundle.axis.y), "+s"(bundle.axis.z), "+s"(bundle.rho), "+s"(bundle.ro), "+s"(bundle.onorm));
    for (uint32_t p0 = 0; p0 < cnt; p0 += 64u) {
      const uint32_t at = p0 + (uint32_t)lane;
      bool pass = false;
      const v4f *R = reinterpret_cast<const v4f *>(recs + beg + at);
      if (at < cnt) {
        const v4f hc = *reinterpret_cast<const v4f *>(g.cell_hdrs + hbeg + (at >> 3));  // the batch's box centre and radius
        const v4f r0 = R[0], r1 = R[1], r2 = R[2], r3 = R[3], r4 = R[4];
        RecA A; A.n = mk3(r0.x, r0.y, r0.z); A.ac = mk3(r1.x, r1.y, r1.z); A.q = mk3(r1.w, r2.x, r2.y);
        const f3 ab = mk3(r3.x, r3.y, r3.z), pv = mk3(r3.w, r4.x, r4.y);
        // tv >= |o - v| for every origin o of the bundle and every vertex v of the batch, and the spread of the bundle's lines
        // there: the expressions of box_test_bundle
        const BoxRay b = box_ray(mk3(hc.x, hc.y, hc.z), bundle.axis, bundle.m, bundle.odr);
        const float d2 = __builtin_fmaf(b.ex, b.ex, __builtin_fmaf(b.ey, b.ey, b.ez * b.ez));
        const float on5 = 0.00001f * bundle.onorm;
        float tv = __builtin_fmaf(__builtin_amdgcn_sqrtf(__builtin_fmaf(b.tc, b.tc, d2)), 1.001f, on5 + hc.w);
        tv = __builtin_fmaf(bundle.ro, 1.001f, tv);
        float spread = __builtin_fmaf(tv, bundle.rho, 1.000001f * bundle.ro);
        if (ENDS) {
          const float te = bundle.T - b.tc;
          const float tb = __builtin_fmaf(__builtin_amdgcn_sqrtf(__builtin_fmaf(te, te, d2)), 1.001f, on5 + hc.w);
          spread = __builtin_fminf(spread, __builtin_fmaf(tb, bundle.rho, 1.000001f * bundle.re));
        }
        const float kc_max = __builtin_fmaf(r0.w, bundle.onorm, r2.w) + g.slack;  // the record's own k1, k2
        pass = bundle_slabs_pass(A, ab, pv, kc_max, spread, bundle);
      }
      const unsigned long long left = __ballot(pass);
      {
        const uint32_t nbp = (cnt - p0 < 64u ? cnt - p0 + kBatch - 1 : 64u) / kBatch;
        int culled = 0;
        for (uint32_t b = 0; b < nbp; b++) culled += ((left >> (8u * b)) & 0xffull) == 0ull ? 1 : 0;
        tally((int)nbp, culled);
      }
      ph.template end<3>();
      // up to 48 records fit the wavefront's slice; more (rare) go in a second round
      constexpr int kFlatSlots = kSliceFloatsC / kRecFloats;
      unsigned long long todo = left;
      while (todo != 0) {
        const int before = __popcll(todo & lanes_below(lane));
        const bool mine = ((todo >> lane) & 1ull) != 0 && before < kFlatSlots;
        if (mine) {  // (read again, from the vector cache, rather than held in twenty registers through the pass)
          v4f *D = reinterpret_cast<v4f *>(slice + before * kRecFloats);
          const v4f r0 = R[0], r1 = R[1], r2 = R[2], r3 = R[3], r4 = R[4];
          D[0] = r0; D[1] = r1; D[2] = r2; D[3] = r3; D[4] = r4;
        }
        const unsigned long long placed = __ballot(mine);
        todo &= ~placed;
        __builtin_amdgcn_wave_barrier();
        const int n = __popcll(placed);
        tally(-2, n);
        ph.template end<4, true>();
        for (int j = 0; j < n; j++) {
          const float *Rj = slice + j * kRecFloats;
          batch(Rj[3], Rj[11]);
          f(Rj, 0u, true);
          if (!more()) { ph.template end<5>(); return; }
        }
        __builtin_amdgcn_wave_barrier();
      }
      ph.template end<5>();
    }
    return;
  }
#endif
  for (uint32_t h0 = 0; h0 < nb; h0 += kHdrChunk) {
    // The bundle's numbers are wave-uniform and live in scalar registers; their products with constants (axis / 511,
    // 1e-5 onorm, ...) are vector arithmetic, and hoisted out of this loop each holds a vector register for the whole
    // visit.  Opaque copies keep that arithmetic where it is used.
    Bundle bundle = bundle_;
    asm volatile("" : "+s"(bundle.axis.x), "+s"(bundle.axis.y), "+s"(bundle.axis.z), "+s"(bundle.rho), "+s"(bundle.ro), "+s"(bundle.onorm));
    asm volatile("" : "+s"(bundle.tmax), "+s"(bundle.odr));
    if (RECTEST && !ENDS && bundle.fan.on) {
      asm volatile("" : "+s"(bundle.fan.smax), "+s"(bundle.fan.hsum), "+s"(bundle.fan.lnorm));
    }
    if (ENDS) asm volatile("" : "+s"(bundle.re), "+s"(bundle.T));
    const int nh = (int)(nb - h0 < (uint32_t)kHdrChunk ? nb - h0 : (uint32_t)kHdrChunk);
    constexpr int kHdrVecs = kHdrFloats / 4;
    static_assert(kHdrChunk * kHdrVecs <= 192, "up to three 16-byte loads per lane stage a header chunk");
    if (lane < kHdrVecs * nh) glds16(hsrc + kHdrVecs * h0 + lane, hs);
    if (lane + 64 < kHdrVecs * nh) glds16(hsrc + kHdrVecs * h0 + lane + 64, hs + 256);
    if (lane + 128 < kHdrVecs * nh) glds16(hsrc + kHdrVecs * h0 + lane + 128, hs + 512);
    wait_staged();
    ph.template end<2>();
    // one lane per batch: can any ray of the bundle meet the batch's box at all?  If so the lane rewrites its header
    // for the ray-by-ray test: c, -+delta, h + W, W (box_test_ray)
    static_assert(kHdrChunk == 32, "two lanes per batch of a chunk");
    float c0 = INFINITY;
    {
      int b = lane & 31;
      asm volatile("" : "+v"(b));  // (worked out here every time: hoisted out of the loops, the header's LDS address is spilled and reloaded per chunk)
      if (b < nh) {
        const uint32_t left = cnt - (h0 + (uint32_t)b) * kBatch;
        c0 = batch_cos_part(hs + b * kHdrFloats, lane < 32 ? 0 : 4, left < (uint32_t)kBatch ? (int)left : kBatch, bundle);
      }
      c0 = __builtin_fminf(c0, __shfl_xor(c0, 32));
    }
    bool maybe = false;
    if (lane < nh) {
      float *H = hs + lane * kHdrFloats;
      const v4f hc = *reinterpret_cast<const v4f *>(H), hh = *reinterpret_cast<const v4f *>(H + 4), k = *reinterpret_cast<const v4f *>(H + 8);
      float w_ray, delta, spread;
      const float inv_cos = batch_inv_cos(c0, k.w, bundle);
      maybe = box_test_bundle(mk3(hc.x, hc.y, hc.z), hc.w, mk3(hh.x, hh.y, hh.z), k.y < 0.0f, inv_cos, k.z, k.w, g.slack, bundle,
                              transparent_counts_anywhere, ENDS, w_ray, delta, spread);
      if (maybe) {
        H[3] = k.y < 0.0f ? -delta : delta;
        *reinterpret_cast<v4f *>(H + 4) = v4f{hh.x + w_ray, hh.y + w_ray, hh.z + w_ray, w_ray};
        H[10] = spread;  // (ga, gb have done their work: the slot keeps the spread of the bundle's lines at the batch for the record-level test)
      }
    }
    __builtin_amdgcn_wave_barrier();
    const unsigned long long candidates = __ballot(maybe);
    tally(nh, nh - __popcll(candidates));
#ifdef RT_HIST
    if (lane == 0) { atomicAdd(&g_cand[transparent_counts_anywhere ? 1 : 0][0], (unsigned long long)nh); atomicAdd(&g_cand[transparent_counts_anywhere ? 1 : 0][1], (unsigned long long)__popcll(candidates)); }
#endif
    // ray by ray: which of the candidates can any ray of the wavefront meet
    // (not where the bundle is a tile's primary rays and the record-level test follows: that one drops what this would, 3 % faster)
    unsigned long long kept = RAYCULL ? 0ull : candidates;
    for (unsigned long long cm = RAYCULL ? candidates : 0ull; cm != 0; cm &= cm - 1) {
      const int b = __ffsll(cm) - 1;
      const v4f hc = *reinterpret_cast<const v4f *>(hs + b * kHdrFloats);      // c, -+delta
      const v4f hw = *reinterpret_cast<const v4f *>(hs + b * kHdrFloats + 4);  // h + W, W
      const unsigned long long hit_by = cull(mk3(hc.x, hc.y, hc.z), hc.w, mk3(hw.x, hw.y, hw.z), hw.w);  // the rays that may meet the batch's box
      const bool keep = hit_by != 0;
#ifdef RT_HIST
      if (lane == 0 && keep) atomicAdd(&g_cand[transparent_counts_anywhere ? 1 : 0][3], (unsigned long long)__popcll(hit_by));
#endif
      tally(0, keep ? 0 : 1);
      tally(-1, 0);  // one more batch reached the ray-by-ray test
      if (keep) kept |= 1ull << b;
    }
#ifdef RT_HIST
    if (lane == 0) atomicAdd(&g_cand[transparent_counts_anywhere ? 1 : 0][2], (unsigned long long)__popcll(kept));
    const unsigned long long hist_kept_boxes = kept;
#endif
    // Record by record, for the bundle: the kept batches are numbered in list order (their numbers go through LDS), lane
    // 8 i + j takes record j of kept batch p0 + i and asks whether any ray of the bundle can pass the record's first slab
    // (bundle_slabs_pass).  The batch's header keeps one bit per surviving record (in the slot of gb, which has done its
    // work); a kept batch without a surviving record is not even staged.
    static_assert(kBatch == 8 && kHdrChunk <= 32, "the record-level bundle test maps 8 batches of 8 records onto the 64 lanes");
    if (RECTEST) {
      const uint32_t kept_all = (uint32_t)kept;
      const int nkept = __popc(kept_all);
      int *number = reinterpret_cast<int *>(rs);  // (the record buffers are not in use yet)
      if (lane < kHdrChunk && ((kept_all >> lane) & 1u)) number[__builtin_amdgcn_mbcnt_lo(kept_all, 0u)] = lane;  // (mbcnt: the kept batches below this lane)
      __builtin_amdgcn_wave_barrier();
#pragma unroll 1
      for (int p0 = 0; p0 < nkept; p0 += 8) {
        const int q = p0 + (lane >> 3);
        bool pass = false;
        int b = 0;
        if (q < nkept) {
          b = number[q];
          const uint32_t pos = beg + (h0 + (uint32_t)b) * kBatch + (uint32_t)(lane & 7);
          if (pos < beg + cnt) {
            const float *R = reinterpret_cast<const float *>(recs + pos);
            const RecA A = rec_a(R);
            const RecB Rb = rec_b(R);
            const float *H = hs + b * kHdrFloats;
            const float kc_max = __builtin_fmaf(H[8], bundle.onorm, __builtin_fabsf(H[9])) + g.slack;
            pass = bundle_slabs_pass(A, Rb.ab, Rb.p, kc_max, H[10], bundle);
            if (!ENDS && bundle.fan.on && pass) pass = fan_slabs_pass(A, Rb.ab, Rb.p, kc_max, H[8], __builtin_fabsf(H[9]), bundle.fan);
          }
        }
        const unsigned long long m = __ballot(pass);
        if (q < nkept && (lane & 7) == 0) hs[b * kHdrFloats + 11] = __uint_as_float((uint32_t)(m >> (lane & 56)) & 0xffu);
      }
      __builtin_amdgcn_wave_barrier();
      kept = __ballot(lane < kHdrChunk && ((kept_all >> lane) & 1u) && __float_as_uint(hs[lane * kHdrFloats + 11]) != 0u);
    }
#ifdef RT_HIST
    if (lane == 0) {
      unsigned long long *q = g_chunk[transparent_counts_anywhere ? 1 : 0];
      atomicAdd(&q[0], 1ull); if (candidates == 0) atomicAdd(&q[1], 1ull); if (hist_kept_boxes == 0) atomicAdd(&q[2], 1ull); if (kept == 0) atomicAdd(&q[3], 1ull);
      if (h0 == 0) { const int cls = nb <= 32 ? 0 : nb <= 64 ? 1 : nb <= 128 ? 2 : 3; atomicAdd(&q[4 + 2 * cls], 1ull); atomicAdd(&q[5 + 2 * cls], (unsigned long long)nb); }
    }
#endif
    ph.template end<3>();
    // the kept batches through two LDS buffers: the next one is on its way while this one is tested
    auto first_record = [&](int b) { return beg + (h0 + (uint32_t)b) * kBatch; };
    auto records_of = [&](uint32_t c0) { return (int)(beg + cnt - c0 < (uint32_t)kBatch ? beg + cnt - c0 : (uint32_t)kBatch); };
    int buf = 0;
    if (kept != 0) {
      const uint32_t c0 = first_record(__ffsll(kept) - 1);
      stage_batch(recs, c0, records_of(c0), rs, lane);
    }
    while (kept != 0) {
      const int b = __ffsll(kept) - 1;
      kept &= kept - 1;
      const uint32_t c0 = first_record(b);
      const float *rb = rs + buf * (kBatch * kRecFloats);
      wait_staged();
      if (kept != 0) {
        const uint32_t c1 = first_record(__ffsll(kept) - 1);
        stage_batch(recs, c1, records_of(c1), rs + (buf ^ 1) * (kBatch * kRecFloats), lane);
      }
      buf ^= 1;
      ph.template end<4, true>();
      const v2f k = *reinterpret_cast<const v2f *>(hs + b * kHdrFloats + 8);  // k1, -+k2
      batch(k.x, __builtin_fabsf(k.y));
      // the batch's records that some ray of the bundle may pass
      uint32_t todo = RECTEST ? (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(hs[b * kHdrFloats + 11])) : (1u << records_of(c0)) - 1u;
      tally(-2, __popc(todo));
#ifdef RT_HIST
      {
        int hrec = 0, hpairs = 0;
        for (uint32_t q = todo; q != 0; q &= q - 1) { const unsigned long long mm = __ballot(pass1(rb + (__ffs(q) - 1) * kRecFloats)); hrec += mm != 0; hpairs += __popcll(mm); }
        if (lane == 0) { atomicAdd(&g_hist[transparent_counts_anywhere ? 1 : 0][0][hrec], 1ull); atomicAdd(&g_hist[transparent_counts_anywhere ? 1 : 0][1][hrec], (unsigned long long)hpairs);
                         atomicAdd(&g_hist[transparent_counts_anywhere ? 1 : 0][1][0], (unsigned long long)__popc(todo)); }  // (no pairs in a batch without a passing record: the slot counts the records left by the bundle's slab test)
      }
#endif
      if (!RECTEST || __popc(todo) >= RT_DENSE) {
        // most records survive: all of them, four at a time through the first slab in straight-line code (LDS offsets
        // are immediates, the reads and FMA chains of four records overlap), then the whole test for those some lane passed
        const int n = records_of(c0);
        int j = 0;
        for (; j + 4 <= n; j += 4) {
          unsigned long long m[4], any = 0;
#pragma unroll
          for (int q = 0; q < 4; q++) { m[q] = __ballot(pass1(rb + (j + q) * kRecFloats)); any |= m[q]; }
          asm volatile("" : "+s"(any));  // (an opaque scalar: the branch stays a scalar branch)
          if (any != 0) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
              if (m[q] != 0) f(rb + (j + q) * kRecFloats, c0 + (uint32_t)(j + q), false);
            }
          }
        }
        todo &= ~0u << j;  // a partial batch's last records: one by one
      }
      for (; todo != 0; todo &= todo - 1) {
        const int j = __ffs(todo) - 1;
        f(rb + j * kRecFloats, c0 + (uint32_t)j, true);
      }
      // (asked once per batch, not per record: the records that follow a walk's last blocker test nothing, and the loops
      // above keep a simple shape)
      if (!more()) { wait_staged(); ph.template end<5>(); return; }
      __builtin_amdgcn_wave_barrier();
      ph.template end<5, true>();
    }
    __builtin_amdgcn_wave_barrier();
    ph.template end<5>();
  }
}
constexpr int kSliceFloats = kSliceFloatsC;

RT_D int rec_id(const float *R) { return __float_as_int(R[10]); }
// The exact test of a triangle the filter could not decide: i, ij, ik by index.
RT_D bool exact_test(const GridP &g, int id, f3 o, f3 r, float &d, float &u, float &v, bool *full = nullptr) {
  const float *G = g.geom + 9 * (long long)id;
  return tri_test(ld3(G), ld3(G + 3), ld3(G + 6), o, r, d, u, v, full);
}

// What one nearest-hit test concludes from the later stages, for a lane that passed the first slab.
// true: tri_test has to be run; false: whatever tri_test answers, the triangle cannot replace the current
// best (rejected, behind the origin, or surely not nearer than `best`).
RT_D bool nearest_needs_exact(const float *R, const RecA &A, PlTest &t, f3 o, f3 r, f3 m, float kc, float lmax, float best,
                              bool &s3) {
  s3 = false;
  const RecB B = rec_b(R);
  if (!pl_stage2(B.ab, B.p, r, m, t)) return false;
  s3 = true;
  const float sd = pl_sdist(A.n, B.an, o, t), e = lmax * kc, ad = __builtin_fabsf(t.det);
  const bool sure = pl_sign_sure(t, kc);
  const bool behind = sd + e < 0.0f;
  const bool farther = (sd - e > 0.0f) & (sd - e > best * (ad + kc) * 1.00001f);
  return !(sure & (behind | farther));
}
// The same for a shadow test.  Returns 0 = no effect on the shadow ray, 1 = tri_test has to be run,
// 2 = surely an accepted hit of a transparent triangle (ts *= kt), 3 = surely an opaque blocker in (1e-5, dmax).
RT_D int shadow_verdict(const float *R, const RecA &A, PlTest &t, f3 o, f3 r, f3 m, float kc, float lmax, float dmax,
                        float &kt, bool &s3) {
  s3 = false;
  const RecB B = rec_b(R);
  if (!pl_stage2(B.ab, B.p, r, m, t)) return 0;
  s3 = true;
  kt = B.kt;
  const float sd = pl_sdist(A.n, B.an, o, t), e = lmax * kc, ad = __builtin_fabsf(t.det);
  const bool sure = pl_sign_sure(t, kc), inside = pl_surely_inside(t, kc);
  if (kt > 0.0f) {
    if (sure & (sd + e < 0.0f)) return 0;             // d < 0: rejected
    if (inside & (sd - e > 0.0f)) return 2;
    return 1;
  }
  const bool out = (sd + e < 0.0000099f * (ad - kc)) | (sd - e > dmax * (ad + kc) * 1.00001f);
  if (sure & out) return 0;                           // wherever it is hit, not inside (1e-5, dmax)
  const bool in = (sd - e > 0.0000101f * (ad + kc)) & (sd + e < dmax * (ad - kc) * 0.99999f);
  if (inside & in) return 3;
  return 1;
}

// How a wavefront spends its lanes on triangle tests.  Three modes, chosen per step from what its
// unfinished rays are standing in (all give the reference's result for every ray; only the order in
// which independent tests are evaluated differs):
//   cell mode  many rays share one cell: the cell's list is staged in LDS once, every ray of the cell
//              tests every triangle (1 lane = 1 ray).  Coherent rays, any list length.
//   lane mode  rays scattered over many short lists: every lane walks its own list from L1/L2.
//   ray mode   a ray alone in a long list: the 64 lanes test 64 different triangles for that one ray
//              and the winner is found with a cross-lane reduction.
#ifndef RT_CELL_MIN_GROUP
#define RT_CELL_MIN_GROUP 32  // rays sharing the leader's cell from which cell mode is taken unconditionally
#endif
#ifndef RT_SHORT_LIST
#define RT_SHORT_LIST 12      // lists up to this length are walked in lane mode (32: ulica's shadow walks 2 % slower; temple the same)
#endif
#ifndef RT_LANE_MIN
#define RT_LANE_MIN 8         // lane mode needs at least this many rays with short lists
#endif
#ifndef RT_CELL_COST
#define RT_CELL_COST 6u       // cell mode's instructions per listed triangle, for the choice against ray mode
#endif
enum WalkMode { kCellMode = 0, kLaneMode = 1, kRayMode = 2 };
RT_D int pick_mode(int group, uint32_t leader_cnt, int short_lanes) {
  if (group >= RT_CELL_MIN_GROUP) return kCellMode;
  if (short_lanes >= RT_LANE_MIN) return kLaneMode;
  // The leader's group alone, in instructions the wavefront issues.  Cell mode: ~40 per batch header plus,
  // for the batches that survive the culling (about a third), staging and kBatch tests of ~20: ~10 per
  // listed triangle.  Ray mode, per ray of the group: ~80 per 64 triangles plus ~150 for the reduction.
  const uint32_t ray_cost = (uint32_t)group * (((leader_cnt + 63u) / 64u) * 80u + 150u);
  return leader_cnt * RT_CELL_COST <= ray_cost ? kCellMode : kRayMode;
}

// Work distribution of the persistent walk kernels: a ticket is 64 consecutive queue entries, taken from device-wide
// counters.  Atomics on one address are served one after another at the memory side (about 15 ns each, measured:
// 506 000 tickets from one counter took 9 ms whatever the kernels did in between), so a wavefront takes two tickets
// per atomic when there are many (1/RT_GRAB_DIV of its fair share, at most RT_GRAB_MAX), and the atomics are spread
// over several counters (below).  The tickets of one grab lie far apart in the queue (neighbouring tiles cost about
// the same, and a run of expensive ones in one wavefront would be the last to finish).  The last RT_GRAB_TAIL /
// RT_GRAB_TAIL_DEN of the queue are handed out one ticket at a time, so that the wavefronts finish together whatever
// a ticket costs.
#ifndef RT_GRAB_DIV
#define RT_GRAB_DIV 4
#endif
#ifndef RT_GRAB_MAX
#define RT_GRAB_MAX 2
#endif
#ifndef RT_GRAB_TAIL
#define RT_GRAB_TAIL 2
#endif
#ifndef RT_GRAB_TAIL_DEN
#define RT_GRAB_TAIL_DEN 16
#endif
// The counters: a walk kernel's slots are dealt out by RT_COUNTERS counters instead of one (counter k hands out the slots
// k, k + RT_COUNTERS, ...: each walks the whole queue, so the queue is still worked through front to back).  With a
// single counter the single-ticket tail of a shadow walk (an eighth of 506 000 tickets, 15 ns each: 0.95 ms) took
// longer to hand out than to work through, and four tickets per atomic were needed before it (temple 4K: shadow
// walk 4.14 -> 3.70 ms, nearest-hit walk 1.13 -> 1.03 ms with eight counters and two per atomic).  The counters lie 4 KB apart
// behind the control block (different memory channels).  A wavefront starts with the counter of its block and moves on
// to the next one when a counter has nothing left; it is done when all have run out.
#ifndef RT_COUNTERS
#define RT_COUNTERS 8
#endif
constexpr int kCounters = RT_COUNTERS, kCounterStride = 1024, kTicketArea = 4096;  // (words between counters; bytes from the control block to the first)
constexpr size_t kCtlBytes = kTicketArea + sizeof(uint32_t) * 2 * kMaxLevels * kCounters * kCounterStride;
RT_D uint32_t *ticket_counters(Ctl *ctl, int which, int level) {  // which: 0 nearest walk, 1 shadow walk
  return reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(ctl) + kTicketArea) + (size_t)((which * kMaxLevels + level) * kCounters) * kCounterStride;
}
struct Tickets {
  uint32_t per_grab, groups, singles, slot = 0, taken;  // slots 0..groups-1: per_grab tickets, `groups` apart; then single tickets
  uint32_t tried = 0;                                   // counters this wavefront has found empty
  RT_D explicit Tickets(unsigned long long total) {
    const unsigned long long tickets = (total + 63) / 64, waves = (unsigned long long)gridDim.x * (blockDim.x / 64);
    const unsigned long long share = tickets / (waves * RT_GRAB_DIV);
    per_grab = (uint32_t)(share < 1 ? 1 : share > RT_GRAB_MAX ? RT_GRAB_MAX : share);
    groups = per_grab > 1 ? (uint32_t)(tickets * (RT_GRAB_TAIL_DEN - RT_GRAB_TAIL) / RT_GRAB_TAIL_DEN / per_grab) : 0u;
    singles = (uint32_t)(tickets - (unsigned long long)groups * per_grab);
    taken = per_grab;
  }
  // base of the next ticket of this wavefront, false when nothing is left
  RT_D bool take(uint32_t *counters, int lane, uint32_t &base) {
    if (taken >= per_grab) {
      for (;;) {
        if (tried >= (uint32_t)kCounters) return false;
        const uint32_t k = (blockIdx.x + tried) % (uint32_t)kCounters;
        uint32_t s = 0;
        if (lane == 0) s = atomicAdd(counters + k * kCounterStride, 1u);
        s = __builtin_amdgcn_readfirstlane(s);
        // (slots are numbered below 2^32 / 64; a counter that has run out is asked at most once more by every wavefront)
        const unsigned long long global_slot = (unsigned long long)s * kCounters + k;
        if (global_slot < (unsigned long long)groups + singles) { slot = (uint32_t)global_slot; break; }
        tried++;  // the counters only grow: this one stays empty
      }
      taken = slot < groups ? 0u : per_grab - 1u;
    }
    const uint32_t ticket = slot < groups ? slot + taken * groups : groups * per_grab + (slot - groups);
    taken++;
    base = 64u * ticket;
    return true;
  }
};

// ------------------------------------------------------------------------------------------------
// rsqrt_ref and sqrt_ref (rt_math.h) against the double-precision idioms they stand for, for every one of the 2^32
// binary32 inputs: bad[0], bad[1] count the inputs whose results differ in any bit (two NaNs count as equal).
__global__ void __launch_bounds__(256) k_math_selfcheck(unsigned long long *__restrict__ bad) {
  unsigned long long n0 = 0, n1 = 0;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
    const float s = __uint_as_float((uint32_t)i);
    const float a = rsqrt_ref(s), b = rsqrt_exact(s), c = sqrt_ref(s), d = sqrt_exact(s);
    if (__float_as_uint(a) != __float_as_uint(b) && !(a != a && b != b)) n0++;
    if (__float_as_uint(c) != __float_as_uint(d) && !(c != c && d != d)) n1++;
  }
  for (int off = 32; off > 0; off >>= 1) { n0 += __shfl_down(n0, off); n1 += __shfl_down(n1, off); }
  if (lane_id() == 0) { atomicAdd(&bad[0], n0); atomicAdd(&bad[1], n1); }
}

// ------------------------------------------------------------------------------------------------
// Per-triangle preprocess for one camera (preprocess.c:17-32): ij, ik, unit normal flipped towards
// the observer, plane offset d.  geom: 9 floats (i, ij, ik); aux: 4 floats (n, d).
// Also the per-triangle record of the staged filter (rt_math.h), evaluated in double from the float edges
// and rounded once, with the triangle's own error constants k1, k2, and the surface's kt.
__global__ void k_preprocess(int nt, const float *__restrict__ verts, f3 ob, const int32_t *__restrict__ sid,
                             const SurfaceD *__restrict__ surf, float *__restrict__ geom, float *__restrict__ aux,
                             float *__restrict__ filt) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nt) return;
  f3 a = ld3(verts + 9 * t), b = ld3(verts + 9 * t + 3), c = ld3(verts + 9 * t + 6);
  f3 ab = sub3(b, a), ac = sub3(c, a);
  f3 io = normalize3(sub3(ob, a));
  f3 n = normalize3(cross3(ab, ac));
  if (dot3(n, io) < 0.0f) n = neg3(n);
  float d = -dot3(a, n);
  st3(geom + 9 * t, a); st3(geom + 9 * t + 3, ab); st3(geom + 9 * t + 6, ac);
  st3(aux + 4 * t, n); aux[4 * t + 3] = d;
  // The filter's constants, in double from the float i, ij, ik the reference computes with.  Vertices
  // A = i, B = i + ij, C = i + ik; relabelled (a', b', c') so that c' - a' is the longest edge.
  const double A[3] = {a.x, a.y, a.z}, AB[3] = {ab.x, ab.y, ab.z}, AC[3] = {ac.x, ac.y, ac.z};
  double B[3], C[3], BC[3];
  for (int k = 0; k < 3; k++) { B[k] = A[k] + AB[k]; C[k] = A[k] + AC[k]; BC[k] = AC[k] - AB[k]; }
  auto len = [](const double *v) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
  const double lab = len(AB), lac = len(AC), lbc = len(BC);
  double a2[3], ab2[3], ac2[3];
  if (lac >= lab && lac >= lbc) {
    for (int k = 0; k < 3; k++) { a2[k] = A[k]; ab2[k] = AB[k]; ac2[k] = AC[k]; }
  } else if (lab >= lbc) {
    for (int k = 0; k < 3; k++) { a2[k] = A[k]; ab2[k] = AC[k]; ac2[k] = AB[k]; }
  } else {
    for (int k = 0; k < 3; k++) { a2[k] = B[k]; ab2[k] = -AB[k]; ac2[k] = BC[k]; }
  }
  auto cross = [](const double *x, const double *y, double *z) {
    z[0] = x[1] * y[2] - x[2] * y[1]; z[1] = x[2] * y[0] - x[0] * y[2]; z[2] = x[0] * y[1] - x[1] * y[0];
  };
  double N[3], Q[3], P[3];
  cross(ac2, ab2, N); cross(ac2, a2, Q); cross(a2, ab2, P);
  const double L = fmax(lab, fmax(lac, lbc)), amax = fmax(len(A), fmax(len(B), len(C)));
  const double eps = 5.9604644775390625e-8;  // 2^-24
  float *F = filt + kRecFloats * t;
  F[0] = (float)N[0]; F[1] = (float)N[1]; F[2] = (float)N[2];
  F[3] = (float)(144.0 * eps * L * 1.001);
  F[4] = (float)ac2[0]; F[5] = (float)ac2[1]; F[6] = (float)ac2[2];
  F[7] = (float)Q[0]; F[8] = (float)Q[1]; F[9] = (float)Q[2];
  F[10] = __int_as_float(t);
  F[11] = (float)((144.0 * eps * L * amax + 56.0 * eps * L * L) * 1.001 + 1e-30);
  F[12] = (float)ab2[0]; F[13] = (float)ab2[1]; F[14] = (float)ab2[2];
  F[15] = (float)P[0]; F[16] = (float)P[1]; F[17] = (float)P[2];
  F[18] = (float)(a2[0] * N[0] + a2[1] * N[1] + a2[2] * N[2]);
  F[19] = surf[sid[t]].kt;
}

// ------------------------------------------------------------------------------------------------
// Grid build, pass 1 (voxelize.c:252-284): voxel index box of every triangle and the number of
// references each cell will receive.  box: lo[3], hi[3] per triangle, followed by the union box of every
// group of 64 consecutive triangles (at box + 6 nt), which lets k_grid_fill skip whole groups.
__global__ void __launch_bounds__(256) k_tri_boxes(int nt, const float *__restrict__ verts, GridP g, int *__restrict__ box,
                                                   uint32_t *__restrict__ counts) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = t < nt;
  int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {INT_MIN, INT_MIN, INT_MIN};
  if (valid) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      int a = f2i_x86((verts[9 * t + k] - g.dmin[k]) / g.cs[k]);
      int b = f2i_x86((verts[9 * t + 3 + k] - g.dmin[k]) / g.cs[k]);
      int c = f2i_x86((verts[9 * t + 6 + k] - g.dmin[k]) / g.cs[k]);
      lo[k] = max(min3i(a, b, c), 0);            // a vertex outside the domain would index outside the
      hi[k] = min(max3i(a, b, c), g.nv[k] - 1);  // reference's voxel array; clamp instead of faulting
      box[6 * t + k] = lo[k];
      box[6 * t + 3 + k] = hi[k];
    }
  }
  // a triangle's own lane counts it into a small box of cells; the 64 lanes share the cells of a large one
  const bool some = valid && hi[0] >= lo[0] && hi[1] >= lo[1] && hi[2] >= lo[2];  // (a triangle outside the domain covers no cell)
  const int ncover = some ? (hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1) : 0;
  if (ncover > 0 && ncover <= 32) {
    for (int i = lo[0]; i <= hi[0]; i++)
      for (int j = lo[1]; j <= hi[1]; j++)
        for (int k = lo[2]; k <= hi[2]; k++) atomicAdd(&counts[cell_index(g, i, j, k)], 1u);
  }
  for (unsigned long long big = __ballot(ncover > 32); big != 0; big &= big - 1) {
    const int src = __ffsll(big) - 1;
    int l[3], ext[3];
#pragma unroll
    for (int q = 0; q < 3; q++) { l[q] = __shfl(lo[q], src); ext[q] = __shfl(hi[q], src) - l[q] + 1; }
    const int total = ext[0] * ext[1] * ext[2];
    for (int e = lane_id(); e < total; e += 64) {
      const int k = e % ext[2], j = (e / ext[2]) % ext[1], i = e / (ext[2] * ext[1]);
      atomicAdd(&counts[cell_index(g, l[0] + i, l[1] + j, l[2] + k)], 1u);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int k = 0; k < 3; k++) { lo[k] = min(lo[k], __shfl_xor(lo[k], off)); hi[k] = max(hi[k], __shfl_xor(hi[k], off)); }
  }
  if (valid && lane_id() == 0) {
    int *gb = box + 6 * (long long)nt + 6 * (t >> 6);
#pragma unroll
    for (int k = 0; k < 3; k++) { gb[k] = lo[k]; gb[3 + k] = hi[k]; }
  }
}

// Everything that follows from the per-cell counts, in one 1024-thread block: exclusive scans into
// cell_start[0..ncells] (records) and hdr_start[0..ncells] (batch headers, ceil(count / kBatch) per cell), the list
// of the non-empty cells (nonempty[0..n), n stored at nonempty[ncells]), the occupancy bitmap (one bit per cell) and
// the voxel index box of the non-empty cells.
// Four consecutive cells per thread and step.
__global__ void __launch_bounds__(1024) k_grid_offsets(int n, int nv1, int nv2, const uint32_t *__restrict__ counts,
                                                       uint32_t *__restrict__ start, uint32_t *__restrict__ hstart,
                                                       uint32_t *__restrict__ nonempty, uint32_t *__restrict__ occ,
                                                       int32_t *__restrict__ occ_box) {
  __shared__ uint32_t wave_sum[3][16];
  __shared__ uint32_t carry_s[3];
  __shared__ int box_s[6];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid < 3) { carry_s[tid] = 0; box_s[tid] = INT_MAX; box_s[3 + tid] = INT_MIN; }
  int blo[3] = {INT_MAX, INT_MAX, INT_MAX}, bhi[3] = {INT_MIN, INT_MIN, INT_MIN};
  __syncthreads();
  for (int base = 0; base < n; base += 4096) {
    const int idx = base + 4 * tid;
    uint32_t v[4], x[3] = {0, 0, 0};  // records, headers, non-empty cells of this thread's four cells
    uint32_t bits = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      v[q] = idx + q < n ? counts[idx + q] : 0u;
      x[0] += v[q]; x[1] += (v[q] + kBatch - 1) / kBatch; x[2] += v[q] > 0u ? 1u : 0u;
      bits |= (v[q] > 0u ? 1u : 0u) << q;
      if (v[q] > 0u) {
        const int c = idx + q, ijk[3] = {c / (nv2 * nv1), (c / nv2) % nv1, c % nv2};
#pragma unroll
        for (int a = 0; a < 3; a++) { blo[a] = min(blo[a], ijk[a]); bhi[a] = max(bhi[a], ijk[a]); }
      }
    }
    // occupancy: 8 threads' four bits make one word (cells idx..idx+3 are bits 4 (tid % 8) .. of word idx / 32)
    bits <<= 4 * (tid & 7);
    bits |= __shfl_xor(bits, 1); bits |= __shfl_xor(bits, 2); bits |= __shfl_xor(bits, 4);
    if ((tid & 7) == 0 && idx < n) occ[idx >> 5] = bits;
    uint32_t mine[3], incl[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
      mine[a] = incl[a] = x[a];
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = __shfl_up(incl[a], off);
        if (lane >= off) incl[a] += y;
      }
      if (lane == 63) wave_sum[a][wid] = incl[a];
    }
    __syncthreads();
    uint32_t pre[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
      pre[a] = carry_s[a] + incl[a] - mine[a];
      for (int wv = 0; wv < wid; wv++) pre[a] += wave_sum[a][wv];
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (idx + q < n) {
        start[idx + q] = pre[0]; hstart[idx + q] = pre[1];
        if (v[q] > 0u) nonempty[pre[2]] = (uint32_t)(idx + q);
      }
      pre[0] += v[q]; pre[1] += (v[q] + kBatch - 1) / kBatch; pre[2] += v[q] > 0u ? 1u : 0u;
    }
    __syncthreads();
    if (tid == 1023) { carry_s[0] = pre[0]; carry_s[1] = pre[1]; carry_s[2] = pre[2]; }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 3; a++) {
    if (blo[a] != INT_MAX) { atomicMin(&box_s[a], blo[a]); atomicMax(&box_s[3 + a], bhi[a]); }
  }
  __syncthreads();
  if (tid < 6) occ_box[tid] = box_s[tid];
  if (tid == 0) { start[n] = carry_s[0]; hstart[n] = carry_s[1]; nonempty[n] = carry_s[2]; }
}

// Grid build, pass 2: one wavefront per cell walks the triangles in index order, 64 at a time, and
// appends those whose voxel box contains the cell; ballot + prefix count keeps the list in ascending
// triangle order, which is the reference's insertion order and therefore its tie-break.
__global__ void __launch_bounds__(256) k_grid_fill(int nt, GridP g, const uint32_t *__restrict__ nonempty, const int *__restrict__ box,
                                                   TriRec *__restrict__ recs, int32_t *__restrict__ refs) {
  const uint32_t slot = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (slot >= nonempty[g.ncells]) return;
  const int cell = (int)nonempty[slot];
  uint32_t beg = g.cell_start[cell], end = g.cell_start[cell + 1];
  const int lane = lane_id();
  int ck = cell % g.nv[2], cj = (cell / g.nv[2]) % g.nv[1], ci = cell / (g.nv[2] * g.nv[1]);
  uint32_t pos = beg;
  const int ngroups = (nt + 63) / 64;
  const int *gbox = box + 6 * (long long)nt;
  for (int g0 = 0; g0 < ngroups && pos < end; g0 += 64) {
    // which of the next 64 groups of 64 triangles can contain a triangle of this cell at all
    bool cand = false;
    if (g0 + lane < ngroups) {
      const int *b = gbox + 6 * (g0 + lane);
      cand = ci >= b[0] && ci <= b[3] && cj >= b[1] && cj <= b[4] && ck >= b[2] && ck <= b[5];
    }
    for (unsigned long long gm = __ballot(cand); gm != 0 && pos < end; gm &= gm - 1) {
      const int t = (g0 + __ffsll(gm) - 1) * 64 + lane;
      bool in = false;
      if (t < nt) {
        const int *b = box + 6 * t;
        in = ci >= b[0] && ci <= b[3] && cj >= b[1] && cj <= b[4] && ck >= b[2] && ck <= b[5];
      }
      unsigned long long m = __ballot(in);
      if (in) {
        uint32_t p = pos + __popcll(m & lanes_below(lane));
        const float4 *F = reinterpret_cast<const float4 *>(g.filt + kRecFloats * (long long)t);
        float4 *R = reinterpret_cast<float4 *>(recs + p);
#pragma unroll
        for (int q = 0; q < kRecVecs; q++) R[q] = F[q];
        refs[p] = t;
      }
      pos += __popcll(m);
    }
  }
}
// Grid build, pass 3: the walk order of every cell's list.  The reference's result does not depend on the
// order a list is walked in, except that equal distances go to the smaller triangle index (the nearest walk
// compares indices for that) and that the transparency product of a shadow ray is taken in list order.
// Lists of 33..4096 opaque triangles are therefore re-ordered along a Morton curve of the triangles'
// centroids inside the cell (triangles larger than half a cell first), so that the 16 records of a batch
// lie close together and their bounding sphere is small; all other lists keep the ascending order.
// One block per cell; 64-bit keys (class, Morton code, position) sorted with a bitonic network in LDS.
constexpr int kSortMax = 4096, kSortThreads = 1024;
RT_D uint32_t morton10(uint32_t x) {
  x &= 0x3ffu;
  x = (x | (x << 16)) & 0x30000ffu;
  x = (x | (x << 8)) & 0x300f00fu;
  x = (x | (x << 4)) & 0x30c30c3u;
  x = (x | (x << 2)) & 0x9249249u;
  return x;
}
__global__ void __launch_bounds__(kSortThreads) k_grid_sort(GridP g, const uint32_t *__restrict__ nonempty, const TriRec *__restrict__ src,
                                                   TriRec *__restrict__ dst) {
  __shared__ unsigned long long keys[kSortMax];
  __shared__ int transparent_s;
  if (blockIdx.x >= nonempty[g.ncells]) return;
  const int cell = (int)nonempty[blockIdx.x];
  const uint32_t beg = g.cell_start[cell], end = g.cell_start[cell + 1];
  const int cnt = (int)(end - beg), tid = threadIdx.x;
  bool sorted = cnt > 32 && cnt <= kSortMax;
  if (sorted) {
    if (tid == 0) transparent_s = 0;
    __syncthreads();
    int any = 0;
    for (int j = tid; j < cnt; j += kSortThreads) any |= src[beg + j].kt > 0.0f ? 1 : 0;
    if (any) atomicOr(&transparent_s, 1);
    __syncthreads();
    sorted = transparent_s == 0;
  }
  if (sorted) {
    int n2 = 64;
    while (n2 < cnt) n2 <<= 1;
    const int ck = cell % g.nv[2], cj = (cell / g.nv[2]) % g.nv[1], ci = cell / (g.nv[2] * g.nv[1]);
    const float lo[3] = {g.dmin[0] + ci * g.cs[0], g.dmin[1] + cj * g.cs[1], g.dmin[2] + ck * g.cs[2]};
    const float big = 0.5f * fmaxf(g.cs[0], fmaxf(g.cs[1], g.cs[2]));
    for (int j = tid; j < n2; j += kSortThreads) {
      unsigned long long key = ~0ull;
      if (j < cnt) {
        const float *G = g.geom + 9 * (long long)src[beg + j].id;
        uint32_t code = 0;
        float extent = 0.0f;
#pragma unroll
        for (int q = 0; q < 3; q++) {
          const float a = G[q], b = G[q] + G[3 + q], c = G[q] + G[6 + q];
          extent = fmaxf(extent, fmaxf(a, fmaxf(b, c)) - fminf(a, fminf(b, c)));
          const float rel = ((a + b + c) * (1.0f / 3.0f) - lo[q]) / g.cs[q];
          const uint32_t u = (uint32_t)fminf(fmaxf(rel * 1024.0f, 0.0f), 1023.0f);  // NaN -> 0: any order is a valid order
          code |= morton10(u) << (2 - q);
        }
        key = ((unsigned long long)((extent > big ? 0u : 1u) << 30 | code) << 32) | (unsigned)j;
      }
      keys[j] = key;
    }
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1) {
      for (int jj = k >> 1; jj > 0; jj >>= 1) {
        for (int i = tid; i < n2; i += kSortThreads) {
          const int l = i ^ jj;
          if (l > i) {
            const unsigned long long x = keys[i], y = keys[l];
            const bool up = (i & k) == 0;
            if ((x > y) == up) { keys[i] = y; keys[l] = x; }
          }
        }
        __syncthreads();
      }
    }
  }
  for (int j = tid; j < cnt; j += kSortThreads) {
    const int from = sorted ? (int)(keys[j] & 0xffffffffull) : j;
    const float4 *S = reinterpret_cast<const float4 *>(src + beg + from);
    float4 *D = reinterpret_cast<float4 *>(dst + beg + j);
#pragma unroll
    for (int q = 0; q < kRecVecs; q++) D[q] = S[q];
  }
}

// Grid build, pass 4: the header of every batch of kBatch consecutive records of a cell's list: bounding
// box and sphere of the batch's triangles (vertices i, i+ij, i+ik as the exact test sees them), maxima of the
// records' error constants, transparency flag.  One block per non-empty cell, each wavefront 64 / kBatch batches at a time.
__global__ void __launch_bounds__(256) k_grid_bounds(GridP g, const uint32_t *__restrict__ nonempty, BatchHdr *__restrict__ hdrs) {
  if (blockIdx.x >= nonempty[g.ncells]) return;
  const int cell = (int)nonempty[blockIdx.x];
  const uint32_t beg = g.cell_start[cell], end = g.cell_start[cell + 1];
  const uint32_t hbeg = g.hdr_start[cell];
  const int lane = lane_id();
  for (uint32_t c0 = beg + 64u * (threadIdx.x >> 6); c0 < end; c0 += 256) {
    const uint32_t p = c0 + lane;
    const bool valid = p < end;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    float v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    float k1 = 0.0f, k2 = 0.0f, transparent = 0.0f, bad = 0.0f, ga = 0.0f, gb = 0.0f;
    if (valid) {
      const TriRec &R = g.cell_tris[p];
      k1 = R.k1; k2 = R.k2; transparent = R.kt > 0.0f ? 1.0f : 0.0f;
      // gb = 32 eps L^2 / |n|, ga = 0.9 gb L  (see box_test_bundle); L from k1 = 144 eps L
      const float L = R.k1 * (1.0f / (144.0f * 5.9604644775390625e-8f));
      const float nn = sqrtf(R.n[0] * R.n[0] + R.n[1] * R.n[1] + R.n[2] * R.n[2]), inv = 1.0f / nn;
      hdrs[hbeg + (p - beg) / kBatch].normal[(p - beg) % kBatch] = pack_normal(R.n[0] * inv, R.n[1] * inv, R.n[2] * inv);
      gb = (32.0f * 5.9604644775390625e-8f) * L * L / nn * 1.0001f;
      ga = 0.9f * gb * L * 1.0001f;
      if (!(gb >= 0.0f) || !(ga >= 0.0f)) bad = 1.0f;  // NaN (degenerate triangle): never cull this batch
      const float *G = g.geom + 9 * (long long)R.id;
#pragma unroll
      for (int q = 0; q < 3; q++) {
        v[q] = G[q]; v[3 + q] = G[q] + G[3 + q]; v[6 + q] = G[q] + G[6 + q];
        lo[q] = fminf(v[q], fminf(v[3 + q], v[6 + q]));
        hi[q] = fmaxf(v[q], fmaxf(v[3 + q], v[6 + q]));
      }
#pragma unroll
      for (int q = 0; q < 9; q++) if (!(fabsf(v[q]) < INFINITY)) bad = 1.0f;  // NaN or infinite vertex: never cull this batch
    }
#pragma unroll
    for (int off = kBatch / 2; off > 0; off >>= 1) {
#pragma unroll
      for (int q = 0; q < 3; q++) { lo[q] = fminf(lo[q], __shfl_xor(lo[q], off)); hi[q] = fmaxf(hi[q], __shfl_xor(hi[q], off)); }
      k1 = fmaxf(k1, __shfl_xor(k1, off)); k2 = fmaxf(k2, __shfl_xor(k2, off));
      ga = fmaxf(ga, __shfl_xor(ga, off)); gb = fmaxf(gb, __shfl_xor(gb, off));
      transparent = fmaxf(transparent, __shfl_xor(transparent, off)); bad = fmaxf(bad, __shfl_xor(bad, off));
    }
    const float cx = 0.5f * (lo[0] + hi[0]), cy = 0.5f * (lo[1] + hi[1]), cz = 0.5f * (lo[2] + hi[2]);
    float r2 = 0.0f;
    if (valid) {
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const float dx = v[3 * q] - cx, dy = v[3 * q + 1] - cy, dz = v[3 * q + 2] - cz;
        r2 = fmaxf(r2, dx * dx + dy * dy + dz * dz);
      }
    }
#pragma unroll
    for (int off = kBatch / 2; off > 0; off >>= 1) r2 = fmaxf(r2, __shfl_xor(r2, off));
    if (valid && (lane & (kBatch - 1)) == 0) {
      BatchHdr &h = hdrs[hbeg + (p - beg) / kBatch];  // (its normals are written above, one per lane)
      h.c[0] = cx; h.c[1] = cy; h.c[2] = cz;
      const float cn = sqrtf(cx * cx + cy * cy + cz * cz);
      h.radius = bad != 0.0f ? INFINITY : (sqrtf(r2) * 1.001f + 0.00001f * cn) * 1.0001f + 1e-30f;
#pragma unroll
      for (int q = 0; q < 3; q++) h.h[q] = bad != 0.0f ? INFINITY : (0.5f * (hi[q] - lo[q]) * 1.001f + 0.00001f * cn) * 1.0001f + 1e-30f;
      h.inv_cos = INFINITY;
      h.k1 = k1; h.k2 = transparent != 0.0f ? -k2 : k2;  // k2 > 0 (rt_math.h)
      h.ga = ga; h.gb = gb;
    }
  }
}
// One bit per cell (k_grid_offsets): does it list any triangle?  The walk kernels keep this map in LDS (kOccWords words, grids
// up to 49 152 cells) so that stepping through empty cells, most of a shadow ray's way to a distant light,
// costs an LDS read instead of a trip to L2 per step.
constexpr int kOccWords = 1536;
// Copies the occupancy map into the block's LDS if it fits; returns whether it did.
// occ_box_s: the kernel's copy of the box of the non-empty cells (g.occ_box), for walk_init.
RT_D bool load_occupancy(const GridP &g, uint32_t *occ_s, int32_t *occ_box_s) {
  if (threadIdx.x < 6) occ_box_s[threadIdx.x] = g.occ_box[threadIdx.x];
  const bool fits = g.ncells <= kOccWords * 32;
  if (fits)
    for (int i = threadIdx.x; i * 32 < g.ncells; i += blockDim.x) occ_s[i] = g.occ[i];
  __syncthreads();
  return fits;
}
RT_D bool cell_occupied(const GridP &g, const uint32_t *occ_s, bool in_lds, int cell) {
  const uint32_t wd = in_lds ? occ_s[cell >> 5] : g.occ[cell >> 5];
  return (wd >> (cell & 31)) & 1u;
}
// Steps the walk to its next non-empty cell; false = it left its range first.  steps counts the cells entered.
// Two instances, so that the map is read with an LDS or a global load, not through a generic pointer.
template <bool IN_LDS, bool COUNT>
RT_D bool walk_to_occupied(const GridP &g, const uint32_t *occ_s, f3 r, Walk &w, unsigned long long &steps) {
  for (;;) {
    if (!walk_step(g, r, w)) return false;
    if (COUNT) steps++;
    const uint32_t wd = IN_LDS ? occ_s[w.cell >> 5] : g.occ[w.cell >> 5];
    if ((wd >> (w.cell & 31)) & 1u) return true;
  }
}
template <bool COUNT>
RT_D bool walk_to_occupied(const GridP &g, const uint32_t *occ_s, bool in_lds, f3 r, Walk &w, unsigned long long &steps) {
  return in_lds ? walk_to_occupied<true, COUNT>(g, occ_s, r, w, steps) : walk_to_occupied<false, COUNT>(g, occ_s, r, w, steps);
}

// ------------------------------------------------------------------------------------------------
// Primary rays of a chunk of tiles (vectormath.h:108-118, voxelize.c:417-471).  Ray index = local
// tile * 64 + (y&7)*8 + (x&7), so one wavefront of the nearest-hit kernel is one 8x8 pixel tile.
// primary_dir: the pixel's ray direction, false = index past the chunk or pixel outside the image.
RT_D bool primary_dir(const PrimaryP &pp, int idx, f3 &dir) {
  dir = mk3(0.0f, 0.0f, 0.0f);
  if (idx >= pp.ntiles * kTilePixels) return false;
  const CameraD &cam = pp.cam;
  const int lt = pp.tile0 + (idx >> 6), slot = idx & 63;
  const long long gt = (long long)lt * pp.world + pp.rank;
  const int x = (int)(gt % pp.tiles_x) * kTile + (slot & 7), y = (int)(gt / pp.tiles_x) * kTile + (slot >> 3);
  if (!(x < cam.sw && y < cam.sh)) return false;
  float w_inv = 1.0f / cam.sw, h_inv = 1.0f / cam.sh;
  float xc = x * w_inv, yc = y * h_inv;
  f3 ul = ld3(cam.ul), ur = ld3(cam.ur), bl = ld3(cam.bl), ob = ld3(cam.ob);
  dir = mk3(xc * (ur.x - ul.x) + yc * (bl.x - ul.x) + ul.x - ob.x,
            xc * (ur.y - ul.y) + yc * (bl.y - ul.y) + ul.y - ob.y,
            xc * (ur.z - ul.z) + yc * (bl.z - ul.z) + ul.z - ob.z);
  dir = normalize3(dir);
  return true;
}
// The ray as the walk wants it: with its start voxel (i = -1: it never enters the grid, -2: no pixel).
RT_D Ray primary_ray(const PrimaryP &pp, const GridP &g, int idx, bool with_voxel) {
  Ray ry;
  ry.o[0] = pp.cam.ob[0]; ry.o[1] = pp.cam.ob[1]; ry.o[2] = pp.cam.ob[2];
  ry.current = -1; ry.pad[0] = ry.pad[1] = 0;
  ry.i = -2; ry.j = ry.k = 0;
  f3 dir;
  const bool pixel = primary_dir(pp, idx, dir);
  st3(ry.r, dir);
  if (pixel) {
    ry.i = 0;
    int i, j, k;
    if (with_voxel) { if (startup_voxel(g, ld3(pp.cam.ob), dir, i, j, k)) { ry.i = i; ry.j = j; ry.k = k; } else ry.i = -1; }
  }
  return ry;
}
// Start of a chunk: resets its control block (queue lengths, tickets, statistics); n0 rays at level 0.
__global__ void __launch_bounds__(256) k_chunk_init(int n0, QueueCaps caps, Ctl *__restrict__ ctl) {
  static_assert(sizeof(Ctl) % 4 == 0 && sizeof(Ctl) / 4 <= 256, "one thread per word of the control block");
  if (threadIdx.x < sizeof(Ctl) / 4) reinterpret_cast<uint32_t *>(ctl)[threadIdx.x] = 0u;
  static_assert(2 * kMaxLevels * kCounters <= 256, "one thread per ticket counter");
  if (threadIdx.x < 2 * kMaxLevels * kCounters) ticket_counters(ctl, 0, 0)[(size_t)threadIdx.x * kCounterStride] = 0u;
  __syncthreads();
  if (threadIdx.x == 0) ctl->count[0] = n0;
  if (threadIdx.x <= kMaxLevels) ctl->cap[threadIdx.x] = caps.cap[threadIdx.x];
}
// Entries of level L's queue: what the level above produced, cut at the queue's capacity (the host sizes the queues of
// the secondary rays from what earlier chunks produced and renders a chunk again, with larger queues, if one was cut).
RT_D int queue_len(const Ctl *__restrict__ ctl, int level) {
  const int n = ctl->count[level], cap = ctl->cap[level];
  return n < cap ? n : cap;
}

// ------------------------------------------------------------------------------------------------
// Nearest-hit walk (voxelize.c:536-613) as a persistent wavefront kernel.
//
// Each wavefront draws tickets of 64 rays from a counter (Tickets); one lane owns one ray and runs the
// reference's 3D-DDA for it (Walk).  The triangle tests are organised per visit, not per lane: the wavefront
// picks the cell of its least advanced unfinished lane and one of three modes (pick_mode).  In cell mode
// every lane standing in that cell joins, the cell's batch headers and the records of the batches that
// survive the culling are staged in LDS (stream_cell) and every joined lane tests every staged record with
// the staged filter, tri_test deciding what the filter cannot; lane mode and ray mode read the records from
// memory.  A lane that found a hit before the voxel's exit distance retires; otherwise it steps to its next
// non-empty cell.  Every listed triangle of a visited cell is accounted for (tested, or proven irrelevant by
// the culling), and of equal distances the smallest triangle index wins, as in the reference's scan of its
// ascending list.
// The shaded hit of a ray without its texture and its secondary rays (raytrace.c:170-182): hit point, the normal turned
// against the ray, the surface's colour.  A miss keeps its marker (h.tri < 0) and zeros.  k_hit, and the nearest-hit walk
// itself where a level neither spawns rays nor meets a textured triangle.
RT_D Node plain_node(const ShadeP &sp, const Hit &h, f3 o, f3 r, int &sid) {
  Node nd;
  nd.tri = h.tri; nd.refl = -1; nd.refr = -1;
  nd.p[0] = nd.p[1] = nd.p[2] = 0.0f; nd.n[0] = nd.n[1] = nd.n[2] = 0.0f; nd.c[0] = nd.c[1] = nd.c[2] = 0.0f;
  sid = -1;
  if (h.tri >= 0) {
    const f3 onew = raypoint3(o, r, h.d);  // voxelize.c:587
    f3 norm = ld3(sp.tri_aux + 4 * h.tri);
    if (dot3(r, norm) > 0.0f) norm = neg3(norm);  // raytrace.c:175-178
    sid = sp.tri_sid[h.tri];
    st3(nd.p, onew); st3(nd.n, norm); st3(nd.c, ld3(sp.surf[sid].color));
  }
  return nd;
}
// A kernel's arguments read again from its argument segment (scalar loads that hit the scalar cache) instead of being
// held in scalar registers from entry to exit: the walk kernels have more arguments than registers for them, and what
// does not fit is parked in the lanes of a vector register and fetched back one v_readlane per word and use (in units
// of the eight-word tuples it was loaded in).  The parts of a ticket outside its cell visits read what they need this
// way; the opaque pointer keeps every read where it is written.
template <typename T>
RT_D T fresh_arg(size_t offset) {
#if defined(__HIP_DEVICE_COMPILE__)
  const char __attribute__((address_space(4))) *p = (const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return *reinterpret_cast<const T __attribute__((address_space(4))) *>(p + offset);
#else
  (void)offset;
  return T();  // (the host pass only parses the kernels)
#endif
}
#include "rt_view_kernels.h"
// k_nearest's arguments: one structure, so that a member's offset in it is its offset in the argument segment
struct NearestArgs {
  GridP g; const Ray *rays; Hit *hits; Ctl *ctl; int level; PrimaryP prim;
  ShadeP sp; Node *nodes; SOrigin *sorg; int node_base;  // for the instances that shade their hits themselves (FUSE)
  ViewListsP vl;  // for the instance that takes the primary rays' hits from the view lists (VL)
};
// VL: the tickets of primary rays are answered from the view lists where they can be (view_candidates, view_verify)
template <bool COUNT, bool RECTEST, bool FUSE, bool VL = false>
__global__ void __launch_bounds__(256, VL ? RT_WAVES_VIEW : RT_WAVES_NEAREST) k_nearest(NearestArgs args) {
  static_assert(!VL || !COUNT, "the counting variant reports the reference's walks");
  const GridP &g = args.g;
  const Ray *__restrict__ rays = args.rays;
  Hit *__restrict__ hits = args.hits;
  Ctl *__restrict__ ctl = args.ctl;
  const int level = args.level;
  const PrimaryP &prim = args.prim;
  __shared__ __attribute__((aligned(16))) float lds_batches[4][kSliceFloats];
  __shared__ uint32_t occ_s[kOccWords];
  float *slice = lds_batches[threadIdx.x >> 6];
  __shared__ int32_t occ_box_s[8];
  const bool occ_in_lds = load_occupancy(g, occ_s, occ_box_s);
  const int lane = lane_id();
  const int nrays = queue_len(ctl, level);
  const uint32_t *__restrict__ cstart = g.cell_start;
  const TriRec *__restrict__ recs = g.cell_tris;
  unsigned long long nS = 0, nT = 0, nLane = 0, nWave = 0, nQ = 0, nS2 = 0, nS3 = 0, nEx = 0, nVisit = 0, nBatch = 0, nCull = 0, nFull = 0;
  uint32_t nPixels = 0, nTraced = 0, nHits = 0;  // wave-uniform tallies, one atomic each per wavefront
  uint32_t nViewed = 0, nViewFell = 0;           // tickets answered from the view lists / tried there and walked after all
  Phases ph;
  ph.begin();
  RT_TICKET(const unsigned long long born = __builtin_amdgcn_s_memtime(); unsigned long long busy = 0;)
  // One triangle for one ray: the filter, then tri_test for what it cannot decide.  Everything after the
  // first slab sits behind one scalar branch that is taken only if some lane of the wavefront passed it.
  auto test = [&](const float *R, f3 to, f3 tr, f3 tm, float kc, int cur, float &b, int &bid, float &bu, float &bv,
                  uint32_t pos, uint32_t &bpos) {
    const RecA A = rec_a(R);
    PlTest t;
    const bool p1 = pl_stage1(A.n, A.ac, A.q, tr, tm, kc, t);
    if (wave_any(p1)) {
      asm volatile("; some lane passed the first slab");  // keeps the scalar branch from being merged into the per-lane one
      bool s3 = false, ex = false;
      if (p1) ex = nearest_needs_exact(R, A, t, to, tr, tm, kc, g.lmax, b, s3);
      if (COUNT) { nS2++; nS3 += __ballot(s3) != 0; nEx += __ballot(ex) != 0; }
      if (ex) {
        const int id = rec_id(R);
        float d, u, v;
        // the reference keeps the first of equal distances in its ascending list = the smallest triangle index; written
        // on the index, the choice does not depend on the order the list is walked in (bid = -1: the voxel's exit distance)
        bool full = false;
        if (exact_test(g, id, to, tr, d, u, v, COUNT ? &full : nullptr) && id != cur && (d < b || (d == b && id < bid))) { b = d; bid = id; bu = u; bv = v; bpos = pos; }
        if (COUNT && full) nFull++;
      }
      return true;
    }
    return false;
  };
  Tickets tickets((unsigned long long)nrays);
  for (;;) {
    uint32_t base;
    if (!tickets.take(ticket_counters(fresh_arg<Ctl *>(offsetof(NearestArgs, ctl)), 0, fresh_arg<int>(offsetof(NearestArgs, level))), lane, base)) break;
    RT_TICKET(TicketLog tl; tl.begin();)
    const int idx = base + lane;
    Ray ry;
    ry.i = -2;
    {  // (the ticket's rays, with its own copies of the arguments)
      const GridP g = fresh_arg<GridP>(offsetof(NearestArgs, g));
      const PrimaryP prim = fresh_arg<PrimaryP>(offsetof(NearestArgs, prim));
      const Ray *__restrict__ rays = fresh_arg<const Ray *>(offsetof(NearestArgs, rays));
      if (idx < nrays) { if (prim.on) ry = primary_ray(prim, g, idx, true); else ry = rays[idx]; }
    }
    bool alive = ry.i >= 0;
    nPixels += __popcll(__ballot(ry.i != -2));
    nTraced += __popcll(__ballot(alive));
    const f3 o = ld3(ry.o), r = ld3(ry.r);
    f3 pm = mk3(0.0f, 0.0f, 0.0f);  // o x r, |o| and o.r: the ray's side of the filter and of the batch culling
    float onorm = 0.0f, odr = 0.0f;
    if (alive) pl_ray(o, r, pm, onorm, odr);
    Hit h;
    h.tri = alive ? -1 : -2;
    h.d = h.u = h.v = 0.0f; h.i = h.j = h.k = 0; h.pad = 0;
    Walk w;
    w.cell = 0;
    uint32_t mybeg = 0, mycnt = 0, myhbeg = 0;  // triangle list of the cell the walk stands in and its first batch header
    if (alive) {
      const GridP g = fresh_arg<GridP>(offsetof(NearestArgs, g));
      const uint32_t *__restrict__ cstart = g.cell_start;
      if (COUNT) nQ++;
      const int glo[3] = {0, 0, 0}, ghi[3] = {g.nv[0] - 1, g.nv[1] - 1, g.nv[2] - 1};
      walk_init(g, o, r, ry.i, ry.j, ry.k, glo, ghi, !COUNT, occ_box_s, w);  // (the counting variant reports the reference's step counts)
      // stand on the first non-empty cell of the walk (the start voxel counts as visited)
      if (COUNT) nS++;
      if (!VL) {
        if (!cell_occupied(g, occ_s, occ_in_lds, w.cell)) alive = walk_to_occupied<COUNT>(g, occ_s, occ_in_lds, r, w, nS);
        if (alive) {  // the list of the cell the walk stands on, fetched once the lanes are together again
          mybeg = cstart[w.cell]; mycnt = cstart[w.cell + 1] - mybeg; myhbeg = g.hdr_start[w.cell];
          if (COUNT) nT += mycnt;
        }
      }
    }
    if (VL) {
      ph.end<0>();
      // The view lists instead of the walk, for a ticket of primary rays.  Where they do not apply (the frame's lists did
      // not fit their buffer) or do not agree with a ray's walk, the ticket takes the walk below, from its untouched state.
      const GridP g = fresh_arg<GridP>(offsetof(NearestArgs, g));
      const ViewListsP vl = fresh_arg<ViewListsP>(offsetof(NearestArgs, vl));
      const PrimaryP prim = fresh_arg<PrimaryP>(offsetof(NearestArgs, prim));
      if (vl.on && prim.on && __ballot(alive) != 0) {
        const long long gt = (long long)(prim.tile0 + (int)(base >> 6)) * prim.world + prim.rank;
        const int tcol = (int)(gt % prim.tiles_x) >> vl.shift, trow = (int)(gt / prim.tiles_x) >> vl.shift;
        const uint32_t texel = (uint32_t)(trow * vl.ntx + tcol);
        const uint32_t lbeg = vl.offsets[texel], lend = vl.offsets[texel + 1], lall = vl.offsets[vl.ntx * vl.nty];
        if (lall <= vl.cap && tcol < vl.ntx && trow < vl.nty) {
          nViewFell++;
          float best = INFINITY, best_u = 0.0f, best_v = 0.0f;
          int best_id = -1;
          uint32_t diag[3] = {0u, 0u, 0u};
          ph.end<1>();
          // (the walk's test with the triangle's i, ij, ik read from the slot instead of from memory)
          view_candidates(g, vl.refs, lbeg, lend, alive, o, r, onorm, slice, lane, diag, ph, [&](const float *R) {
            const float kc = alive ? __builtin_fmaf(R[3], onorm, R[11]) + g.slack : -INFINITY;
            const RecA A = rec_a(R);
            PlTest t;
            const bool p1 = pl_stage1(A.n, A.ac, A.q, r, pm, kc, t);
            if (wave_any(p1)) {
              asm volatile("; some lane passed the first slab");
              bool s3 = false, ex = false;
              if (p1) ex = nearest_needs_exact(R, A, t, o, r, pm, kc, g.lmax, best, s3);
              if (ex) {
                const int id = rec_id(R);
                float d, u, v;
                if (tri_test(ld3(R + kRecFloats), ld3(R + kRecFloats + 3), ld3(R + kRecFloats + 6), o, r, d, u, v) && id != ry.current &&
                    (d < best || (d == best && id < best_id))) { best = d; best_id = id; best_u = u; best_v = v; }
              }
            }
          });
          Walk v = w;
          ph.end<5>();
          bool agreed = view_verify(g, vl.tri_box, alive && best_id >= 0, best_id, best, r, ry.i, ry.j, ry.k, v);
          if (vl.pad != 0 && ((base >> 6) & 1u) != 0u) agreed = false;  // (a test's way to see every other ticket take the walk after the lists)
          ph.end<6>();
          if (agreed) {
            if (alive && best_id >= 0) { h.tri = best_id; h.d = best; h.u = best_u; h.v = best_v; w = v; }
            alive = false;
            nViewed++; nViewFell--;
          }
          RT_TICKET(tl.visits = diag[0]; tl.batches = diag[1]; tl.tests = diag[2];)  // (diagnostic build: what the ticket looked at)
        }
      }
      if (__ballot(alive) != 0) {  // the ticket walks after all: what the set-up left out
        if (alive && !cell_occupied(g, occ_s, occ_in_lds, w.cell)) alive = walk_to_occupied<COUNT>(g, occ_s, occ_in_lds, r, w, nS);
        if (alive) { mybeg = g.cell_start[w.cell]; mycnt = g.cell_start[w.cell + 1] - mybeg; myhbeg = g.hdr_start[w.cell]; }
      }
      ph.end<7>();
    }
    ph.end<0>();
    for (;;) {
      const unsigned long long m = __ballot(alive);
      if (m == 0) break;
      const int leader = least_advanced(m, alive, w.tx, w.ty, w.tz);
      const int cell = __builtin_amdgcn_readlane(w.cell, leader);
      const uint32_t beg = __builtin_amdgcn_readlane(mybeg, leader), cnt = __builtin_amdgcn_readlane(mycnt, leader);
      const uint32_t hbeg = __builtin_amdgcn_readlane(myhbeg, leader);
      const bool in_group = alive && w.cell == cell;
      const bool is_short = alive && mycnt <= RT_SHORT_LIST;
      const int mode = pick_mode(__popcll(__ballot(in_group)), cnt, __popcll(__ballot(is_short)));
      if (COUNT) nVisit++;
      RT_TICKET(tl.visits++;)
#ifdef RT_HIST
      const unsigned long long hist_w0 = nWave, hist_l0 = nLane;
#endif
      // exit distance of this voxel: a hit only counts if it lies before it (voxelize.c:569,574)
      float best = min3_ref(w.tx + w.dtx, w.ty + w.dty, w.tz + w.dtz);
      int best_id = -1;
      float best_u = 0.0f, best_v = 0.0f;
      bool proc;  // this lane's current cell is fully tested after this step
      uint32_t nopos = 0;
      if (mode == kCellMode) {
        proc = in_group;
        // lanes outside the cell test with kc = -inf: rejected by the first slab
        float kc = -INFINITY;
        stream_cell<RECTEST, false, !RECTEST>(g, hbeg, beg, cnt, slice, lane, make_bundle<false>(proc, o, r, pm, odr, onorm, best, INFINITY, central_lane(proc, lane, prim.on != 0, leader)), false, ph,
                    [&](f3 c, float delta, f3 hw, float W) {
                      return wave_mask(proc & box_test_ray(c, delta, hw, W, r, pm, odr, best, false));
                    },
                    [&](int n, int culled) {
                      if (n >= 0) { if (COUNT) { nBatch += n; nCull += culled; } }
                      RT_TICKET(else if (n == -1) tl.batches++; else tl.tests += culled;)
                    },
                    [&](float k1, float k2) { kc = proc ? __builtin_fmaf(k1, onorm, k2) + g.slack : -INFINITY; },
                    [&](const float *R) {
                      if (COUNT) { nWave++; if (proc) nLane++; }
                      const RecA A = rec_a(R);
                      PlTest t;
                      return pl_stage1(A.n, A.ac, A.q, r, pm, kc, t);
                    },
                    [&](const float *R, uint32_t, bool first) {
                      if (COUNT && first) { nWave++; if (proc) nLane++; }
                      return test(R, o, r, pm, kc, ry.current, best, best_id, best_u, best_v, 0u, nopos);
                    },
                    [] { return true; });
      } else if (mode == kLaneMode) {
        proc = is_short;
        const uint32_t pn = proc ? mycnt : 0u;
        for (uint32_t i = 0; __ballot(i < pn) != 0; ++i) {
          if (COUNT) { nWave++; if (i < pn) nLane++; }
          if (i < pn) {
            const float *R = reinterpret_cast<const float *>(recs + mybeg + i);
            test(R, o, r, pm, __builtin_fmaf(R[3], onorm, R[11]) + g.slack, ry.current, best, best_id, best_u, best_v, 0u, nopos);
          }
        }
      } else {
        // ray mode: the leader's ray against 64 triangles per step; lane L takes list positions L, L+64, ...
        proc = lane == leader;
        const f3 lo = lane_bcast3(o, leader), lr = lane_bcast3(r, leader), lm = lane_bcast3(pm, leader);
        const float lon = lane_bcast(onorm, leader);
        const int lcur = __builtin_amdgcn_readlane(ry.current, leader);
        float cb = lane_bcast(best, leader), cu = 0.0f, cv = 0.0f;
        int cid = -1;
        uint32_t cpos = 0xffffffffu;
        for (uint32_t t0 = beg; t0 < beg + cnt; t0 += 64) {
          const uint32_t t = t0 + lane;
          if (COUNT) { nWave++; if (t < beg + cnt) nLane++; }
          if (t < beg + cnt) {
            const float *R = reinterpret_cast<const float *>(recs + t);
            test(R, lo, lr, lm, __builtin_fmaf(R[3], lon, R[11]) + g.slack, lcur, cb, cid, cu, cv, t, cpos);  // ascending t: earliest wins in a lane
          }
        }
        // smallest d, ties to the smallest triangle index: the reference's strict `d < dmin` scan of its ascending list
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          const float od = __shfl_xor(cb, off), ou = __shfl_xor(cu, off), ov = __shfl_xor(cv, off);
          const uint32_t op = __shfl_xor(cpos, off);
          const int oi = __shfl_xor(cid, off);
          if (od < cb || (od == cb && oi >= 0 && (cid < 0 || oi < cid))) { cb = od; cpos = op; cid = oi; cu = ou; cv = ov; }
        }
        if (proc) { best = cb; best_id = cid; best_u = cu; best_v = cv; }
      }
#ifdef RT_HIST
      if (COUNT) {
        unsigned long long dl = nLane - hist_l0;
        for (int off = 32; off > 0; off >>= 1) dl += __shfl_down(dl, off);
        if (lane == 0) { atomicAdd(&g_mode[0][mode][0], 1ull); atomicAdd(&g_mode[0][mode][1], nWave - hist_w0); atomicAdd(&g_mode[0][mode][2], dl); atomicAdd(&g_mode[0][mode][3], (unsigned long long)__popcll(__ballot(proc))); }
      }
#endif
      ph.end<7>();
      if (proc) {
        const GridP g = fresh_arg<GridP>(offsetof(NearestArgs, g));
        const uint32_t *__restrict__ cstart = g.cell_start;
        if (best_id >= 0) {
          h.tri = best_id; h.d = best; h.u = best_u; h.v = best_v;  // (its voxel: w.cell, where the walk now stays)
          alive = false;
        } else {
          alive = walk_to_occupied<COUNT>(g, occ_s, occ_in_lds, r, w, nS);
        }
        if (alive) {
          mybeg = cstart[w.cell]; mycnt = cstart[w.cell + 1] - mybeg; myhbeg = g.hdr_start[w.cell];
          if (COUNT) nT += mycnt;
        }
      }
      ph.end<6>();
    }
    Hit *__restrict__ hits = fresh_arg<Hit *>(offsetof(NearestArgs, hits));
    // the ray's result: its hit record for k_hit, or (FUSE) the shaded hit and the origin of its shadow walks themselves
    auto emit = [&](const Hit &hh, Hit *__restrict__ out) {
      if (FUSE) {
        const ShadeP fsp = fresh_arg<ShadeP>(offsetof(NearestArgs, sp));
        Node *__restrict__ nodes = fresh_arg<Node *>(offsetof(NearestArgs, nodes));
        SOrigin *__restrict__ sorg = fresh_arg<SOrigin *>(offsetof(NearestArgs, sorg));
        const int node_base = fresh_arg<int>(offsetof(NearestArgs, node_base));
        if (idx < nrays) {
          int sid;
          const Node nd = plain_node(fsp, hh, o, r, sid);
          nodes[node_base + idx] = nd;
          SOrigin so;
          so.p[0] = nd.p[0]; so.p[1] = nd.p[1]; so.p[2] = nd.p[2]; so.tri = nd.tri;
          sorg[node_base + idx] = so;
        }
      } else if (idx < nrays) {
        out[idx] = hh;
      }
    };
    {
      const GridP g = fresh_arg<GridP>(offsetof(NearestArgs, g));
      if (!FUSE && h.tri >= 0) cell_to_voxel(g, w.cell, h.i, h.j, h.k);
      emit(h, hits);
      nHits += __popcll(__ballot(h.tri >= 0));
    }
    RT_TICKET(ph.end<0>(); if (level == 0) tl.end(0, base, lane, ph); busy += __builtin_amdgcn_s_memtime() - tl.t0;)
  }
  RT_TICKET(if (level == 0) TicketLog::life(0, lane, busy, born);)
  ph.end<0>();
  ph.flush(0, lane);
  if (lane == 0) {
    if (level == 0) {
      atomicAdd(&ctl->stat[kPixels], (unsigned long long)nPixels);
      atomicAdd(&ctl->stat[kPixelsTraced], (unsigned long long)nTraced);
    }
    atomicAdd(&ctl->stat[kShaded], (unsigned long long)nHits);
    if (VL && (nViewed | nViewFell) != 0u) {
      atomicAdd(&ctl->stat[kViewListTickets], (unsigned long long)nViewed);
      atomicAdd(&ctl->stat[kViewListFallbacks], (unsigned long long)nViewFell);
    }
  }
  if (COUNT) {
    // wave-level sums, one atomic per counter per wavefront
    for (int off = 32; off > 0; off >>= 1) {
      nS += __shfl_down(nS, off); nT += __shfl_down(nT, off); nLane += __shfl_down(nLane, off); nQ += __shfl_down(nQ, off);
      nFull += __shfl_down(nFull, off);
    }
    if (lane == 0) {
      atomicAdd(&ctl->stat[kNearS], nS); atomicAdd(&ctl->stat[kNearT], nT);
      atomicAdd(&ctl->stat[kNearLaneTests], nLane); atomicAdd(&ctl->stat[kNearWaveTests], nWave * 64ull);
      atomicAdd(&ctl->stat[kNearestQ], nQ); atomicAdd(&ctl->stat[kNearFull], nFull);
      atomicAdd(&ctl->stat[kNearStage2], nS2); atomicAdd(&ctl->stat[kNearStage3], nS3); atomicAdd(&ctl->stat[kNearExact], nEx);
      atomicAdd(&ctl->stat[kNearVisits], nVisit); atomicAdd(&ctl->stat[kNearBatches], nBatch); atomicAdd(&ctl->stat[kNearCulled], nCull);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The order in which a level's nodes get their shadow walks and their colours: by the cell of the hit point, so that the
// 64 nodes of a shadow ticket start their walks close together whatever pixels they come from (the nodes of a level
// below 0 are appended in the order their rays were spawned: neighbours in the queue are neighbours on the surface the
// parent rays left, not on the one they hit).  Nodes without a hit come last.  A counting sort: nodes per bucket (a
// part of a cell), an exclusive scan over the buckets, positions handed out bucket by bucket.  The order inside a cell is
// whatever the atomics make it; no result depends on it (a node's walks and colour are its own).
// parts: a cell is cut into parts^3 buckets (the host picks what the bucket array affords)
RT_D uint32_t node_cell(const GridP &g, const SOrigin &nd, int parts) {
  int i, j, k;
  if (nd.tri < 0 || !voxel_of(g, ld3(nd.p), i, j, k)) return (uint32_t)g.ncells * (uint32_t)(parts * parts * parts);  // no hit (or a hit point outside the grid): the last bucket
  // the part of the cell, from the fraction of the same quotient that gave the cell (any rounding only moves a node to a neighbouring bucket)
  const float fx = (nd.p[0] - g.dmin[0]) / g.cs[0] - (float)i, fy = (nd.p[1] - g.dmin[1]) / g.cs[1] - (float)j, fz = (nd.p[2] - g.dmin[2]) / g.cs[2] - (float)k;
  const int si = min(max((int)(fx * parts), 0), parts - 1), sj = min(max((int)(fy * parts), 0), parts - 1), sk = min(max((int)(fz * parts), 0), parts - 1);
  return (uint32_t)cell_index(g, i, j, k) * (uint32_t)(parts * parts * parts) + (uint32_t)((si * parts + sj) * parts + sk);
}
// adds one per active lane to bucket[key], one atomic per distinct key of the wavefront; returns the lane's position
RT_D uint32_t bucket_take(uint32_t *bucket, uint32_t key, bool active, int lane) {
  uint32_t pos = 0;
  unsigned long long todo = __ballot(active);
  while (todo != 0) {
    const int first = __ffsll(todo) - 1;
    const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, first);
    const unsigned long long same = __ballot(active && key == k);
    uint32_t base = 0;
    if (lane == first) base = atomicAdd(&bucket[k], (uint32_t)__popcll(same));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, first);
    if (active && key == k) pos = base + (uint32_t)__popcll(same & lanes_below(lane));
    todo &= ~same;
  }
  return pos;
}
__global__ void __launch_bounds__(256) k_order_count(GridP g, const SOrigin *__restrict__ sorg, const Ctl *__restrict__ ctl, int level,
                                                     int node_base, int parts, uint32_t *__restrict__ bucket) {
  const int n = queue_len(ctl, level), lane = lane_id();
  for (int i0 = blockIdx.x * blockDim.x; i0 < n; i0 += gridDim.x * blockDim.x) {
    const int i = i0 + threadIdx.x;
    const bool valid = i < n;
    const uint32_t key = valid ? node_cell(g, sorg[node_base + i], parts) : 0u;
    (void)bucket_take(bucket, key, valid, lane);
  }
}
// Exclusive scan of bucket[0 .. nb-1] in place, in two launches: every block scans its kScanTile buckets and leaves
// their sum in totals[block]; one block then scans the totals.  k_order_place adds a bucket's block total to its start.
constexpr int kScanTile = 4096;
__global__ void __launch_bounds__(1024) k_order_scan(uint32_t *__restrict__ bucket, int nb, uint32_t *__restrict__ totals) {
  __shared__ uint32_t part[1024];
  const int base = blockIdx.x * kScanTile + threadIdx.x * 4;
  uint32_t v[4], sum = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) { v[q] = base + q < nb ? bucket[base + q] : 0u; sum += v[q]; }
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const uint32_t add = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  uint32_t run = part[threadIdx.x] - sum;
#pragma unroll
  for (int q = 0; q < 4; q++) { if (base + q < nb) bucket[base + q] = run; run += v[q]; }
  if (threadIdx.x == 1023) totals[blockIdx.x] = part[1023];
}
__global__ void __launch_bounds__(1024) k_order_scan_totals(uint32_t *__restrict__ totals, int n) {
  __shared__ uint32_t part[1024];
  uint32_t carry = 0;
  for (int t0 = 0; t0 < n; t0 += 1024) {
    const int t = t0 + threadIdx.x;
    const uint32_t v = t < n ? totals[t] : 0u;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const uint32_t add = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0u;
      __syncthreads();
      part[threadIdx.x] += add;
      __syncthreads();
    }
    if (t < n) totals[t] = carry + part[threadIdx.x] - v;
    carry += part[1023];
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256) k_order_place(GridP g, const SOrigin *__restrict__ sorg, const Ctl *__restrict__ ctl, int level,
                                                     int node_base, int parts, uint32_t *__restrict__ bucket, const uint32_t *__restrict__ totals,
                                                     uint32_t *__restrict__ order) {
  const int n = queue_len(ctl, level), lane = lane_id();
  for (int i0 = blockIdx.x * blockDim.x; i0 < n; i0 += gridDim.x * blockDim.x) {
    const int i = i0 + threadIdx.x;
    const bool valid = i < n;
    const uint32_t key = valid ? node_cell(g, sorg[node_base + i], parts) : 0u;
    const uint32_t pos = bucket_take(bucket, key, valid, lane);
    if (valid) order[node_base + pos + totals[key / kScanTile]] = (uint32_t)i;
  }
}

// ------------------------------------------------------------------------------------------------
// Hit kernel (raytrace.c:170-204): turns the hits of one level into shaded-hit records and emits the
// secondary rays.  Normal flip, surface colour, procedural texture + bump (fused here), reflected and
// refracted rays.  Children are appended to the next level's queue with one atomic per wavefront
// (ballot + prefix count), so a child's slot is its parent's link.
__global__ void __launch_bounds__(256, RT_WAVES_HIT) k_hit(GridP g, ShadeP sp, const Ray *__restrict__ rays, const Hit *__restrict__ hits,
                                             Node *__restrict__ nodes, SOrigin *__restrict__ sorg, const float *__restrict__ geom,
                                             Ray *__restrict__ next_rays, Ctl *__restrict__ ctl, int level,
                                             int node_base, int next_base, int next_cap, bool spawn, PrimaryP prim) {
  const int n = queue_len(ctl, level);
  const int lane = lane_id();
  // block-stride loop: the launch is sized for the queue's capacity, cut at a few thousand blocks; a block's
  // wavefronts stay together (the child slots come from a ballot)
  for (int idx0 = blockIdx.x * blockDim.x; idx0 < n; idx0 += gridDim.x * blockDim.x) {
  const int idx = idx0 + threadIdx.x;
  const bool valid = idx < n;
  Hit h; h.tri = -2;
  Ray ry;
  if (valid) { h = hits[idx]; if (prim.on) ry = primary_ray(prim, g, idx, false); else ry = rays[idx]; }
  int sid;
  Node nd = plain_node(sp, h, ld3(ry.o), ld3(ry.r), sid);
  bool want_refl = false, want_refr = false;
  f3 refl_dir = mk3(0, 0, 0), refr_dir = mk3(0, 0, 0);
  if (h.tri >= 0) {
    const f3 r = ld3(ry.r);
    const SurfaceD sf = sp.surf[sid];
    f3 norm = ld3(nd.n);
    if (sid == 7) {  // raytrace.c:183
      int slot = h.tri == sp.tex_tri[0] ? 0 : h.tri == sp.tex_tri[1] ? 1 : h.tri == sp.tex_tri[2] ? 2 : h.tri == sp.tex_tri[3] ? 3 : -1;
      if (slot >= 0) {
        f3 nc = ld3(nd.c);
        apply_texture(slot, ld3(geom + 9 * h.tri + 3), ld3(geom + 9 * h.tri + 6), h.u, h.v, norm, nc);
        st3(nd.n, norm); st3(nd.c, nc);
      }
    }
    if (spawn) {
      const f3 inv = neg3(r);
      if (sf.kr > 0.0f) { want_refl = true; refl_dir = reflect3(norm, inv, dot3(norm, inv)); }
      if (sf.kt > 0.0f) { want_refr = true; refr_dir = refract3(norm, inv, sf.eta); }
    }
  }
  // append children (uniform control flow from here on)
#pragma unroll
  for (int kind = 0; kind < 2; kind++) {
    const bool want = kind == 0 ? want_refl : want_refr;
    unsigned long long m = __ballot(want);
    if (m == 0) continue;
    int base = 0;
    const int first = __ffsll(m) - 1;
    if (lane == first) base = atomicAdd(&ctl->count[level + 1], __popcll(m));
    base = __builtin_amdgcn_readlane(base, first);
    if (want) {
      int slot = base + __popcll(m & lanes_below(lane));
      if (slot < next_cap) {
        Ray c;
        st3(c.o, ld3(nd.p));
        st3(c.r, kind == 0 ? refl_dir : refr_dir);
        c.i = h.i; c.j = h.j; c.k = h.k;
        c.current = h.tri;
        c.pad[0] = c.pad[1] = 0;
        next_rays[slot] = c;
        if (kind == 0) nd.refl = next_base + slot; else nd.refr = next_base + slot;
      }
    }
  }
  if (valid) {
    nodes[node_base + idx] = nd;
    SOrigin so;
    so.p[0] = nd.p[0]; so.p[1] = nd.p[1]; so.p[2] = nd.p[2]; so.tri = nd.tri;
    sorg[node_base + idx] = so;
  }
  }
}

// Replay of the order in which the reference consumes rand() for planar lights (raytrace.c:245-251).
// rtRayTrace shades a hit after it has traced the hit's reflected and then its refracted ray
// (raytrace.c:192-204 before :241), pixels in row-major order: the hits of a frame draw their
// 32*npl values in post-order of each pixel's ray tree, trees in pixel order.  The shaded hit with
// rank k uses the values starting at 32*npl*k.  Three passes over the node arrays:
//   k_subtree_hits   deepest level first: shaded hits in the subtree of every node
//   k_tree_starts    exclusive prefix of the primary nodes' subtree sizes in pixel order (one block)
//   k_assign_ranks   top level first: start of each child's subtree, then the node's own rank
__global__ void __launch_bounds__(256) k_subtree_hits(const Node *__restrict__ nodes, int32_t *__restrict__ size,
                                                      const Ctl *__restrict__ ctl, int level, int node_base) {
  const int n = queue_len(ctl, level);
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
    const Node v = nodes[node_base + idx];
    int32_t s = v.tri >= 0 ? 1 : 0;
    if (v.refl >= 0) s += size[v.refl];
    if (v.refr >= 0) s += size[v.refr];
    size[node_base + idx] = s;
  }
}
// Pixel-order scan of the level-0 subtree sizes of the whole image.  sizes: per rank `per_rank` entries, tile-major in
// the rank's own tile numbering (tile t of the image is tile t / world of rank t % world); the ranks of this rank's own
// pixels go to `rank` (its level-0 nodes).  world = 1: sizes is the rank's own array.
__global__ void __launch_bounds__(1024) k_tree_starts(int w, int h, int tiles_x, const int32_t *__restrict__ sizes,
                                                      long long per_rank, int world, int me, int32_t *__restrict__ rank) {
  __shared__ uint32_t wave_sum[16];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n = w * h;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    int p = base + tid;
    uint32_t v = 0;
    int node = 0, owner = 0;
    if (p < n) {
      int x = p % w, y = p / w;
      const int tile = (y >> 3) * tiles_x + (x >> 3);
      owner = tile % world;
      node = (tile / world) * kTilePixels + (y & 7) * kTile + (x & 7);
      v = (uint32_t)sizes[owner * per_rank + node];
    }
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t y = __shfl_up(x, off);
      if (lane >= off) x += y;
    }
    if (lane == 63) wave_sum[wid] = x;
    __syncthreads();
    uint32_t prefix = carry_s;
    for (int wv = 0; wv < wid; wv++) prefix += wave_sum[wv];
    if (p < n && owner == me) rank[node] = (int32_t)(prefix + x - v);
    __syncthreads();
    if (tid == 1023) carry_s = prefix + x;
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256) k_assign_ranks(const Node *__restrict__ nodes, const int32_t *__restrict__ size,
                                                      int32_t *__restrict__ rank, const Ctl *__restrict__ ctl, int level,
                                                      int node_base) {
  const int n = queue_len(ctl, level);
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
    const int gidx = node_base + idx;
    const Node v = nodes[gidx];
    const int32_t start = rank[gidx];  // set by k_tree_starts (level 0) or by the parent
    const int32_t a = v.refl >= 0 ? size[v.refl] : 0, b = v.refr >= 0 ? size[v.refr] : 0;
    if (v.refl >= 0) rank[v.refl] = start;
    if (v.refr >= 0) rank[v.refr] = start + a;
    rank[gidx] = start + a + b;
  }
}

// Position of shadow-query slot `slot` of a shaded hit: a point light, or sample d of planar light c
// at a + eta*ab + psi*ac with eta, psi = rand()/(float)RAND_MAX (raytrace.c:250-261).
RT_D void slot_light(const ShadeP &sp, int slot, int rand_rank, f3 &pos, float &flux, f3 &color) {
  if (slot < sp.nl) {
    const LightD L = sp.lights[slot];
    pos = ld3(L.p); flux = L.flux; color = ld3(L.color);
  } else {
    const int q = slot - sp.nl, c = q / kPlanarSamples;
    const PlanarD P = sp.planar[c];
    const int32_t *rv = sp.rand_tab + ((long long)rand_rank * sp.npl * kPlanarSamples + q) * 2;
    const float eta = rv[0] / 2147483648.0f, psi = rv[1] / 2147483648.0f;  // (float)RAND_MAX == 2^31
    pos = ld3(P.a);
    pos = add3(pos, mul3(ld3(P.ab), eta));
    pos = add3(pos, mul3(ld3(P.ac), psi));
    flux = P.flux / kPlanarSamples;
    color = ld3(P.color);
  }
}

#include "rt_light_kernels.h"

// ------------------------------------------------------------------------------------------------
// Shadow walk (voxelize.c:615-738), any-hit, persistent wavefront kernel with the same per-cell
// organisation as k_nearest.  Work item = (shadow slot, shaded hit) in slot-major order, so the 64
// lanes of a ticket are neighbouring hit points looking at the same light.  Result per item: the
// transparency product ts (>= 0) if the light is visible, -1 if the point is in shadow, stored slot-major too
// (ts_out[slot * ts_stride + node]: one full line per ticket).
// The reference's per-triangle shadow cache (voxelize.c:647-655) is scan-order state; it is dropped.
// ENDS: the bundle of a visit knows that its rays end at their lights (Bundle.re, .T): sharper culling where the rays of
// a ticket start far apart and meet at the light; where they start on one 8x8 tile it costs more than it saves.
// k_shadow's arguments (one structure, like NearestArgs)
struct ShadowArgs {
  GridP g; ShadeP sp; const SOrigin *sorg; const int32_t *rand_rank; float *ts_out; long long ts_stride; Ctl *ctl;
  int level, node_base; const uint32_t *order;
  LightListsP ll;
  const uint32_t *ticket_list;  // null, or (LL instance, queue length a multiple of 64): [0] how many tickets, then their numbers: only these are done
};
// LL: tickets whose walks all end at one point light are answered from that light's lists where they can be (below)
template <bool COUNT, bool RECTEST, bool ENDS, bool LL = false>
__global__ void __launch_bounds__(256, ENDS ? RT_WAVES_ENDS : LL ? RT_WAVES_LL : RT_WAVES) k_shadow(ShadowArgs args) {
  static_assert(!LL || (!COUNT && RECTEST && !ENDS), "the light lists go with the fan of a point light's shadow rays, and not with the counting variant");
  const GridP &g = args.g;
  const ShadeP &sp = args.sp;
  const SOrigin *__restrict__ sorg = args.sorg;
  const int32_t *__restrict__ rand_rank = args.rand_rank;
  float *__restrict__ ts_out = args.ts_out;
  const long long ts_stride = args.ts_stride;
  Ctl *__restrict__ ctl = args.ctl;
  const int level = args.level, node_base = args.node_base;
  const uint32_t *__restrict__ order = args.order;
  __shared__ __attribute__((aligned(16))) float lds_batches[4][kSliceFloats];
  __shared__ uint32_t occ_s[kOccWords];
  float *slice = lds_batches[threadIdx.x >> 6];
  __shared__ int32_t occ_box_s[8];
  const bool occ_in_lds = load_occupancy(g, occ_s, occ_box_s);
  const int lane = lane_id();
  const int n = queue_len(ctl, level);
  const long long total = (long long)n * sp.nslots;
  const uint32_t *__restrict__ cstart = g.cell_start;
  const TriRec *__restrict__ recs = g.cell_tris;
  unsigned long long nS = 0, nT = 0, nQ = 0, nW = 0, nLane = 0, nWave = 0, nS2 = 0, nS3 = 0, nEx = 0, nVisit = 0, nBatch = 0, nCull = 0, nFull = 0;
  Phases ph;
  ph.begin();
  RT_TICKET(const unsigned long long born = __builtin_amdgcn_s_memtime(); unsigned long long busy = 0;)
  const uint32_t *__restrict__ ticket_list = LL ? args.ticket_list : nullptr;
  Tickets tickets(ticket_list ? 64ull * ticket_list[0] : (unsigned long long)total);
  uint32_t nListed = 0, nListFell = 0;  // tickets answered from the light lists / tried and walked after all
  for (;;) {
    uint32_t base;
    if (!tickets.take(ticket_counters(fresh_arg<Ctl *>(offsetof(ShadowArgs, ctl)), 1, fresh_arg<int>(offsetof(ShadowArgs, level))), lane, base)) break;
    if (LL && ticket_list) base = 64u * fresh_arg<const uint32_t *>(offsetof(ShadowArgs, ticket_list))[1 + (base >> 6)];
    RT_TICKET(TicketLog tl; tl.begin();)
    const long long item = (long long)base + lane;
    bool alive = false;
    int slot = 0, nidx = 0, current = -1, where = 0;
    f3 a = mk3(0, 0, 0), r = mk3(0, 0, 0), pm = mk3(0, 0, 0);  // pm = a x r, onorm >= |a|: the filter's ray side
    float dmax = 0.0f, ts = 1.0f, result = 1.0f, onorm = 0.0f, odr = 0.0f;
    bool have_result = false;
    Walk w;
    w.cell = 0;
    uint32_t mybeg = 0, mycnt = 0, myhbeg = 0;  // triangle list of the cell the walk stands in and its first batch header
    f3 light = mk3(0.0f, 0.0f, 0.0f);  // where this lane's ray ends
    {  // (the set-up of the ticket's walks, with its own copies of the arguments)
    const GridP g = fresh_arg<GridP>(offsetof(ShadowArgs, g));
    const ShadeP sp = fresh_arg<ShadeP>(offsetof(ShadowArgs, sp));
    const SOrigin *__restrict__ sorg = fresh_arg<const SOrigin *>(offsetof(ShadowArgs, sorg));
    const int32_t *__restrict__ rand_rank = fresh_arg<const int32_t *>(offsetof(ShadowArgs, rand_rank));
    const uint32_t *__restrict__ order = fresh_arg<const uint32_t *>(offsetof(ShadowArgs, order));
    const int node_base = fresh_arg<int>(offsetof(ShadowArgs, node_base));
    const uint32_t *__restrict__ cstart = g.cell_start;
    if (item < total) {
      // slot = item / n: one division per wavefront; its 64 items span at most two slots unless the queue is tiny
      const uint32_t slot0 = base / (uint32_t)n, rest = base - slot0 * (uint32_t)n + (uint32_t)lane;
      if (n >= 64) {
        slot = (int)(slot0 + (rest >= (uint32_t)n ? 1u : 0u));
        nidx = (int)(rest >= (uint32_t)n ? rest - (uint32_t)n : rest);
      } else {
        slot = (int)(slot0 + rest / (uint32_t)n);
        nidx = (int)(rest % (uint32_t)n);
      }
      const int at = nidx;  // the node's place in the level's order: where its results go
      if (order) nidx = (int)order[node_base + at];
      where = node_base + at;
      const SOrigin nd = sorg[node_base + nidx];
      current = nd.tri;
      if (current >= 0) {
        if (COUNT) nQ++;
        have_result = true;
        a = ld3(nd.p);
        f3 b, lcol; float lflux;
        slot_light(sp, slot, rand_rank ? rand_rank[node_base + nidx] : 0, b, lflux, lcol);
        light = b;
        r = ray3(a, b);
        const float kt_cur = LL ? 0.0f : sp.surf[sp.tri_sid[current]].kt;  // (the instance with the light lists runs in scenes without a transparent triangle)
        bool lit_walk = true;
        if (kt_cur == 0.0f && dot3(r, ld3(sp.tri_aux + 4 * current)) <= 0.0f) { result = -1.0f; lit_walk = false; }  // :640-644
        if (lit_walk) {
          dmax = dist3(a, b);
          int ai[3], bi[3];
          // a point outside the grid: the reference reports an error and treats the light as visible (:661-670)
          if (voxel_of(g, a, ai[0], ai[1], ai[2]) && voxel_of(g, b, bi[0], bi[1], bi[2])) {
            int lo[3], hi[3];
#pragma unroll
            for (int c = 0; c < 3; c++) { lo[c] = ai[c] < bi[c] ? ai[c] : bi[c]; hi[c] = ai[c] < bi[c] ? bi[c] : ai[c]; }
            walk_init(g, a, r, ai[0], ai[1], ai[2], lo, hi, !COUNT, occ_box_s, w);
            pl_ray(a, r, pm, onorm, odr);
            alive = true;
            if (COUNT) { nW++; nS++; }
            // (with the light lists a ticket seldom walks: its walks go to their first non-empty cells only if it does)
            if (!LL && !cell_occupied(g, occ_s, occ_in_lds, w.cell)) alive = walk_to_occupied<COUNT>(g, occ_s, occ_in_lds, r, w, nS);
          }
        }
      }
    }
    if (!LL && alive) {  // the list of the cell the walk stands on, fetched once the lanes are together again
      mybeg = cstart[w.cell]; mycnt = cstart[w.cell + 1] - mybeg; myhbeg = g.hdr_start[w.cell];
      if (COUNT) nT += mycnt;
    }
    }
    // The ticket's walks as the fan of one point light, if they all end at the same one (tickets that are a tile's hit
    // points, in the kernels built without ENDS, i.e. for scenes without secondary rays: with the fan compiled into ulica's
    // kernel its shadow walks took 114 instead of 107 ms, whichever tickets used it).  Built once per ticket, from all its
    // walks: the groups of its cell visits are nearly the whole ticket.
    // Bounds over all the ticket's walks, once per ticket (make_bundle): |a|, the rays' lengths, and whether they all end at
    // one point light.
    float onorm_all = -1.0f, tmax_all = -1.0f;
    bool one_end = false;
    if (ENDS) {
      float on = alive ? onorm : 0.0f, tm = alive ? dmax : 0.0f;
      if (!(on >= 0.0f)) on = INFINITY;
      if (!(tm >= 0.0f)) tm = INFINITY;
      onorm_all = wave_max(on);
      tmax_all = wave_max(tm);
      const unsigned long long walking = __ballot(alive);
      const int slot_first = __builtin_amdgcn_readlane(slot, walking != 0 ? __ffsll(walking) - 1 : 0);
      one_end = slot_first < fresh_arg<ShadeP>(offsetof(ShadowArgs, sp)).nl && __ballot(alive && slot != slot_first) == 0;
    }
    Fan ticket_fan = no_fan();
    if (RECTEST && !ENDS && order == nullptr) {
      const unsigned long long walking = __ballot(alive);
      if (walking != 0) {
        const int first = __ffsll(walking) - 1;
        const int slot_t = __builtin_amdgcn_readlane(slot, first);
        if (slot_t < sp.nl && __ballot(alive && slot != slot_t) == 0)
          ticket_fan = make_fan(alive, a, lane_bcast3(light, first), wave_max(alive ? dmax : 0.0f) * 1.00001f);
      }
    }
    // One triangle for one shadow ray: 0 = no effect, 2 = accepted hit of a transparent triangle, 3 = opaque
    // blocker inside (1e-5, dmax); the filter decides what it can prove, tri_test the rest (:700-711).
    // react(verdict, kt) runs, with all lanes of the caller active, only when some lane passed the first slab.
    auto test = [&](const float *R, f3 to, f3 tr, f3 tm, float kc, int cur, float tdmax, auto &&react) {
      const RecA A = rec_a(R);
      PlTest t;
      const bool p1 = pl_stage1(A.n, A.ac, A.q, tr, tm, kc, t);
      if (wave_any(p1)) {
        asm volatile("; some lane passed the first slab");  // keeps the scalar branch from being merged into the per-lane one
        bool s3 = false;
        int verdict = 0;
        float kt = 0.0f;
        if (p1) verdict = shadow_verdict(R, A, t, to, tr, tm, kc, g.lmax, tdmax, kt, s3);
        if (COUNT) { nS2++; nS3 += __ballot(s3) != 0; nEx += __ballot(verdict == 1) != 0; }
        if (verdict != 0) {
          const int id = rec_id(R);
          if (id == cur) {
            verdict = 0;
          } else if (verdict == 1) {
            float d, u, v;
            bool full = false;
            verdict = !exact_test(g, id, to, tr, d, u, v, COUNT ? &full : nullptr) ? 0 : kt > 0.0f ? 2 : (d > 0.00001f && d < tdmax) ? 3 : 0;
            if (COUNT && full) nFull++;
          }
        }
        react(verdict, kt);
        return true;
      }
      return false;
    };
    if (LL) {
      ph.end<0>();
      // The light's lists instead of the walk (lists_lookup, lists_verify), for a ticket whose walks all end at one point
      // light (ticket_fan).  Where they do not apply or do not agree, the ticket takes the walk below, from its untouched state.
      const LightListsP ll = fresh_arg<LightListsP>(offsetof(ShadowArgs, ll));
      const unsigned long long walking = __ballot(alive);
      if (ll.on && ticket_fan.on && walking != 0) {
        nListFell++;
        const int light_i = __builtin_amdgcn_readlane(slot, __ffsll(walking) - 1);
        int hit_id = -1;
        uint32_t diag[3] = {0u, 0u, 0u};
        if (lists_lookup(g, ll, light_i, ticket_fan, alive, a, r, pm, onorm, current, dmax, slice, lane, ll.levels - 1, hit_id, diag) &&
            lists_verify(ll, alive, hit_id, r, [&](Walk &v, int &vi, int &vj, int &vk) {
              v = w;
              (void)voxel_of(g, a, vi, vj, vk);  // (the walk still stands in its first cell, the hit point's voxel)
            })) {
          if (alive && hit_id >= 0) result = -1.0f;
          alive = false;
          nListed++; nListFell--;
        }
        RT_TICKET(tl.visits = diag[0]; tl.batches = diag[1]; tl.tests = diag[2];)  // (diagnostic build: what the ticket looked at)
      }
      if (__ballot(alive) != 0) {  // the ticket walks after all: what the set-up left out
        const GridP g = fresh_arg<GridP>(offsetof(ShadowArgs, g));
        if (alive && !cell_occupied(g, occ_s, occ_in_lds, w.cell)) alive = walk_to_occupied<COUNT>(g, occ_s, occ_in_lds, r, w, nS);
        if (alive) { mybeg = g.cell_start[w.cell]; mycnt = g.cell_start[w.cell + 1] - mybeg; myhbeg = g.hdr_start[w.cell]; }
      }
      ph.end<7>();
    }
    ph.end<0>();
    for (;;) {
      const unsigned long long m = __ballot(alive);
      if (m == 0) break;
      const int leader = least_advanced(m, alive, w.tx, w.ty, w.tz);
      const int cell = __builtin_amdgcn_readlane(w.cell, leader);
      const uint32_t beg = __builtin_amdgcn_readlane(mybeg, leader), cnt = __builtin_amdgcn_readlane(mycnt, leader);
      const uint32_t hbeg = __builtin_amdgcn_readlane(myhbeg, leader);
      const bool in_group = alive && w.cell == cell;
      const bool is_short = alive && mycnt <= RT_SHORT_LIST;
      const int mode = pick_mode(__popcll(__ballot(in_group)), cnt, __popcll(__ballot(is_short)));
      if (COUNT) nVisit++;
      RT_TICKET(tl.visits++;)
#ifdef RT_HIST
      const unsigned long long hist_w0 = nWave, hist_l0 = nLane;
#endif
      bool proc;  // this lane's current cell is fully tested (or the ray blocked) after this step
      if (mode == kCellMode) {
        proc = in_group;
        // lanes outside the cell, and blocked ones, test with kc = -inf: rejected by the first slab
        float kc = -INFINITY;
        bool go = __ballot(in_group) != 0;  // some ray of the group is still unblocked
        Bundle bundle = make_bundle<ENDS, ENDS>(in_group, a, r, pm, odr, onorm, dmax, ENDS ? dmax : INFINITY, central_lane(in_group, lane, order == nullptr, leader),
                                                onorm_all, tmax_all, one_end);
        bundle.fan = ticket_fan;
        if (go)
          stream_cell<RECTEST, ENDS, RT_SHADOW_RAYCULL(RECTEST, ENDS)>(g, hbeg, beg, cnt, slice, lane, bundle, true, ph,
                      [&](f3 c, float delta, f3 hw, float W) {
                        return wave_mask(in_group & (result >= 0.0f) & box_test_ray(c, delta, hw, W, r, pm, odr, dmax, true));
                      },
                      [&](int n, int culled) {
                        if (n >= 0) { if (COUNT) { nBatch += n; nCull += culled; } }
                        RT_TICKET(else if (n == -1) tl.batches++; else tl.tests += culled;)
                      },
                      [&](float k1, float k2) { kc = (in_group & (result >= 0.0f)) ? __builtin_fmaf(k1, onorm, k2) + g.slack : -INFINITY; },
                      [&](const float *R) {
                        if (COUNT) { nWave++; if (kc != -INFINITY) nLane++; }
                        const RecA A = rec_a(R);
                        PlTest t;
                        return pl_stage1(A.n, A.ac, A.q, r, pm, kc, t);
                      },
                      [&](const float *R, uint32_t, bool first) {
                        if (COUNT && first) { nWave++; if (kc != -INFINITY) nLane++; }
                        return test(R, a, r, pm, kc, current, dmax, [&](int verdict, float kt) {
                          if (verdict == 2) {
                            ts *= kt;  // every listing of a transparent triangle attenuates (:703-706)
                          } else if (verdict == 3) {
#ifdef RT_HIST
                            result = -(float)(rec_id(R) + 2); kc = -INFINITY;  // (diagnostic build: which triangle blocked the ray)
#else
                            result = -1.0f; kc = -INFINITY;
#endif
                          }
                          go = __ballot(kc != -INFINITY) != 0;
                        });
                      },
                      [&] { return go; });
        alive = alive & (result >= 0.0f);
      } else if (mode == kLaneMode) {
        proc = is_short;
        uint32_t pn = proc ? mycnt : 0u;
        for (uint32_t i = 0; __ballot(i < pn) != 0; ++i) {
          if (COUNT) { nWave++; if (i < pn) nLane++; }
          if (i < pn) {
            const float *R = reinterpret_cast<const float *>(recs + mybeg + i);
            test(R, a, r, pm, __builtin_fmaf(R[3], onorm, R[11]) + g.slack, current, dmax, [&](int verdict, float kt) {
              if (verdict == 2) ts *= kt;
              else if (verdict == 3) { result = -1.0f; alive = false; pn = 0u; }
            });
          }
        }
      } else {
        // ray mode: the leader's shadow ray against 64 triangles per step
        proc = lane == leader;
        const f3 la = lane_bcast3(a, leader), lr = lane_bcast3(r, leader), lm = lane_bcast3(pm, leader);
        const float lon = lane_bcast(onorm, leader);
        const int lcur = __builtin_amdgcn_readlane(current, leader);
        const float ldmax = lane_bcast(dmax, leader);
        float cts = lane_bcast(ts, leader);
        bool blocked = false;
        for (uint32_t t0 = beg; t0 < beg + cnt && !blocked; t0 += 64) {
          const uint32_t t = t0 + lane;
          if (COUNT) { nWave++; if (t < beg + cnt) nLane++; }
          int verdict = 0;
          float kt = 0.0f;
          if (t < beg + cnt) {
            const float *R = reinterpret_cast<const float *>(recs + t);
            test(R, la, lr, lm, __builtin_fmaf(R[3], lon, R[11]) + g.slack, lcur, ldmax, [&](int vd, float k) { verdict = vd; kt = k; });
          }
          // the transparency product is taken in list order, like the reference's scan
          for (unsigned long long tm = __ballot(verdict == 2); tm != 0; tm &= tm - 1) cts *= lane_bcast(kt, __ffsll(tm) - 1);
          blocked = __ballot(verdict == 3) != 0;
        }
        if (proc) {
          if (blocked) { result = -1.0f; alive = false; } else ts = cts;
        }
      }
#ifdef RT_HIST
      if (COUNT) {
        unsigned long long dl = nLane - hist_l0;
        for (int off = 32; off > 0; off >>= 1) dl += __shfl_down(dl, off);
        const unsigned long long np = (unsigned long long)__popcll(__ballot(proc));
        if (lane == 0) { atomicAdd(&g_mode[1][mode][0], 1ull); atomicAdd(&g_mode[1][mode][1], nWave - hist_w0); atomicAdd(&g_mode[1][mode][2], dl); atomicAdd(&g_mode[1][mode][3], np); }
      }
#endif
      ph.end<7>();
      if (proc && alive) {
        const GridP g = fresh_arg<GridP>(offsetof(ShadowArgs, g));
        const uint32_t *__restrict__ cstart = g.cell_start;
        alive = walk_to_occupied<COUNT>(g, occ_s, occ_in_lds, r, w, nS);
        if (alive) {
          mybeg = cstart[w.cell]; mycnt = cstart[w.cell + 1] - mybeg; myhbeg = g.hdr_start[w.cell];
          if (COUNT) nT += mycnt;
        }
      }
      ph.end<6>();
    }
    float *__restrict__ ts_out = fresh_arg<float *>(offsetof(ShadowArgs, ts_out));
    const long long ts_stride = fresh_arg<long long>(offsetof(ShadowArgs, ts_stride));
    if (have_result) {
      // slot-major: the 64 lanes of a ticket are consecutive hits of one slot, so the store is one 256-byte line
#ifdef RT_HIST
      ts_out[(long long)slot * ts_stride + where] = result < 0.0f ? result : ts;  // (diagnostic build: the blocker stays visible)
#else
      ts_out[(long long)slot * ts_stride + where] = result < 0.0f ? -1.0f : ts;
#endif
    }
    RT_TICKET(ph.end<0>(); if (level == 0) tl.end(1, base, lane, ph); busy += __builtin_amdgcn_s_memtime() - tl.t0;)
  }
  RT_TICKET(if (level == 0) TicketLog::life(1, lane, busy, born);)
  ph.end<0>();
  ph.flush(8, lane);
  if (LL && lane == 0 && (nListed | nListFell) != 0) {
    Ctl *ctl = fresh_arg<Ctl *>(offsetof(ShadowArgs, ctl));
    atomicAdd(&ctl->stat[kShadListTickets], (unsigned long long)nListed);
    atomicAdd(&ctl->stat[kShadListFallbacks], (unsigned long long)nListFell);
  }
  if (COUNT) {
    for (int off = 32; off > 0; off >>= 1) {
      nS += __shfl_down(nS, off); nT += __shfl_down(nT, off); nQ += __shfl_down(nQ, off); nW += __shfl_down(nW, off);
      nLane += __shfl_down(nLane, off); nFull += __shfl_down(nFull, off);
    }
    if (lane == 0) {
      atomicAdd(&ctl->stat[kShadFull], nFull);
      atomicAdd(&ctl->stat[kShadLaneTests], nLane); atomicAdd(&ctl->stat[kShadWaveTests], nWave * 64ull);
      atomicAdd(&ctl->stat[kShadS], nS); atomicAdd(&ctl->stat[kShadT], nT);
      atomicAdd(&ctl->stat[kShadowQ], nQ); atomicAdd(&ctl->stat[kShadowWalks], nW);
      atomicAdd(&ctl->stat[kShadStage2], nS2); atomicAdd(&ctl->stat[kShadStage3], nS3); atomicAdd(&ctl->stat[kShadExact], nEx);
      atomicAdd(&ctl->stat[kShadVisits], nVisit); atomicAdd(&ctl->stat[kShadBatches], nBatch); atomicAdd(&ctl->stat[kShadCulled], nCull);
    }
  }
}

RT_D void mm_fold(float *mn, float *mx, const float *omn, const float *omx) {
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (omx[k] > mx[k]) mx[k] = omx[k];
    if (omn[k] < mn[k]) mn[k] = omn[k];
  }
}

// One light's contribution (raytrace.c:217-235 / 271-289).
// SPECULAR = false: for scenes none of whose surfaces has ks > 0 (the pow of the specular term and its registers are compiled out)
template <bool SPECULAR>
RT_D f3 light_term(const SurfaceD &sf, f3 norm, f3 r, f3 rnew, f3 onew, f3 lpos, float lflux, f3 lcol, f3 nc,
                   float ts, float distmod) {
  float ndl = dot3(norm, rnew);
  float df = sf.kd * ndl;
  if (df < 0.0f && sf.kt > 0.0f) df = -df;
  float rf = 0.0f;
  if (SPECULAR && sf.ks > 0.0f) {
    f3 t = reflect3(norm, rnew, ndl);
    rf = (float)((double)sf.ks * pow((double)dot3(r, t), (double)sf.g));
    if (rf < 0.0f && sf.kt > 0.0f) rf = -rf;
  }
  float scale = ts * lflux * (df + rf) / (dist3(onew, lpos) + distmod);
  return mk3((lcol.x + nc.x) * scale, (lcol.y + nc.y) * scale, (lcol.z + nc.z) * scale);
}

// ------------------------------------------------------------------------------------------------
// Colour of every shaded hit of one level, in the reference's summation order (raytrace.c:188-296):
// ambient, + kr * reflected child, + kt * refracted child, + point lights in order, + planar lights
// (16 samples each, accumulated exactly as written, quirks included).  Levels run deepest first so a
// node finds its children's colours ready.  Level 0 also writes the tile-major framebuffer.
template <bool SPECULAR>
__global__ void __launch_bounds__(256, SPECULAR ? RT_WAVES_ACC : RT_WAVES_ACC_PLAIN) k_accumulate(ShadeP sp, const Ray *__restrict__ rays, const Node *__restrict__ nodes,
                                                    const float *__restrict__ ts_in, long long ts_stride, const int32_t *__restrict__ rand_rank,
                                                    float *__restrict__ colors, Ctl *__restrict__ ctl, int level,
                                                    int node_base, float *__restrict__ fb_rgb, int32_t *__restrict__ fb_ids,
                                                    long long fb_base, float *__restrict__ partial, PrimaryP prim, const uint32_t *__restrict__ order) {
  const int n = queue_len(ctl, level);
  const bool fold = level == 0 && partial != nullptr;  // level 0 also folds the colour range of the block's pixels
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {FLT_MIN, FLT_MIN, FLT_MIN};
  // block-stride loop (the launches of the levels below 0 are sized for the queue's capacity, cut at a few thousand blocks)
  for (int idx0 = blockIdx.x * blockDim.x; idx0 < n; idx0 += gridDim.x * blockDim.x) {
  const int at = idx0 + threadIdx.x;  // place in the level's order: where the shadow walks left this node's results
  const bool valid = at < n;
  if (!valid) continue;
  const int idx = order ? (int)order[node_base + at] : at;
  const int gidx = node_base + idx;
  const Node nd = nodes[gidx];
  f3 res = mk3(0.0f, 0.0f, 0.0f);
  if (nd.tri >= 0) {
    const SurfaceD sf = sp.surf[sp.tri_sid[nd.tri]];
    const f3 nc = ld3(nd.c), norm = ld3(nd.n), onew = ld3(nd.p);
    f3 r;
    if (prim.on) primary_dir(prim, idx, r); else r = ld3(rays[idx].r);
    if (sf.ka > 0.0f) res = mul3(nc, sf.ka * sp.total_flux);
    if (sf.kr > 0.0f) {
      f3 rc = nd.refl >= 0 ? ld3(colors + 4ll * nd.refl) : mk3(0.0f, 0.0f, 0.0f);
      res = add3(res, mul3(rc, sf.kr));
    }
    if (sf.kt > 0.0f) {
      f3 rc = nd.refr >= 0 ? ld3(colors + 4ll * nd.refr) : mk3(0.0f, 0.0f, 0.0f);
      res = add3(res, mul3(rc, sf.kt));
    }
    const float *ts = ts_in + node_base + at;  // slot s of this node: ts[s * ts_stride]
    f3 tmp = mk3(0.0f, 0.0f, 0.0f);
    // the shadow results of the first lights are asked for together: one memory round trip instead of one per light
    constexpr int kAhead = 8;
    float ts_first[kAhead];
#pragma unroll
    for (int c = 0; c < kAhead; c++) ts_first[c] = c < sp.nl ? ts[c * ts_stride] : -1.0f;
    for (int c = 0; c < sp.nl; c++) {
      float t;
      if (c < kAhead) {
        t = ts_first[0];
#pragma unroll
        for (int q = 1; q < kAhead; q++) t = c == q ? ts_first[q] : t;
      } else {
        t = ts[c * ts_stride];
      }
      if (t >= 0.0f) {
        const LightD L = sp.lights[c];
        const f3 lp = ld3(L.p);
        tmp = light_term<SPECULAR>(sf, norm, r, ray3(onew, lp), onew, lp, L.flux, ld3(L.color), nc, t, sp.distmod);
        res = add3(res, tmp);
      }
    }
    for (int c = 0; c < sp.npl; c++) {
      f3 sum = mk3(0.0f, 0.0f, 0.0f);
      for (int d = 0; d < kPlanarSamples; d++) {
        const int slot = sp.nl + c * kPlanarSamples + d;
        const float t = ts[slot * ts_stride];
        if (t >= 0.0f) {
          f3 lp, lcol; float lflux;
          slot_light(sp, slot, rand_rank[gidx], lp, lflux, lcol);
          sum = light_term<SPECULAR>(sf, norm, r, ray3(onew, lp), onew, lp, lflux, lcol, nc, t, sp.distmod);
        }
        sum = mul3(sum, 1.0f / kPlanarSamples);
        tmp = add3(tmp, sum);
        res = add3(res, tmp);
      }
    }
  }
  if (valid) {
    if (level != 0) {  // only a parent reads a node's colour, and level 0 has none
      colors[4ll * gidx] = res.x; colors[4ll * gidx + 1] = res.y; colors[4ll * gidx + 2] = res.z;
    } else {
      long long p = fb_base + idx;
      fb_rgb[3 * p] = res.x; fb_rgb[3 * p + 1] = res.y; fb_rgb[3 * p + 2] = res.z;
      fb_ids[p] = nd.tri >= 0 ? nd.tri : -1;
    }
  }
  if (fold && nd.tri >= -1) {  // (a pixel whose ray never entered the grid takes no part, raytrace.c:381-382)
    const float c[3] = {res.x, res.y, res.z};
    mm_fold(mn, mx, c, c);
  }
  }
  if (fold) {
    // per-channel min/max over the traced pixels of this block (raytrace.c:396-399): partial[blockIdx.x][6]
    __shared__ float sm[4][6];
    for (int off = 32; off > 0; off >>= 1) {
      float omn[3], omx[3];
#pragma unroll
      for (int k = 0; k < 3; k++) { omn[k] = __shfl_down(mn[k], off); omx[k] = __shfl_down(mx[k], off); }
      mm_fold(mn, mx, omn, omx);
    }
    const int wid = threadIdx.x >> 6;
    if (lane_id() == 0) { for (int k = 0; k < 3; k++) { sm[wid][k] = mn[k]; sm[wid][3 + k] = mx[k]; } }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int wv = 1; wv < 4; wv++) mm_fold(mn, mx, &sm[wv][0], &sm[wv][3]);
      for (int k = 0; k < 3; k++) { partial[blockIdx.x * 6 + k] = mn[k]; partial[blockIdx.x * 6 + 3 + k] = mx[k]; }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Per-channel min/max over the traced pixels of a chunk (raytrace.c:396-399; NaN never wins because
// the comparisons are the reference's `>` and `<`).  Pixels whose ray never entered the grid do not
// take part.  partial: [blocks][6]; k_minmax_final folds partials into the running d_minmax.
// End of a chunk: folds the blocks' colour ranges into the running range and, if ctl is given, the chunk's statistics
// and queue lengths into the frame totals (k_fold_stats).
__global__ void __launch_bounds__(1024) k_minmax_final(const float *__restrict__ partial, int nblocks, float *__restrict__ minmax,
                                                       const Ctl *__restrict__ ctl = nullptr, unsigned long long *__restrict__ totals = nullptr) {
  __shared__ float sm[16][6];
  if (ctl != nullptr) {
    const int t = threadIdx.x;
    if (t < kStatCount) totals[t] += ctl->stat[t];
    else if (t < kStatCount + kMaxLevels) totals[t] += (unsigned long long)queue_len(ctl, t - kStatCount);
  }
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {FLT_MIN, FLT_MIN, FLT_MIN};
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) mm_fold(mn, mx, partial + 6 * b, partial + 6 * b + 3);
  for (int off = 32; off > 0; off >>= 1) {
    float omn[3], omx[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { omn[k] = __shfl_down(mn[k], off); omx[k] = __shfl_down(mx[k], off); }
    mm_fold(mn, mx, omn, omx);
  }
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < 3; k++) { sm[threadIdx.x >> 6][k] = mn[k]; sm[threadIdx.x >> 6][3 + k] = mx[k]; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float fmn[3] = {minmax[0], minmax[1], minmax[2]}, fmx[3] = {minmax[3], minmax[4], minmax[5]};  // the running range of earlier chunks
    for (int wv = 0; wv < (int)(blockDim.x >> 6); wv++) mm_fold(fmn, fmx, sm[wv], sm[wv] + 3);
    for (int k = 0; k < 3; k++) { minmax[k] = fmn[k]; minmax[3 + k] = fmx[k]; }
  }
}
// Start of a frame: the running colour range and the frame's statistics.
__global__ void k_frame_init(float *minmax, unsigned long long *totals) {
  if (threadIdx.x < 3) minmax[threadIdx.x] = FLT_MAX;       // raytrace.c:318
  else if (threadIdx.x < 6) minmax[threadIdx.x] = FLT_MIN;  // raytrace.c:319 (smallest positive normal, as written)
  if (threadIdx.x < kTotals) totals[threadIdx.x] = 0ull;
}
__global__ void k_minmax_init(float *minmax) {
  if (threadIdx.x < 3) minmax[threadIdx.x] = FLT_MAX;       // raytrace.c:318
  else if (threadIdx.x < 6) minmax[threadIdx.x] = FLT_MIN;  // raytrace.c:319 (smallest positive normal, as written)
}

// Adds a chunk's statistics and queue lengths to the frame totals: totals[0..kStatCount) are the Stat
// counters, totals[kStatCount + L] the number of rays of recursion level L.
__global__ void k_fold_stats(const Ctl *__restrict__ ctl, unsigned long long *__restrict__ totals) {
  int t = threadIdx.x;
  if (t < kStatCount) totals[t] += ctl->stat[t];
  else if (t < kStatCount + kMaxLevels) totals[t] += (unsigned long long)queue_len(ctl, t - kStatCount);
}

// ------------------------------------------------------------------------------------------------
// Tone map (raytrace.c:433-466): v = pow((c - min) * delta, gamma) * 255 in double, rounded to float,
// truncated to an integer, low byte kept; packed R<<24 | G<<16 | B<<8 (bitmap.h:70).
// When 2 gamma is a small integer (the default gamma is 2.5) the power is a few products and a square root: in
// binary32 first, then, for the values within 2e-6 of a rounding boundary, in double, within 1e-14 of the true value:
// if every value within 1e-12 of it gives the same byte, so does any correctly working pow, and the call is
// skipped; the few values next to a rounding boundary take the pow path like all others.
// stride: floats from one pixel's colour to the next (3 for a packed colour buffer; a caller's pixel records may be wider).
__global__ void __launch_bounds__(256) k_tonemap(const float *__restrict__ rgb_, long long stride, long long n, const float *__restrict__ minmax,
                                                 float gamma, uint32_t *__restrict__ rgba) {
  long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const float *__restrict__ rgb = rgb_ + (stride - 3) * p;  // (so that rgb[3 p + k] below is channel k of pixel p)
  const float twice = gamma + gamma;
  const int halves = (twice >= 1.0f && twice <= 8.0f && twice == (float)(int)twice) ? (int)twice : 0;  // gamma = halves / 2
  uint32_t ch[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    float delta = 1.0f / (minmax[3 + k] - minmax[k]);
    const float x = (rgb[3 * p + k] - minmax[k]) * delta;
    bool done = false;
    if (halves != 0 && x >= 0.0f && x <= 4.0f) {
      // first in binary32: the hardware's square root (1 ulp), up to four products and the factor 255 (half an ulp each)
      // are within 4.2e-7 of the true value, the reference's rounded value within 0.6e-7 of that, the window's ends are
      // rounded once more: if both ends of a window of +-8e-7 give the same byte, that is it
      float pf = (halves & 1) ? __builtin_amdgcn_sqrtf(x) : 1.0f;
      for (int q = 0; q < (halves >> 1); q++) pf *= x;
      const float t = pf * 255.0f;
      const int flo = (int)(t * 0.9999992f), fhi = (int)(t * 1.0000008f);  // (0 <= t < 2^17: the truncation of f2u_x86)
      if (flo == fhi) { ch[k] = (uint32_t)flo & 0xffu; done = true; }
    }
    if (!done && halves != 0 && x >= 0.0f && x <= 4.0f) {
      const double xd = (double)x;
      // (the square root from the hardware's reciprocal estimate and two Newton steps: within 1e-14, a hundredth of the window)
      double pw = 1.0;
      if (halves & 1) {
        double y = __builtin_amdgcn_rsq(xd);
#pragma unroll
        for (int it = 0; it < 2; it++) { const double g2 = xd * y, h2 = 0.5 * y; y = __builtin_fma(y, __builtin_fma(-g2, h2, 0.5), y); }
        pw = xd > 0.0 ? xd * y : 0.0;
      }
      for (int q = 0; q < (halves >> 1); q++) pw *= xd;
      const uint32_t lo = f2u_x86((float)(pw * (1.0 - 1e-12) * 255.0) / 1.0f), hi = f2u_x86((float)(pw * (1.0 + 1e-12) * 255.0) / 1.0f);
      if (lo == hi) { ch[k] = lo & 0xffu; done = true; }
    }
    if (!done) {
      float acc = (float)(pow((double)x, (double)gamma) * 255.0);
      ch[k] = f2u_x86(acc / 1.0f) & 0xffu;
    }
  }
  rgba[p] = (ch[0] << 24) | (ch[1] << 16) | (ch[2] << 8);
}

// The tone map with a list of gammas (raytrace.c:439-466): per channel `r += pow(x, gamma_k) * 255.0f` for every gamma of
// the list (product and sum in double, rounded to float at every +=), then r / (float)ng, converted like a single gamma's
// value.  An empty list gives 0.0f / 0.0f = NaN, which the conversion turns into 0.
__global__ void __launch_bounds__(256) k_tonemap_multi(const float *__restrict__ rgb, long long stride, long long n, const float *__restrict__ minmax,
                                                       const float *__restrict__ gammas, int ng, uint32_t *__restrict__ rgba) {
  long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  uint32_t ch[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float delta = 1.0f / (minmax[3 + k] - minmax[k]);
    const float x = (rgb[stride * p + k] - minmax[k]) * delta;
    float acc = 0.0f;
    for (int q = 0; q < ng; q++) acc = (float)((double)acc + pow((double)x, (double)gammas[q]) * 255.0);
    ch[k] = f2u_x86(acc / (float)ng) & 0xffu;
  }
  rgba[p] = (ch[0] << 24) | (ch[1] << 16) | (ch[2] << 8);
}

// BMP pixel rows on the device (bitmap.c:453-469): bottom-up, B G R [0] per pixel, every row padded with zeros to a
// multiple of four bytes.  One thread per 32-bit word of the output, so the stores coalesce; words = row_bytes / 4.
__global__ void __launch_bounds__(256) k_bmp_rows(const uint32_t *__restrict__ image, int w, int h, int bytes_pp, int words,
                                                  uint32_t *__restrict__ rows) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (long long)words * h) return;
  const int y_out = (int)(q / words), wd = (int)(q - (long long)y_out * words);
  const uint32_t *src = image + (long long)(h - 1 - y_out) * w;
  uint32_t out = 0;
#pragma unroll
  for (int b = 0; b < 4; b++) {
    const int byte = 4 * wd + b, x = byte / bytes_pp, ch = byte - x * bytes_pp;
    if (x < w && ch < 3) out |= ((src[x] >> (8 + 8 * ch)) & 0xffu) << (8 * b);  // pixel = R<<24 | G<<16 | B<<8 (bitmap.h:70)
  }
  rows[q] = out;
}

// The frame of one rank in the caller's own pixel records (the reference's RT_VisualizedScenePixel, raytrace.h:23-27: a
// colour of four floats and a pointer to the primary-hit triangle), row-major, from the tile-major buffers of the pixel
// loop.  The pointer is an address of the CALLER's address space, tri_base + id x tri_stride, or null without a hit: it is
// worked out here as a number and never followed.  The colour's fourth lane, and every byte of a record that neither
// member covers, is 0.  tri_offset < 0: records without a pointer.
struct PixelLayoutD { long long stride, color_offset, tri_offset; unsigned long long tri_base; long long tri_stride; };
__global__ void __launch_bounds__(256) k_pixelmap(const float *__restrict__ rgb_t, const int32_t *__restrict__ ids_t, int w, int h, int tiles_x,
                                                  PixelLayoutD lay, unsigned char *__restrict__ map) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (long long)w * h) return;
  const int x = (int)(p % w), y = (int)(p / w);
  const long long s = ((long long)(y >> 3) * tiles_x + (x >> 3)) * kTilePixels + (y & 7) * kTile + (x & 7);
  // the record word by word: colour, pointer halves, and 0 for what no member covers
  uint32_t *rec = reinterpret_cast<uint32_t *>(map + p * lay.stride);
  const int32_t id = lay.tri_offset >= 0 ? ids_t[s] : -1;
  const unsigned long long ptr = id >= 0 ? lay.tri_base + (unsigned long long)id * (unsigned long long)lay.tri_stride : 0ull;
  const int cw = (int)(lay.color_offset >> 2), tw = lay.tri_offset >= 0 ? (int)(lay.tri_offset >> 2) : -2, nw = (int)(lay.stride >> 2);
  for (int q = 0; q < nw; q++) {
    uint32_t v = 0u;
    if (q >= cw && q < cw + 3) v = __float_as_uint(rgb_t[3 * s + (q - cw)]);
    else if (q == tw) v = (uint32_t)ptr;
    else if (q == tw + 1) v = (uint32_t)(ptr >> 32);
    rec[q] = v;
  }
}

// Tile-major blocks of all ranks (rank-major, then local tile, then 64 slots) -> row-major image.
__global__ void __launch_bounds__(256) k_untile(const uint32_t *__restrict__ src, int world, int w, int h, int tiles_x,
                                                long long ntiles, int words, uint32_t *__restrict__ dst) {
  long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (long long)w * h) return;
  int x = (int)(p % w), y = (int)(p / w);
  long long gt = (long long)(y >> 3) * tiles_x + (x >> 3);
  int rank = (int)(gt % world);
  long long lt = gt / world;
  // tiles owned by ranks below `rank`: ceil((ntiles - q) / world) for q < rank
  long long before = 0;
  for (int q = 0; q < rank; q++) before += (ntiles - q + world - 1) / world;
  long long s = ((before + lt) * kTilePixels + (y & 7) * kTile + (x & 7)) * words;
  for (int q = 0; q < words; q++) dst[p * words + q] = src[s + q];
}

}  // namespace rt
