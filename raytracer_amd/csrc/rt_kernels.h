// HIP kernels of the trace path (gfx950, wave64).  Included once by rtgpu.hip.
//
// Frame set-up          k_preprocess, k_tri_boxes, k_scan_cells, k_grid_fill
// Pixel loop (a chunk)  k_primary -> [k_nearest -> k_hit] per recursion level -> k_shadow per level
//                       -> k_accumulate per level, deepest first -> k_minmax_*
// After the exchange    k_tonemap, k_untile
#pragma once
#include "rt_math.h"
#include "rt_texture.h"

#ifndef RT_SOURCE
#define RT_SOURCE 1  // how a cell's triangle list reaches the lanes: 0 = scalar loads, 1 = LDS-staged batches
#endif
#ifndef RT_LDS_PIPELINE
#define RT_LDS_PIPELINE 0
#endif
#ifndef RT_UNROLL
#define RT_UNROLL 2
#endif
#ifndef RT_WAVES
#define RT_WAVES 8  // waves per SIMD the two walk kernels are compiled for (caps their VGPR allocation)
#endif

namespace rt {

RT_D int lane_id() { return threadIdx.x & 63; }
RT_D unsigned long long lanes_below(int lane) { return (1ull << lane) - 1ull; }

// Streams the packed triangle list [beg, end) of one cell through every lane of the wavefront.
// f(a, ab, ac, record, index) is called once per record, in list order, with wave-uniform arguments;
// more() is a wave-uniform predicate evaluated after each record (false = stop early).
//
// RT_SOURCE 1: the wavefront copies the list to its private LDS slice in batches of kBatch records with
// coalesced 16-byte vector loads (64 lanes x 16 B = 21 records per instruction), then every lane reads the
// same record back (LDS broadcast, two 16-byte reads + one 4-byte read per test; id and kt only on a hit).
// LDS operations of one wavefront execute in issue order, so no barrier is needed between the copy and
// the reads, nor before the next batch overwrites the slice.
// RT_SOURCE 0: wave-uniform global loads (scalar cache path).
constexpr int kBatch = 64;
constexpr int kRecFloats = 12;
using v4f = float __attribute__((ext_vector_type(4)));

template <typename F, typename G>
RT_D void stream_cell(const TriRec *__restrict__ recs, uint32_t beg, uint32_t end, float *slice, int lane, F &&f, G &&more) {
#if RT_SOURCE == 1
  for (uint32_t c0 = beg; c0 < end; c0 += kBatch) {
    const int n = (int)(end - c0 < (uint32_t)kBatch ? end - c0 : (uint32_t)kBatch);
    const int nvec = n * 3;
    const float4 *src = reinterpret_cast<const float4 *>(recs + c0);
    float4 *dst = reinterpret_cast<float4 *>(slice);
    float4 v0, v1, v2;
    if (lane < nvec) v0 = src[lane];
    if (lane + 64 < nvec) v1 = src[lane + 64];
    if (lane + 128 < nvec) v2 = src[lane + 128];
    if (lane < nvec) dst[lane] = v0;
    if (lane + 64 < nvec) dst[lane + 64] = v1;
    if (lane + 128 < nvec) dst[lane + 128] = v2;
    __builtin_amdgcn_wave_barrier();
    // all nine floats of a record feed the branch-free part of the test, so the three reads
    // (16 + 16 + 4 bytes, same address in every lane = LDS broadcast) sit together at the top
#if RT_LDS_PIPELINE
    v4f x = *reinterpret_cast<const v4f *>(slice);
    v4f y = *reinterpret_cast<const v4f *>(slice + 4);
    float z = slice[8];
    for (int j = 0; j < n; ++j) {
      const float *R = slice + j * kRecFloats;
      const float *N = slice + (j + 1 < kBatch ? j + 1 : j) * kRecFloats;  // next record, read one test ahead
      const v4f nx = *reinterpret_cast<const v4f *>(N);
      const v4f ny = *reinterpret_cast<const v4f *>(N + 4);
      const float nz = N[8];
      f(mk3(x.x, x.y, x.z), mk3(x.w, y.x, y.y), mk3(y.z, y.w, z), R, c0 + (uint32_t)j);
      if (!more()) return;
      x = nx; y = ny; z = nz;
    }
#else
#pragma unroll RT_UNROLL
    for (int j = 0; j < n; ++j) {
      const float *R = slice + j * kRecFloats;
      const v4f x = *reinterpret_cast<const v4f *>(R);
      const v4f y = *reinterpret_cast<const v4f *>(R + 4);
      const float z = R[8];
      f(mk3(x.x, x.y, x.z), mk3(x.w, y.x, y.y), mk3(y.z, y.w, z), R, c0 + (uint32_t)j);
      if (!more()) return;
    }
#endif
    __builtin_amdgcn_wave_barrier();
  }
#else
  for (uint32_t t = beg; t < end; ++t) {
    const TriRec *T = recs + t;
    f(ld3(T->a), ld3(T->ab), ld3(T->ac), reinterpret_cast<const float *>(T), t);
    if (!more()) return;
  }
#endif
}
// id and kt of the record handed to f (read only when a test succeeds)
RT_D int rec_id(const float *R) { return __float_as_int(R[9]); }
RT_D float rec_kt(const float *R) { return R[10]; }

// One lane's own record fetch (per-lane and one-ray-per-wave modes): 16 + 16 + 4 bytes from L1/L2.
RT_D const float *load_rec(const TriRec *__restrict__ recs, uint32_t idx, f3 &a, f3 &ab, f3 &ac) {
  const float *R = reinterpret_cast<const float *>(recs + idx);
  const v4f x = *reinterpret_cast<const v4f *>(R);
  const v4f y = *reinterpret_cast<const v4f *>(R + 4);
  const float z = R[8];
  a = mk3(x.x, x.y, x.z); ab = mk3(x.w, y.x, y.y); ac = mk3(y.z, y.w, z);
  return R;
}
RT_D float lane_bcast(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
RT_D f3 lane_bcast3(f3 v, int l) { return mk3(lane_bcast(v.x, l), lane_bcast(v.y, l), lane_bcast(v.z, l)); }

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
#define RT_SHORT_LIST 32      // lists up to this length are walked in lane mode
#endif
#ifndef RT_LANE_MIN
#define RT_LANE_MIN 8         // lane mode needs at least this many rays with short lists
#endif
enum WalkMode { kCellMode = 0, kLaneMode = 1, kRayMode = 2 };
RT_D int pick_mode(int group, uint32_t leader_cnt, int short_lanes) {
  if (group >= RT_CELL_MIN_GROUP) return kCellMode;
  if (short_lanes >= RT_LANE_MIN) return kLaneMode;
  // the leader's group alone: cell mode costs cnt steps, ray mode group * (ceil(cnt/64) + 1 for the reduction)
  return leader_cnt <= (uint32_t)group * ((leader_cnt + 63u) / 64u + 1u) ? kCellMode : kRayMode;
}

// ------------------------------------------------------------------------------------------------
// Per-triangle preprocess for one camera (preprocess.c:17-32): ij, ik, unit normal flipped towards
// the observer, plane offset d.  geom: 9 floats (i, ij, ik); aux: 4 floats (n, d).
__global__ void k_preprocess(int nt, const float *__restrict__ verts, f3 ob, float *__restrict__ geom,
                             float *__restrict__ aux) {
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
}

// ------------------------------------------------------------------------------------------------
// Grid build, pass 1 (voxelize.c:252-284): voxel index box of every triangle and the number of
// references each cell will receive.  box: lo[3], hi[3] per triangle.
__global__ void k_tri_boxes(int nt, const float *__restrict__ verts, GridP g, int *__restrict__ box,
                            uint32_t *__restrict__ counts) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nt) return;
  int lo[3], hi[3];
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
  for (int i = lo[0]; i <= hi[0]; i++)
    for (int j = lo[1]; j <= hi[1]; j++)
      for (int k = lo[2]; k <= hi[2]; k++) atomicAdd(&counts[cell_index(g, i, j, k)], 1u);
}

// Exclusive scan of the per-cell counts into cell_start[0..ncells] (one 1024-thread block).
__global__ void __launch_bounds__(1024) k_scan_cells(int n, const uint32_t *__restrict__ counts,
                                                     uint32_t *__restrict__ start) {
  __shared__ uint32_t wave_sum[16];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    int idx = base + tid;
    uint32_t v = idx < n ? counts[idx] : 0u, x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t y = __shfl_up(x, off);
      if (lane >= off) x += y;
    }
    if (lane == 63) wave_sum[wid] = x;
    __syncthreads();
    uint32_t prefix = carry_s;
    for (int wv = 0; wv < wid; wv++) prefix += wave_sum[wv];
    if (idx < n) start[idx] = prefix + x - v;
    __syncthreads();
    if (tid == 1023) carry_s = prefix + x;
    __syncthreads();
  }
  if (tid == 0) start[n] = carry_s;
}

// Grid build, pass 2: one wavefront per cell walks the triangles in index order, 64 at a time, and
// appends those whose voxel box contains the cell; ballot + prefix count keeps the list in ascending
// triangle order, which is the reference's insertion order and therefore its tie-break.
__global__ void __launch_bounds__(256) k_grid_fill(int nt, GridP g, const int *__restrict__ box,
                                                   const float *__restrict__ geom, const int32_t *__restrict__ sid,
                                                   const SurfaceD *__restrict__ surf, TriRec *__restrict__ recs,
                                                   int32_t *__restrict__ refs) {
  int cell = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (cell >= g.ncells) return;
  uint32_t beg = g.cell_start[cell], end = g.cell_start[cell + 1];
  if (beg == end) return;
  const int lane = lane_id();
  int ck = cell % g.nv[2], cj = (cell / g.nv[2]) % g.nv[1], ci = cell / (g.nv[2] * g.nv[1]);
  uint32_t pos = beg;
  for (int t0 = 0; t0 < nt && pos < end; t0 += 64) {
    int t = t0 + lane;
    bool in = false;
    if (t < nt) {
      const int *b = box + 6 * t;
      in = ci >= b[0] && ci <= b[3] && cj >= b[1] && cj <= b[4] && ck >= b[2] && ck <= b[5];
    }
    unsigned long long m = __ballot(in);
    if (in) {
      uint32_t p = pos + __popcll(m & lanes_below(lane));
      TriRec r;
#pragma unroll
      for (int q = 0; q < 3; q++) { r.a[q] = geom[9 * t + q]; r.ab[q] = geom[9 * t + 3 + q]; r.ac[q] = geom[9 * t + 6 + q]; }
      r.id = t;
      r.kt = surf[sid[t]].kt;
      r.pad = 0;
      recs[p] = r;
      refs[p] = t;
    }
    pos += __popcll(m);
  }
}

// ------------------------------------------------------------------------------------------------
// Primary rays of a chunk of tiles (vectormath.h:108-118, voxelize.c:417-471).  Ray index = local
// tile * 64 + (y&7)*8 + (x&7), so one wavefront of the nearest-hit kernel is one 8x8 pixel tile.
__global__ void __launch_bounds__(256) k_primary(CameraD cam, GridP g, int tile0, int ntiles, int rank, int world,
                                                 int tiles_x, Ray *__restrict__ rays, Ctl *__restrict__ ctl) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx == 0) ctl->count[0] = ntiles * kTilePixels;
  if (idx >= ntiles * kTilePixels) return;
  int lt = tile0 + (idx >> 6), slot = idx & 63;
  long long gt = (long long)lt * world + rank;
  int x = (int)(gt % tiles_x) * kTile + (slot & 7), y = (int)(gt / tiles_x) * kTile + (slot >> 3);
  Ray ry;
  ry.o[0] = cam.ob[0]; ry.o[1] = cam.ob[1]; ry.o[2] = cam.ob[2];
  ry.current = -1; ry.pad[0] = ry.pad[1] = 0;
  ry.r[0] = ry.r[1] = ry.r[2] = 0.0f;
  ry.i = -2; ry.j = ry.k = 0;
  if (x < cam.sw && y < cam.sh) {
    float w_inv = 1.0f / cam.sw, h_inv = 1.0f / cam.sh;
    float xc = x * w_inv, yc = y * h_inv;
    f3 ul = ld3(cam.ul), ur = ld3(cam.ur), bl = ld3(cam.bl), ob = ld3(cam.ob);
    f3 dir = mk3(xc * (ur.x - ul.x) + yc * (bl.x - ul.x) + ul.x - ob.x,
                 xc * (ur.y - ul.y) + yc * (bl.y - ul.y) + ul.y - ob.y,
                 xc * (ur.z - ul.z) + yc * (bl.z - ul.z) + ul.z - ob.z);
    dir = normalize3(dir);
    st3(ry.r, dir);
    int i, j, k;
    if (startup_voxel(g, ob, dir, i, j, k)) { ry.i = i; ry.j = j; ry.k = k; } else ry.i = -1;
  }
  rays[idx] = ry;
}

// ------------------------------------------------------------------------------------------------
// Nearest-hit walk (voxelize.c:536-613) as a persistent wavefront kernel.
//
// Each wavefront draws batches of 64 rays from a ticket counter; one lane owns one ray and runs the
// reference's 3D-DDA for it.  The triangle tests are organised per CELL, not per lane: the wavefront
// picks the cell of its first unfinished lane, every lane currently standing in that cell joins, and
// the cell's packed triangle list is streamed once through wave-uniform (scalar) loads while the
// joined lanes evaluate Möller–Trumbore against SGPR operands.  Lanes in other cells wait their turn.
// A lane that found a hit inside the voxel's exit distance retires; otherwise it steps its DDA to the
// next non-empty cell.  All triangles of a visited cell are tested (the reference's test ignores the
// running minimum), and ties keep the lowest triangle index because lists are in ascending order.
template <bool COUNT>
__global__ void __launch_bounds__(256, RT_WAVES) k_nearest(GridP g, const Ray *__restrict__ rays, Hit *__restrict__ hits,
                                                 Ctl *__restrict__ ctl, int level) {
  __shared__ __attribute__((aligned(16))) float lds_batches[4][kBatch * kRecFloats];
  float *slice = lds_batches[threadIdx.x >> 6];
  const int lane = lane_id();
  const int nrays = ctl->count[level];
  const uint32_t *__restrict__ cstart = g.cell_start;
  const TriRec *__restrict__ recs = g.cell_tris;
  unsigned long long nS = 0, nT = 0, nLane = 0, nWave = 0, nQ = 0;
  uint32_t nPixels = 0, nTraced = 0, nHits = 0;  // wave-uniform tallies, one atomic each per wavefront
  for (;;) {
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&ctl->ticket_nearest[level], 64u);
    base = __builtin_amdgcn_readfirstlane(base);
    if (base >= (uint32_t)nrays) break;
    const int idx = base + lane;
    Ray ry;
    ry.i = -2;
    if (idx < nrays) ry = rays[idx];
    bool alive = ry.i >= 0;
    nPixels += __popcll(__ballot(ry.i != -2));
    nTraced += __popcll(__ballot(alive));
    const f3 o = ld3(ry.o), r = ld3(ry.r);
    Hit h;
    h.tri = alive ? -1 : -2;
    h.d = h.u = h.v = 0.0f; h.i = h.j = h.k = 0; h.pad = 0;
    Dda w;
    int mycell = 0;
    uint32_t mybeg = 0, mycnt = 0;  // triangle list of mycell
    if (alive) {
      if (COUNT) nQ++;
      dda_init(g, o, r, ry.i, ry.j, ry.k, w);
      // stand on the first non-empty cell of the walk (the start voxel counts as visited)
      for (;;) {
        mycell = cell_index(g, w.i, w.j, w.k);
        mybeg = cstart[mycell]; mycnt = cstart[mycell + 1] - mybeg;
        if (COUNT) { nS++; nT += mycnt; }
        if (mycnt > 0) break;
        dda_step(w);
        if (w.i < 0 || w.i >= g.nv[0] || w.j < 0 || w.j >= g.nv[1] || w.k < 0 || w.k >= g.nv[2]) { alive = false; break; }
      }
    }
    for (;;) {
      const unsigned long long m = __ballot(alive);
      if (m == 0) break;
      const int leader = __ffsll(m) - 1;
      const int cell = __builtin_amdgcn_readlane(mycell, leader);
      const uint32_t beg = __builtin_amdgcn_readlane(mybeg, leader), cnt = __builtin_amdgcn_readlane(mycnt, leader);
      const bool in_group = alive && mycell == cell;
      const bool is_short = alive && mycnt <= RT_SHORT_LIST;
      const int mode = pick_mode(__popcll(__ballot(in_group)), cnt, __popcll(__ballot(is_short)));
      // exit distance of this voxel: a hit only counts if it lies before it (voxelize.c:569,574)
      float best = min3_ref(w.tx + w.dtx, w.ty + w.dty, w.tz + w.dtz);
      int best_id = -1;
      float best_u = 0.0f, best_v = 0.0f;
      bool proc;  // this lane's current cell is fully tested after this step
      if (mode == kCellMode) {
        proc = in_group;
        if (COUNT) { nWave += (unsigned long long)cnt; if (proc) nLane += cnt; }
        const f3 rr = proc ? r : mk3(0.0f, 0.0f, 0.0f);  // lanes outside the cell test with r = 0: never a candidate
        stream_cell(recs, beg, beg + cnt, slice, lane,
                    [&](f3 ta, f3 tab, f3 tac, const float *R, uint32_t) {
                      float det, tp; f3 tv;
                      if (tri_candidate(ta, tab, tac, o, rr, det, tp, tv)) {
                        float d, u, v;
                        if (tri_exact(tab, tac, r, det, tp, tv, d, u, v)) {
                          const int id = rec_id(R);
                          if (id != ry.current && d < best) { best = d; best_id = id; best_u = u; best_v = v; }
                        }
                      }
                    },
                    [] { return true; });
      } else if (mode == kLaneMode) {
        proc = is_short;
        const uint32_t pn = proc ? mycnt : 0u;
        for (uint32_t i = 0; __ballot(i < pn) != 0; ++i) {
          if (COUNT) { nWave++; if (i < pn) nLane++; }
          if (i < pn) {
            f3 ta, tab, tac; float det, tp; f3 tv;
            const float *R = load_rec(recs, mybeg + i, ta, tab, tac);
            if (tri_candidate(ta, tab, tac, o, r, det, tp, tv)) {
              float d, u, v;
              if (tri_exact(tab, tac, r, det, tp, tv, d, u, v)) {
                const int id = rec_id(R);
                if (id != ry.current && d < best) { best = d; best_id = id; best_u = u; best_v = v; }
              }
            }
          }
        }
      } else {
        // ray mode: the leader's ray against 64 triangles per step; lane L takes list positions L, L+64, ...
        proc = lane == leader;
        const f3 lo = lane_bcast3(o, leader), lr = lane_bcast3(r, leader);
        const int lcur = __builtin_amdgcn_readlane(ry.current, leader);
        float cb = lane_bcast(best, leader), cu = 0.0f, cv = 0.0f;
        int cid = -1;
        uint32_t cpos = 0xffffffffu;
        for (uint32_t t0 = beg; t0 < beg + cnt; t0 += 64) {
          const uint32_t t = t0 + lane;
          if (COUNT) { nWave++; if (t < beg + cnt) nLane++; }
          if (t < beg + cnt) {
            f3 ta, tab, tac; float det, tp; f3 tv;
            const float *R = load_rec(recs, t, ta, tab, tac);
            if (tri_candidate(ta, tab, tac, lo, lr, det, tp, tv)) {
              float d, u, v;
              if (tri_exact(tab, tac, lr, det, tp, tv, d, u, v)) {
                const int id = rec_id(R);
                if (id != lcur && d < cb) { cb = d; cpos = t; cid = id; cu = u; cv = v; }  // ascending t: earliest wins in a lane
              }
            }
          }
        }
        // smallest d, ties to the earliest list position: the reference's strict `d < dmin` scan order
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          const float od = __shfl_xor(cb, off), ou = __shfl_xor(cu, off), ov = __shfl_xor(cv, off);
          const uint32_t op = __shfl_xor(cpos, off);
          const int oi = __shfl_xor(cid, off);
          if (od < cb || (od == cb && op < cpos)) { cb = od; cpos = op; cid = oi; cu = ou; cv = ov; }
        }
        if (proc) { best = cb; best_id = cid; best_u = cu; best_v = cv; }
      }
      if (proc) {
        if (best_id >= 0) {
          h.tri = best_id; h.d = best; h.u = best_u; h.v = best_v; h.i = w.i; h.j = w.j; h.k = w.k;
          alive = false;
        } else {
          for (;;) {
            dda_step(w);
            if (w.i < 0 || w.i >= g.nv[0] || w.j < 0 || w.j >= g.nv[1] || w.k < 0 || w.k >= g.nv[2]) { alive = false; break; }
            mycell = cell_index(g, w.i, w.j, w.k);
            mybeg = cstart[mycell]; mycnt = cstart[mycell + 1] - mybeg;
            if (COUNT) { nS++; nT += mycnt; }
            if (mycnt > 0) break;
          }
        }
      }
    }
    if (idx < nrays) hits[idx] = h;
    nHits += __popcll(__ballot(h.tri >= 0));
  }
  if (lane == 0) {
    if (level == 0) {
      atomicAdd(&ctl->stat[kPixels], (unsigned long long)nPixels);
      atomicAdd(&ctl->stat[kPixelsTraced], (unsigned long long)nTraced);
    }
    atomicAdd(&ctl->stat[kShaded], (unsigned long long)nHits);
  }
  if (COUNT) {
    // wave-level sums, one atomic per counter per wavefront
    for (int off = 32; off > 0; off >>= 1) {
      nS += __shfl_down(nS, off); nT += __shfl_down(nT, off); nLane += __shfl_down(nLane, off); nQ += __shfl_down(nQ, off);
    }
    if (lane == 0) {
      atomicAdd(&ctl->stat[kNearS], nS); atomicAdd(&ctl->stat[kNearT], nT);
      atomicAdd(&ctl->stat[kNearLaneTests], nLane); atomicAdd(&ctl->stat[kNearWaveTests], nWave * 64ull);
      atomicAdd(&ctl->stat[kNearestQ], nQ);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Hit kernel (raytrace.c:170-204): turns the hits of one level into shaded-hit records and emits the
// secondary rays.  Normal flip, surface colour, procedural texture + bump (fused here), reflected and
// refracted rays.  Children are appended to the next level's queue with one atomic per wavefront
// (ballot + prefix count), so a child's slot is its parent's link.
__global__ void __launch_bounds__(256) k_hit(GridP g, ShadeP sp, const Ray *__restrict__ rays, const Hit *__restrict__ hits,
                                             Node *__restrict__ nodes, const float *__restrict__ geom,
                                             Ray *__restrict__ next_rays, Ctl *__restrict__ ctl, int level,
                                             int node_base, int next_base, int next_cap, bool spawn) {
  const int n = ctl->count[level];
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = lane_id();
  const bool valid = idx < n;
  Hit h; h.tri = -2;
  Ray ry;
  if (valid) { h = hits[idx]; ry = rays[idx]; }
  Node nd;
  nd.tri = h.tri; nd.refl = -1; nd.refr = -1;
  nd.p[0] = nd.p[1] = nd.p[2] = 0.0f; nd.n[0] = nd.n[1] = nd.n[2] = 0.0f; nd.c[0] = nd.c[1] = nd.c[2] = 0.0f;
  bool want_refl = false, want_refr = false;
  f3 refl_dir = mk3(0, 0, 0), refr_dir = mk3(0, 0, 0);
  if (h.tri >= 0) {
    const f3 o = ld3(ry.o), r = ld3(ry.r);
    const f3 onew = raypoint3(o, r, h.d);  // voxelize.c:587
    f3 norm = ld3(sp.tri_aux + 4 * h.tri);
    if (dot3(r, norm) > 0.0f) norm = neg3(norm);  // raytrace.c:175-178
    const int sid = sp.tri_sid[h.tri];
    const SurfaceD sf = sp.surf[sid];
    f3 nc = ld3(sf.color);
    if (sid == 7) {  // raytrace.c:183
      int slot = h.tri == sp.tex_tri[0] ? 0 : h.tri == sp.tex_tri[1] ? 1 : h.tri == sp.tex_tri[2] ? 2 : h.tri == sp.tex_tri[3] ? 3 : -1;
      if (slot >= 0) apply_texture(slot, ld3(geom + 9 * h.tri + 3), ld3(geom + 9 * h.tri + 6), h.u, h.v, norm, nc);
    }
    st3(nd.p, onew); st3(nd.n, norm); st3(nd.c, nc);
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
  if (valid) nodes[node_base + idx] = nd;
}

// Exclusive prefix count of primary hits in row-major pixel order, used to replay the order in which
// the reference consumes rand() for planar lights (raytrace.c:245-251): the shaded hit of pixel p
// uses the 32*npl values starting at 32*npl*rank(p).  Single 1024-thread block over the image.
__global__ void __launch_bounds__(1024) k_hit_rank(int w, int h, int tiles_x, const Node *__restrict__ nodes,
                                                   int32_t *__restrict__ rank) {
  __shared__ uint32_t wave_sum[16];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n = w * h;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    int p = base + tid;
    uint32_t v = 0;
    int node = 0;
    if (p < n) {
      int x = p % w, y = p / w;
      node = ((y >> 3) * tiles_x + (x >> 3)) * kTilePixels + (y & 7) * kTile + (x & 7);
      v = nodes[node].tri >= 0 ? 1u : 0u;
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
    if (p < n) rank[node] = (int32_t)(prefix + x - v);
    __syncthreads();
    if (tid == 1023) carry_s = prefix + x;
    __syncthreads();
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

// ------------------------------------------------------------------------------------------------
// Shadow walk (voxelize.c:615-738), any-hit, persistent wavefront kernel with the same per-cell
// organisation as k_nearest.  Work item = (shadow slot, shaded hit) in slot-major order, so the 64
// lanes of a batch are neighbouring hit points looking at the same light.  Result per item: the
// transparency product ts (>= 0) if the light is visible, -1 if the point is in shadow.
// The reference's per-triangle shadow cache (voxelize.c:647-655) is scan-order state; it is dropped.
template <bool COUNT>
__global__ void __launch_bounds__(256, RT_WAVES) k_shadow(GridP g, ShadeP sp, const Node *__restrict__ nodes,
                                                const int32_t *__restrict__ rand_rank, float *__restrict__ ts_out,
                                                Ctl *__restrict__ ctl, int level, int node_base) {
  __shared__ __attribute__((aligned(16))) float lds_batches[4][kBatch * kRecFloats];
  float *slice = lds_batches[threadIdx.x >> 6];
  const int lane = lane_id();
  const int n = ctl->count[level];
  const long long total = (long long)n * sp.nslots;
  const uint32_t *__restrict__ cstart = g.cell_start;
  const TriRec *__restrict__ recs = g.cell_tris;
  unsigned long long nS = 0, nT = 0, nQ = 0, nW = 0, nLane = 0, nWave = 0;
  for (;;) {
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&ctl->ticket_shadow[level], 64u);
    base = __builtin_amdgcn_readfirstlane(base);
    if ((long long)base >= total) break;
    const long long item = (long long)base + lane;
    bool alive = false;
    int slot = 0, nidx = 0, current = -1;
    f3 a = mk3(0, 0, 0), r = mk3(0, 0, 0);
    float dmax = 0.0f, ts = 1.0f, result = 1.0f;
    bool have_result = false;
    Dda w;
    int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    int mycell = 0;
    uint32_t mybeg = 0, mycnt = 0;  // triangle list of mycell
    if (item < total) {
      slot = (int)(item / n);
      nidx = (int)(item - (long long)slot * n);
      const Node nd = nodes[node_base + nidx];
      current = nd.tri;
      if (current >= 0) {
        if (COUNT) nQ++;
        have_result = true;
        a = ld3(nd.p);
        f3 b, lcol; float lflux;
        slot_light(sp, slot, rand_rank ? rand_rank[nidx] : 0, b, lflux, lcol);
        r = ray3(a, b);
        const float kt_cur = sp.surf[sp.tri_sid[current]].kt;
        bool lit_walk = true;
        if (kt_cur == 0.0f && dot3(r, ld3(sp.tri_aux + 4 * current)) <= 0.0f) { result = -1.0f; lit_walk = false; }  // :640-644
        if (lit_walk) {
          dmax = dist3(a, b);
          int ai[3], bi[3];
          // a point outside the grid: the reference reports an error and treats the light as visible (:661-670)
          if (voxel_of(g, a, ai[0], ai[1], ai[2]) && voxel_of(g, b, bi[0], bi[1], bi[2])) {
#pragma unroll
            for (int c = 0; c < 3; c++) { lo[c] = ai[c] < bi[c] ? ai[c] : bi[c]; hi[c] = ai[c] < bi[c] ? bi[c] : ai[c]; }
            dda_init(g, a, r, ai[0], ai[1], ai[2], w);
            alive = true;
            if (COUNT) nW++;
            for (;;) {
              mycell = cell_index(g, w.i, w.j, w.k);
              mybeg = cstart[mycell]; mycnt = cstart[mycell + 1] - mybeg;
              if (COUNT) { nS++; nT += mycnt; }
              if (mycnt > 0) break;
              dda_step(w);
              if (w.i < lo[0] || w.i > hi[0] || w.j < lo[1] || w.j > hi[1] || w.k < lo[2] || w.k > hi[2]) { alive = false; break; }
            }
          }
        }
      }
    }
    for (;;) {
      const unsigned long long m = __ballot(alive);
      if (m == 0) break;
      const int leader = __ffsll(m) - 1;
      const int cell = __builtin_amdgcn_readlane(mycell, leader);
      const uint32_t beg = __builtin_amdgcn_readlane(mybeg, leader), cnt = __builtin_amdgcn_readlane(mycnt, leader);
      const bool in_group = alive && mycell == cell;
      const bool is_short = alive && mycnt <= RT_SHORT_LIST;
      const int mode = pick_mode(__popcll(__ballot(in_group)), cnt, __popcll(__ballot(is_short)));
      bool proc;  // this lane's current cell is fully tested (or the ray blocked) after this step
      if (mode == kCellMode) {
        proc = in_group;
        bool in = in_group;
        f3 rr = in ? r : mk3(0.0f, 0.0f, 0.0f);  // lanes outside the cell, and blocked ones, test with r = 0
        auto test = [&](f3 ta, f3 tab, f3 tac, const float *R, uint32_t) {
          float det, tp; f3 tv;
          if (tri_candidate(ta, tab, tac, a, rr, det, tp, tv)) {
            float d, u, v;
            if (tri_exact(tab, tac, r, det, tp, tv, d, u, v) && rec_id(R) != current) {
              const float kt = rec_kt(R);
              if (kt > 0.0f) {
                ts *= kt;  // every listing of a transparent triangle attenuates (:703-706)
              } else if (d > 0.00001f && d < dmax) {
                result = -1.0f; alive = false; in = false; rr = mk3(0.0f, 0.0f, 0.0f);  // opaque blocker (:707-711)
              }
            }
          }
        };
        if (__ballot(in) != 0)
          stream_cell(recs, beg, beg + cnt, slice, lane,
                      [&](f3 ta, f3 tab, f3 tac, const float *R, uint32_t pos) {
                        if (COUNT) { nWave++; if (in) nLane++; }
                        test(ta, tab, tac, R, pos);
                      },
                      [&] { return __ballot(in) != 0; });
      } else if (mode == kLaneMode) {
        proc = is_short;
        uint32_t pn = proc ? mycnt : 0u;
        for (uint32_t i = 0; __ballot(i < pn) != 0; ++i) {
          if (COUNT) { nWave++; if (i < pn) nLane++; }
          if (i < pn) {
            f3 ta, tab, tac; float det, tp; f3 tv;
            const float *R = load_rec(recs, mybeg + i, ta, tab, tac);
            if (tri_candidate(ta, tab, tac, a, r, det, tp, tv)) {
              float d, u, v;
              if (tri_exact(tab, tac, r, det, tp, tv, d, u, v) && rec_id(R) != current) {
                const float kt = rec_kt(R);
                if (kt > 0.0f) {
                  ts *= kt;
                } else if (d > 0.00001f && d < dmax) {
                  result = -1.0f; alive = false; pn = 0u;
                }
              }
            }
          }
        }
      } else {
        // ray mode: the leader's shadow ray against 64 triangles per step
        proc = lane == leader;
        const f3 la = lane_bcast3(a, leader), lr = lane_bcast3(r, leader);
        const int lcur = __builtin_amdgcn_readlane(current, leader);
        const float ldmax = lane_bcast(dmax, leader);
        float cts = lane_bcast(ts, leader);
        bool blocked = false;
        for (uint32_t t0 = beg; t0 < beg + cnt && !blocked; t0 += 64) {
          const uint32_t t = t0 + lane;
          if (COUNT) { nWave++; if (t < beg + cnt) nLane++; }
          bool transparent = false, opaque = false;
          float kt = 0.0f;
          if (t < beg + cnt) {
            f3 ta, tab, tac; float det, tp; f3 tv;
            const float *R = load_rec(recs, t, ta, tab, tac);
            if (tri_candidate(ta, tab, tac, la, lr, det, tp, tv)) {
              float d, u, v;
              if (tri_exact(tab, tac, lr, det, tp, tv, d, u, v) && rec_id(R) != lcur) {
                kt = rec_kt(R);
                if (kt > 0.0f) transparent = true;
                else if (d > 0.00001f && d < ldmax) opaque = true;
              }
            }
          }
          // the transparency product is taken in list order, like the reference's scan
          for (unsigned long long tm = __ballot(transparent); tm != 0; tm &= tm - 1) cts *= lane_bcast(kt, __ffsll(tm) - 1);
          blocked = __ballot(opaque) != 0;
        }
        if (proc) {
          if (blocked) { result = -1.0f; alive = false; } else ts = cts;
        }
      }
      if (proc && alive) {
        for (;;) {
          dda_step(w);
          if (w.i < lo[0] || w.i > hi[0] || w.j < lo[1] || w.j > hi[1] || w.k < lo[2] || w.k > hi[2]) { alive = false; break; }
          mycell = cell_index(g, w.i, w.j, w.k);
          mybeg = cstart[mycell]; mycnt = cstart[mycell + 1] - mybeg;
          if (COUNT) { nS++; nT += mycnt; }
          if (mycnt > 0) break;
        }
      }
    }
    if (have_result) ts_out[(long long)(node_base + nidx) * sp.nslots + slot] = result < 0.0f ? -1.0f : ts;
  }
  if (COUNT) {
    for (int off = 32; off > 0; off >>= 1) {
      nS += __shfl_down(nS, off); nT += __shfl_down(nT, off); nQ += __shfl_down(nQ, off); nW += __shfl_down(nW, off);
      nLane += __shfl_down(nLane, off);
    }
    if (lane == 0) {
      atomicAdd(&ctl->stat[kShadLaneTests], nLane); atomicAdd(&ctl->stat[kShadWaveTests], nWave * 64ull);
      atomicAdd(&ctl->stat[kShadS], nS); atomicAdd(&ctl->stat[kShadT], nT);
      atomicAdd(&ctl->stat[kShadowQ], nQ); atomicAdd(&ctl->stat[kShadowWalks], nW);
    }
  }
}

// One light's contribution (raytrace.c:217-235 / 271-289).
RT_D f3 light_term(const SurfaceD &sf, f3 norm, f3 r, f3 rnew, f3 onew, f3 lpos, float lflux, f3 lcol, f3 nc,
                   float ts, float distmod) {
  float ndl = dot3(norm, rnew);
  float df = sf.kd * ndl;
  if (df < 0.0f && sf.kt > 0.0f) df = -df;
  float rf = 0.0f;
  if (sf.ks > 0.0f) {
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
__global__ void __launch_bounds__(256) k_accumulate(ShadeP sp, const Ray *__restrict__ rays, const Node *__restrict__ nodes,
                                                    const float *__restrict__ ts_in, const int32_t *__restrict__ rand_rank,
                                                    float *__restrict__ colors, Ctl *__restrict__ ctl, int level,
                                                    int node_base, float *__restrict__ fb_rgb, int32_t *__restrict__ fb_ids,
                                                    long long fb_base) {
  const int n = ctl->count[level];
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int gidx = node_base + idx;
  const Node nd = nodes[gidx];
  f3 res = mk3(0.0f, 0.0f, 0.0f);
  if (nd.tri >= 0) {
    const SurfaceD sf = sp.surf[sp.tri_sid[nd.tri]];
    const f3 nc = ld3(nd.c), norm = ld3(nd.n), onew = ld3(nd.p), r = ld3(rays[idx].r);
    if (sf.ka > 0.0f) res = mul3(nc, sf.ka * sp.total_flux);
    if (sf.kr > 0.0f) {
      f3 rc = nd.refl >= 0 ? ld3(colors + 4ll * nd.refl) : mk3(0.0f, 0.0f, 0.0f);
      res = add3(res, mul3(rc, sf.kr));
    }
    if (sf.kt > 0.0f) {
      f3 rc = nd.refr >= 0 ? ld3(colors + 4ll * nd.refr) : mk3(0.0f, 0.0f, 0.0f);
      res = add3(res, mul3(rc, sf.kt));
    }
    const float *ts = ts_in + (long long)gidx * sp.nslots;
    f3 tmp = mk3(0.0f, 0.0f, 0.0f);
    for (int c = 0; c < sp.nl; c++) {
      const float t = ts[c];
      if (t >= 0.0f) {
        const LightD L = sp.lights[c];
        const f3 lp = ld3(L.p);
        tmp = light_term(sf, norm, r, ray3(onew, lp), onew, lp, L.flux, ld3(L.color), nc, t, sp.distmod);
        res = add3(res, tmp);
      }
    }
    for (int c = 0; c < sp.npl; c++) {
      f3 sum = mk3(0.0f, 0.0f, 0.0f);
      for (int d = 0; d < kPlanarSamples; d++) {
        const int slot = sp.nl + c * kPlanarSamples + d;
        const float t = ts[slot];
        if (t >= 0.0f) {
          f3 lp, lcol; float lflux;
          slot_light(sp, slot, rand_rank[idx], lp, lflux, lcol);
          sum = light_term(sf, norm, r, ray3(onew, lp), onew, lp, lflux, lcol, nc, t, sp.distmod);
        }
        sum = mul3(sum, 1.0f / kPlanarSamples);
        tmp = add3(tmp, sum);
        res = add3(res, tmp);
      }
    }
  }
  colors[4ll * gidx] = res.x; colors[4ll * gidx + 1] = res.y; colors[4ll * gidx + 2] = res.z;
  if (level == 0) {
    long long p = fb_base + idx;
    fb_rgb[3 * p] = res.x; fb_rgb[3 * p + 1] = res.y; fb_rgb[3 * p + 2] = res.z;
    fb_ids[p] = nd.tri >= 0 ? nd.tri : -1;
  }
}

// ------------------------------------------------------------------------------------------------
// Per-channel min/max over the traced pixels of a chunk (raytrace.c:396-399; NaN never wins because
// the comparisons are the reference's `>` and `<`).  Pixels whose ray never entered the grid do not
// take part.  partial: [blocks][6]; k_minmax_final folds partials into the running d_minmax.
RT_D void mm_fold(float *mn, float *mx, const float *omn, const float *omx) {
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (omx[k] > mx[k]) mx[k] = omx[k];
    if (omn[k] < mn[k]) mn[k] = omn[k];
  }
}
__global__ void __launch_bounds__(256) k_minmax_partial(const Node *__restrict__ nodes, const float *__restrict__ colors,
                                                        int n, float *__restrict__ partial) {
  __shared__ float sm[4][6];
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {FLT_MIN, FLT_MIN, FLT_MIN};
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
    if (nodes[idx].tri >= -1) {
      float c[3] = {colors[4ll * idx], colors[4ll * idx + 1], colors[4ll * idx + 2]};
      mm_fold(mn, mx, c, c);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    float omn[3], omx[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { omn[k] = __shfl_down(mn[k], off); omx[k] = __shfl_down(mx[k], off); }
    mm_fold(mn, mx, omn, omx);
  }
  const int lane = lane_id(), wid = threadIdx.x >> 6;
  if (lane == 0) { for (int k = 0; k < 3; k++) { sm[wid][k] = mn[k]; sm[wid][3 + k] = mx[k]; } }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int wv = 1; wv < 4; wv++) mm_fold(mn, mx, &sm[wv][0], &sm[wv][3]);
    for (int k = 0; k < 3; k++) { partial[blockIdx.x * 6 + k] = mn[k]; partial[blockIdx.x * 6 + 3 + k] = mx[k]; }
  }
}
__global__ void k_minmax_final(const float *__restrict__ partial, int nblocks, float *__restrict__ minmax) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float mn[3] = {minmax[0], minmax[1], minmax[2]}, mx[3] = {minmax[3], minmax[4], minmax[5]};
  for (int b = 0; b < nblocks; b++) mm_fold(mn, mx, partial + 6 * b, partial + 6 * b + 3);
  for (int k = 0; k < 3; k++) { minmax[k] = mn[k]; minmax[3 + k] = mx[k]; }
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
  else if (t < kStatCount + kMaxLevels) totals[t] += (unsigned long long)ctl->count[t - kStatCount];
}

// ------------------------------------------------------------------------------------------------
// Tone map (raytrace.c:433-466): v = pow((c - min) * delta, gamma) * 255 in double, rounded to float,
// truncated to an integer, low byte kept; packed R<<24 | G<<16 | B<<8 (bitmap.h:70).
__global__ void __launch_bounds__(256) k_tonemap(const float *__restrict__ rgb, long long n, const float *__restrict__ minmax,
                                                 float gamma, uint32_t *__restrict__ rgba) {
  long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  uint32_t ch[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    float delta = 1.0f / (minmax[3 + k] - minmax[k]);
    float acc = (float)(pow((double)((rgb[3 * p + k] - minmax[k]) * delta), (double)gamma) * 255.0);
    ch[k] = f2u_x86(acc / 1.0f) & 0xffu;
  }
  rgba[p] = (ch[0] << 24) | (ch[1] << 16) | (ch[2] << 8);
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
