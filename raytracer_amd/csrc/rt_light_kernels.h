// The device side of the light lists (rt_lightlists.h): the kernels that fill the lists from the host's rectangles, what the
// shadow walk (k_shadow's LL instance) uses to answer a ticket from them, and the lists' own kernel, k_shadow_lists.  Part of
// rt_kernels.h, which includes this file inside namespace rt in front of the shadow walk (it uses the record types, the fan,
// the wavefront helpers, the tickets and the scan's tile size defined above that place); not a header to include by itself.
#pragma once
// Light lists (rt_lightlists.h, LightListsP): the lists are filled from the host's rectangles, once per level (level l has
// texels of 2^l x 2^l texels of level 0: a texel index shifted right by l).  k_ll_count: references per
// texel (one wavefront per rectangle); k_order_scan + k_order_scan_totals: exclusive scan; k_ll_starts: the lists' starts, and
// a copy of them as fill cursors; k_ll_fill: the references, in the order the atomics grant (any order will do: a shadow
// query asks whether some triangle of the list blocks, not which).
__global__ void __launch_bounds__(256) k_ll_count(const LLRectD *__restrict__ rects, int n, const LLFaceD *__restrict__ faces, int R, int level,
                                                  uint32_t level_base, uint32_t *__restrict__ counts) {
  const int w = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = lane_id();
  if (w >= n) return;
  const LLRectD r = rects[w];
  const int a0 = (int)(r.rect & 255u) >> level, a1 = (int)((r.rect >> 8) & 255u) >> level, b0 = (int)((r.rect >> 16) & 255u) >> level, b1 = (int)(r.rect >> 24) >> level;
  const int nw = a1 - a0 + 1, total = nw * (b1 - b0 + 1), Rl = R >> level;
  const uint32_t base = level_base + (faces[r.face].base >> (2 * level));
  for (int e = lane; e < total; e += 64) atomicAdd(&counts[base + (uint32_t)((b0 + e / nw) * Rl + a0 + e % nw)], 1u);
}
__global__ void __launch_bounds__(256) k_ll_starts(uint32_t *__restrict__ offsets, const uint32_t *__restrict__ totals, int n, uint32_t all,
                                                   uint32_t *__restrict__ cursor) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const uint32_t v = offsets[i] + totals[i / kScanTile]; offsets[i] = v; cursor[i] = v; }
  else if (i == n) offsets[n] = all;
}
__global__ void __launch_bounds__(256) k_ll_fill(const LLRectD *__restrict__ rects, int n, const LLFaceD *__restrict__ faces, int R, int level,
                                                 uint32_t level_base, uint32_t *__restrict__ cursor, LLRef *__restrict__ refs) {
  const int w = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = lane_id();
  if (w >= n) return;
  const LLRectD r = rects[w];
  const int a0 = (int)(r.rect & 255u) >> level, a1 = (int)((r.rect >> 8) & 255u) >> level, b0 = (int)((r.rect >> 16) & 255u) >> level, b1 = (int)(r.rect >> 24) >> level;
  const int nw = a1 - a0 + 1, total = nw * (b1 - b0 + 1), Rl = R >> level;
  const uint32_t base = level_base + (faces[r.face].base >> (2 * level));
  LLRef ref; ref.tri = r.tri; ref.rect = (uint32_t)a0 | (uint32_t)a1 << 8 | (uint32_t)b0 << 16 | (uint32_t)b1 << 24;  // the rectangle in this level's texels
  for (int e = lane; e < total; e += 64) refs[atomicAdd(&cursor[base + (uint32_t)((b0 + e / nw) * Rl + a0 + e % nw)], 1u)] = ref;
}
// ------------------------------------------------------------------------------------------------
// Shadow queries from the light lists (rt_lightlists.h; LightListsP), for the 64 hit points of a wavefront that look at one
// point light.  Every walking lane (alive) finds its texel on the cube around the light; the lists of the texels the
// wavefront covers hold every triangle the exact test can accept for any of its rays.  First the references: the texels'
// lists are read 64 positions per pass and the triangles the wavefront takes (each once: at the first texel of its box
// that the triangle's rectangle covers) are collected in LDS; then the collected triangles, 64 per pass: one lane per
// triangle gathers its record and culls it for the whole fan (fan_plane_behind, fan_slabs_pass), what is left goes
// through LDS to the test every lane runs.  A ray none of these triangles blocks is lit (hit_id stays -1); a ray one of
// them blocks (hit_id: the triangle) is in shadow if lists_verify agrees.  false: the wavefront does not fit the lists
// (a hit point outside the face's window or off the face, an origin beyond the norm the lists were built for, its box of
// texels too large even at level max_level); nothing has been decided then.
// The shadow test of one record for all lanes (the filter decides what it can prove, tri_test the rest, voxelize.c:700-711);
// true if some lane passed the first slab.  For scenes without a transparent triangle: a hit counts if it blocks.
RT_D bool lists_test(const GridP &g, const float *R, f3 a, f3 r, f3 pm, float kc, int cur, float dmax, int &hit_id, bool &open) {
  const RecA A = rec_a(R);
  PlTest t;
  const bool p1 = pl_stage1(A.n, A.ac, A.q, r, pm, kc, t);
  if (wave_any(p1)) {
    asm volatile("; some lane passed the first slab");  // keeps the scalar branch from being merged into the per-lane one
    int verdict = 0;
    float kt = 0.0f;
    bool s3 = false;
    if (p1) verdict = shadow_verdict(R, A, t, a, r, pm, kc, g.lmax, dmax, kt, s3);
    if (verdict != 0) {
      const int id = rec_id(R);
      if (id == cur) {
        verdict = 0;
      } else if (verdict == 1) {
        float d, u, v;
        verdict = (exact_test(g, id, a, r, d, u, v) && d > 0.00001f && d < dmax) ? 3 : 0;
      }
      if (verdict == 3) { hit_id = id; open = false; }
    }
    return true;
  }
  return false;
}
RT_D bool lists_lookup(const GridP &g, const LightListsP &ll, int light_i, const Fan &fan, bool alive, f3 a, f3 r, f3 pm, float onorm, int current,
                       float dmax, float *slice, int lane, int max_level, int &hit_id, uint32_t (&diag)[3]) {
  const unsigned long long walking = __ballot(alive);
  const int first = __ffsll(walking) - 1;
  // the face of the first walking lane's hit point, for all lanes: faces overlap (coordinates up to kLLFaceExt), so a
  // tile across a seam of the cube is still looked up on one face
  const f3 d = sub3(a, fan.L);
  const float adx = __builtin_fabsf(d.x), ady = __builtin_fabsf(d.y), adz = __builtin_fabsf(d.z);
  const int ax_own = (adx >= ady && adx >= adz) ? 0 : (ady >= adz ? 1 : 2);
  const int ax = __builtin_amdgcn_readlane(ax_own, first);
  const float dw = ax == 0 ? d.x : ax == 1 ? d.y : d.z;
  const float du = ax == 0 ? d.y : d.x, dv = ax == 2 ? d.y : d.z;
  const bool neg = lane_bcast(dw, first) < 0.0f;
  const float aw = __builtin_fabsf(dw);
  const float p = du / aw, q = dv / aw;
  const bool onface = (dw < 0.0f) == neg && __builtin_fabsf(p) <= kLLFaceExt && __builtin_fabsf(q) <= kLLFaceExt;  // (a NaN: false)
  const int face0 = ax * 2 + (neg ? 1 : 0);
  LLFaceD F = ll.faces[(light_i * 6 + face0) * 2];
  // window 0 if every hit point lies in it, else window 1 (the whole face)
  if (F.on == 0u || __ballot(alive && !(p >= F.p0 && p <= F.p1 && q >= F.q0 && q <= F.q1)) != 0) F = ll.faces[(light_i * 6 + face0) * 2 + 1];
  const bool fits = !alive || (onface && p >= F.p0 && p <= F.p1 && q >= F.q0 && q <= F.q1 && onorm <= ll.olim);
  int tu = (int)((p - F.p0) * F.sp), tv = (int)((q - F.q0) * F.sq);
  tu = tu < 0 ? 0 : tu > ll.R - 1 ? ll.R - 1 : tu;
  tv = tv < 0 ? 0 : tv > ll.R - 1 ? ll.R - 1 : tv;
  if (F.on == 0u || __ballot(!fits) != 0) return false;
  int U0 = -(int)wave_max(alive ? -(float)tu : -1e9f), U1 = (int)wave_max(alive ? (float)tu : -1.0f);
  int V0 = -(int)wave_max(alive ? -(float)tv : -1e9f), V1 = (int)wave_max(alive ? (float)tv : -1.0f);
  // the finest level at which the wavefront's box of texels is small (a texel of level l is 2^l x 2^l of level 0's)
  int level = 0;
  while (level < max_level && ((U1 >> level) - (U0 >> level) + 1) * ((V1 >> level) - (V0 >> level) + 1) > RT_LL_TEXELS) level++;
  U0 >>= level; U1 >>= level; V0 >>= level; V1 >>= level;
  const int nw = U1 - U0 + 1, ntex = nw * (V1 - V0 + 1), Rl = ll.R >> level;
  if (ntex > (level < ll.levels - 1 ? RT_LL_TEXELS : 64)) return false;
  uint32_t off = 0u, cnt = 0u;
  if (lane < ntex) {
    const uint32_t t = ll_level_base(ll.texels0, level) + (F.base >> (2 * level)) + (uint32_t)((V0 + lane / nw) * Rl + U0 + lane % nw);
    off = ll.offsets[t]; cnt = ll.offsets[t + 1] - off;
  }
  const float onorm_all = wave_max(alive ? onorm : 0.0f);
  diag[0] = (uint32_t)(level * 1000 + ntex);
  bool open = alive, go = true;
  int tx = U0, ty = V0;
  constexpr int kSlots = 32;                                   // records of the slice for a pass's survivors ...
  constexpr int kIds = kSliceFloats - kSlots * kRecFloats;     // ... and the rest for the collected triangles
  static_assert(kIds >= 128, "room for two passes' references");
  int *ids = reinterpret_cast<int *>(slice + kSlots * kRecFloats);
  int nids = 0;
  int j = 0;
  uint32_t k0 = 0u;
  for (;;) {
    while (j < ntex && nids <= kIds - 64) {
      const uint32_t off_j = (uint32_t)__builtin_amdgcn_readlane((int)off, j), cnt_j = (uint32_t)__builtin_amdgcn_readlane((int)cnt, j);
      if (k0 < cnt_j) {
        const uint32_t at = k0 + (uint32_t)lane;
        bool own = false;
        int id = 0;
        if (at < cnt_j) {
          const LLRef ref = ll.refs[off_j + at];
          const int ra0 = (int)(ref.rect & 255u), rb0 = (int)((ref.rect >> 16) & 255u);
          own = tx == (ra0 > U0 ? ra0 : U0) && ty == (rb0 > V0 ? rb0 : V0);
          id = ref.tri;
        }
        const unsigned long long m = __ballot(own);
        if (own) ids[nids + __popcll(m & lanes_below(lane))] = id;
        nids += __popcll(m);
        diag[1] += (uint32_t)__popcll(m);
        k0 += 64u;
      }
      if (k0 >= cnt_j) {
        j++; k0 = 0u;
        if (++tx > U1) { tx = U0; ty++; }
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (nids == 0) break;
    for (int b0 = 0; b0 < nids && go; b0 += 64) {
      bool pass = false;
      const v4f *R = nullptr;
      if (b0 + lane < nids) {
        R = reinterpret_cast<const v4f *>(g.filt + (size_t)kRecFloats * (size_t)ids[b0 + lane]);
        const v4f r0 = R[0], r1 = R[1], r2 = R[2], r3 = R[3], r4 = R[4];
        RecA A; A.n = mk3(r0.x, r0.y, r0.z); A.ac = mk3(r1.x, r1.y, r1.z); A.q = mk3(r1.w, r2.x, r2.y);
        const float kc_max = __builtin_fmaf(r0.w, onorm_all, r2.w);
        pass = !fan_plane_behind(A, r4.z, kc_max, fan) &&
               fan_slabs_pass(A, mk3(r3.x, r3.y, r3.z), mk3(r3.w, r4.x, r4.y), kc_max, r0.w, r2.w, fan);
      }
      unsigned long long todo = __ballot(pass);
      while (todo != 0 && go) {
        const int before = __popcll(todo & lanes_below(lane));
        const bool mine = ((todo >> lane) & 1ull) != 0 && before < kSlots;
        if (mine) {
          v4f *D = reinterpret_cast<v4f *>(slice + before * kRecFloats);
          const v4f r0 = R[0], r1 = R[1], r2 = R[2], r3 = R[3], r4 = R[4];
          D[0] = r0; D[1] = r1; D[2] = r2; D[3] = r3; D[4] = r4;
        }
        const unsigned long long placed = __ballot(mine);
        todo &= ~placed;
        __builtin_amdgcn_wave_barrier();
        const int n = __popcll(placed);
        diag[2] += (uint32_t)n;
        for (int q2 = 0; q2 < n && go; q2++) {
          const float *Rj = slice + q2 * kRecFloats;
          const float kc = open ? __builtin_fmaf(Rj[3], onorm, Rj[11]) : -INFINITY;
          if (lists_test(g, Rj, a, r, pm, kc, current, dmax, hit_id, open)) go = __ballot(open) != 0;
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    nids = 0;
    if (!go || j >= ntex) break;
  }
  return true;
}
// The candidate blockers against the walks: a lane's blocker counts if its own walk (the reference's accumulated-sum DDA,
// stepped without looking at any list) passes a cell of the triangle's voxel box, i.e. a cell that lists it: the reference's
// scan finds a blocker there at the latest (voxelize.c:615-738 stops at the first).  start(v, vi, vj, vk) gives a lane's
// walk standing in the voxel of its hit point, and that voxel.  false: some lane's blocker is not on its walk (nothing is
// decided then).
template <typename S>
RT_D bool lists_verify(const LightListsP &ll, bool alive, int hit_id, f3 r, S &&start) {
  bool need = alive && hit_id >= 0, failed = false;
  if (__ballot(need) != 0) {
    int lo0 = 0, lo1 = 0, lo2 = 0, hi0 = -1, hi1 = -1, hi2 = -1, vi = 0, vj = 0, vk = 0;
    Walk v;
    v.tx = v.ty = v.tz = v.dtx = v.dty = v.dtz = 0.0f; v.cell = 0; v.nx = v.ny = v.nz = 0;
    if (need) {
      const int32_t *bx = ll.tri_box + 6 * (size_t)hit_id;
      lo0 = bx[0]; lo1 = bx[1]; lo2 = bx[2]; hi0 = bx[3]; hi1 = bx[4]; hi2 = bx[5];
      start(v, vi, vj, vk);
    }
    const int si = r.x > 0.0f ? 1 : -1, sj = r.y > 0.0f ? 1 : -1, sk = r.z > 0.0f ? 1 : -1;
    for (;;) {
      if (need && vi >= lo0 && vi <= hi0 && vj >= lo1 && vj <= hi1 && vk >= lo2 && vk <= hi2) need = false;
      if (__ballot(need) == 0) break;
      if (need) {  // walk_step, with the voxel's indices instead of the cell's
        const float xn = v.tx + v.dtx, yn = v.ty + v.dty, zn = v.tz + v.dtz;
        const bool xy = xn < yn;
        const bool sx = xy & (xn < zn), sy = !xy & (yn < zn);
        v.tx = sx ? xn : v.tx; v.ty = sy ? yn : v.ty; v.tz = (sx | sy) ? v.tz : zn;
        v.nx -= sx ? 1 : 0; v.ny -= sy ? 1 : 0; v.nz -= (sx | sy) ? 0 : 1;
        vi += sx ? si : 0; vj += sy ? sj : 0; vk += (sx | sy) ? 0 : sk;
        if ((v.nx | v.ny | v.nz) < 0) { need = false; failed = true; }
      }
    }
  }
  return __ballot(failed) == 0;
}
// ------------------------------------------------------------------------------------------------
// Shadow queries from the light lists alone (scenes without transparent triangles, point lights only, one recursion
// level): what k_shadow<..., LL> does inside its ticket loop, without the walk's code and state around it.  A ticket is 64
// consecutive shaded hits (a tile's hit points) and goes through all lights in turn: the hit points, their normals,
// voxels and bounding box are fetched and worked out once for all lights.  Only the narrow cases are settled here (the
// tile's hit points cover few texels at a level up to RT_LISTS_MAX_LEVEL): their cost is even, so the wavefronts finish
// together.  A (ticket, light) that is wide, or that the lists cannot settle, is appended to `deferred` (deferred[0]: how
// many; entries light * (n / 64) + ticket: k_shadow's own ticket numbers) and left to k_shadow, where every such pair is a
// ticket of its own.
#ifndef RT_WAVES_LISTS
#define RT_WAVES_LISTS 6
#endif
#ifndef RT_LISTS_MAX_LEVEL
#define RT_LISTS_MAX_LEVEL 2  // the coarsest level of texels a ticket may need to be settled by k_shadow_lists (temple 4K, both kernels: 0: 2.47 ms, 1: 1.50, 2: 1.42, 3: 1.94)
#endif
struct ListsArgs {
  GridP g; ShadeP sp; const SOrigin *sorg; float *ts_out; long long ts_stride; Ctl *ctl; int node_base; LightListsP ll;
  uint32_t *deferred;
};
__global__ void __launch_bounds__(256, RT_WAVES_LISTS) k_shadow_lists(ListsArgs args) {
  __shared__ __attribute__((aligned(16))) float lds_batches[4][kSliceFloats];
  float *slice = lds_batches[threadIdx.x >> 6];
  const int lane = lane_id();
  const GridP &g = args.g;
  const LightListsP &ll = args.ll;
  __shared__ int32_t occ_box_s[8];
  if (threadIdx.x < 6) occ_box_s[threadIdx.x] = g.occ_box[threadIdx.x];
  __syncthreads();
  const int n = queue_len(args.ctl, 0);
  uint32_t nListed = 0, nDeferred = 0;
  Tickets tickets((unsigned long long)n);
  for (;;) {
    uint32_t base;
    if (!tickets.take(ticket_counters(args.ctl, 1, 1), lane, base)) break;  // (the counters of level 1's shadow walk: there is no level 1)
    const int node = (int)base + lane;
    int current = -1;
    f3 a = mk3(0.0f, 0.0f, 0.0f), nrm = mk3(0.0f, 0.0f, 0.0f);
    if (node < n) {
      const SOrigin nd = args.sorg[args.node_base + node];
      current = nd.tri;
      a = ld3(nd.p);
    }
    const bool has = current >= 0;
    if (__ballot(has) == 0) continue;
    if (has) nrm = ld3(args.sp.tri_aux + 4 * current);
    int ai[3] = {0, 0, 0};
    const bool a_in = has && voxel_of(g, a, ai[0], ai[1], ai[2]);
    // the box of the ticket's hit points, for the fans of all lights (make_fan; a NaN among them: no fan, the walk)
    const bool nan_any = __ballot(has && (a.x != a.x || a.y != a.y || a.z != a.z)) != 0;
    const f3 blo = mk3(-wave_max(has ? -a.x : -INFINITY), -wave_max(has ? -a.y : -INFINITY), -wave_max(has ? -a.z : -INFINITY));
    const f3 bhi = mk3(wave_max(has ? a.x : -INFINITY), wave_max(has ? a.y : -INFINITY), wave_max(has ? a.z : -INFINITY));
    const f3 bc = mk3(0.5f * (blo.x + bhi.x), 0.5f * (blo.y + bhi.y), 0.5f * (blo.z + bhi.z));
    const f3 bh = mk3(__builtin_fmaf(__builtin_fmaxf(bhi.x - bc.x, bc.x - blo.x), 1.000001f, 1e-30f),
                      __builtin_fmaf(__builtin_fmaxf(bhi.y - bc.y, bc.y - blo.y), 1.000001f, 1e-30f),
                      __builtin_fmaf(__builtin_fmaxf(bhi.z - bc.z, bc.z - blo.z), 1.000001f, 1e-30f));
    for (int light_i = 0; light_i < args.sp.nl; light_i++) {
      const f3 b = ld3(args.sp.lights[light_i].p);
      const f3 r = ray3(a, b);
      float result = 1.0f, dmax = 0.0f;
      bool alive = false;
      int bi[3] = {0, 0, 0};
      if (has) {
        if (dot3(r, nrm) <= 0.0f) result = -1.0f;  // voxelize.c:640-644 (no triangle is transparent here)
        else {
          dmax = dist3(a, b);
          // a point outside the grid: the reference reports an error and treats the light as visible (voxelize.c:661-670)
          alive = a_in && voxel_of(g, b, bi[0], bi[1], bi[2]);
        }
      }
      bool settled = true;
      if (__ballot(alive) != 0) {
        settled = false;
        f3 pm; float onorm, odr;
        pl_ray(a, r, pm, onorm, odr);
        Fan fan;
        {
          const f3 rc = sub3(b, bc);
          const f3 mc = mk3(__builtin_fmaf(bc.y, rc.z, -(bc.z * rc.y)), __builtin_fmaf(bc.z, rc.x, -(bc.x * rc.z)), __builtin_fmaf(bc.x, rc.y, -(bc.y * rc.x)));
          fan.on = !nan_any;
          fan.c = lane_bcast3(bc, 0); fan.h = lane_bcast3(bh, 0); fan.rc = lane_bcast3(rc, 0); fan.mc = lane_bcast3(mc, 0); fan.L = lane_bcast3(b, 0);
          fan.smax = lane_bcast(wave_max(alive ? dmax : 0.0f) * 1.00001f, 0);
          fan.hsum = lane_bcast(bh.x + bh.y + bh.z, 0);
          fan.lnorm = lane_bcast(__builtin_fabsf(b.x) + __builtin_fabsf(b.y) + __builtin_fabsf(b.z), 0);
        }
        int hit_id = -1;
        uint32_t diag[3] = {0u, 0u, 0u};
        if (fan.on && lists_lookup(g, ll, light_i, fan, alive, a, r, pm, onorm, current, dmax, slice, lane, ll.levels - 1 < RT_LISTS_MAX_LEVEL ? ll.levels - 1 : RT_LISTS_MAX_LEVEL, hit_id, diag) &&
            lists_verify(ll, alive, hit_id, r, [&](Walk &v, int &vi, int &vj, int &vk) {
              int lo[3], hi[3];
#pragma unroll
              for (int c = 0; c < 3; c++) { lo[c] = ai[c] < bi[c] ? ai[c] : bi[c]; hi[c] = ai[c] < bi[c] ? bi[c] : ai[c]; }
              walk_init(g, a, r, ai[0], ai[1], ai[2], lo, hi, true, occ_box_s, v);
              vi = ai[0]; vj = ai[1]; vk = ai[2];
            })) {
          if (alive && hit_id >= 0) result = -1.0f;
          settled = true;
          nListed++;
        }
      }
      if (settled) {
        // slot-major, like k_shadow: one 256-byte line per ticket and light
        if (has) args.ts_out[(long long)light_i * args.ts_stride + args.node_base + node] = result < 0.0f ? -1.0f : 1.0f;
      } else {
        nDeferred++;
        if (lane == 0) {
          const uint32_t at = atomicAdd(args.deferred, 1u);
          args.deferred[1 + at] = (uint32_t)light_i * (uint32_t)(n >> 6) + (base >> 6);
        }
      }
    }
  }
  if (lane == 0 && nListed != 0) atomicAdd(&args.ctl->stat[kShadListTickets], (unsigned long long)nListed);
}

