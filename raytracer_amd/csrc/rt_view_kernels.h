// The device side of the view lists (rt_viewlists.h): the kernels that build a frame's lists, and what the nearest-hit walk
// (k_nearest's VL instance) uses to answer a tile's primary rays from them.  Part of rt_kernels.h, which includes this file
// inside namespace rt where it used to stand (it uses the record types, the wavefront helpers, the phase clocks and the walk
// state defined above that place); not a header to include by itself.
#pragma once
// View lists (rt_viewlists.h, ViewListsP), built for every frame.  k_view_rects: every triangle's rectangle of texels and
// the number of 64-texel pieces it is cut into (scanned by k_order_scan + k_order_scan_totals + k_ll_starts, like the
// texels' counts afterwards).  k_view_pieces writes down which triangle every piece belongs to; k_view_scatter: one
// wavefront per piece, whatever triangles they belong to (a floor across the whole screen is thousands of texels, most
// triangles are one or two).
// FILL = false: references per texel; FILL = true: the references, unless the frame needs more of them than there is room
// for (k_nearest sees the same number and leaves the lists alone).
__global__ void __launch_bounds__(256) k_view_rects(int nt, const float *__restrict__ verts, CameraD cam, int shift, ViewRectD *__restrict__ rects,
                                                    uint32_t *__restrict__ pieces) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t > nt) return;
  if (t == nt) { pieces[nt] = 0u; return; }
  ViewRect r;
  ViewRectD o; o.ab = 1u; o.cd = 0u;
  uint32_t n = 0u;
  if (view_rect(verts + 9 * (size_t)t, cam, r)) {
    const int s = 3 + shift;
    const uint32_t a0 = (uint32_t)(r.a0 >> s), a1 = (uint32_t)(r.a1 >> s), b0 = (uint32_t)(r.b0 >> s), b1 = (uint32_t)(r.b1 >> s);
    o.ab = a0 | a1 << 16; o.cd = b0 | b1 << 16;
    n = ((a1 - a0 + 1u) * (b1 - b0 + 1u) + 63u) / 64u;
  }
  rects[t] = o;
  pieces[t] = n;
}
// piece_tri[p] = the triangle piece p belongs to (one thread per triangle writes its pieces' entries)
__global__ void __launch_bounds__(256) k_view_pieces(int nt, const uint32_t *__restrict__ piece_start, uint32_t cap_pieces, int32_t *__restrict__ piece_tri) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nt) return;
  const uint32_t p0 = piece_start[t], p1 = piece_start[t + 1] < cap_pieces ? piece_start[t + 1] : cap_pieces;
  for (uint32_t p = p0; p < p1; p++) piece_tri[p] = t;
}
template <bool FILL>
__global__ void __launch_bounds__(256) k_view_scatter(const ViewRectD *__restrict__ rects, const uint32_t *__restrict__ piece_start, const int32_t *__restrict__ piece_tri,
                                                      uint32_t cap_pieces, int nt, int ntx, int ntexels, uint32_t cap, uint32_t *__restrict__ counts_or_cursor,
                                                      const uint32_t *__restrict__ offsets, int32_t *__restrict__ refs) {
  const uint32_t total = piece_start[nt];
  if (total > cap_pieces) {  // (more pieces than piece_tri holds: then more references than their buffer holds, too; the count says so)
    if (!FILL && blockIdx.x == 0 && threadIdx.x == 0) counts_or_cursor[0] = cap + 1u;
    return;
  }
  if (FILL && offsets[ntexels] > cap) return;
  const int lane = lane_id();
  const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
  for (uint32_t W = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); W < total; W += nwaves) {
    const int t = piece_tri[W];
    if ((unsigned)t >= (unsigned)nt) continue;  // (cannot happen while the pieces' scan holds; a stray entry must not become an address)
    const ViewRectD r = rects[t];
    const uint32_t a0 = r.ab & 0xffffu, a1 = r.ab >> 16, b0 = r.cd & 0xffffu, b1 = r.cd >> 16;
    const uint32_t nw = a1 - a0 + 1u, cells = nw * (b1 - b0 + 1u);
    const uint32_t e = (W - piece_start[t]) * 64u + (uint32_t)lane;
    if (e < cells) {
      const uint32_t texel = (b0 + e / nw) * (uint32_t)ntx + a0 + e % nw;
      if (texel >= (uint32_t)ntexels) continue;
      if (FILL) { const uint32_t at = atomicAdd(&counts_or_cursor[texel], 1u); if (at < cap) refs[at] = t; }
      else atomicAdd(&counts_or_cursor[texel], 1u);
    }
  }
}
// ------------------------------------------------------------------------------------------------
// Primary rays from the view lists (rt_viewlists.h; ViewListsP), for the 64 pixels of a tile.  The texel's list holds
// every triangle the exact test can accept for any of the tile's rays.  64 references per pass: one lane per triangle
// gathers its record and culls it for the whole tile (apex_slabs_pass), what is left goes through LDS to the test every
// lane runs (test(record): the walk's own, with no voxel's exit distance in force).
//
// Can ANY ray of a bundle that starts at ONE point o pass the three slabs of this record, with a hit in front of o?
// false only if surely none can.  With a common origin the filter's quantities are linear in the direction:
// det = r.N, tp = r.W1, nv = r.W2 with W1 = ik x o - Q, W2 = ij x o + P (rt_math.h: tp = m.ik - r.Q, nv = m.ij + r.P, m = o x r), so
// X1 = r.(2 W1 - N), X2 = r.(-2 W2 - N), X3 = r.(2 W1 - 2 W2 - N) and over |r - axis| <= rho each stays within rho |gradient| of
// its value at the axis.  tri_test accepts a ray only if |X*| <= |det*| (1 + 7 eps) + kc for each (bundle_slabs_pass); the
// values worked out here are within kc / 4 of the exact ones (W1, W2 in binary32: 4 eps L (|o| + A) each), so no ray passes if
//     |X(axis)| - 1.001 rho |gradient| > 1.00001 (|det(axis)| + rho |N|) + 2 kc_max.
// In front: d = dp / det with dp = i.N - o.N the same for all rays; as in fan_plane_behind no ray is accepted if
// |det(axis)| - rho |N| > 1.5 kc_max (one sign, large enough) and sign(det) dp < -2 |ik| kc_max.  A NaN compares false: passes.
// tests/test_filter_cpu.py restates it in binary32 and checks that no accepted triangle is dropped.
RT_D bool apex_slabs_pass(const RecA &A, f3 ab, f3 P, float an, float kc_max, f3 o, f3 axis, float rho) {
  const f3 w1 = mk3(__builtin_fmaf(A.ac.y, o.z, __builtin_fmaf(-A.ac.z, o.y, -A.q.x)), __builtin_fmaf(A.ac.z, o.x, __builtin_fmaf(-A.ac.x, o.z, -A.q.y)),
                    __builtin_fmaf(A.ac.x, o.y, __builtin_fmaf(-A.ac.y, o.x, -A.q.z)));
  const f3 w2 = mk3(__builtin_fmaf(ab.y, o.z, __builtin_fmaf(-ab.z, o.y, P.x)), __builtin_fmaf(ab.z, o.x, __builtin_fmaf(-ab.x, o.z, P.y)),
                    __builtin_fmaf(ab.x, o.y, __builtin_fmaf(-ab.y, o.x, P.z)));
  const f3 g1 = mk3(__builtin_fmaf(2.0f, w1.x, -A.n.x), __builtin_fmaf(2.0f, w1.y, -A.n.y), __builtin_fmaf(2.0f, w1.z, -A.n.z));
  const f3 g2 = mk3(__builtin_fmaf(-2.0f, w2.x, -A.n.x), __builtin_fmaf(-2.0f, w2.y, -A.n.y), __builtin_fmaf(-2.0f, w2.z, -A.n.z));
  const f3 g3 = mk3(__builtin_fmaf(-2.0f, w2.x, g1.x), __builtin_fmaf(-2.0f, w2.y, g1.y), __builtin_fmaf(-2.0f, w2.z, g1.z));
  auto along = [&](f3 v) { return __builtin_fmaf(axis.x, v.x, __builtin_fmaf(axis.y, v.y, axis.z * v.z)); };
  auto norm = [&](f3 v) { return __builtin_amdgcn_sqrtf(__builtin_fmaf(v.x, v.x, __builtin_fmaf(v.y, v.y, v.z * v.z))) * 1.001f; };
  const float det = along(A.n), rn = rho * norm(A.n);
  const float T = __builtin_fmaf(__builtin_fabsf(det) + rn, 1.00001f, kc_max + kc_max);
  const bool out = (__builtin_fabsf(along(g1)) - rho * norm(g1) > T) | (__builtin_fabsf(along(g2)) - rho * norm(g2) > T) |
                   (__builtin_fabsf(along(g3)) - rho * norm(g3) > T);
  const float dp = __builtin_fmaf(-o.x, A.n.x, __builtin_fmaf(-o.y, A.n.y, __builtin_fmaf(-o.z, A.n.z, an)));
  const float sdp = __int_as_float(__float_as_int(dp) ^ (__float_as_int(det) & (int)0x80000000));
  const bool behind = (__builtin_fabsf(det) - rn > 1.5f * kc_max) & (sdp < -2.0f * norm(A.ac) * kc_max);
  return !(out | behind);
}
// in: the lanes whose rays take part (all with the origin o); beg .. end: the texel's references
constexpr int kViewSlotFloats = 32;
template <typename T>
RT_D void view_candidates(const GridP &g, const int32_t *__restrict__ refs, uint32_t beg, uint32_t end, bool in, f3 o, f3 r, float onorm,
                          float *slice, int lane, uint32_t (&diag)[3], Phases &ph, T &&test) {
  const int mid = central_lane(in, lane, true, 0);
  const f3 axis = lane_bcast3(r, mid), o_all = lane_bcast3(o, mid);
  const f3 dr = sub3(r, axis);
  float rho_own = in ? __builtin_fmaf(__builtin_amdgcn_sqrtf(dot3(dr, dr)), 1.001f, 1e-7f) : 0.0f;
  if (!(rho_own >= 0.0f)) rho_own = INFINITY;  // (a NaN would be dropped by the maximum)
  const float rho = lane_bcast(wave_max(rho_own), 0);
  float on_own = in ? onorm : 0.0f;
  if (!(on_own >= 0.0f)) on_own = INFINITY;
  const float onorm_all = lane_bcast(wave_max(on_own), 0);
  // a survivor's slot in the slice: its record, then i, ij, ik for the exact test (so that no test waits for memory)
  constexpr int kSlots = kSliceFloats / kViewSlotFloats;
  static_assert(kSlots >= 24 && kViewSlotFloats % 4 == 0 && kViewSlotFloats >= kRecFloats + 9, "most passes fit the slice at once; records stay 16-byte aligned");
  for (uint32_t e0 = beg; e0 < end; e0 += 64u) {
    bool pass = false;
    const v4f *R = nullptr;
    const float *G = nullptr;
    if (e0 + (uint32_t)lane < end) {
      const int id = refs[e0 + lane];
      R = reinterpret_cast<const v4f *>(g.filt + (size_t)kRecFloats * (size_t)id);
      G = g.geom + 9 * (size_t)id;
      const v4f r0 = R[0], r1 = R[1], r2 = R[2], r3 = R[3], r4 = R[4];
      RecA A; A.n = mk3(r0.x, r0.y, r0.z); A.ac = mk3(r1.x, r1.y, r1.z); A.q = mk3(r1.w, r2.x, r2.y);
      const float kc_max = __builtin_fmaf(r0.w, onorm_all, r2.w) + g.slack;
      pass = apex_slabs_pass(A, mk3(r3.x, r3.y, r3.z), mk3(r3.w, r4.x, r4.y), r4.z, kc_max, o_all, axis, rho);
    }
    diag[1] += (uint32_t)__popcll(__ballot(e0 + (uint32_t)lane < end));
    unsigned long long todo = __ballot(pass);
    ph.end<3, true>();
    while (todo != 0) {
      const int before = __popcll(todo & lanes_below(lane));
      const bool mine = ((todo >> lane) & 1ull) != 0 && before < kSlots;
      if (mine) {
        v4f *D = reinterpret_cast<v4f *>(slice + before * kViewSlotFloats);
        const v4f r0 = R[0], r1 = R[1], r2 = R[2], r3 = R[3], r4 = R[4];
        const float g0 = G[0], g1 = G[1], g2 = G[2], g3 = G[3], g4 = G[4], g5 = G[5], g6 = G[6], g7 = G[7], g8 = G[8];
        D[0] = r0; D[1] = r1; D[2] = r2; D[3] = r3; D[4] = r4;
        v4f x; x.x = g0; x.y = g1; x.z = g2; x.w = g3; D[5] = x;
        x.x = g4; x.y = g5; x.z = g6; x.w = g7; D[6] = x;
        slice[before * kViewSlotFloats + 28] = g8;
      }
      const unsigned long long placed = __ballot(mine);
      todo &= ~placed;
      __builtin_amdgcn_wave_barrier();
      const int n = __popcll(placed);
      diag[2] += (uint32_t)n;
      ph.end<4, true>();
      for (int q2 = 0; q2 < n; q2++) test(slice + q2 * kViewSlotFloats);
      __builtin_amdgcn_wave_barrier();
      ph.end<5, true>();
    }
  }
}
// The nearest candidates against the walks.  The reference's walk (voxelize.c:536-613) takes, in the first cell of the walk
// that has one, the nearest accepted triangle that the cell lists and that lies before the cell's exit distance.  With
// best the smallest distance among ALL accepted triangles of the scene (ties: the smallest index) no cell whose exit
// distance is at most best can have such a triangle; so if the first cell with a larger exit distance lists that
// triangle, the walk stops there with it.  Steps every lane that has a candidate (need) to that cell with the reference's
// accumulated-sum DDA and looks the cell up in the triangle's voxel box (= the cells that list it); the walk comes out
// standing in that cell.  false: some lane's candidate is not confirmed that way (nothing is decided then).
RT_D bool view_verify(const GridP &g, const int32_t *__restrict__ tri_box, bool need, int best_id, float best, f3 r, int vi, int vj, int vk, Walk &v) {
  bool failed = false;
  if (__ballot(need) != 0) {
    int lo0 = 0, lo1 = 0, lo2 = 0, hi0 = -1, hi1 = -1, hi2 = -1;
    if (need) {
      const int32_t *bx = tri_box + 6 * (size_t)best_id;
      lo0 = bx[0]; lo1 = bx[1]; lo2 = bx[2]; hi0 = bx[3]; hi1 = bx[4]; hi2 = bx[5];
      if ((v.nx | v.ny | v.nz) < 0) { need = false; failed = true; }
    }
    const int si = r.x > 0.0f ? 1 : -1, sj = r.y > 0.0f ? 1 : -1, sk = r.z > 0.0f ? 1 : -1;
    for (;;) {
      const float xn = v.tx + v.dtx, yn = v.ty + v.dty, zn = v.tz + v.dtz;
      if (need && best < min3_ref(xn, yn, zn)) {
        need = false;
        failed = !(vi >= lo0 && vi <= hi0 && vj >= lo1 && vj <= hi1 && vk >= lo2 && vk <= hi2);
      }
      if (__ballot(need) == 0) break;
      if (need) {  // walk_step, with the voxel's indices instead of the cell's
        const bool xy = xn < yn;
        const bool sx = xy & (xn < zn), sy = !xy & (yn < zn);
        v.tx = sx ? xn : v.tx; v.ty = sy ? yn : v.ty; v.tz = (sx | sy) ? v.tz : zn;
        v.nx -= sx ? 1 : 0; v.ny -= sy ? 1 : 0; v.nz -= (sx | sy) ? 0 : 1;
        vi += sx ? si : 0; vj += sy ? sj : 0; vk += (sx | sy) ? 0 : sk;
        if ((v.nx | v.ny | v.nz) < 0) { need = false; failed = true; }
      }
    }
    v.cell = cell_index(g, vi, vj, vk);
  }
  return __ballot(failed) == 0;
}
