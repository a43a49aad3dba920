// Plain-old-data records shared by the HIP kernels and the host side of librtgpu.
#pragma once
#include <stdint.h>

namespace rt {

constexpr int kTile = 8;          // image tiles are 8x8 pixels = one wavefront of primary rays
constexpr int kTilePixels = 64;
constexpr int kMaxLevels = 5;     // recursion depth of the reference (src/raytrace.c:390)
constexpr int kPlanarSamples = 16; // src/raytrace.c:242

struct f3 { float x, y, z; };

// One triangle reference of a grid cell, self-contained so that a cell's list is a linear stream: the
// per-triangle constants of the staged filter (rt_math.h: n = ik x ij, ik, q = ik x i for the first slab;
// ij, p = i x ij for the other two; an = i.n for the distance; vertices labelled so that ik is the longest
// edge), the filter's error constants, the triangle index and the transparency kt of its surface (what
// the shadow walk needs, src/voxelize.c:703).  The few triangles the filter cannot decide fetch i, ij, ik by index for tri_test.
struct __attribute__((aligned(16))) TriRec {
  float n[3];
  float k1;          // the triangle's own error constants: k1 here, next to n (what batch_fuzz reads), k2 below
  float ac[3];
  float q[3];
  int32_t id;
  float k2;
  float ab[3];
  float p[3];
  float an;
  float kt;
};
static_assert(sizeof(TriRec) == 80, "TriRec is 80 bytes");
constexpr int kRecFloats = 20;
#ifndef RT_BATCH
#define RT_BATCH 8
#endif
constexpr int kBatch = RT_BATCH; // records per batch of a cell's list: the unit of culling and of LDS staging
static_assert(kBatch == 8 || kBatch == 16 || kBatch == 32 || kBatch == 64, "a wavefront builds the bounds of 64 / kBatch batches at a time");
#ifndef RT_HDR_CHUNK
#define RT_HDR_CHUNK 32
#endif
constexpr int kHdrChunk = RT_HDR_CHUNK;  // batch headers staged in LDS at a time (one lane per header in the bundle test)
static_assert(kHdrChunk == 32 || kHdrChunk == 64, "the masks of a header chunk are 64 bits wide");
constexpr int kHdrFloats = 20;

// Header of one batch (positions kBatch*b .. kBatch*b + kBatch-1 of a cell's list): the axis-aligned box
// (centre c, half extents h) and the sphere (c, radius) that contain every triangle of the batch, both already
// inflated for rounding; the maxima of the records' k1, k2, with k2 negated if any triangle of the batch is
// transparent (kt > 0); the batch's constants ga, gb of the bound of what rounding can add to the box
// (box_test_bundle in rt_kernels.h); and the unit normals of the batch's triangles, three signed 10-bit fractions
// (units of 1/511) in a word each: within 0.0017 of the normal, all zero for a degenerate triangle.  inv_cos is scratch:
// the walk kernels keep the visit's bound of 1 / |cos(ray, normal)| for the batch there, in their LDS copy, and for a
// batch the bundle test lets through they overwrite radius, h and inv_cos of that copy with -+delta, h + W and W for the
// ray-by-ray test, ga with the bundle's distance bound and gb with one bit per record the bundle's record-level test leaves.
struct __attribute__((aligned(16))) BatchHdr {
  float c[3];
  float radius;
  float h[3];
  float inv_cos;
  float k1, k2;
  float ga, gb;
  uint32_t normal[8];
};
static_assert(kBatch == 8, "a batch header holds the normals of 8 records");
static_assert(sizeof(BatchHdr) == 4 * kHdrFloats, "BatchHdr is 80 bytes");

// Uniform grid ("UDD", src/voxelize.h:21-25) plus the grown scene domain.
struct GridP {
  float dmin[3], dmax[3], cs[3];
  int32_t nv[3];
  int32_t ncells;
  const uint32_t *cell_start;  // ncells+1 offsets into cell_tris
  const uint32_t *occ;         // one bit per cell: 1 = its list is not empty
  const int32_t *occ_box;      // lo[3], hi[3]: voxel index box of the non-empty cells (lo > hi: every cell is empty)
  const TriRec *cell_tris;
  const uint32_t *hdr_start;   // ncells+1 offsets into cell_hdrs (one header per kBatch records of a cell)
  const BatchHdr *cell_hdrs;
  const float *geom;           // per triangle i, ij, ik (9 floats)
  const float *filt;           // per triangle kRecFloats floats: a TriRec with the triangle's own k1, k2
  float lmax;                  // longest triangle edge of the scene (upper bound)
  float slack;                 // 0, or +inf to switch the filter and the batch culling off (every test is then the exact one)
};

// A ray waiting for its nearest-hit walk.  i < 0 marks a lane without work:
// -1 = primary ray that never enters the grid (src/raytrace.c:381-382), -2 = padding outside the image.
struct __attribute__((aligned(16))) Ray {
  float o[3];
  float r[3];
  int32_t i, j, k;   // start voxel
  int32_t current;   // triangle the ray leaves (excluded by identity, src/voxelize.c:574), -1 for primary rays
  int32_t pad[2];
};
static_assert(sizeof(Ray) == 48, "Ray is 48 bytes");

// Result of a nearest-hit walk (src/voxelize.c:583-590).
struct __attribute__((aligned(16))) Hit {
  int32_t tri;       // >= 0 triangle, -1 miss, -2 no ray
  float d, u, v;
  int32_t i, j, k;   // voxel the hit was found in: start voxel of the secondary rays (src/raytrace.c:195)
  int32_t pad;
};
static_assert(sizeof(Hit) == 32, "Hit is 32 bytes");

// A shaded hit: everything the light loops of src/raytrace.c:211-296 need.
struct __attribute__((aligned(16))) Node {
  float p[3];        // onew, the hit point
  float n[3];        // shading normal (flipped towards the ray, bump-mapped)
  float c[3];        // nc, surface or texture colour
  int32_t tri;       // >= 0 triangle, -1 miss, -2 no ray
  int32_t refl;      // global node index of the reflected child or -1
  int32_t refr;      // global node index of the refracted child or -1
};
static_assert(sizeof(Node) == 48, "Node is 48 bytes");

// What a shadow query needs of a shaded hit (src/voxelize.c:615-644): the hit point and the triangle it lies on
// (>= 0 triangle, -1 miss, -2 no ray).  The shadow walk reads one of these per (hit, light) instead of the 48-byte node.
struct __attribute__((aligned(16))) SOrigin {
  float p[3];
  int32_t tri;
};
static_assert(sizeof(SOrigin) == 16, "SOrigin is 16 bytes");

struct SurfaceD { float color[3]; float kd, ks, g, ka, kt, eta, kr; };
struct LightD { float p[3]; float flux; float color[3]; };
struct PlanarD { float flux; float color[3]; float a[3], ab[3], ac[3]; };

struct CameraD { float ob[3], ul[3], bl[3], ur[3]; int32_t sw, sh; };

// The primary rays of a chunk of tiles are not stored: the kernels that need one (nearest-hit walk, hit and colour
// kernels of level 0) compute it from the camera and the ray's index in the chunk.  on = 0: rays come from a queue.
struct PrimaryP { CameraD cam; int32_t tile0, ntiles, rank, world, tiles_x, on; };

// Scene-wide constants handed to the shading kernels.
struct ShadeP {
  const SurfaceD *surf;
  const LightD *lights;
  const PlanarD *planar;
  const float *tri_aux;     // per triangle n[3], d
  const int32_t *tri_sid;
  const int32_t *rand_tab;  // glibc rand() stream for the planar-light replay, or nullptr
  int32_t nl, npl, nslots;  // nslots = nl + 16*npl shadow queries per shaded hit
  int32_t tex_tri[4];       // triangles carrying the brick texture (src/preprocess.c:45-71), -1 = none
  float total_flux, distmod;
};

// Light lists (rt_lightlists.h): around every point light the six faces of a cube, a window of each face cut into
// R x R texels, a texel listing the triangles a shadow ray seen from the light in that texel can be blocked by.
// refs[offsets[t] .. offsets[t + 1]) is texel t's list; a reference carries its triangle's whole texel rectangle, so that a
// wavefront that looks at several texels takes each triangle once (at the first texel of the rectangle it looks at).
struct LLRef { int32_t tri; uint32_t rect; };  // rect = a0 | a1 << 8 | b0 << 16 | b1 << 24 (texel columns a0..a1, rows b0..b1)
struct LLFaceD { float p0, q0, sp, sq, p1, q1; uint32_t base, on; };  // column floor((p - p0) sp), row floor((q - q0) sq) at level 0; base: the face's first texel there (base >> 2 l at level l)
struct LLRectD { uint32_t face; uint32_t rect; int32_t tri; };
struct LightListsP {
  const uint32_t *offsets;
  const LLRef *refs;
  const LLFaceD *faces;   // [(light * 6 + face) * 2 + window]
  const int32_t *tri_box; // per triangle the voxel index box lo[3], hi[3] of this frame's grid
  float olim;             // the lists are complete for rays whose origin's norm is at most this
  int32_t R;              // texels per face edge at level 0 (<= 256, a power of two)
  int32_t on;
  int32_t levels;         // 1..4: level l has R >> l texels per face edge; its texels follow those of the levels before it
  uint32_t texels0;       // texels of level 0 (level l: texels0 >> 2 l)
  uint32_t pad[3];
};
// first texel of level l
__host__ __device__ inline uint32_t ll_level_base(uint32_t texels0, int level) {
  return level == 0 ? 0u : level == 1 ? texels0 : level == 2 ? texels0 + (texels0 >> 2) : texels0 + (texels0 >> 2) + (texels0 >> 4);
}

// View lists (rt_viewlists.h): the screen cut into ntx x nty texels of (8 << shift)^2 pixels, a texel listing the triangles
// the primary ray of one of its pixels can hit.  refs[offsets[t] .. offsets[t + 1]) is texel t's list, t = row * ntx + column;
// offsets[ntx * nty] is the number of references the frame's lists need: beyond cap they were not written and are not used.
struct ViewRectD { uint32_t ab, cd; };  // a triangle's texels: columns a0 | a1 << 16, rows b0 | b1 << 16; a0 > a1: none
struct ViewListsP {
  const uint32_t *offsets;
  const int32_t *refs;
  const int32_t *tri_box;  // per triangle the voxel index box lo[3], hi[3] of this frame's grid
  uint32_t cap;
  int32_t ntx, nty, shift, on;
  int32_t pad;  // (not 0: every other ticket is treated as if its walks had not confirmed the lists' answer; for tests)
};

// Device-side control block of one chunk: queue lengths and statistics (the allocation continues with the ticket counters
// of the persistent kernels, kCtlBytes in all).
struct Ctl {
  int32_t count[kMaxLevels + 1];        // rays/nodes per level as produced; what was produced beyond cap[] is dropped
  int32_t cap[kMaxLevels + 1];          // capacity of every level's queue: readers use min(count, cap) (queue_len)
  uint32_t pad[2 * kMaxLevels + 1];     // (the ticket counters of the walk kernels lie behind the block, 4 KB apart: ticket_counters)
  unsigned long long stat[32];
};
struct QueueCaps { int32_t cap[kMaxLevels + 1]; };
enum Stat { kPixels = 0, kPixelsTraced, kNearestQ, kShadowQ, kShadowWalks, kNearS, kNearT, kShadS, kShadT,
            kNearLaneTests, kNearWaveTests, kShaded, kShadLaneTests, kShadWaveTests,
            kNearStage2, kNearStage3, kNearExact, kShadStage2, kShadStage3, kShadExact,
            kNearVisits, kShadVisits, kNearBatches, kNearCulled, kShadBatches, kShadCulled, kNearFull, kShadFull, kShadListTickets, kShadListFallbacks,
            kViewListTickets, kViewListFallbacks, kStatCount };
constexpr int kTotals = 40;  // frame totals: kStatCount counters, then the queue length of every level
static_assert(kStatCount <= 32 && kStatCount + kMaxLevels <= kTotals, "the statistics fit the control block and the totals buffer");

}  // namespace rt
