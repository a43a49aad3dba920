// librtgpu: host side of the C ABI declared in include/rtgpu.h.
// Owns the device copies of the scene, the per-frame grid and the ray/node queues of a chunk, and
// sequences the kernels of rt_kernels.h on one HIP stream.  There is no CPU rendering path here:
// without a HIP device rtgpu_create fails.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rtgpu.h"
#include "rt_kernels.h"
#include "rt_lightlists.h"

using namespace rt;

// Device memory of rtgpu_render_static, the entry point that allocates nothing: the queues of one chunk live in this
// array instead of memory from hipMalloc.  Every GPU the library runs on holds its own copy, at its own address, until
// the process ends.  In host code the array's name is only a handle for hipGetSymbolAddress (rtgpu_create asks once
// per device, with that device current, and keeps the answer in g_scratch_ptr); everything that needs the memory
// (kernel arguments) takes that per-device pointer.
constexpr size_t kScratchBytes = 64ull << 20;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];
constexpr int kMaxDevices = 64;
static void *g_scratch_ptr[kMaxDevices];  // device address of `scratch` on every device a context was created on
// Every context of a GPU shares that GPU's array.  Who enqueued into it last, and an event recorded behind that call: a
// call from another context or on another stream first makes its stream wait for that event, so that calls which
// follow each other on the host do not overlap on the device whichever streams they use.  (Calls issued from several
// host threads at once, and calls recorded under stream capture, are outside of this: see rtgpu_render_static.)
struct ScratchOwner { const void *ctx = nullptr; hipStream_t stream = nullptr; hipEvent_t done = nullptr; bool valid = false; };
static ScratchOwner g_scratch_owner[kMaxDevices];
static std::mutex g_scratch_mutex;

namespace {

struct TimedSpan { hipEvent_t a, b; int what; };
enum What { W_PRE = 0, W_GRID, W_PRIMARY, W_NEAREST, W_HIT, W_SHADOW, W_ACCUM, W_MINMAX, W_COUNT };

}  // namespace

struct rtgpu_ctx {
  int device = 0;
  hipStream_t stream = nullptr, own_stream = nullptr;
  int num_cu = 256;
  std::string msg;
  // options
  bool opt_count = false;
  int opt_timing = 0;  // 0 off, 1 events around every kernel family, 2 around the two walk kernels only
  int64_t chunk_bytes = 8ll << 30;
  int waves_per_cu = 0;  // wavefronts per CU of the persistent walk kernels; 0 = what each is compiled for (its register budget)
  int contexts_on_gpu = 1;  // option "contexts_on_gpu": a hint, how many contexts keep frames in flight on this GPU (waves_of)
  bool opt_pipeline = true;  // frame set-up on its own stream, into the set of buffers the previous frame does not use
  bool filter = true;        // staged triangle filter + batch culling (off: every listed triangle gets the exact test)
  // scene, host side
  bool have_scene = false, have_frame = false;
  int nt = 0, ns = 0, nl = 0, npl = 0;
  std::vector<rtgpu_light> lights;
  std::vector<rtgpu_surface> surfaces;
  float dmin0[3], dmax0[3], gamma = 2.5f, distmod = 2.0f, vcoeff[3] = {0, 0, 0};
  int vmode = 0, texture_present = 0;
  int tex_tri[4] = {-1, -1, -1, -1};
  int fan = 0, levels = 1;
  // scene, device side
  float *d_verts = nullptr, *d_geom = nullptr, *d_aux = nullptr, *d_filt = nullptr;
  int32_t *d_sid = nullptr, *d_box = nullptr, *d_refs = nullptr, *d_rand = nullptr;
  SurfaceD *d_surf = nullptr;
  LightD *d_lights = nullptr;
  PlanarD *d_planar = nullptr;
  uint32_t *d_bucket = nullptr;  // nodes per cell, then positions: the order of a level's shadow work (k_order_*)
  uint32_t *d_counts = nullptr, *d_cell_start = nullptr, *d_nbatch = nullptr /* list of the non-empty cells */, *d_hdr_start = nullptr, *d_occ = nullptr;
  BatchHdr *d_hdrs = nullptr;
  size_t cap_hdrs = 0;
  TriRec *d_recs = nullptr, *d_recs_unsorted = nullptr;
  size_t cap_cells = 0, cap_refs = 0, cap_rand = 0, cap_bucket = 0;
  float lmax = 0.0f;         // longest triangle edge of the uploaded scene
  int64_t known_nrefs = -1;  // reference count of the uploaded scene's grid (it does not depend on the camera)
  int64_t known_nonempty = 0;  // and its number of non-empty cells
  // Two sets of everything the trace kernels read of a frame's set-up (per-camera triangle data, grid), so that
  // rtgpu_frame_begin can build the next frame's set on its own stream while the previous frame's kernels still
  // run: the members above are the set in use, `alt` is the other one.  released[p]: the renders that read the set
  // of parity p are enqueued up to here; built: the set just built is complete.
  struct GridSet {
    float *d_geom = nullptr, *d_aux = nullptr;
    uint32_t *d_cell_start = nullptr, *d_hdr_start = nullptr, *d_occ = nullptr;
    BatchHdr *d_hdrs = nullptr;
      TriRec *d_recs = nullptr;
    int32_t *d_refs = nullptr;
    uint32_t *d_vl_offsets = nullptr;
    int32_t *d_vl_refs = nullptr;
  } alt;
  hipStream_t build_stream = nullptr;
  hipEvent_t ev_built = nullptr, ev_released[2] = {nullptr, nullptr};
  bool released_valid[2] = {false, false};
  int parity = 0;
  // a frame rendered in two halves (rtgpu_render_trace ... rtgpu_render_shade): tiles of the share, -1 = none pending
  int64_t split_tiles = -1;
  int split_rank = 0, split_world = 1;
  int known_nv[3] = {0, 0, 0};
  // frame
  CameraD cam = {};
  GridP grid = {};
  rtgpu_grid_info ginfo = {};
  ShadeP shade = {};
  // chunk queues
  size_t chunk_tiles_cap = 0;
  // Sizing of the secondary-ray queues.  A hit spawns up to `fan` rays, so level L could hold fan^L entries per pixel
  // (31 nodes per pixel at fan 2); scenes spawn far fewer (ulica: 2.9).  fac[L] = entries of level L per pixel the queues
  // are planned for; fac_alloc[L] what the allocated queues were planned for; ratio[L] the largest count[L] / count[0]
  // a chunk of this scene has produced so far (known after a first, small chunk with worst-case queues).
  double fac[kMaxLevels] = {1, 0, 0, 0, 0}, fac_alloc[kMaxLevels] = {0, 0, 0, 0, 0}, ratio[kMaxLevels] = {1, 0, 0, 0, 0};
  bool have_ratio = false;
  int64_t last_chunks = 0, last_rerendered = 0;  // of the last rtgpu_render call
  int queue_margin_permille = 1500;  // option "queue_margin": planned entries = observed ratio x this / 1000 (+ 0.02 per pixel)
  bool adaptive_queues = true;  // option "queue_sizing": 1 = from what the scene produces (one host sync per chunk), 0 = worst case
  size_t cap[kMaxLevels] = {0, 0, 0, 0, 0}, base[kMaxLevels] = {0, 0, 0, 0, 0}, total_nodes = 0;
  Ray *d_rays[kMaxLevels] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  Hit *d_hits = nullptr;  // sized for the largest level
  Node *d_nodes = nullptr;
  SOrigin *d_sorg = nullptr;  // hit point + triangle of every node: what the shadow walk reads
  int opt_record_test = 1;  // option "record_test": the bundle's record-level test in the walk kernels
  bool opt_fuse_hit = true; // option "fuse_hit": the nearest-hit walk shades its own hits where it can
  int opt_bundle_ends = -1;  // option "bundle_ends": the shadow walk's bundles narrow towards their lights; -1: in scenes with secondary rays
  bool mostly_opaque = true;  // fewer than half of the scene's triangles are transparent (kt > 0)
  bool any_specular = true;   // some surface a triangle uses has ks > 0 (k_accumulate's variant)
  float *d_colors = nullptr, *d_ts = nullptr, *d_partial = nullptr;
  PrimaryP prim{};  // the chunk whose level-0 rays the kernels compute themselves (set by chunk_trace, used by chunk_shade)
  int32_t *d_rank = nullptr, *d_size = nullptr;  // planar lights: rand() replay rank of every node, hits per subtree
  uint32_t *d_order = nullptr;  // per level below 0: its nodes by the cell of their hit point (k_order_place)
  int opt_order = 1;  // option "order": 0 queue order everywhere, 1 the levels below 0 by cell, 2 level 0 too
  int opt_order_parts = 8;  // option "order_parts": the cells are cut into this many parts per axis for the order (1, 2, 4 or 8)
  Ctl *d_ctl = nullptr;
  unsigned char *scratch = nullptr;  // this device's copy of the static scratch array (rtgpu_render_static)
  unsigned long long *d_totals = nullptr;  // [kStatCount + kMaxLevels + 1]
  // device buffers of the host-array conveniences (rtgpu_render_host, rtgpu_tonemap_host): kept and grown, not allocated per call
  float *h_rgb_t = nullptr, *h_rgb = nullptr, *h_mm = nullptr;
  int32_t *h_ids_t = nullptr, *h_ids = nullptr;
  uint32_t *h_rgba = nullptr;
  size_t h_cap_tiles = 0, h_cap_px = 0, h_cap_tone = 0;
  float *d_gammas = nullptr;  // the gamma list of rtgpu_tonemap_multi
  size_t cap_gammas = 0;
  unsigned char *h_map = nullptr;  // device copy of a caller's pixel records (rtgpu_render_pixelmap, rtgpu_tonemap_pixelmap)
  size_t h_cap_map = 0;
  // host copy of the uploaded scene's arrays: an upload of the same scene again is recognised and skipped
  std::vector<float> up_verts;
  std::vector<int32_t> up_sid;
  std::vector<rtgpu_planar_light> up_planar;
  size_t cap_nt = 0, cap_ns = 0, cap_nl = 0, cap_npl = 0;  // what the scene's device arrays were allocated for
  // light lists (rt_lightlists.h): built at the first frame of a scene without transparent triangles that has point lights
  bool opt_light_lists = true;  // option "light_lists"
  int opt_ll_levels = 4;        // option "light_list_levels": 1..4 levels of texels, each twice as coarse as the one before
  int opt_ll_texels = 256;      // option "light_list_texels": texels per face edge of the finest level (a power of two, 8..256)
  bool ll_tried = false;        // the uploaded scene's lists were built, or found not to apply
  // option "light_lists_after" (its default from the environment, RTGPU_LIGHT_LISTS_AFTER, else 1): frames of an uploaded
  // scene that are set up before its lists are built.  The rectangles are worked out on the host (temple: 0.14 s), which a
  // caller that renders one frame of a scene, like the reference's command line, would pay for nothing
  int opt_ll_after = 1;
  int frames_of_scene = 0;      // rtgpu_frame_begin calls since the scene was uploaded
  bool all_opaque = false;      // no triangle of the scene is transparent
  LightListsP ll = {};
  uint32_t *d_ll_offsets = nullptr;
  LLRef *d_ll_refs = nullptr;
  LLFaceD *d_ll_faces = nullptr;
  size_t cap_ll_texels = 0, cap_ll_refs = 0, cap_ll_faces = 0;
  uint32_t *d_ll_deferred = nullptr;  // (ticket, light) pairs k_shadow_lists leaves to k_shadow (one of the chunk's queues)
  // view lists (rt_viewlists.h): built with every frame's set-up; offsets and references belong to the frame's set (alt: the other set's)
  bool opt_view_lists = true;  // option "view_lists"
  bool opt_vl_test_fallback = false;  // option "view_list_test_fallback": every other ticket walks after its lookup, as if the walks had not confirmed it; for tests
  int64_t opt_vl_capacity = 0; // option "view_list_capacity": room for that many references instead of the planned number (0); for tests
  int opt_vl_shift = 1;        // option "view_list_texel_shift": a texel is at least 2^shift tiles a side (temple 3840x2160: 2.11, 2.04, 2.03 ms a frame with 0, 1, 2)
  ViewListsP vl = {};
  uint32_t *d_vl_offsets = nullptr, *d_vl_pieces = nullptr, *d_vl_cursor = nullptr, *d_vl_totals = nullptr;
  int32_t *d_vl_refs = nullptr, *d_vl_piece_tri = nullptr;
  ViewRectD *d_vl_rects = nullptr;
  size_t cap_vl_texels = 0, cap_vl_refs = 0, cap_vl_tris = 0, cap_vl_cursor = 0;  // (triangles the rectangles', pieces' and piece map's buffers are for; entries of the cursors)
  bool vl_no_memory = false;  // an allocation for them failed once: this context goes without
  int opt_ll_kernel = 1;  // option "light_list_kernel": the lists' own kernel where it applies: 0 never (inside k_shadow only), 1 where a wavefront gets many tickets, 2 always
  // timing
  std::vector<TimedSpan> spans;
  rtgpu_timings timings = {};
};

namespace {

constexpr int kPartialBlocks = 1024;

int fail(rtgpu_ctx *c, int code, const std::string &m) {
  if (c) c->msg = m;
  return code;
}

#define HIPCHK(ctx, expr)                                                                          \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail(ctx, e_ == hipErrorOutOfMemory ? RTGPU_E_MEMORY : RTGPU_E_NO_DEVICE,             \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                              \
  } while (0)

// A kernel launch whose failure (bad configuration, no code object for this GPU, ...) is reported with the kernel's name.
#define KLAUNCH(ctx, kern, grid, block, st, ...)                                                   \
  do {                                                                                             \
    hipLaunchKernelGGL(kern, grid, block, 0, st, __VA_ARGS__);                                     \
    hipError_t e_ = hipGetLastError();                                                             \
    if (e_ != hipSuccess)                                                                          \
      return fail(ctx, RTGPU_E_NO_DEVICE, std::string("launch of " #kern ": ") + hipGetErrorString(e_)); \
  } while (0)

template <typename T>
int dev_alloc(rtgpu_ctx *c, T **p, size_t n) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  HIPCHK(c, hipMalloc((void **)p, (n ? n : 1) * sizeof(T)));
  return RTGPU_OK;
}
template <typename T>
void dev_free(T **p) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
}

void swap_grid_sets(rtgpu_ctx *c) {
  std::swap(c->d_geom, c->alt.d_geom); std::swap(c->d_aux, c->alt.d_aux);
  std::swap(c->d_cell_start, c->alt.d_cell_start); std::swap(c->d_hdr_start, c->alt.d_hdr_start); std::swap(c->d_occ, c->alt.d_occ);
  std::swap(c->d_hdrs, c->alt.d_hdrs); std::swap(c->d_recs, c->alt.d_recs);
  std::swap(c->d_refs, c->alt.d_refs);
  std::swap(c->d_vl_offsets, c->alt.d_vl_offsets); std::swap(c->d_vl_refs, c->alt.d_vl_refs);
  c->parity ^= 1;
}
// The kernels that read the current set are enqueued up to here.
void mark_released(rtgpu_ctx *c) {
  if (hipEventRecord(c->ev_released[c->parity], c->stream) == hipSuccess) c->released_valid[c->parity] = true;
}

struct Span {  // HIP events around launches, on the stream the kernels go to
  rtgpu_ctx *c; int what; hipStream_t st; hipEvent_t a = nullptr, b = nullptr; bool on;
  Span(rtgpu_ctx *ctx, int w, hipStream_t stream = nullptr) : c(ctx), what(w), st(stream ? stream : ctx->stream) {
    on = c->opt_timing == 1 || (c->opt_timing == 2 && (w == W_NEAREST || w == W_SHADOW));
    if (on) {
      (void)hipEventCreate(&a); (void)hipEventCreate(&b);
      (void)hipEventRecord(a, st);
    }
  }
  ~Span() {
    if (on) { (void)hipEventRecord(b, st); c->spans.push_back({a, b, what}); }
  }
};

void resolve_spans(rtgpu_ctx *c) {
  float acc[W_COUNT] = {0};
  int nn = 0, ns = 0;
  for (auto &s : c->spans) {
    float ms = 0.0f;
    (void)hipEventSynchronize(s.b);
    (void)hipEventElapsedTime(&ms, s.a, s.b);
    acc[s.what] += ms;
    if (s.what == W_NEAREST) nn++;
    if (s.what == W_SHADOW) ns++;
    (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b);
  }
  // (nothing recorded since the previous call: all zero)
  c->timings = {acc[W_PRE], acc[W_GRID], acc[W_PRIMARY], acc[W_NEAREST], acc[W_HIT], acc[W_SHADOW], acc[W_ACCUM], acc[W_MINMAX], nn, ns};
  c->spans.clear();
}

// glibc rand() with the default seed: random_r TYPE_3 (r[i] = r[i-3] + r[i-31], output >> 1) after
// srandom_r(1)'s 310 warm-up draws.  The planar-light sampling of the reference consumes this stream.
void glibc_rand_stream(std::vector<int32_t> &out, size_t n) {
  uint32_t ring[31];
  long word = 1;
  ring[0] = 1;
  for (int i = 1; i < 31; i++) {
    long hi = word / 127773, lo = word % 127773;
    word = 16807 * lo - 2836 * hi;
    if (word < 0) word += 2147483647;
    ring[i] = (uint32_t)word;
  }
  int f = 3, b = 0;
  out.resize(n);
  for (long long q = -310; q < (long long)n; q++) {
    uint32_t v = ring[f] += ring[b];
    if (q >= 0) out[q] = (int32_t)(v >> 1);
    if (++f >= 31) f = 0;
    if (++b >= 31) b = 0;
  }
}

// The order of a level's shadow work (k_order_*): parts per cell and axis that keep the bucket array within 2^25 entries,
// and the buckets that makes (node_cell's keys).
int order_parts_of(int ncells, int most) {
  int parts = most;
  while (parts > 1 && (double)ncells * parts * parts * parts > (double)(1 << 25)) parts >>= 1;
  return parts;
}
size_t order_buckets_of(int ncells, int parts) { return (size_t)ncells * (size_t)(parts * parts * parts) + 1; }
int64_t tiles_x_of(const rtgpu_ctx *c) { return (c->cam.sw + kTile - 1) / kTile; }
int64_t tiles_y_of(const rtgpu_ctx *c) { return (c->cam.sh + kTile - 1) / kTile; }
int64_t tiles_of_rank(int64_t ntiles, int rank, int world) { return rank < ntiles ? (ntiles - rank + world - 1) / world : 0; }

// Queues of one chunk: level L holds at most cap[L] rays/nodes; a node has at most `fan` children.  plan_queues
// sets the capacities for `tiles` tiles and asks give(slot, bytes) for the memory of every queue: ensure_chunk
// allocates it, rtgpu_render_static carves it from the library's static scratch array.
template <typename F>
int plan_queues(rtgpu_ctx *c, size_t tiles, F &&give) {
  size_t total = 0, maxcap = 0;
  for (int L = 0; L < c->levels; L++) {
    c->cap[L] = L == 0 ? tiles * kTilePixels : (size_t)ceil((double)(tiles * kTilePixels) * c->fac[L]) + 64;
    c->base[L] = total;
    total += c->cap[L];
    maxcap = c->cap[L] > maxcap ? c->cap[L] : maxcap;
  }
  c->total_nodes = total;
  int rc;
  for (int L = 0; L < c->levels; L++)
    if ((rc = give((void **)&c->d_rays[L], sizeof(Ray) * (L == 0 ? (size_t)1 : c->cap[L])))) return rc;  // (level 0: primary rays are not stored)
  if ((rc = give((void **)&c->d_hits, sizeof(Hit) * maxcap))) return rc;
  if ((rc = give((void **)&c->d_nodes, sizeof(Node) * total))) return rc;
  if ((rc = give((void **)&c->d_sorg, sizeof(SOrigin) * total))) return rc;
  if ((rc = give((void **)&c->d_colors, sizeof(float) * total * 4))) return rc;
  if ((rc = give((void **)&c->d_ts, sizeof(float) * total * (size_t)(c->shade.nslots > 0 ? c->shade.nslots : 1)))) return rc;
  if ((rc = give((void **)&c->d_partial, sizeof(float) * ((tiles * kTilePixels + 255) / 256 * 6)))) return rc;  // colour range per block of level 0
  const size_t nrank = c->npl > 0 ? total : 1;
  if ((rc = give((void **)&c->d_rank, sizeof(int32_t) * nrank)) || (rc = give((void **)&c->d_size, sizeof(int32_t) * nrank))) return rc;
  if ((rc = give((void **)&c->d_order, sizeof(uint32_t) * (c->levels > 1 || c->opt_order > 1 ? total : 1)))) return rc;
  // the (ticket, light) pairs k_shadow_lists leaves to k_shadow: at most every pair of the chunk
  if ((rc = give((void **)&c->d_ll_deferred, sizeof(uint32_t) * (c->levels == 1 && c->npl == 0 ? tiles * (size_t)(c->nl > 0 ? c->nl : 0) + 1 : 1)))) return rc;
  return RTGPU_OK;
}

// Factors of the worst case (every hit spawns `fan` rays), and of the current knowledge of the scene.
void worst_case_factors(rtgpu_ctx *c) {
  double f = 1.0;
  for (int L = 0; L < kMaxLevels; L++) { c->fac[L] = L < c->levels ? f : 0.0; f *= c->fan; }
}
void observed_factors(rtgpu_ctx *c) {
  double worst = 1.0;
  for (int L = 0; L < kMaxLevels; L++) {
    const double want = L == 0 ? 1.0 : c->ratio[L] * (c->queue_margin_permille * 0.001) + 0.02;  // default: half as much again as the fullest chunk so far
    c->fac[L] = L < c->levels ? (want < worst ? want : worst) : 0.0;
    worst *= c->fan;
  }
}

int ensure_chunk(rtgpu_ctx *c, size_t tiles) {
  bool fits = tiles <= c->chunk_tiles_cap;
  for (int L = 0; L < c->levels && fits; L++)  // (queues planned for more tiles hold more entries per level than asked for here)
    fits = (double)c->chunk_tiles_cap * c->fac_alloc[L] >= (double)tiles * c->fac[L];
  if (fits) return RTGPU_OK;
  c->chunk_tiles_cap = 0;
  int rc = plan_queues(c, tiles, [&](void **slot, size_t bytes) {
    if (*slot) { (void)hipFree(*slot); *slot = nullptr; }
    HIPCHK(c, hipMalloc(slot, bytes ? bytes : 1));
    return (int)RTGPU_OK;
  });
  if (rc) return rc;
  c->chunk_tiles_cap = tiles;
  for (int L = 0; L < kMaxLevels; L++) c->fac_alloc[L] = c->fac[L];
  return RTGPU_OK;
}

// The chunk queues of a context, as a value: rtgpu_render_static puts the carved ones in and the context's own back.
struct QueueSet {
  size_t chunk_tiles_cap, cap[kMaxLevels], base[kMaxLevels], total_nodes;
  Ray *d_rays[kMaxLevels]; Hit *d_hits; Node *d_nodes; SOrigin *d_sorg; float *d_colors, *d_ts, *d_partial; int32_t *d_rank, *d_size; uint32_t *d_order, *d_ll_deferred;
};
QueueSet take_queues(rtgpu_ctx *c) {
  QueueSet q;
  q.chunk_tiles_cap = c->chunk_tiles_cap; q.total_nodes = c->total_nodes;
  for (int L = 0; L < kMaxLevels; L++) { q.cap[L] = c->cap[L]; q.base[L] = c->base[L]; q.d_rays[L] = c->d_rays[L]; c->d_rays[L] = nullptr; }
  q.d_hits = c->d_hits; q.d_nodes = c->d_nodes; q.d_sorg = c->d_sorg; q.d_colors = c->d_colors; q.d_ts = c->d_ts; q.d_partial = c->d_partial;
  q.d_rank = c->d_rank; q.d_size = c->d_size; q.d_order = c->d_order; c->d_order = nullptr;
  q.d_ll_deferred = c->d_ll_deferred; c->d_ll_deferred = nullptr;
  c->d_hits = nullptr; c->d_nodes = nullptr; c->d_sorg = nullptr; c->d_colors = c->d_ts = c->d_partial = nullptr; c->d_rank = c->d_size = nullptr;
  return q;
}
void put_queues(rtgpu_ctx *c, const QueueSet &q) {
  c->chunk_tiles_cap = q.chunk_tiles_cap; c->total_nodes = q.total_nodes;
  for (int L = 0; L < kMaxLevels; L++) { c->cap[L] = q.cap[L]; c->base[L] = q.base[L]; c->d_rays[L] = q.d_rays[L]; }
  c->d_hits = q.d_hits; c->d_nodes = q.d_nodes; c->d_sorg = q.d_sorg; c->d_colors = q.d_colors; c->d_ts = q.d_ts; c->d_partial = q.d_partial;
  c->d_rank = q.d_rank; c->d_size = q.d_size; c->d_order = q.d_order; c->d_ll_deferred = q.d_ll_deferred;
}

// The glibc rand() stream from seed 1, at least `need` values, on the device.
int ensure_rand(rtgpu_ctx *c, size_t need) {
  if (need <= c->cap_rand) return RTGPU_OK;
  std::vector<int32_t> tab;
  glibc_rand_stream(tab, need);
  int rc;
  if ((rc = dev_alloc(c, &c->d_rand, need))) return rc;
  HIPCHK(c, hipMemcpy(c->d_rand, tab.data(), need * sizeof(int32_t), hipMemcpyHostToDevice));
  c->cap_rand = need;
  c->shade.rand_tab = c->d_rand;
  return RTGPU_OK;
}

size_t bytes_per_tile(const rtgpu_ctx *c) {
  double nodes = 0, maxcap = 0;
  for (int L = 0; L < c->levels; L++) {
    const double capL = kTilePixels * c->fac[L] + (L > 0 ? 1.0 : 0.0);
    nodes += capL;
    maxcap = capL > maxcap ? capL : maxcap;
  }
  const double b = nodes * (double)(sizeof(Ray) + sizeof(Node) + sizeof(SOrigin) + 16 + 4 + 4 * (size_t)(c->shade.nslots > 0 ? c->shade.nslots : 1)) -
                   (double)(kTilePixels * sizeof(Ray)) + maxcap * (double)sizeof(Hit) + kTilePixels * 4.0;  // (no ray queue for level 0)
  return (size_t)ceil(b);
}

}  // namespace

extern "C" {

const char *rtgpu_error_string(int code) {
  switch (code) {
    case RTGPU_OK: return "ok";
    case RTGPU_E_IO: return "unable to read or write file";
    case RTGPU_E_MEMORY: return "unable to allocate memory";
    case RTGPU_E_NOT_ENOUGH_SURFACES: return "not enough surfaces to cover entire scene";
    case RTGPU_E_INVALID_PARAM: return "parameter has invalid value";
    case RTGPU_E_NO_DEVICE: return "no usable HIP device / HIP runtime error";
    case RTGPU_E_STATE: return "call order violated";
    case RTGPU_E_UNSUPPORTED: return "unsupported combination";
    default: return "unknown error";
  }
}

const char *rtgpu_last_message(const rtgpu_ctx *ctx) { return ctx ? ctx->msg.c_str() : ""; }

int rtgpu_create(int device, rtgpu_ctx **out) {
  if (!out) return RTGPU_E_INVALID_PARAM;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return RTGPU_E_NO_DEVICE;
  rtgpu_ctx *c = new rtgpu_ctx();
  c->device = device;
  if (const char *e = getenv("RTGPU_LIGHT_LISTS_AFTER")) { const long v = strtol(e, nullptr, 10); if (v >= 0 && v <= 1000000) c->opt_ll_after = (int)v; }
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return RTGPU_E_NO_DEVICE;
  }
  c->stream = c->own_stream;
  if (hipStreamCreateWithFlags(&c->build_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_built, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_released[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_released[1], hipEventDisableTiming) != hipSuccess) {
    rtgpu_destroy(c);
    return RTGPU_E_NO_DEVICE;
  }
  // this device's address of the static scratch array, asked once per device while that device is current
  if (device < kMaxDevices) {
    if (!g_scratch_ptr[device]) {
      void *p = nullptr;
      if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) { rtgpu_destroy(c); return RTGPU_E_NO_DEVICE; }
      g_scratch_ptr[device] = p;
    }
    {
      std::lock_guard<std::mutex> lock(g_scratch_mutex);
      if (!g_scratch_owner[device].done && hipEventCreateWithFlags(&g_scratch_owner[device].done, hipEventDisableTiming) != hipSuccess) {
        g_scratch_owner[device].done = nullptr;
        rtgpu_destroy(c);
        return RTGPU_E_NO_DEVICE;
      }
    }
    c->scratch = static_cast<unsigned char *>(g_scratch_ptr[device]);
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->num_cu = prop.multiProcessorCount;
  if (hipMalloc((void **)&c->d_ctl, kCtlBytes) != hipSuccess ||
      hipMalloc((void **)&c->d_totals, sizeof(unsigned long long) * kTotals) != hipSuccess ||
      hipMalloc((void **)&c->d_partial, sizeof(float) * 6 * kPartialBlocks) != hipSuccess) {
    rtgpu_destroy(c);
    return RTGPU_E_MEMORY;
  }
  memset(&c->timings, 0, sizeof c->timings);
  memset(&c->ginfo, 0, sizeof c->ginfo);
  *out = c;
  return RTGPU_OK;
}

void rtgpu_destroy(rtgpu_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->build_stream) (void)hipStreamSynchronize(c->build_stream);
  resolve_spans(c);
  dev_free(&c->alt.d_geom); dev_free(&c->alt.d_aux); dev_free(&c->alt.d_cell_start); dev_free(&c->alt.d_hdr_start); dev_free(&c->alt.d_occ);
  dev_free(&c->alt.d_hdrs); dev_free(&c->alt.d_recs); dev_free(&c->alt.d_refs);
  dev_free(&c->alt.d_vl_offsets); dev_free(&c->alt.d_vl_refs); dev_free(&c->d_vl_offsets); dev_free(&c->d_vl_refs);
  dev_free(&c->d_vl_pieces); dev_free(&c->d_vl_cursor); dev_free(&c->d_vl_totals); dev_free(&c->d_vl_rects); dev_free(&c->d_vl_piece_tri);
  if (c->ev_built) (void)hipEventDestroy(c->ev_built);
  for (int k = 0; k < 2; k++) if (c->ev_released[k]) (void)hipEventDestroy(c->ev_released[k]);
  if (c->build_stream) (void)hipStreamDestroy(c->build_stream);
  dev_free(&c->d_verts); dev_free(&c->d_geom); dev_free(&c->d_aux); dev_free(&c->d_filt); dev_free(&c->d_sid); dev_free(&c->d_box);
  dev_free(&c->d_refs); dev_free(&c->d_rand); dev_free(&c->d_surf); dev_free(&c->d_lights); dev_free(&c->d_planar);
  dev_free(&c->d_counts); dev_free(&c->d_cell_start); dev_free(&c->d_recs); dev_free(&c->d_recs_unsorted);
  dev_free(&c->d_nbatch); dev_free(&c->d_hdr_start); dev_free(&c->d_occ); dev_free(&c->d_hdrs); dev_free(&c->d_bucket);
  for (int L = 0; L < kMaxLevels; L++) dev_free(&c->d_rays[L]);
  dev_free(&c->d_hits); dev_free(&c->d_nodes); dev_free(&c->d_sorg); dev_free(&c->d_colors); dev_free(&c->d_ts); dev_free(&c->d_partial);
  dev_free(&c->h_rgb_t); dev_free(&c->h_rgb); dev_free(&c->h_mm); dev_free(&c->h_ids_t); dev_free(&c->h_ids); dev_free(&c->h_rgba); dev_free(&c->d_gammas); dev_free(&c->h_map);
  dev_free(&c->d_rank); dev_free(&c->d_size); dev_free(&c->d_order); dev_free(&c->d_ctl); dev_free(&c->d_totals);
  dev_free(&c->d_ll_offsets); dev_free(&c->d_ll_refs); dev_free(&c->d_ll_faces); dev_free(&c->d_ll_deferred);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

int rtgpu_set_stream(rtgpu_ctx *c, void *hip_stream) {
  if (!c) return RTGPU_E_INVALID_PARAM;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
  return RTGPU_OK;
}

int rtgpu_set_null_stream(rtgpu_ctx *c) {
  if (!c) return RTGPU_E_INVALID_PARAM;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->stream = nullptr;  // every launch, event and wait below takes the null stream as it is
  return RTGPU_OK;
}

int rtgpu_set_option(rtgpu_ctx *c, const char *name, int64_t value) {
  if (!c || !name) return RTGPU_E_INVALID_PARAM;
  if (!strcmp(name, "count")) c->opt_count = value != 0;
  else if (!strcmp(name, "timing")) { if (value < 0 || value > 2) return fail(c, RTGPU_E_INVALID_PARAM, "timing is 0, 1 or 2"); c->opt_timing = (int)value; }
  else if (!strcmp(name, "chunk_bytes")) { if (value < (1 << 20)) return fail(c, RTGPU_E_INVALID_PARAM, "chunk_bytes < 1 MiB"); c->chunk_bytes = value; c->chunk_tiles_cap = 0; }
  else if (!strcmp(name, "filter")) c->filter = value != 0;
  else if (!strcmp(name, "record_test")) { if (value < -1 || value > 1) return fail(c, RTGPU_E_INVALID_PARAM, "record_test: -1 (by the scene), 0 or 1"); c->opt_record_test = (int)value; }
  else if (!strcmp(name, "fuse_hit")) c->opt_fuse_hit = value != 0;
  else if (!strcmp(name, "bundle_ends")) { if (value < -1 || value > 1) return fail(c, RTGPU_E_INVALID_PARAM, "bundle_ends: -1 (by the scene), 0 or 1"); c->opt_bundle_ends = (int)value; }
  else if (!strcmp(name, "order")) { if (value < 0 || value > 2) return fail(c, RTGPU_E_INVALID_PARAM, "order: 0, 1 or 2"); if ((int)value != c->opt_order) c->chunk_tiles_cap = 0; c->opt_order = (int)value; }
  else if (!strcmp(name, "order_parts")) { if (value != 1 && value != 2 && value != 4 && value != 8) return fail(c, RTGPU_E_INVALID_PARAM, "order_parts: 1, 2, 4 or 8"); c->opt_order_parts = (int)value; }
  else if (!strcmp(name, "queue_margin")) { if (value < 1 || value > 100000) return fail(c, RTGPU_E_INVALID_PARAM, "queue_margin out of range"); c->queue_margin_permille = (int)value; }
  else if (!strcmp(name, "queue_sizing")) { c->adaptive_queues = value != 0; c->have_ratio = false; }
  else if (!strcmp(name, "pipeline")) c->opt_pipeline = value != 0;
  else if (!strcmp(name, "light_lists")) c->opt_light_lists = value != 0;
  else if (!strcmp(name, "light_lists_after")) { if (value < 0 || value > 1000000) return fail(c, RTGPU_E_INVALID_PARAM, "light_lists_after: 0 .. 1000000 frames"); c->opt_ll_after = (int)value; }
  else if (!strcmp(name, "view_lists")) c->opt_view_lists = value != 0;
  else if (!strcmp(name, "view_list_test_fallback")) c->opt_vl_test_fallback = value != 0;
  else if (!strcmp(name, "view_list_capacity")) { if (value < 0 || value > (1ll << 30)) return fail(c, RTGPU_E_INVALID_PARAM, "view_list_capacity: 0 (planned) .. 2^30"); c->opt_vl_capacity = value; }
  else if (!strcmp(name, "view_list_texel_shift")) { if (value < 0 || value > 4) return fail(c, RTGPU_E_INVALID_PARAM, "view_list_texel_shift: 0..4"); c->opt_vl_shift = (int)value; }
  else if (!strcmp(name, "light_list_kernel")) { if (value < 0 || value > 2) return fail(c, RTGPU_E_INVALID_PARAM, "light_list_kernel: 0, 1 or 2"); c->opt_ll_kernel = (int)value; }
  else if (!strcmp(name, "light_list_levels")) {
    if (value < 1 || value > 4) return fail(c, RTGPU_E_INVALID_PARAM, "light_list_levels: 1..4");
    if (c->opt_ll_levels != (int)value) { c->opt_ll_levels = (int)value; c->ll_tried = false; c->ll.on = 0; }
  }
  else if (!strcmp(name, "light_list_texels")) {
    if (value < 8 || value > 256 || (value & (value - 1)) != 0) return fail(c, RTGPU_E_INVALID_PARAM, "light_list_texels: a power of two, 8..256");
    if (c->opt_ll_texels != (int)value) { c->opt_ll_texels = (int)value; c->ll_tried = false; c->ll.on = 0; }
  }
  else if (!strcmp(name, "contexts_on_gpu")) { if (value < 1 || value > 16) return fail(c, RTGPU_E_INVALID_PARAM, "contexts_on_gpu: 1..16"); c->contexts_on_gpu = (int)value; }
  else if (!strcmp(name, "waves_per_cu")) { if (value < 0 || value > 64) return fail(c, RTGPU_E_INVALID_PARAM, "waves_per_cu out of range"); c->waves_per_cu = (int)value; }
  else return fail(c, RTGPU_E_INVALID_PARAM, std::string("unknown option ") + name);
  return RTGPU_OK;
}

int rtgpu_scene_upload(rtgpu_ctx *c, const rtgpu_scene_desc *s) {
  if (!c || !s || s->nt <= 0 || s->ns <= 0 || !s->vertices || !s->sid || !s->surfaces || s->nl < 0 || s->npl < 0 ||
      (s->nl > 0 && !s->lights) || (s->npl > 0 && !s->planar))
    return fail(c, RTGPU_E_INVALID_PARAM, "scene description incomplete");
  HIPCHK(c, hipSetDevice(c->device));
  // The scene that is already there, again (a host that calls the reference's frame function once per frame hands the
  // same scene over every time): nothing to do, and what is known about it (grid sizes, queue sizes) stays known.  The
  // same arrays with another domain or configuration (the reference's frame function grows RT_Scene.dmin / dmax at every
  // call, src/raytrace.c:350-353, src/voxelize.c:140-141, and its drop-in hands that back): only the numbers
  // rtgpu_frame_begin works from change, and the size of the next grid has to be read back once.
  if (c->have_scene && c->nt == s->nt && c->ns == s->ns && c->nl == s->nl && c->npl == s->npl && c->texture_present == s->texture_present &&
      !memcmp(c->up_verts.data(), s->vertices, sizeof(float) * 9 * (size_t)s->nt) && !memcmp(c->up_sid.data(), s->sid, sizeof(int32_t) * (size_t)s->nt) &&
      !memcmp(c->surfaces.data(), s->surfaces, sizeof(rtgpu_surface) * (size_t)s->ns) &&
      (s->nl == 0 || !memcmp(c->lights.data(), s->lights, sizeof(rtgpu_light) * (size_t)s->nl)) &&
      (s->npl == 0 || !memcmp(c->up_planar.data(), s->planar, sizeof(rtgpu_planar_light) * (size_t)s->npl))) {
    const bool same_numbers = !memcmp(c->dmin0, s->dmin, sizeof c->dmin0) && !memcmp(c->dmax0, s->dmax, sizeof c->dmax0) &&
                              !memcmp(&c->gamma, &s->gamma, sizeof(float)) && !memcmp(&c->distmod, &s->distmod, sizeof(float)) &&
                              c->vmode == s->vmode && !memcmp(c->vcoeff, s->vcoeff, sizeof c->vcoeff);
    if (!same_numbers) {
      memcpy(c->dmin0, s->dmin, sizeof c->dmin0); memcpy(c->dmax0, s->dmax, sizeof c->dmax0);
      c->gamma = s->gamma; c->distmod = s->distmod; c->vmode = s->vmode;
      memcpy(c->vcoeff, s->vcoeff, sizeof c->vcoeff);
      c->known_nrefs = -1;
      c->have_frame = false;
    }
    return RTGPU_OK;
  }
  for (int t = 0; t < s->nt; t++)
    if (s->sid[t] < 0 || s->sid[t] >= s->ns) return fail(c, RTGPU_E_NOT_ENOUGH_SURFACES, "surface id out of range");
  c->have_scene = c->have_frame = false;
  c->known_nrefs = -1;
  c->nt = s->nt; c->ns = s->ns; c->nl = s->nl; c->npl = s->npl;
  c->lights.assign(s->lights, s->lights + s->nl);
  c->surfaces.assign(s->surfaces, s->surfaces + s->ns);
  c->up_verts.assign(s->vertices, s->vertices + 9 * (size_t)s->nt);
  c->up_sid.assign(s->sid, s->sid + s->nt);
  c->up_planar.assign(s->planar, s->planar + s->npl);
  memcpy(c->dmin0, s->dmin, sizeof c->dmin0); memcpy(c->dmax0, s->dmax, sizeof c->dmax0);
  c->gamma = s->gamma; c->distmod = s->distmod; c->vmode = s->vmode;
  memcpy(c->vcoeff, s->vcoeff, sizeof c->vcoeff);
  c->texture_present = s->texture_present;
  // The texture goes to the 1st, 2nd, 7th and 8th triangle whose surface id is 7 (preprocess.c:41-79).
  int seen = 0;
  for (int q = 0; q < 4; q++) c->tex_tri[q] = -1;
  for (int t = 0; t < s->nt; t++) {
    if (s->sid[t] == 7) {
      int slot = seen == 0 ? 0 : seen == 1 ? 1 : seen == 6 ? 2 : seen == 7 ? 3 : -1;
      if (slot >= 0 && s->texture_present) c->tex_tri[slot] = t;
      seen++;
    }
  }
  // longest triangle edge of the scene, from the float edges the kernels compute with (the filter's distance bound)
  double lmax = 0.0;
  for (int t = 0; t < s->nt; t++) {
    const float *v = s->vertices + 9 * (size_t)t;
    double e[3] = {0.0, 0.0, 0.0};
    for (int k = 0; k < 3; k++) {
      const double ab = (double)(float)(v[3 + k] - v[k]), ac = (double)(float)(v[6 + k] - v[k]);
      e[0] += ab * ab; e[1] += ac * ac; e[2] += (ac - ab) * (ac - ab);
    }
    for (int k = 0; k < 3; k++) if (e[k] > lmax) lmax = e[k];  // a NaN edge never wins; NaN triangles end in the exact test anyway
  }
  c->lmax = (float)(sqrt(lmax) * 1.001);
  // Secondary rays: a hit spawns a reflected ray if kr > 0 and a refracted one if kt > 0 (raytrace.c:193-204).
  // Only surfaces some triangle uses can spawn anything.
  std::vector<char> used(s->ns, 0);
  for (int t = 0; t < s->nt; t++) used[s->sid[t]] = 1;
  c->fan = 0;
  for (int q = 0; q < s->ns; q++) {
    if (!used[q]) continue;
    int f = (s->surfaces[q].kr > 0.0f) + (s->surfaces[q].kt > 0.0f);
    c->fan = f > c->fan ? f : c->fan;
  }
  c->levels = c->fan > 0 ? kMaxLevels : 1;
  c->any_specular = false;
  for (int q = 0; q < s->ns; q++) if (used[q] && s->surfaces[q].ks > 0.0f) c->any_specular = true;
  // Scenes of mostly transparent triangles keep their cells' lists in ascending order (the transparency product is
  // taken in list order) and their shadow rays leave scattered points: bundles are wide against the triangles there.
  size_t transparent = 0;
  for (int t = 0; t < s->nt; t++) transparent += s->surfaces[s->sid[t]].kt > 0.0f ? 1 : 0;
  c->mostly_opaque = 2 * transparent < (size_t)s->nt;
  c->all_opaque = transparent == 0;
  c->ll_tried = false;
  c->ll.on = 0;
  c->frames_of_scene = 0;
  c->chunk_tiles_cap = 0;
  c->have_ratio = false;
  for (int L = 0; L < kMaxLevels; L++) c->ratio[L] = L == 0 ? 1.0 : 0.0;
  worst_case_factors(c);
  int rc;
  // the kernels of earlier frames may still read what is replaced below
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipStreamSynchronize(c->build_stream));
  c->released_valid[0] = c->released_valid[1] = false;
  // the scene's device arrays are kept from upload to upload and only grow
  const size_t nt = (size_t)s->nt;
  if (nt > c->cap_nt) {
    c->cap_nt = 0;
    if ((rc = dev_alloc(c, &c->d_verts, nt * 9)) || (rc = dev_alloc(c, &c->d_geom, nt * 9)) ||
        (rc = dev_alloc(c, &c->d_aux, nt * 4)) || (rc = dev_alloc(c, &c->d_filt, nt * kRecFloats)) || (rc = dev_alloc(c, &c->d_sid, nt)) ||
        (rc = dev_alloc(c, &c->d_box, nt * 6 + (nt + 63) / 64 * 6)) ||
        (rc = dev_alloc(c, &c->alt.d_geom, nt * 9)) || (rc = dev_alloc(c, &c->alt.d_aux, nt * 4)))
      return rc;
    c->cap_nt = nt;
  }
  if ((size_t)s->ns > c->cap_ns) { c->cap_ns = 0; if ((rc = dev_alloc(c, &c->d_surf, (size_t)s->ns))) return rc; c->cap_ns = (size_t)s->ns; }
  if ((size_t)s->nl > c->cap_nl || !c->d_lights) { c->cap_nl = 0; if ((rc = dev_alloc(c, &c->d_lights, (size_t)s->nl))) return rc; c->cap_nl = (size_t)s->nl; }
  if ((size_t)s->npl > c->cap_npl || !c->d_planar) { c->cap_npl = 0; if ((rc = dev_alloc(c, &c->d_planar, (size_t)s->npl))) return rc; c->cap_npl = (size_t)s->npl; }
  static_assert(sizeof(rtgpu_surface) == sizeof(SurfaceD) && sizeof(rtgpu_light) == sizeof(LightD) &&
                    sizeof(rtgpu_planar_light) == sizeof(PlanarD) && sizeof(rtgpu_camera) == sizeof(CameraD),
                "ABI structs and device structs share one layout");
  HIPCHK(c, hipMemcpyAsync(c->d_verts, s->vertices, sizeof(float) * 9 * s->nt, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_sid, s->sid, sizeof(int32_t) * s->nt, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_surf, s->surfaces, sizeof(SurfaceD) * s->ns, hipMemcpyHostToDevice, c->stream));
  if (s->nl) HIPCHK(c, hipMemcpyAsync(c->d_lights, s->lights, sizeof(LightD) * s->nl, hipMemcpyHostToDevice, c->stream));
  if (s->npl) HIPCHK(c, hipMemcpyAsync(c->d_planar, s->planar, sizeof(PlanarD) * s->npl, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->have_scene = true;
  return RTGPU_OK;
}

namespace {
// The light lists of the uploaded scene (rt_lightlists.h), once per scene: the rectangles on the host, the lists on the
// device.  A scene they do not apply to (a transparent triangle, no point light), or whose lists would be larger than
// kLLMaxRefs references even with coarser texels, goes without: its shadow queries all take the walk.
constexpr uint64_t kLLMaxRefs = 128ull << 20;
constexpr uint64_t kLLMaxPairs = 1ull << 20;
int build_light_lists(rtgpu_ctx *c) {
  if (c->ll_tried) return RTGPU_OK;
  c->ll_tried = true;
  c->ll.on = 0;
  if (!c->all_opaque || c->nl <= 0 || c->nt <= 0) return RTGPU_OK;
  // (the rectangles are worked out on the host, about a microsecond per triangle and light: a scene of very many of both
  // keeps its walks rather than wait seconds for its first frame)
  if ((uint64_t)c->nt * (uint64_t)c->nl > kLLMaxPairs) return RTGPU_OK;
  std::vector<float> lp(3 * (size_t)c->nl);
  for (int k = 0; k < c->nl; k++) for (int q = 0; q < 3; q++) lp[3 * (size_t)k + q] = c->lights[k].p[q];
  LightListsHost h;
  int R = c->opt_ll_texels;
  for (;;) {
    build_light_rects(c->nt, c->up_verts.data(), c->nl, lp.data(), R, 0.0f, h);
    if (h.refs <= kLLMaxRefs || R <= 32) break;
    R /= 2;
  }
  if (h.refs > kLLMaxRefs || h.rects.empty() || h.refs == 0) return RTGPU_OK;
  // levels: texels of 2^l x 2^l texels of level 0, down to 8 per face edge, four levels at most
  size_t ntex0 = 0;
  for (const LLFaceH &F : h.faces) if (F.on) ntex0 += (size_t)R * R;
  int levels = 1;
  const int max_levels = c->opt_ll_levels;
  while (levels < max_levels && (R >> levels) >= 8) levels++;
  uint32_t level_base[4] = {0, 0, 0, 0};
  size_t ntexels = 0;
  uint64_t refs_all = 0;
  for (int l = 0; l < levels; l++) {
    level_base[l] = ll_level_base((uint32_t)ntex0, l);
    ntexels += ntex0 >> (2 * l);
    for (const LLRect &r : h.rects) {
      const int a0 = (int)(r.rect & 255u) >> l, a1 = (int)((r.rect >> 8) & 255u) >> l, b0 = (int)((r.rect >> 16) & 255u) >> l, b1 = (int)(r.rect >> 24) >> l;
      refs_all += (uint64_t)(a1 - a0 + 1) * (uint64_t)(b1 - b0 + 1);
    }
  }
  if (refs_all > 2 * kLLMaxRefs) return RTGPU_OK;
  const int nscan = (int)((ntexels + kScanTile - 1) / kScanTile);
  int rc;
  if (ntexels + 1 + (size_t)nscan + 2 > c->cap_ll_texels) {
    c->cap_ll_texels = 0;
    if ((rc = dev_alloc(c, &c->d_ll_offsets, ntexels + 1 + (size_t)nscan + 2))) return rc;
    c->cap_ll_texels = ntexels + 1 + (size_t)nscan + 2;
  }
  if ((size_t)refs_all > c->cap_ll_refs) {
    c->cap_ll_refs = 0;
    if ((rc = dev_alloc(c, &c->d_ll_refs, (size_t)refs_all))) return rc;
    c->cap_ll_refs = (size_t)refs_all;
  }
  if (h.faces.size() > c->cap_ll_faces) {
    c->cap_ll_faces = 0;
    if ((rc = dev_alloc(c, &c->d_ll_faces, h.faces.size()))) return rc;
    c->cap_ll_faces = h.faces.size();
  }
  static_assert(sizeof(LLRect) == sizeof(LLRectD) && sizeof(LLFaceH) == sizeof(LLFaceD), "host and device forms share one layout");
  LLRectD *d_rects = nullptr;
  uint32_t *d_cursor = nullptr;
  if ((rc = dev_alloc(c, &d_rects, h.rects.size())) || (rc = dev_alloc(c, &d_cursor, ntexels))) { dev_free(&d_rects); dev_free(&d_cursor); return rc; }
  uint32_t *totals = c->d_ll_offsets + ntexels + 1;
  const int nr = (int)h.rects.size();
  auto build = [&]() -> int {
    HIPCHK(c, hipMemsetAsync(c->d_ll_offsets, 0, sizeof(uint32_t) * (ntexels + 1), c->stream));
    HIPCHK(c, hipMemcpyAsync(d_rects, h.rects.data(), sizeof(LLRectD) * h.rects.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_ll_faces, h.faces.data(), sizeof(LLFaceD) * h.faces.size(), hipMemcpyHostToDevice, c->stream));
    const int wblocks = (int)(((size_t)nr * 64 + 255) / 256);
    for (int l = 0; l < levels; l++)
      KLAUNCH(c, k_ll_count, dim3(wblocks), dim3(256), c->stream, (const LLRectD *)d_rects, nr, (const LLFaceD *)c->d_ll_faces, R, l, level_base[l], c->d_ll_offsets);
    KLAUNCH(c, k_order_scan, dim3(nscan), dim3(1024), c->stream, c->d_ll_offsets, (int)ntexels, totals);
    KLAUNCH(c, k_order_scan_totals, dim3(1), dim3(1024), c->stream, totals, nscan);
    KLAUNCH(c, k_ll_starts, dim3((unsigned)((ntexels + 1 + 255) / 256)), dim3(256), c->stream, c->d_ll_offsets, (const uint32_t *)totals, (int)ntexels,
            (uint32_t)refs_all, d_cursor);
    for (int l = 0; l < levels; l++)
      KLAUNCH(c, k_ll_fill, dim3(wblocks), dim3(256), c->stream, (const LLRectD *)d_rects, nr, (const LLFaceD *)c->d_ll_faces, R, l, level_base[l], d_cursor, c->d_ll_refs);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RTGPU_OK;
  };
  rc = build();
  dev_free(&d_rects); dev_free(&d_cursor);
  if (rc) return rc;
  c->ll.offsets = c->d_ll_offsets; c->ll.refs = c->d_ll_refs; c->ll.faces = c->d_ll_faces; c->ll.tri_box = c->d_box;
  c->ll.olim = h.olim; c->ll.R = R; c->ll.on = 1; c->ll.levels = levels;
  c->ll.texels0 = (uint32_t)ntex0;
  return RTGPU_OK;
}
// The lists are an accelerator, not a need: if they cannot be built (no memory for them), the scene keeps its walks.
int ensure_light_lists(rtgpu_ctx *c) {
  if (c->ll_tried) return RTGPU_OK;
  const int rc = build_light_lists(c);
  if (rc == RTGPU_E_MEMORY) {
    (void)hipGetLastError();
    dev_free(&c->d_ll_offsets); dev_free(&c->d_ll_refs); dev_free(&c->d_ll_faces);
    c->cap_ll_texels = c->cap_ll_refs = c->cap_ll_faces = 0;
    c->ll.on = 0;
    return RTGPU_OK;
  }
  return rc;
}
// The view lists of the frame being set up (rt_viewlists.h), on the stream bs, into the set of buffers swap_grid_sets has
// just made current.  The screen's texels: one per tile of 8 x 8 pixels, or 2^shift tiles a side where that would be more
// than 2^18 texels.  The references get a buffer planned from the scene and the screen (64 per texel and 16 per triangle,
// a million at least); a frame that needs more goes without (the kernels see the number the frame needs next to the
// buffer's size).  Buffers that cannot be had: this context keeps its walks.
int build_view_lists(rtgpu_ctx *c, hipStream_t bs) {
  c->vl.on = 0;
  if (!c->opt_view_lists || c->vl_no_memory || c->nt <= 0) return RTGPU_OK;
  const int tiles_x = (c->cam.sw + kTile - 1) / kTile, tiles_y = (c->cam.sh + kTile - 1) / kTile;
  int shift = c->opt_vl_shift;
  while ((int64_t)((tiles_x >> shift) + 1) * ((tiles_y >> shift) + 1) > (1 << 18) && shift < 12) shift++;
  const int ntx = ((tiles_x - 1) >> shift) + 1, nty = ((tiles_y - 1) >> shift) + 1;
  if (ntx > 65535 || nty > 65535 || (int64_t)ntx * nty > (1 << 18)) return RTGPU_OK;
  const size_t ntexels = (size_t)ntx * nty;
  // (the pieces of all rectangles are numbered in 32 bits: a scene of so many triangles that they could not be keeps its walks)
  if ((uint64_t)c->nt * ((ntexels + 63) / 64 + 1) >= (1ull << 32)) return RTGPU_OK;
  const size_t want_refs = std::max<size_t>((size_t)1 << 20, 64 * ntexels + 16 * (size_t)c->nt);
  const int nscan_t = (int)((ntexels + 1 + kScanTile - 1) / kScanTile), nscan_p = (int)(((size_t)c->nt + 1 + kScanTile - 1) / kScanTile);
  const size_t want_cursor = std::max(ntexels, (size_t)c->nt) + 2;  // (the cursors of the two scans: pieces per triangle, references per texel)
  auto grow = [&]() -> int {
    int rc;
    if (ntexels + 2 > c->cap_vl_texels) {
      c->cap_vl_texels = 0;
      if ((rc = dev_alloc(c, &c->d_vl_offsets, ntexels + 2)) || (rc = dev_alloc(c, &c->alt.d_vl_offsets, ntexels + 2))) return rc;
      c->cap_vl_texels = ntexels + 2;
    }
    if (want_refs > c->cap_vl_refs) {
      c->cap_vl_refs = 0;
      if ((rc = dev_alloc(c, &c->d_vl_refs, want_refs)) || (rc = dev_alloc(c, &c->alt.d_vl_refs, want_refs))) return rc;
      c->cap_vl_refs = want_refs;
      c->cap_vl_tris = 0;  // (the piece map's size goes with the references')
    }
    if ((size_t)c->nt > c->cap_vl_tris) {
      c->cap_vl_tris = 0;
      const size_t pieces = c->cap_vl_refs / 64 + (size_t)c->nt + 1;
      if ((rc = dev_alloc(c, &c->d_vl_rects, (size_t)c->nt)) || (rc = dev_alloc(c, &c->d_vl_pieces, (size_t)c->nt + 2)) ||
          (rc = dev_alloc(c, &c->d_vl_piece_tri, pieces)) ||
          (hipMemset(c->d_vl_piece_tri, 0, sizeof(int32_t) * pieces) != hipSuccess ? (rc = RTGPU_E_MEMORY) : 0))
        return rc;
      c->cap_vl_tris = (size_t)c->nt;
    }
    if (want_cursor > c->cap_vl_cursor) {
      c->cap_vl_cursor = 0;
      if ((rc = dev_alloc(c, &c->d_vl_cursor, want_cursor)) || (rc = dev_alloc(c, &c->d_vl_totals, want_cursor / kScanTile + 2))) return rc;
      c->cap_vl_cursor = want_cursor;
    }
    return RTGPU_OK;
  };
  // (a buffer about to be replaced may still be read by the previous frame's kernels)
  if (ntexels + 2 > c->cap_vl_texels || want_refs > c->cap_vl_refs || (size_t)c->nt > c->cap_vl_tris || want_cursor > c->cap_vl_cursor) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(bs));
    const int rc = grow();
    if (rc == RTGPU_E_MEMORY) {
      (void)hipGetLastError();
      dev_free(&c->d_vl_offsets); dev_free(&c->alt.d_vl_offsets); dev_free(&c->d_vl_refs); dev_free(&c->alt.d_vl_refs);
      dev_free(&c->d_vl_rects); dev_free(&c->d_vl_pieces); dev_free(&c->d_vl_cursor); dev_free(&c->d_vl_totals); dev_free(&c->d_vl_piece_tri);
      c->cap_vl_texels = c->cap_vl_refs = c->cap_vl_tris = c->cap_vl_cursor = 0;
      c->vl_no_memory = true;
      return RTGPU_OK;
    }
    if (rc) return rc;
  }
  const int nt = c->nt;
  const uint32_t cap = (uint32_t)std::min<size_t>(c->opt_vl_capacity > 0 ? std::min<size_t>((size_t)c->opt_vl_capacity, c->cap_vl_refs) : c->cap_vl_refs, 0x7fffffffu);
  KLAUNCH(c, k_view_rects, dim3((unsigned)((nt + 1 + 255) / 256)), dim3(256), bs, nt, (const float *)c->d_verts, c->cam, shift, c->d_vl_rects, c->d_vl_pieces);
  KLAUNCH(c, k_order_scan, dim3(nscan_p), dim3(1024), bs, c->d_vl_pieces, nt + 1, c->d_vl_totals);
  KLAUNCH(c, k_order_scan_totals, dim3(1), dim3(1024), bs, c->d_vl_totals, nscan_p);
  KLAUNCH(c, k_ll_starts, dim3((unsigned)((nt + 2 + 255) / 256)), dim3(256), bs, c->d_vl_pieces, (const uint32_t *)c->d_vl_totals, nt + 1, 0u, c->d_vl_cursor);
  HIPCHK(c, hipMemsetAsync(c->d_vl_offsets, 0, sizeof(uint32_t) * (ntexels + 2), bs));
  // (every triangle's pieces: at most a 64th of its references, and one)
  const uint32_t cap_pieces = (uint32_t)std::min<size_t>(c->cap_vl_refs / 64 + c->cap_vl_tris + 1, 0xffffffffu);  // (what the piece map was allocated for)
  KLAUNCH(c, k_view_pieces, dim3((unsigned)((nt + 255) / 256)), dim3(256), bs, nt, (const uint32_t *)c->d_vl_pieces, cap_pieces, c->d_vl_piece_tri);
  const int sblocks = 32 * c->num_cu;
  KLAUNCH(c, k_view_scatter<false>, dim3(sblocks), dim3(256), bs, (const ViewRectD *)c->d_vl_rects, (const uint32_t *)c->d_vl_pieces, (const int32_t *)c->d_vl_piece_tri,
          cap_pieces, nt, ntx, (int)ntexels, cap, c->d_vl_offsets, (const uint32_t *)nullptr, (int32_t *)nullptr);
  KLAUNCH(c, k_order_scan, dim3(nscan_t), dim3(1024), bs, c->d_vl_offsets, (int)ntexels + 1, c->d_vl_totals);
  KLAUNCH(c, k_order_scan_totals, dim3(1), dim3(1024), bs, c->d_vl_totals, nscan_t);
  KLAUNCH(c, k_ll_starts, dim3((unsigned)((ntexels + 2 + 255) / 256)), dim3(256), bs, c->d_vl_offsets, (const uint32_t *)c->d_vl_totals, (int)ntexels + 1, 0u, c->d_vl_cursor);
  KLAUNCH(c, k_view_scatter<true>, dim3(sblocks), dim3(256), bs, (const ViewRectD *)c->d_vl_rects, (const uint32_t *)c->d_vl_pieces, (const int32_t *)c->d_vl_piece_tri,
          cap_pieces, nt, ntx, (int)ntexels, cap, c->d_vl_cursor, (const uint32_t *)c->d_vl_offsets, c->d_vl_refs);
  c->vl.offsets = c->d_vl_offsets; c->vl.refs = c->d_vl_refs; c->vl.tri_box = c->d_box; c->vl.cap = cap;
  c->vl.ntx = ntx; c->vl.nty = nty; c->vl.shift = shift; c->vl.on = 1; c->vl.pad = c->opt_vl_test_fallback ? 1 : 0;
  return RTGPU_OK;
}
}  // namespace

int rtgpu_frame_begin(rtgpu_ctx *c, const rtgpu_camera *cam) {
  if (!c || !cam) return RTGPU_E_INVALID_PARAM;
  if (!c->have_scene) return fail(c, RTGPU_E_STATE, "no scene uploaded");
  if (cam->sw <= 0 || cam->sh <= 0) return fail(c, RTGPU_E_INVALID_PARAM, "camera resolution must be positive");
  HIPCHK(c, hipSetDevice(c->device));
  c->have_frame = false;
  memcpy(&c->cam, cam, sizeof(CameraD));
  // raytrace.c:346-355: total flux, and grow the domain so that it contains every point light
  float dmin[3], dmax[3], total_flux = 0.0f;
  memcpy(dmin, c->dmin0, sizeof dmin); memcpy(dmax, c->dmax0, sizeof dmax);
  for (int k = 0; k < c->nl; k++) {
    total_flux += c->lights[k].flux;
    for (int i = 0; i < 3; i++) {
      if (c->lights[k].p[i] < dmin[i]) dmin[i] = c->lights[k].p[i] - 0.001f;
      if (c->lights[k].p[i] > dmax[i]) dmax[i] = c->lights[k].p[i] + 0.001f;
    }
  }
  // rtUddCreate (voxelize.c:138-199), same expressions, same mixed float/double evaluation
  float ds[3], cs[3];
  int nv[3];
  for (int k = 0; k < 3; k++) {
    dmin[k] -= 0.001f;
    dmax[k] += 0.001f;
    ds[k] = (float)((double)(dmax[k] - dmin[k]) + 0.001);
  }
  if (c->vmode != 0)
    for (int k = 0; k < 3; k++)
      if (c->vcoeff[k] <= 0.0f) return fail(c, RTGPU_E_INVALID_PARAM, "voxelization coefficients must be > 0 in this voxmode");
  float v = (float)pow((double)((float)c->nt / (ds[0] * ds[1] * ds[2])), (double)0.33333f);
  for (int k = 0; k < 3; k++) {
    float tmp;
    if (c->vmode == 0) tmp = (float)ceil((double)(ds[k] * v));
    else if (c->vmode == 1) tmp = (float)ceil((double)(ds[k] * v * c->vcoeff[k]));
    else tmp = (float)ceil((double)c->vcoeff[k]);
    nv[k] = (int)tmp;
    cs[k] = ds[k] / tmp;
  }
  if (!(nv[0] > 0 && nv[1] > 0 && nv[2] > 0) || (double)nv[0] * nv[1] * nv[2] > 5e8)
    return fail(c, RTGPU_E_INVALID_PARAM, "degenerate voxel grid");
  const int ncells = nv[0] * nv[1] * nv[2];
  int rc;
  if ((size_t)ncells + 1 > c->cap_cells) {
    if ((rc = dev_alloc(c, &c->d_counts, (size_t)ncells + 1)) || (rc = dev_alloc(c, &c->d_cell_start, (size_t)ncells + 1)) ||
        (rc = dev_alloc(c, &c->d_nbatch, (size_t)ncells + 1)) || (rc = dev_alloc(c, &c->d_hdr_start, (size_t)ncells + 1)) ||
        (rc = dev_alloc(c, &c->d_occ, (size_t)ncells / 32 + 1 + 8)) ||  // + the box of the non-empty cells
        (rc = dev_alloc(c, &c->alt.d_cell_start, (size_t)ncells + 1)) || (rc = dev_alloc(c, &c->alt.d_hdr_start, (size_t)ncells + 1)) ||
        (rc = dev_alloc(c, &c->alt.d_occ, (size_t)ncells / 32 + 1 + 8))) return rc;
    c->cap_cells = (size_t)ncells + 1;
  }
  {  // the buckets of the shading order, then their scan blocks' totals (only where some level is shaded in that order)
    const bool ordered = c->nl + kPlanarSamples * c->npl > 0 && (c->opt_order > 1 || (c->opt_order == 1 && c->levels > 1));
    const size_t want = ordered ? order_buckets_of(ncells, order_parts_of(ncells, 8)) * (kScanTile + 1) / kScanTile + 2 : 0;
    if (want > c->cap_bucket) {
      if ((rc = dev_alloc(c, &c->d_bucket, want))) return rc;
      c->cap_bucket = want;
    }
  }
  // The set-up of this frame goes into the set the previous frame does not use, on the build stream: it overlaps
  // the previous frame's kernels (their tail, a few slow wavefronts, leaves most of the GPU idle).
  const bool piped = c->opt_pipeline;
  hipStream_t bs = piped ? c->build_stream : c->stream;
  swap_grid_sets(c);
  if (piped && c->released_valid[c->parity]) HIPCHK(c, hipStreamWaitEvent(bs, c->ev_released[c->parity], 0));
  GridP g;
  for (int k = 0; k < 3; k++) { g.dmin[k] = dmin[k]; g.dmax[k] = dmax[k]; g.cs[k] = cs[k]; g.nv[k] = nv[k]; }
  g.ncells = ncells; g.cell_start = c->d_cell_start; g.cell_tris = nullptr; g.hdr_start = c->d_hdr_start; g.occ = c->d_occ; g.occ_box = reinterpret_cast<const int32_t *>(c->d_occ + ((size_t)ncells / 32 + 1)); g.cell_hdrs = nullptr; g.geom = c->d_geom; g.filt = c->d_filt; g.lmax = c->lmax; g.slack = 0.0f;
  const int tb = 256, tblocks = (c->nt + tb - 1) / tb;
  {
    Span sp(c, W_PRE, bs);
    KLAUNCH(c, k_preprocess, dim3(tblocks), dim3(tb), bs, c->nt, c->d_verts, f3{cam->ob[0], cam->ob[1], cam->ob[2]}, c->d_sid, c->d_surf,
                       c->d_geom, c->d_aux, c->d_filt);
  }
  uint32_t nrefs = 0;
  {
    Span sp(c, W_GRID, bs);
    HIPCHK(c, hipMemsetAsync(c->d_counts, 0, sizeof(uint32_t) * ((size_t)ncells + 1), bs));
    KLAUNCH(c, k_tri_boxes, dim3(tblocks), dim3(tb), bs, c->nt, c->d_verts, g, c->d_box, c->d_counts);
    KLAUNCH(c, k_grid_offsets, dim3(1), dim3(1024), bs, ncells, nv[1], nv[2], c->d_counts, c->d_cell_start, c->d_hdr_start, c->d_nbatch, c->d_occ,
                       reinterpret_cast<int32_t *>(c->d_occ + ((size_t)ncells / 32 + 1)));
    // The grid depends on the scene and its lights only, so the numbers of references and of non-empty cells are
    // read back (a host sync) for the first frame of a scene and reused afterwards; the lists themselves are
    // rebuilt every frame.
    uint32_t nonempty = 0;
    if (c->known_nrefs >= 0 && c->known_nv[0] == nv[0] && c->known_nv[1] == nv[1] && c->known_nv[2] == nv[2]) {
      nrefs = (uint32_t)c->known_nrefs;
      nonempty = (uint32_t)c->known_nonempty;
    } else {
      HIPCHK(c, hipMemcpyAsync(&nrefs, c->d_cell_start + ncells, sizeof(uint32_t), hipMemcpyDeviceToHost, bs));
      HIPCHK(c, hipMemcpyAsync(&nonempty, c->d_nbatch + ncells, sizeof(uint32_t), hipMemcpyDeviceToHost, bs));
      HIPCHK(c, hipStreamSynchronize(bs));
      c->known_nrefs = nrefs;
      c->known_nonempty = nonempty;
      for (int k = 0; k < 3; k++) c->known_nv[k] = nv[k];
    }
    if ((size_t)nrefs > c->cap_refs) {
      if ((rc = dev_alloc(c, &c->d_recs, (size_t)nrefs)) || (rc = dev_alloc(c, &c->d_recs_unsorted, (size_t)nrefs)) ||
          (rc = dev_alloc(c, &c->d_refs, (size_t)nrefs)) ||
          (rc = dev_alloc(c, &c->alt.d_recs, (size_t)nrefs)) || (rc = dev_alloc(c, &c->alt.d_refs, (size_t)nrefs))) return rc;
      c->cap_refs = nrefs;
    }
    // one header per kBatch records of a cell: at most nrefs / kBatch full batches plus one partial batch per non-empty cell
    const size_t nhdrs = (size_t)nrefs / kBatch + ((size_t)nrefs < (size_t)ncells ? (size_t)nrefs : (size_t)ncells) + 1;
    if (nhdrs > c->cap_hdrs) {
      if ((rc = dev_alloc(c, &c->d_hdrs, nhdrs)) || (rc = dev_alloc(c, &c->alt.d_hdrs, nhdrs))) return rc;
      c->cap_hdrs = nhdrs;
    }
    g.cell_tris = c->d_recs;
    g.cell_hdrs = c->d_hdrs;
    if (nonempty > 0) {  // d_nbatch holds the list of the non-empty cells, its length at [ncells]
      KLAUNCH(c, k_grid_fill, dim3((nonempty + 3) / 4), dim3(256), bs, c->nt, g, c->d_nbatch, c->d_box, c->d_recs_unsorted, c->d_refs);
      KLAUNCH(c, k_grid_sort, dim3(nonempty), dim3(kSortThreads), bs, g, c->d_nbatch, c->d_recs_unsorted, c->d_recs);
      KLAUNCH(c, k_grid_bounds, dim3(nonempty), dim3(256), bs, g, c->d_nbatch, c->d_hdrs);
    }
    if ((rc = build_view_lists(c, bs))) return rc;
  }
  if (piped) {
    HIPCHK(c, hipEventRecord(c->ev_built, bs));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_built, 0));
  }
  HIPCHK(c, hipGetLastError());
  c->grid = g;
  for (int k = 0; k < 3; k++) { c->ginfo.nv[k] = nv[k]; c->ginfo.cell[k] = cs[k]; c->ginfo.dmin[k] = dmin[k]; c->ginfo.dmax[k] = dmax[k]; }
  c->ginfo.ncells = ncells; c->ginfo.nrefs = (int32_t)nrefs; c->ginfo.total_flux = total_flux;
  // planar lights: the reference draws 2 rand() values per sample, 16 samples per light, per shaded hit
  if (c->npl > 0 && (rc = ensure_rand(c, (size_t)cam->sw * cam->sh * c->npl * kPlanarSamples * 2))) return rc;
  ShadeP sp;
  sp.surf = c->d_surf; sp.lights = c->d_lights; sp.planar = c->d_planar; sp.tri_aux = c->d_aux; sp.tri_sid = c->d_sid;
  sp.rand_tab = c->npl > 0 ? c->d_rand : nullptr;  // (ensure_rand keeps c->shade.rand_tab current when the table grows)
  sp.nl = c->nl; sp.npl = c->npl; sp.nslots = c->nl + kPlanarSamples * c->npl;
  for (int q = 0; q < 4; q++) sp.tex_tri[q] = c->tex_tri[q];
  sp.total_flux = total_flux; sp.distmod = c->distmod;
  if (sp.nslots != c->shade.nslots) c->chunk_tiles_cap = 0;
  c->shade = sp;
  if (c->opt_light_lists && c->frames_of_scene >= c->opt_ll_after && (rc = ensure_light_lists(c))) return rc;
  if (c->frames_of_scene < (1 << 30)) c->frames_of_scene++;
  c->have_frame = true;
  return RTGPU_OK;
}

int rtgpu_grid_info_get(const rtgpu_ctx *c, rtgpu_grid_info *out) {
  if (!c || !out) return RTGPU_E_INVALID_PARAM;
  if (!c->have_frame) return RTGPU_E_STATE;
  *out = c->ginfo;
  return RTGPU_OK;
}

int64_t rtgpu_chunk_tile_limit(int fan, int levels, int nslots) {
  if (fan < 0 || fan > 2 || levels < 1 || levels > kMaxLevels || nslots < 0) return -1;
  // the deepest level holds tiles * 64 * fan^(levels-1) nodes; node indices are int32, and the shadow walk's items,
  // nodes * nslots, are numbered by 32-bit tickets of 64 (Tickets::take): both must stay below 2^31 resp. 2^32 - 64
  int64_t deepest = kTilePixels, all = kTilePixels;
  for (int L = 1; L < levels; L++) { deepest *= fan; all += deepest; }
  int64_t by_nodes = 0x7fffffffll / all;
  const int64_t per_tile_items = (deepest > kTilePixels ? deepest : (int64_t)kTilePixels) * (nslots > 0 ? nslots : 1);
  int64_t by_items = (0xffffffffll - 64) / per_tile_items;
  const int64_t lim = by_nodes < by_items ? by_nodes : by_items;
  return lim < 1 ? 1 : lim;
}

int64_t rtgpu_tile_count(const rtgpu_ctx *c, int rank, int world) {
  if (!c || !c->have_frame || world <= 0 || rank < 0 || rank >= world) return -1;
  return tiles_of_rank(tiles_x_of(c) * tiles_y_of(c), rank, world);
}

namespace {

// Blocks of the one-thread-per-node kernels: level 0 has exactly n0 entries; the levels below are launched for their
// queue's capacity, cut at 4096 blocks: the kernels loop, and the entries actually there are usually far fewer.
unsigned node_blocks(size_t cap, int level) {
  const size_t b = (cap + 255) / 256;
  return (unsigned)(level == 0 || b < 4096 ? (b ? b : 1) : 4096);
}

// Wavefronts per CU of a walk kernel compiled for `compiled` of them.  Contexts that share the GPU (frames in flight, option
// "contexts_on_gpu") launch 12 each in scenes without secondary rays, so that the other frame's kernels and set-up find room
// (two in flight, ms per frame with all / 12: temple 4K one rank of 8: 0.75 / 0.64, of 4: 1.29 / 0.89, of 2: 1.40 / 1.33, the
// whole frame 1.96 / 1.95); with secondary rays a frame is many launches with host waits between them and fills the
// gaps by itself (ulica 1080p one rank of 2: 57.1 / 58.1, of 4: 32.0 / 33.5, though of 8: 23.6 / 19.8).  Three contexts or
// more launch 8 each (with a hardware queue for every stream, GPU_MAX_HW_QUEUES=16; one rank of 8, ms per frame with 8 / 12
// wavefronts a CU: two in flight 0.57 / 0.60, three 0.47 / 0.52; one rank of 4: two 0.81 / 0.72, three 0.77 / 0.77).
int waves_of(const rtgpu_ctx *c, int compiled) {
  if (c->waves_per_cu) return c->waves_per_cu;
  const int shared = c->contexts_on_gpu >= 3 ? 8 : 12;  // (two contexts: 12 a CU each; three or more: 8)
  return c->contexts_on_gpu > 1 && c->levels == 1 && compiled > shared ? shared : compiled;
}

// One chunk of tiles, first half: primary rays, then nearest-hit walk and hit kernel level by level
// (raytrace.c:170-204); with planar lights also the number of shaded hits in every node's subtree.
int chunk_trace(rtgpu_ctx *c, int64_t t0, int nt_chunk, int rank, int world) {
  const int n0 = nt_chunk * kTilePixels;
  const int pgrid = c->num_cu * ((waves_of(c, 4 * RT_WAVES_NEAREST) + 3) / 4);
  const int64_t tx = tiles_x_of(c);
  c->prim = PrimaryP{c->cam, (int32_t)t0, nt_chunk, rank, world, (int32_t)tx, 1};
  PrimaryP queued = c->prim;
  queued.on = 0;
  {
    Span sp(c, W_PRIMARY);  // the control block; the primary rays themselves are computed where they are used
    QueueCaps caps;
    for (int L = 0; L <= kMaxLevels; L++) caps.cap[L] = L == 0 ? n0 : L < c->levels ? (int32_t)c->cap[L] : 0;
    KLAUNCH(c, k_chunk_init, dim3(1), dim3(256), c->stream, n0, caps, c->d_ctl);
  }
  for (int L = 0; L < c->levels; L++) {
    const size_t capL = L == 0 ? (size_t)n0 : c->cap[L];
    bool fused = false;
    {
      Span sp(c, W_NEAREST);
      const bool rectest = c->opt_record_test < 0 ? c->mostly_opaque : c->opt_record_test != 0;
      // A level that neither spawns rays nor meets a textured triangle shades its hits at the end of the walk: the hit
      // records do not go through memory and k_hit is not launched (option "fuse_hit").
      fused = c->opt_fuse_hit && !(L + 1 < c->levels) && c->tex_tri[0] < 0 && c->tex_tri[1] < 0 && c->tex_tri[2] < 0 && c->tex_tri[3] < 0;
      // the primary rays' hits from the view lists where the frame has them (not the counting variant, which reports the walks)
      const bool viewed = L == 0 && c->opt_view_lists && c->vl.on && c->filter && !c->opt_count;
      const int pgrid = c->num_cu * ((waves_of(c, 4 * (viewed ? RT_WAVES_VIEW : RT_WAVES_NEAREST)) + 3) / 4);
      const NearestArgs nargs{c->grid, c->d_rays[L], c->d_hits, c->d_ctl, L, L == 0 ? c->prim : queued,
                              c->shade, c->d_nodes, c->d_sorg, (int)c->base[L], c->vl};
#define RT_LAUNCH_NEAREST(COUNT, RECTEST, FUSE, VL) \
  KLAUNCH(c, (k_nearest<COUNT, RECTEST, FUSE, VL>), dim3(pgrid), dim3(256), c->stream, nargs)
#define RT_LAUNCH_NEAREST2(COUNT, RECTEST) do { if (fused) RT_LAUNCH_NEAREST(COUNT, RECTEST, true, false); else RT_LAUNCH_NEAREST(COUNT, RECTEST, false, false); } while (0)
#define RT_LAUNCH_NEAREST3(RECTEST) do { if (fused) RT_LAUNCH_NEAREST(false, RECTEST, true, true); else RT_LAUNCH_NEAREST(false, RECTEST, false, true); } while (0)
      if (c->opt_count) { if (rectest) RT_LAUNCH_NEAREST2(true, true); else RT_LAUNCH_NEAREST2(true, false); }
      else if (viewed) { if (rectest) RT_LAUNCH_NEAREST3(true); else RT_LAUNCH_NEAREST3(false); }
      else { if (rectest) RT_LAUNCH_NEAREST2(false, true); else RT_LAUNCH_NEAREST2(false, false); }
#undef RT_LAUNCH_NEAREST3
#undef RT_LAUNCH_NEAREST2
#undef RT_LAUNCH_NEAREST
    }
    if (!fused) {
      Span sp(c, W_HIT);
      const bool spawn = L + 1 < c->levels;
      KLAUNCH(c, k_hit, dim3(node_blocks(capL, L)), dim3(256), c->stream, c->grid, c->shade, c->d_rays[L], c->d_hits,
                         c->d_nodes, c->d_sorg, c->d_geom, spawn ? c->d_rays[L + 1] : nullptr, c->d_ctl, L, (int)c->base[L],
                         spawn ? (int)c->base[L + 1] : 0, spawn ? (int)c->cap[L + 1] : 0, spawn, L == 0 ? c->prim : queued);
    }
  }
  if (c->npl > 0) {
    for (int L = c->levels - 1; L >= 0; L--) {
      const size_t capL = L == 0 ? (size_t)n0 : c->cap[L];
      KLAUNCH(c, k_subtree_hits, dim3(node_blocks(capL, L)), dim3(256), c->stream, c->d_nodes, c->d_size, c->d_ctl, L, (int)c->base[L]);
    }
  }
  return RTGPU_OK;
}

// Planar lights: the rand() replay rank of every shaded hit (one chunk = this rank's whole share of the image).
// sizes: the level-0 subtree sizes of every rank (per_rank entries each), or this rank's own for world = 1;
// total_hits: shaded hits of the whole frame if known, else -1 (world = 1: read back from the device).
int chunk_ranks(rtgpu_ctx *c, int n0, const int32_t *sizes, int64_t per_rank, int rank, int world, int64_t total_hits) {
  KLAUNCH(c, k_tree_starts, dim3(1), dim3(1024), c->stream, c->cam.sw, c->cam.sh, (int)tiles_x_of(c), sizes, (long long)per_rank,
                     world, rank, c->d_rank);
  for (int L = 0; L < c->levels; L++) {
    const size_t capL = L == 0 ? (size_t)n0 : c->cap[L];
    KLAUNCH(c, k_assign_ranks, dim3(node_blocks(capL, L)), dim3(256), c->stream, c->d_nodes, c->d_size, c->d_rank, c->d_ctl, L, (int)c->base[L]);
  }
  if (c->fan > 0) {
    // secondary hits draw random numbers too: the table must cover every shaded hit of the frame (a host sync,
    // only for planar lights together with reflective or transparent surfaces)
    unsigned long long shaded = (unsigned long long)(total_hits > 0 ? total_hits : 0);
    if (total_hits < 0) {
      HIPCHK(c, hipMemcpyAsync(&shaded, &c->d_ctl->stat[kShaded], sizeof shaded, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    int rc = ensure_rand(c, (size_t)shaded * c->npl * kPlanarSamples * 2);
    if (rc) return rc;
  }
  return RTGPU_OK;
}

// The order in which level L's nodes are shaded (option "order"), or null: queue order.
const uint32_t *level_order(const rtgpu_ctx *c, int L) {
  if (c->shade.nslots <= 0 || c->opt_order == 0 || (L == 0 && c->opt_order < 2)) return nullptr;
  // (the buckets are set up by rtgpu_frame_begin for the options in force then: an order asked for later starts with the next frame)
  const size_t need = order_buckets_of(c->grid.ncells, order_parts_of(c->grid.ncells, c->opt_order_parts)) * (kScanTile + 1) / kScanTile + 2;
  if (!c->d_bucket || c->cap_bucket < need || !c->d_order) return nullptr;
  return c->d_order;
}

// Second half: shadow walks, colours deepest level first, colour range, statistics.
int chunk_shade(rtgpu_ctx *c, int64_t t0, int nt_chunk, float *d_rgb, int32_t *d_ids, float *d_minmax) {
  const int n0 = nt_chunk * kTilePixels;
  const bool ends = c->opt_bundle_ends < 0 ? c->levels > 1 : c->opt_bundle_ends != 0;
  const int pgrid = c->num_cu * ((waves_of(c, 4 * (ends ? RT_WAVES_ENDS : RT_WAVES)) + 3) / 4);
  if (c->shade.nslots > 0) {
    for (int L = 0; L < c->levels; L++) {
      Span sp(c, W_SHADOW);
      const uint32_t *order = level_order(c, L);
      if (order) {  // this level's nodes by the cell of their hit point
        const size_t capL = L == 0 ? (size_t)n0 : c->cap[L];
        const int parts = order_parts_of(c->grid.ncells, c->opt_order_parts);
        const int nbuckets = (int)order_buckets_of(c->grid.ncells, parts), nscan = (nbuckets + kScanTile - 1) / kScanTile;
        uint32_t *totals = c->d_bucket + nbuckets;
        HIPCHK(c, hipMemsetAsync(c->d_bucket, 0, sizeof(uint32_t) * (size_t)nbuckets, c->stream));
        KLAUNCH(c, k_order_count, dim3(node_blocks(capL, L)), dim3(256), c->stream, c->grid, c->d_sorg, (const Ctl *)c->d_ctl, L, (int)c->base[L], parts, c->d_bucket);
        KLAUNCH(c, k_order_scan, dim3(nscan), dim3(1024), c->stream, c->d_bucket, nbuckets, totals);
        KLAUNCH(c, k_order_scan_totals, dim3(1), dim3(1024), c->stream, totals, nscan);
        KLAUNCH(c, k_order_place, dim3(node_blocks(capL, L)), dim3(256), c->stream, c->grid, c->d_sorg, (const Ctl *)c->d_ctl, L, (int)c->base[L], parts, c->d_bucket, (const uint32_t *)totals, c->d_order);
      }
      const int32_t *rr = c->npl > 0 ? c->d_rank : nullptr;
      const bool rectest = c->opt_record_test < 0 ? c->mostly_opaque : c->opt_record_test != 0;
      LightListsP ll = c->ll;
      ll.tri_box = c->d_box;
      const bool lists = c->opt_light_lists && ll.on && c->filter && !c->opt_count && rectest && !ends;
      if (!lists) ll.on = 0;
      ShadowArgs sargs{c->grid, c->shade, c->d_sorg, rr, c->d_ts, (long long)c->total_nodes, c->d_ctl, L, (int)c->base[L], order, ll, nullptr};
      // The lists' own kernel first, where it applies: one recursion level, point lights only, queue order (a ticket is a
      // tile's 64 hit points, the queue a whole number of tickets).  What it leaves goes to k_shadow as a list of tickets.
      // (its tickets are six times as long as k_shadow's: it pays where a wavefront gets many of them.  One rank's share of the
      // temple frame at 3840x2160, ms with / without it: whole frame 2.62 / 2.76, half 1.77 / 1.61, quarter 1.21 / 1.12, eighth 0.95 / 0.89)
      const bool lists_kernel = lists && c->opt_ll_kernel != 0 && c->levels == 1 && c->npl == 0 && order == nullptr && L == 0 && n0 % 64 == 0 && c->d_ll_deferred &&
                                (c->opt_ll_kernel > 1 || (size_t)(n0 / 64) >= (size_t)16 * 4 * RT_WAVES_LISTS * (size_t)c->num_cu);
      if (lists_kernel) {
        HIPCHK(c, hipMemsetAsync(c->d_ll_deferred, 0, sizeof(uint32_t), c->stream));
        const ListsArgs largs{c->grid, c->shade, c->d_sorg, c->d_ts, (long long)c->total_nodes, c->d_ctl, (int)c->base[L], ll, c->d_ll_deferred};
        KLAUNCH(c, k_shadow_lists, dim3(c->num_cu * RT_WAVES_LISTS), dim3(256), c->stream, largs);
        sargs.ticket_list = c->d_ll_deferred;
      }
#define RT_LAUNCH_SHADOW(COUNT, RECTEST, ENDS) \
  KLAUNCH(c, (k_shadow<COUNT, RECTEST, ENDS>), dim3(pgrid), dim3(256), c->stream, sargs)
#define RT_LAUNCH_SHADOW2(COUNT, RECTEST) do { if (ends) RT_LAUNCH_SHADOW(COUNT, RECTEST, true); else RT_LAUNCH_SHADOW(COUNT, RECTEST, false); } while (0)
      if (c->opt_count) { if (rectest) RT_LAUNCH_SHADOW2(true, true); else RT_LAUNCH_SHADOW2(true, false); }
      else if (lists) KLAUNCH(c, (k_shadow<false, true, false, true>), dim3(c->num_cu * ((waves_of(c, 4 * RT_WAVES_LL) + 3) / 4)), dim3(256), c->stream, sargs);
      else { if (rectest) RT_LAUNCH_SHADOW2(false, true); else RT_LAUNCH_SHADOW2(false, false); }
#undef RT_LAUNCH_SHADOW2
#undef RT_LAUNCH_SHADOW
    }
  }
  PrimaryP queued = c->prim;
  queued.on = 0;
  for (int L = c->levels - 1; L >= 0; L--) {
    Span sp(c, W_ACCUM);
    const size_t capL = L == 0 ? (size_t)n0 : c->cap[L];
#define RT_LAUNCH_ACCUMULATE(SPECULAR) \
    KLAUNCH(c, (k_accumulate<SPECULAR>), dim3(node_blocks(capL, L)), dim3(256), c->stream, c->shade, c->d_rays[L], c->d_nodes, \
                       c->d_ts, (long long)c->total_nodes, c->d_rank, c->d_colors, c->d_ctl, L, (int)c->base[L], d_rgb, d_ids, (long long)t0 * kTilePixels, \
                       L == 0 ? c->d_partial : nullptr,  /* level 0 (the last one) also folds the colour range of its blocks */ \
                       L == 0 ? c->prim : queued, level_order(c, L))
    if (c->any_specular) RT_LAUNCH_ACCUMULATE(true); else RT_LAUNCH_ACCUMULATE(false);
#undef RT_LAUNCH_ACCUMULATE
  }
  {
    Span sp(c, W_MINMAX);
    KLAUNCH(c, k_minmax_final, dim3(1), dim3(1024), c->stream, c->d_partial, (n0 + 255) / 256, d_minmax,
                       (const Ctl *)c->d_ctl, c->d_totals);  // also the chunk's statistics into the frame totals
  }
  return RTGPU_OK;
}

// Tiles per chunk for this rank's share (at most `most`), and the queues for it, with the current sizing factors.
int plan_chunks(rtgpu_ctx *c, int64_t ntiles, bool whole, int64_t *chunk_out) {
  size_t per_tile = bytes_per_tile(c);
  int64_t chunk = (int64_t)(c->chunk_bytes / (int64_t)per_tile);
  if (chunk < 1) chunk = 1;
  if (chunk > ntiles) chunk = ntiles;
  if (whole && chunk < ntiles)
    return fail(c, RTGPU_E_UNSUPPORTED, "planar lights need this rank's whole share of the image in one chunk; raise chunk_bytes");
  const int64_t limit = rtgpu_chunk_tile_limit(c->fan, c->levels, c->shade.nslots);
  if (whole && limit < ntiles)
    return fail(c, RTGPU_E_UNSUPPORTED, "planar lights need this rank's whole share of the image in one chunk, and it has more items than the queues index");
  if (chunk > limit) chunk = limit;
  *chunk_out = chunk;
  return ensure_chunk(c, (size_t)chunk);
}

// After the first half of a chunk: how many rays every level held (one host synchronisation).  Updates the scene's
// observed ratios; *cut = some queue was too small for what the level above produced.
int check_queues(rtgpu_ctx *c, int n0, bool *cut) {
  int32_t count[kMaxLevels + 1];
  HIPCHK(c, hipMemcpyAsync(count, c->d_ctl, sizeof count, hipMemcpyDeviceToHost, c->stream));  // Ctl starts with count[]
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *cut = false;
  for (int L = 1; L < c->levels; L++) {
    const double r = (double)count[L] / (double)(n0 > 0 ? n0 : 1);
    if (r > c->ratio[L]) c->ratio[L] = r;
    if ((size_t)count[L] > c->cap[L]) *cut = true;
  }
  c->have_ratio = true;
  return RTGPU_OK;
}

}  // namespace

int rtgpu_render(rtgpu_ctx *c, int rank, int world, float *d_rgb, int32_t *d_ids, float *d_minmax) {
  if (!c || !d_rgb || !d_ids || !d_minmax || world <= 0 || rank < 0 || rank >= world) return RTGPU_E_INVALID_PARAM;
  if (!c->have_frame) return fail(c, RTGPU_E_STATE, "rtgpu_frame_begin has not been called");
  if (c->npl > 0 && world != 1)
    return fail(c, RTGPU_E_UNSUPPORTED, "planar lights on several GPUs: rtgpu_render_trace, exchange of rtgpu_hit_counts, rtgpu_render_shade");
  HIPCHK(c, hipSetDevice(c->device));
  const int64_t ntiles = tiles_of_rank(tiles_x_of(c) * tiles_y_of(c), rank, world);
  KLAUNCH(c, k_frame_init, dim3(1), dim3(64), c->stream, d_minmax, c->d_totals);
  c->split_tiles = -1;
  if (ntiles == 0) return RTGPU_OK;
  c->grid.slack = c->filter ? 0.0f : INFINITY;
  // Scenes with secondary rays: the queues of the levels below 0 are sized from what the scene's chunks have produced so
  // far (a first chunk of at most 2048 tiles runs with worst-case queues to find out), checked after the first half of
  // every chunk; a chunk whose queues were cut is rendered again with worst-case queues.  Planar lights need the
  // whole share in one chunk with its random numbers: worst case.  Scenes without secondary rays have nothing to size.
  const bool adaptive = c->adaptive_queues && c->fan > 0 && c->npl == 0;
  bool worst_once = false;
  int rc;
  c->last_chunks = c->last_rerendered = 0;
  for (int64_t t0 = 0; t0 < ntiles;) {
    const bool pilot = adaptive && !c->have_ratio;
    if (adaptive && !pilot && !worst_once) observed_factors(c); else worst_case_factors(c);
    int64_t chunk = 0;
    const int64_t left = ntiles - t0;
    if ((rc = plan_chunks(c, pilot && left > 2048 ? 2048 : left, c->npl > 0, &chunk))) return rc;
    const int nt_chunk = (int)chunk;
    if ((rc = chunk_trace(c, t0, nt_chunk, rank, world))) return rc;
    if (adaptive) {
      bool cut = false;
      if ((rc = check_queues(c, nt_chunk * kTilePixels, &cut))) return rc;
      if (cut) { worst_once = true; c->last_rerendered++; continue; }  // (only a chunk sized from earlier ones can be cut; worst-case queues hold everything)
      worst_once = false;
    }
    if (c->npl > 0 && (rc = chunk_ranks(c, nt_chunk * kTilePixels, c->d_size, 0, 0, 1, -1))) return rc;
    if ((rc = chunk_shade(c, t0, nt_chunk, d_rgb, d_ids, d_minmax))) return rc;
    t0 += chunk;
    c->last_chunks++;
  }
  mark_released(c);
  HIPCHK(c, hipGetLastError());
  return RTGPU_OK;
}

int64_t rtgpu_static_scratch_bytes(void) { return (int64_t)kScratchBytes; }

int rtgpu_render_static(rtgpu_ctx *c, int rank, int world, float *d_rgb, int32_t *d_ids, float *d_minmax) {
  if (!c || !d_rgb || !d_ids || !d_minmax || world <= 0 || rank < 0 || rank >= world) return RTGPU_E_INVALID_PARAM;
  if (!c->have_frame) return fail(c, RTGPU_E_STATE, "rtgpu_frame_begin has not been called");
  if (!c->scratch) return fail(c, RTGPU_E_UNSUPPORTED, "rtgpu_render_static: no static scratch array on this device index");
  if (c->npl > 0)
    return fail(c, RTGPU_E_UNSUPPORTED, "rtgpu_render_static: planar lights need a table of random numbers sized at run time; use rtgpu_render");
  HIPCHK(c, hipSetDevice(c->device));
  const int64_t ntiles = tiles_of_rank(tiles_x_of(c) * tiles_y_of(c), rank, world);
  // the whole share is one chunk, and its queues must fit the array: sizes first, nothing is touched on refusal
  if (ntiles > rtgpu_chunk_tile_limit(c->fan, c->levels, c->shade.nslots))
    return fail(c, RTGPU_E_UNSUPPORTED, "rtgpu_render_static: this rank's share has more work items than one chunk can number; use rtgpu_render");
  const QueueSet own = take_queues(c);
  worst_case_factors(c);  // (sizing from earlier chunks would need a look at the queue lengths from the host in mid-frame)
  size_t need = 0;
  (void)plan_queues(c, (size_t)ntiles, [&](void **, size_t bytes) { need += (bytes + 255) & ~(size_t)255; return (int)RTGPU_OK; });
  if (need > kScratchBytes) {
    put_queues(c, own);
    return fail(c, RTGPU_E_UNSUPPORTED, "rtgpu_render_static: the queues of this rank's share need " + std::to_string(need) +
                                            " bytes, the static scratch array holds " + std::to_string(kScratchBytes) +
                                            "; render fewer tiles per rank or use rtgpu_render");
  }
  size_t off = 0;
  (void)plan_queues(c, (size_t)ntiles, [&](void **slot, size_t bytes) {
    *slot = c->scratch + off;
    off += (bytes + 255) & ~(size_t)255;
    return (int)RTGPU_OK;
  });
  // Nothing in the array has to start at zero: every queue entry is written by the kernel that produces it before the
  // kernel that consumes it reads it (hits, nodes, ts and colours of shaded hits, one partial range per block of
  // level 0), and the counters live in the context's control block, which k_chunk_init resets.
  const int timing = c->opt_timing;
  c->opt_timing = 0;  // (the per-kernel timing creates events: an allocating call)
  c->split_tiles = -1;
  // Under stream capture the call only records work: no event of it may be waited for from outside the capture, so
  // neither the array's owner nor the pipelined frame set-up learn of it (the caller orders a later
  // rtgpu_frame_begin, and other users of the array, behind the graph's launches).
  hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
  if (c->stream) (void)hipStreamIsCapturing(c->stream, &capture);
  const bool capturing = capture != hipStreamCaptureStatusNone;
  ScratchOwner &owner = g_scratch_owner[c->device];
  int rc = RTGPU_OK;
  {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!capturing && owner.valid && (owner.ctx != c || owner.stream != c->stream) && hipStreamWaitEvent(c->stream, owner.done, 0) != hipSuccess)
      rc = fail(c, RTGPU_E_NO_DEVICE, "rtgpu_render_static: cannot order the call behind the previous user of the scratch array");
  }
  if (!rc) do {
    hipLaunchKernelGGL(k_frame_init, dim3(1), dim3(64), 0, c->stream, d_minmax, c->d_totals);
    if (ntiles == 0) break;
    c->grid.slack = c->filter ? 0.0f : INFINITY;
    if ((rc = chunk_trace(c, 0, (int)ntiles, rank, world))) break;
    if ((rc = chunk_shade(c, 0, (int)ntiles, d_rgb, d_ids, d_minmax))) break;
    if (capturing) c->released_valid[c->parity] = false; else mark_released(c);
  } while (0);
  {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    owner.valid = !capturing && hipEventRecord(owner.done, c->stream) == hipSuccess;
    owner.ctx = c; owner.stream = c->stream;
  }
  c->opt_timing = timing;
  put_queues(c, own);
  // (c->fac[] keeps the worst-case factors this call planned with; rtgpu_render sets its own before it plans a chunk, and
  // the context's own queues are judged by fac_alloc[], which this call has not touched)
  if (rc) return rc;
  HIPCHK(c, hipGetLastError());
  return RTGPU_OK;
}

int rtgpu_render_trace(rtgpu_ctx *c, int rank, int world) {
  if (!c || world <= 0 || rank < 0 || rank >= world) return RTGPU_E_INVALID_PARAM;
  if (!c->have_frame) return fail(c, RTGPU_E_STATE, "rtgpu_frame_begin has not been called");
  HIPCHK(c, hipSetDevice(c->device));
  const int64_t ntiles = tiles_of_rank(tiles_x_of(c) * tiles_y_of(c), rank, world);
  HIPCHK(c, hipMemsetAsync(c->d_totals, 0, sizeof(unsigned long long) * kTotals, c->stream));
  c->split_tiles = -1;
  if (ntiles == 0) { c->split_tiles = 0; c->split_rank = rank; c->split_world = world; return RTGPU_OK; }
  int64_t chunk = 0;
  worst_case_factors(c);
  int rc = plan_chunks(c, ntiles, true, &chunk);
  if (rc) return rc;
  c->grid.slack = c->filter ? 0.0f : INFINITY;
  if ((rc = chunk_trace(c, 0, (int)ntiles, rank, world))) return rc;
  mark_released(c);
  HIPCHK(c, hipGetLastError());
  c->split_tiles = ntiles; c->split_rank = rank; c->split_world = world;
  return RTGPU_OK;
}

int rtgpu_hit_counts(rtgpu_ctx *c, int32_t *d_counts) {
  if (!c || !d_counts) return RTGPU_E_INVALID_PARAM;
  if (c->split_tiles < 0) return fail(c, RTGPU_E_STATE, "rtgpu_render_trace has not been called for this frame");
  HIPCHK(c, hipSetDevice(c->device));
  if (c->split_tiles == 0) return RTGPU_OK;
  const size_t n = (size_t)c->split_tiles * kTilePixels;
  if (c->npl > 0) HIPCHK(c, hipMemcpyAsync(d_counts, c->d_size, sizeof(int32_t) * n, hipMemcpyDeviceToDevice, c->stream));
  else HIPCHK(c, hipMemsetAsync(d_counts, 0, sizeof(int32_t) * n, c->stream));
  return RTGPU_OK;
}

int rtgpu_hit_total(rtgpu_ctx *c, int64_t *d_total) {
  if (!c || !d_total) return RTGPU_E_INVALID_PARAM;
  if (c->split_tiles < 0) return fail(c, RTGPU_E_STATE, "rtgpu_render_trace has not been called for this frame");
  HIPCHK(c, hipSetDevice(c->device));
  static_assert(sizeof(unsigned long long) == sizeof(int64_t), "the statistics are 64-bit words");
  // the walks of every level add their hits to one counter of the chunk's control block (a rank's share is one chunk here)
  if (c->split_tiles == 0) HIPCHK(c, hipMemsetAsync(d_total, 0, sizeof(int64_t), c->stream));
  else HIPCHK(c, hipMemcpyAsync(d_total, &c->d_ctl->stat[kShaded], sizeof(int64_t), hipMemcpyDeviceToDevice, c->stream));
  return RTGPU_OK;
}

int rtgpu_render_shade(rtgpu_ctx *c, const int32_t *d_all_counts, int64_t per_rank, int64_t total_hits, float *d_rgb, int32_t *d_ids,
                       float *d_minmax) {
  if (!c || !d_rgb || !d_ids || !d_minmax) return RTGPU_E_INVALID_PARAM;
  if (c->split_tiles < 0) return fail(c, RTGPU_E_STATE, "rtgpu_render_trace has not been called for this frame");
  HIPCHK(c, hipSetDevice(c->device));
  KLAUNCH(c, k_minmax_init, dim3(1), dim3(64), c->stream, d_minmax);
  const int64_t ntiles = c->split_tiles;
  c->split_tiles = -1;
  if (ntiles == 0) return RTGPU_OK;
  int rc;
  if (c->npl > 0) {
    if (c->split_world > 1 && (!d_all_counts || per_rank < ntiles * kTilePixels || total_hits < 0))
      return fail(c, RTGPU_E_INVALID_PARAM, "planar lights on several GPUs need every rank's hit counts and their sum");
    if (c->split_world > 1) rc = chunk_ranks(c, (int)ntiles * kTilePixels, d_all_counts, per_rank, c->split_rank, c->split_world, total_hits);
    else rc = chunk_ranks(c, (int)ntiles * kTilePixels, c->d_size, 0, 0, 1, -1);
    if (rc) return rc;
  }
  if ((rc = chunk_shade(c, 0, (int)ntiles, d_rgb, d_ids, d_minmax))) return rc;
  mark_released(c);
  HIPCHK(c, hipGetLastError());
  return RTGPU_OK;
}

namespace {
// The tone map of n pixels whose colours lie `stride` floats apart, for a list of ng gammas (host memory).
int tonemap_strided(rtgpu_ctx *c, const float *d_rgb, int64_t stride, int64_t n, const float *d_minmax, const float *gammas, int32_t ng,
                    uint32_t *d_rgba) {
  if (n == 0) return RTGPU_OK;
  const dim3 blocks((unsigned)((n + 255) / 256));
  if (ng == 1) {  // one gamma: the kernel with the short forms of the power
    KLAUNCH(c, k_tonemap, blocks, dim3(256), c->stream, d_rgb, (long long)stride, (long long)n, d_minmax, gammas[0], d_rgba);
  } else {
    if ((size_t)ng > c->cap_gammas) {
      int rc = dev_alloc(c, &c->d_gammas, (size_t)ng + 16);
      if (rc) { c->cap_gammas = 0; return rc; }
      c->cap_gammas = (size_t)ng + 16;
    }
    // (from pageable memory the copy has left the caller's array when the call returns)
    if (ng > 0) HIPCHK(c, hipMemcpyAsync(c->d_gammas, gammas, sizeof(float) * (size_t)ng, hipMemcpyHostToDevice, c->stream));
    KLAUNCH(c, k_tonemap_multi, blocks, dim3(256), c->stream, d_rgb, (long long)stride, (long long)n, d_minmax, (const float *)c->d_gammas, (int)ng, d_rgba);
  }
  HIPCHK(c, hipGetLastError());
  return RTGPU_OK;
}
}  // namespace

int rtgpu_tonemap(rtgpu_ctx *c, const float *d_rgb, int64_t n, const float *d_minmax, float gamma, uint32_t *d_rgba) {
  if (!c || !d_rgb || !d_minmax || !d_rgba || n < 0) return RTGPU_E_INVALID_PARAM;
  HIPCHK(c, hipSetDevice(c->device));
  return tonemap_strided(c, d_rgb, 3, n, d_minmax, &gamma, 1, d_rgba);
}

int rtgpu_tonemap_multi(rtgpu_ctx *c, const float *d_rgb, int64_t n, const float *d_minmax, const float *gammas, int32_t ngammas,
                        uint32_t *d_rgba) {
  if (!c || !d_rgb || !d_minmax || !d_rgba || n < 0 || ngammas < 0 || (ngammas > 0 && !gammas)) return RTGPU_E_INVALID_PARAM;
  HIPCHK(c, hipSetDevice(c->device));
  return tonemap_strided(c, d_rgb, 3, n, d_minmax, gammas, ngammas, d_rgba);
}

int rtgpu_untile(rtgpu_ctx *c, const void *d_gathered, int world, int elem_bytes, void *d_image) {
  if (!c || !d_gathered || !d_image || world <= 0 || elem_bytes <= 0 || elem_bytes % 4) return RTGPU_E_INVALID_PARAM;
  if (!c->have_frame) return RTGPU_E_STATE;
  HIPCHK(c, hipSetDevice(c->device));
  const long long n = (long long)c->cam.sw * c->cam.sh, tx = tiles_x_of(c), ntiles = tx * tiles_y_of(c);
  KLAUNCH(c, k_untile, dim3((unsigned)((n + 255) / 256)), dim3(256), c->stream, (const uint32_t *)d_gathered, world, c->cam.sw,
                     c->cam.sh, (int)tx, ntiles, elem_bytes / 4, (uint32_t *)d_image);
  HIPCHK(c, hipGetLastError());
  return RTGPU_OK;
}

int64_t rtgpu_bmp_row_bytes(int32_t width, int32_t bpp) {
  if (width <= 0 || (bpp != 24 && bpp != 32)) return -1;
  return (int64_t)(4 * ceil(bpp * width / 32.0));  // bitmap.c:457
}

int rtgpu_bmp_rows(rtgpu_ctx *c, const uint32_t *d_image, int32_t bpp, void *d_rows) {
  if (!c || !d_image || !d_rows || (bpp != 24 && bpp != 32)) return RTGPU_E_INVALID_PARAM;
  if (!c->have_frame) return RTGPU_E_STATE;
  HIPCHK(c, hipSetDevice(c->device));
  const int w = c->cam.sw, h = c->cam.sh, words = (int)(rtgpu_bmp_row_bytes(w, bpp) / 4);
  const long long n = (long long)words * h;
  KLAUNCH(c, k_bmp_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), c->stream, d_image, w, h, bpp / 8, words, (uint32_t *)d_rows);
  HIPCHK(c, hipGetLastError());
  return RTGPU_OK;
}

int rtgpu_bmp_header(int32_t width, int32_t height, int32_t bpp, unsigned char h[54]) {
  if (!h || width <= 0 || height <= 0 || (bpp != 24 && bpp != 32)) return RTGPU_E_INVALID_PARAM;
  const uint32_t total = (uint32_t)(54 + rtgpu_bmp_row_bytes(width, bpp) * height);
  auto wr32 = [](unsigned char *p, uint32_t v) { p[0] = v & 0xff; p[1] = (v >> 8) & 0xff; p[2] = (v >> 16) & 0xff; p[3] = (v >> 24) & 0xff; };
  memset(h, 0, 54);
  h[0] = 'B'; h[1] = 'M';
  wr32(h + 2, total); wr32(h + 10, 54); wr32(h + 14, 40);
  wr32(h + 18, (uint32_t)width); wr32(h + 22, (uint32_t)height);
  h[26] = 1; h[28] = (unsigned char)bpp;
  wr32(h + 34, total - 54);
  return RTGPU_OK;
}

int rtgpu_sync(rtgpu_ctx *c) {
  if (!c) return RTGPU_E_INVALID_PARAM;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return RTGPU_OK;
}

int rtgpu_counters_get(rtgpu_ctx *c, rtgpu_counters *out) {
  if (!c || !out) return RTGPU_E_INVALID_PARAM;
  HIPCHK(c, hipSetDevice(c->device));
  unsigned long long h[kTotals];
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(h, c->d_totals, sizeof h, hipMemcpyDeviceToHost));
  memset(out, 0, sizeof *out);
  out->pixels = (int64_t)h[kPixels];
  out->pixels_traced = (int64_t)h[kPixelsTraced];
  int64_t secondary = 0;
  for (int L = 1; L < kMaxLevels; L++) secondary += (int64_t)h[kStatCount + L];
  out->nearest_queries = out->pixels_traced + secondary;
  out->shadow_queries = (int64_t)h[kShaded] * c->shade.nslots;
  out->shadow_walks = (int64_t)h[kShadowWalks];
  out->nearest_steps = (int64_t)h[kNearS]; out->nearest_tests = (int64_t)h[kNearT];
  out->shadow_steps = (int64_t)h[kShadS]; out->shadow_tests = (int64_t)h[kShadT];
  out->nearest_lane_tests = (int64_t)h[kNearLaneTests]; out->nearest_lane_slots = (int64_t)h[kNearWaveTests];
  out->shadow_lane_tests = (int64_t)h[kShadLaneTests]; out->shadow_lane_slots = (int64_t)h[kShadWaveTests];
  out->nearest_stage2 = (int64_t)h[kNearStage2]; out->nearest_stage3 = (int64_t)h[kNearStage3]; out->nearest_exact = (int64_t)h[kNearExact];
  out->shadow_stage2 = (int64_t)h[kShadStage2]; out->shadow_stage3 = (int64_t)h[kShadStage3]; out->shadow_exact = (int64_t)h[kShadExact];
  out->nearest_visits = (int64_t)h[kNearVisits]; out->shadow_visits = (int64_t)h[kShadVisits];
  out->nearest_batches = (int64_t)h[kNearBatches]; out->nearest_batches_culled = (int64_t)h[kNearCulled];
  out->shadow_batches = (int64_t)h[kShadBatches]; out->shadow_batches_culled = (int64_t)h[kShadCulled];
  out->chunks = c->last_chunks; out->chunks_rerendered = c->last_rerendered;
  out->nearest_full_tests = (int64_t)h[kNearFull]; out->shadow_full_tests = (int64_t)h[kShadFull];
  out->shadow_list_tickets = (int64_t)h[kShadListTickets]; out->shadow_list_fallbacks = (int64_t)h[kShadListFallbacks];
  out->view_list_tickets = (int64_t)h[kViewListTickets]; out->view_list_fallbacks = (int64_t)h[kViewListFallbacks];
  return RTGPU_OK;
}

int rtgpu_phase_cycles(rtgpu_ctx *c, int64_t out[16], int reset) {
  if (!c || !out) return RTGPU_E_INVALID_PARAM;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  unsigned long long h[16];
  HIPCHK(c, hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof h));
  for (int i = 0; i < 16; i++) out[i] = (int64_t)h[i];
  if (reset) {
    memset(h, 0, sizeof h);
    HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(g_phase), h, sizeof h));
  }
  return RTGPU_OK;
}

#ifdef RT_HIST
// the shadow results of the last chunk, slot-major (diagnostic build: a blocked ray holds -(blocking triangle + 2))
int rtgpu_debug_ts(rtgpu_ctx *c, float *out, long long nodes, int nslots) {
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int s = 0; s < nslots; s++)
    HIPCHK(c, hipMemcpy(out + (size_t)s * nodes, c->d_ts + (size_t)s * c->total_nodes, sizeof(float) * (size_t)nodes, hipMemcpyDeviceToHost));
  return RTGPU_OK;
}
int rtgpu_debug_hist(rtgpu_ctx *c, unsigned long long *out) {
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hist), sizeof(unsigned long long) * 36));
  HIPCHK(c, hipMemcpyFromSymbol(out + 36, HIP_SYMBOL(g_cand), sizeof(unsigned long long) * 8));
  HIPCHK(c, hipMemcpyFromSymbol(out + 44, HIP_SYMBOL(g_mode), sizeof(unsigned long long) * 24));
  HIPCHK(c, hipMemcpyFromSymbol(out + 68, HIP_SYMBOL(g_chunk), sizeof(unsigned long long) * 24));
  return RTGPU_OK;
}
#endif

int rtgpu_ticket_log(rtgpu_ctx *c, int which, uint32_t *out, int max_tickets, int *copied) {
  if (!c || !out || !copied || which < 0 || which > 1) return RTGPU_E_INVALID_PARAM;
  *copied = 0;
#ifdef RT_TICKETS
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const int n = max_tickets < kTicketLog ? max_tickets : kTicketLog;
  const size_t entry = sizeof(uint32_t) * kTicketWords;
  HIPCHK(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ticket_log), entry * (size_t)n, entry * (size_t)kTicketLog * which));
  *copied = n;
#else
  (void)max_tickets;
#endif
  return RTGPU_OK;
}

int rtgpu_light_list_pairs(int64_t n, const float *verts, const float *lights, float olim, double *out) {
  if (n < 0 || !verts || !lights || !out || !(olim > 0.0f)) return RTGPU_E_INVALID_PARAM;
  for (int64_t k = 0; k < n; k++) {
    const ll_detail::D3 P = {lights[3 * k], lights[3 * k + 1], lights[3 * k + 2]};
    for (int f = 0; f < 6; f++) {
      ll_detail::URect r;
      double *o = out + (k * 6 + f) * 5;
      if (face_rect(verts + 9 * k, P, (double)olim * 1.0002, f, r)) { o[0] = 1.0; o[1] = r.u0; o[2] = r.u1; o[3] = r.v0; o[4] = r.v1; }
      else o[0] = o[1] = o[2] = o[3] = o[4] = 0.0;
    }
  }
  return RTGPU_OK;
}
int rtgpu_light_list_rects(int32_t nt, const float *verts, int32_t nl, const float *lights, int32_t R, float olim, int64_t cap,
                           uint32_t *rects, float *faces, float *olim_out, int64_t *n_rects, int64_t *refs) {
  if (nt < 0 || nl < 0 || R < 1 || R > 256 || (nt > 0 && !verts) || (nl > 0 && !lights)) return RTGPU_E_INVALID_PARAM;
  LightListsHost h;
  build_light_rects(nt, verts, nl, lights, R, olim, h);
  if (olim_out) *olim_out = h.olim;
  if (n_rects) *n_rects = (int64_t)h.rects.size();
  if (refs) *refs = (int64_t)h.refs;
  if (faces) memcpy(faces, h.faces.data(), sizeof(LLFaceH) * h.faces.size());  // nl * 12
  if (rects) memcpy(rects, h.rects.data(), sizeof(LLRect) * (size_t)std::min<int64_t>(cap > 0 ? cap : 0, (int64_t)h.rects.size()));
  return RTGPU_OK;
}

int rtgpu_view_list_rects(int64_t n, const float *verts, const rtgpu_camera *cams, int64_t ncams, int32_t *out) {
  if (n < 0 || ncams < 1 || (n > 0 && !verts) || !cams || !out || (ncams != 1 && ncams != n)) return RTGPU_E_INVALID_PARAM;
  for (int64_t t = 0; t < n; t++) {
    CameraD cam;
    memcpy(&cam, cams + (ncams == 1 ? 0 : t), sizeof cam);
    ViewRect r;
    int32_t *o = out + 5 * t;
    o[0] = view_rect(verts + 9 * t, cam, r) ? 1 : 0;
    o[1] = o[0] ? r.a0 : 0; o[2] = o[0] ? r.a1 : -1; o[3] = o[0] ? r.b0 : 0; o[4] = o[0] ? r.b1 : -1;
  }
  return RTGPU_OK;
}

int rtgpu_math_selfcheck(rtgpu_ctx *c, uint64_t mismatches[2]) {
  if (!c || !mismatches) return RTGPU_E_INVALID_PARAM;
  HIPCHK(c, hipSetDevice(c->device));
  // two spare words at the end of the frame totals buffer serve as the counters (the frame's statistics stay as they are)
  static_assert(kStatCount + kMaxLevels + 2 <= kTotals, "two spare words behind the statistics and the queue lengths");
  unsigned long long *slots = c->d_totals + (kTotals - 2);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemsetAsync(slots, 0, sizeof(unsigned long long) * 2, c->stream));
  KLAUNCH(c, k_math_selfcheck, dim3(c->num_cu * 8), dim3(256), c->stream, slots);
  unsigned long long bad[2] = {0, 0};
  HIPCHK(c, hipMemcpyAsync(bad, slots, sizeof bad, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  mismatches[0] = bad[0]; mismatches[1] = bad[1];
  return RTGPU_OK;
}

int rtgpu_timings_get(rtgpu_ctx *c, rtgpu_timings *out) {
  if (!c || !out) return RTGPU_E_INVALID_PARAM;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  resolve_spans(c);
  *out = c->timings;
  return RTGPU_OK;
}

int rtgpu_render_host(rtgpu_ctx *c, float *rgb, int32_t *ids, float *minmax) {
  if (!c) return RTGPU_E_INVALID_PARAM;
  if (!c->have_frame) return fail(c, RTGPU_E_STATE, "rtgpu_frame_begin has not been called");
  HIPCHK(c, hipSetDevice(c->device));
  const int64_t nt = rtgpu_tile_count(c, 0, 1), npx = (int64_t)c->cam.sw * c->cam.sh;
  int rc = RTGPU_OK;
  // the tile-major and row-major device copies are kept from call to call and only grow
  if ((size_t)nt > c->h_cap_tiles) {
    c->h_cap_tiles = 0;
    if ((rc = dev_alloc(c, &c->h_rgb_t, (size_t)nt * kTilePixels * 3)) || (rc = dev_alloc(c, &c->h_ids_t, (size_t)nt * kTilePixels))) return rc;
    c->h_cap_tiles = (size_t)nt;
  }
  if ((size_t)npx > c->h_cap_px) {
    c->h_cap_px = 0;
    if ((rc = dev_alloc(c, &c->h_rgb, (size_t)npx * 3)) || (rc = dev_alloc(c, &c->h_ids, (size_t)npx))) return rc;
    c->h_cap_px = (size_t)npx;
  }
  if (!c->h_mm && (rc = dev_alloc(c, &c->h_mm, 6))) return rc;
  if ((rc = rtgpu_render(c, 0, 1, c->h_rgb_t, c->h_ids_t, c->h_mm)) || (rc = rtgpu_untile(c, c->h_rgb_t, 1, 12, c->h_rgb)) ||
      (rc = rtgpu_untile(c, c->h_ids_t, 1, 4, c->h_ids)))
    return rc;
  if (rgb) HIPCHK(c, hipMemcpyAsync(rgb, c->h_rgb, sizeof(float) * 3 * npx, hipMemcpyDeviceToHost, c->stream));
  if (ids) HIPCHK(c, hipMemcpyAsync(ids, c->h_ids, sizeof(int32_t) * npx, hipMemcpyDeviceToHost, c->stream));
  if (minmax) HIPCHK(c, hipMemcpyAsync(minmax, c->h_mm, sizeof(float) * 6, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return RTGPU_OK;
}

void *rtgpu_pinned_alloc(size_t bytes) {
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return p;
}
void rtgpu_pinned_free(void *p) {
  if (p) (void)hipHostFree(p);
}

namespace {
int check_layout(rtgpu_ctx *c, const rtgpu_pixel_layout *lay) {
  if (!lay || lay->stride < 16 || lay->stride % 4 || lay->color_offset < 0 || lay->color_offset % 4 || lay->color_offset + 16 > lay->stride ||
      (lay->tri_offset >= 0 && (lay->tri_offset % 8 || lay->stride % 8 || lay->tri_offset + 8 > lay->stride ||
                                (lay->tri_offset < lay->color_offset + 16 && lay->color_offset < lay->tri_offset + 8))))
    return fail(c, RTGPU_E_INVALID_PARAM, "pixel layout: a colour of four floats (4-byte aligned) and, if there is one, an 8-byte aligned pointer, inside the stride and apart");
  return RTGPU_OK;
}
// the device copy of a caller's pixel records
int ensure_map(rtgpu_ctx *c, size_t bytes) {
  if (bytes <= c->h_cap_map) return RTGPU_OK;
  c->h_cap_map = 0;
  int rc = dev_alloc(c, &c->h_map, bytes);
  if (rc) return rc;
  c->h_cap_map = bytes;
  return RTGPU_OK;
}
}  // namespace

int rtgpu_render_pixelmap(rtgpu_ctx *c, void *map, const rtgpu_pixel_layout *lay, float *minmax) {
  if (!c || !map) return RTGPU_E_INVALID_PARAM;
  if (!c->have_frame) return fail(c, RTGPU_E_STATE, "rtgpu_frame_begin has not been called");
  int rc = check_layout(c, lay);
  if (rc) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  const int64_t nt = rtgpu_tile_count(c, 0, 1), npx = (int64_t)c->cam.sw * c->cam.sh;
  if ((size_t)nt > c->h_cap_tiles) {
    c->h_cap_tiles = 0;
    if ((rc = dev_alloc(c, &c->h_rgb_t, (size_t)nt * kTilePixels * 3)) || (rc = dev_alloc(c, &c->h_ids_t, (size_t)nt * kTilePixels))) return rc;
    c->h_cap_tiles = (size_t)nt;
  }
  if (!c->h_mm && (rc = dev_alloc(c, &c->h_mm, 6))) return rc;
  const size_t bytes = (size_t)npx * (size_t)lay->stride;
  if ((rc = ensure_map(c, bytes)) || (rc = rtgpu_render(c, 0, 1, c->h_rgb_t, c->h_ids_t, c->h_mm))) return rc;
  const PixelLayoutD ld{lay->stride, lay->color_offset, lay->tri_offset, lay->tri_base, lay->tri_stride};
  KLAUNCH(c, k_pixelmap, dim3((unsigned)((npx + 255) / 256)), dim3(256), c->stream, (const float *)c->h_rgb_t, (const int32_t *)c->h_ids_t,
          c->cam.sw, c->cam.sh, (int)tiles_x_of(c), ld, c->h_map);
  HIPCHK(c, hipMemcpyAsync(map, c->h_map, bytes, hipMemcpyDeviceToHost, c->stream));
  if (minmax) HIPCHK(c, hipMemcpyAsync(minmax, c->h_mm, sizeof(float) * 6, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return RTGPU_OK;
}

int rtgpu_tonemap_pixelmap(rtgpu_ctx *c, const void *map, const rtgpu_pixel_layout *lay, int64_t n, const float *minmax, const float *gammas,
                           int32_t ngammas, uint32_t *rgba) {
  if (!c || !map || !minmax || !rgba || n < 0 || ngammas < 0 || (ngammas > 0 && !gammas)) return RTGPU_E_INVALID_PARAM;
  int rc = check_layout(c, lay);
  if (rc) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  if (n == 0) return RTGPU_OK;
  const size_t bytes = (size_t)n * (size_t)lay->stride;
  if ((rc = ensure_map(c, bytes))) return rc;
  if ((size_t)n > c->h_cap_tone) {
    c->h_cap_tone = 0;
    if ((rc = dev_alloc(c, &c->h_rgba, (size_t)n))) return rc;
    c->h_cap_tone = (size_t)n;
  }
  if (!c->h_mm && (rc = dev_alloc(c, &c->h_mm, 6))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->h_map, map, bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->h_mm, minmax, sizeof(float) * 6, hipMemcpyHostToDevice, c->stream));
  if ((rc = tonemap_strided(c, reinterpret_cast<const float *>(c->h_map + lay->color_offset), lay->stride / 4, n, c->h_mm, gammas, ngammas, c->h_rgba))) return rc;
  HIPCHK(c, hipMemcpyAsync(rgba, c->h_rgba, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return RTGPU_OK;
}

int rtgpu_tonemap_host(rtgpu_ctx *c, const float *rgb, int64_t n, const float *minmax, float gamma, uint32_t *rgba) {
  return rtgpu_tonemap_multi_host(c, rgb, n, minmax, &gamma, 1, rgba);
}

int rtgpu_tonemap_multi_host(rtgpu_ctx *c, const float *rgb, int64_t n, const float *minmax, const float *gammas, int32_t ngammas, uint32_t *rgba) {
  if (!c || !rgb || !minmax || !rgba || n < 0 || ngammas < 0 || (ngammas > 0 && !gammas)) return RTGPU_E_INVALID_PARAM;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = RTGPU_OK;
  if ((size_t)n > c->h_cap_px) {  // (shares the row-major colour buffer of rtgpu_render_host)
    c->h_cap_px = 0;
    if ((rc = dev_alloc(c, &c->h_rgb, (size_t)n * 3)) || (rc = dev_alloc(c, &c->h_ids, (size_t)n))) return rc;
    c->h_cap_px = (size_t)n;
  }
  if ((size_t)n > c->h_cap_tone) {
    c->h_cap_tone = 0;
    if ((rc = dev_alloc(c, &c->h_rgba, (size_t)n))) return rc;
    c->h_cap_tone = (size_t)n;
  }
  if (!c->h_mm && (rc = dev_alloc(c, &c->h_mm, 6))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->h_rgb, rgb, sizeof(float) * 3 * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->h_mm, minmax, sizeof(float) * 6, hipMemcpyHostToDevice, c->stream));
  if ((rc = tonemap_strided(c, c->h_rgb, 3, n, c->h_mm, gammas, ngammas, c->h_rgba))) return rc;
  HIPCHK(c, hipMemcpyAsync(rgba, c->h_rgba, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return RTGPU_OK;
}

int rtgpu_grid_download(rtgpu_ctx *c, int32_t *cell_start, int32_t *refs) {
  if (!c) return RTGPU_E_INVALID_PARAM;
  if (!c->have_frame) return RTGPU_E_STATE;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (cell_start) HIPCHK(c, hipMemcpy(cell_start, c->d_cell_start, sizeof(uint32_t) * ((size_t)c->ginfo.ncells + 1), hipMemcpyDeviceToHost));
  if (refs && c->ginfo.nrefs > 0) HIPCHK(c, hipMemcpy(refs, c->d_refs, sizeof(int32_t) * (size_t)c->ginfo.nrefs, hipMemcpyDeviceToHost));
  return RTGPU_OK;
}

int rtgpu_triangles_download(rtgpu_ctx *c, float *out10) {
  if (!c || !out10) return RTGPU_E_INVALID_PARAM;
  if (!c->have_frame) return RTGPU_E_STATE;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::vector<float> geom((size_t)c->nt * 9), aux((size_t)c->nt * 4);
  HIPCHK(c, hipMemcpy(geom.data(), c->d_geom, geom.size() * sizeof(float), hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(aux.data(), c->d_aux, aux.size() * sizeof(float), hipMemcpyDeviceToHost));
  for (int t = 0; t < c->nt; t++) {
    float *o = out10 + 10 * (size_t)t;
    for (int k = 0; k < 3; k++) { o[k] = aux[4 * (size_t)t + k]; o[3 + k] = geom[9 * (size_t)t + 3 + k]; o[6 + k] = geom[9 * (size_t)t + 6 + k]; }
    o[9] = aux[4 * (size_t)t + 3];
  }
  return RTGPU_OK;
}

}  // extern "C"
