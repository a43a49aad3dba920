/* rtgpu — C ABI of the MI355X (gfx950) implementation of the per-pixel trace path.
 *
 * This is the drop-in boundary.  The reference renders a frame with one call,
 *     RT_VisualizedScene* rtVisualizedSceneRaytrace(RT_Scene*, RT_Camera*)   (src/raytrace.h:57)
 * followed by rtVisualizedSceneToBitmap (src/raytrace.h:64).  A host that keeps the reference's
 * loaders and BMP writer replaces the body of those two functions with the calls below
 * (raytracer_amd/host/raytrace.c is a complete host built that way).
 *
 * Only plain-old-data crosses this boundary: counts, float/int arrays, opaque handle.  Every function
 * returns 0 on success or an RTGPU_E_* code; the codes 1..7 are the reference's errno values
 * (src/error.h:19-42) so the C caller can store them into errno unchanged.
 *
 * Pointers named d_* are DEVICE pointers (hipMalloc'd by the caller, e.g. a torch tensor's data_ptr);
 * all other pointers are host memory.  One context drives one GPU; contexts are not thread-safe.
 */
#ifndef RTGPU_H
#define RTGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTGPU_OK 0
#define RTGPU_E_IO 1               /* E_IO */
#define RTGPU_E_MEMORY 2           /* E_MEMORY */
#define RTGPU_E_NOT_ENOUGH_SURFACES 6 /* E_NOT_ENOUGH_SURFACES (src/scene.c:291-294) */
#define RTGPU_E_INVALID_PARAM 7    /* E_INVALID_PARAM_VALUE (src/voxelize.c:169-191) */
#define RTGPU_E_NO_DEVICE 100      /* no HIP device / HIP runtime error */
#define RTGPU_E_STATE 101          /* call order violated (e.g. render before frame_begin) */
#define RTGPU_E_UNSUPPORTED 102    /* combination the GPU path does not implement (see message) */

typedef struct rtgpu_ctx rtgpu_ctx;

/* RT_Surface (src/scene.h:32-36) without the unused alpha lane. */
typedef struct rtgpu_surface {
  float color[3];
  float kd, ks, g, ka, kt, eta, kr;
} rtgpu_surface;

/* RT_Light (src/scene.h:59-63). */
typedef struct rtgpu_light {
  float p[3];
  float flux;
  float color[3];
} rtgpu_light;

/* RT_PlanarLight (src/scene.h:67-73): origin a and the edge vectors ab, ac computed by the loader
 * (src/scene.c:496-498). */
typedef struct rtgpu_planar_light {
  float flux;
  float color[3];
  float a[3], ab[3], ac[3];
} rtgpu_planar_light;

/* RT_Camera (src/scene.h:109-113). */
typedef struct rtgpu_camera {
  float ob[3], ul[3], bl[3], ur[3];
  int32_t sw, sh;
} rtgpu_camera;

/* What rtSceneLoad / rtSceneSet* leave in RT_Scene (src/scene.h:88-105), flattened. */
typedef struct rtgpu_scene_desc {
  int32_t nt, ns, nl, npl;
  const float *vertices;            /* nt*9: i, j, k of each triangle AFTER the loader's vertex nudge (src/scene.c:138-181) */
  const int32_t *sid;               /* nt surface ids */
  const rtgpu_surface *surfaces;    /* ns */
  const rtgpu_light *lights;        /* nl (may be NULL when nl == 0) */
  const rtgpu_planar_light *planar; /* npl (may be NULL when npl == 0) */
  float dmin[3], dmax[3];           /* RT_Scene.dmin/dmax as the loader left them */
  float gamma, distmod;             /* RT_SceneConfig (src/scene.h:77-83) */
  int32_t vmode;                    /* 0 VOX_DEFAULT, 1 VOX_MODIFIED_DEFAULT, 2 VOX_FIXED */
  float vcoeff[3];
  int32_t texture_present;          /* 1 if "textures/brickwall.bmp" could be loaded (src/preprocess.c:13) */
} rtgpu_scene_desc;

/* Grid geometry of the current frame (RT_Udd, src/voxelize.h:21-25, plus the grown domain). */
typedef struct rtgpu_grid_info {
  int32_t nv[3];
  float cell[3];
  float dmin[3], dmax[3];
  int32_t ncells;
  int32_t nrefs;       /* total triangle references over all cells */
  float total_flux;    /* src/raytrace.c:346-348 */
} rtgpu_grid_info;

/* Ray statistics of the last rtgpu_render call.  "Rays" are counted as the reference issues them:
 * nearest-hit queries + shadow queries. */
typedef struct rtgpu_counters {
  int64_t pixels;          /* pixels of this rank's tiles inside the image */
  int64_t pixels_traced;   /* ... whose primary ray entered the grid */
  int64_t nearest_queries; /* rtUddFindNearestTriangle calls (src/voxelize.c:536) */
  int64_t shadow_queries;  /* rtUddFindShadow calls (src/voxelize.c:615), back-face early-outs included */
  int64_t shadow_walks;    /* shadow queries that reached the grid walk */
  int64_t nearest_steps, nearest_tests; /* S and T of the nearest walks (only with option "count" = 1) */
  int64_t shadow_steps, shadow_tests;   /* S and T of the shadow walks   (only with option "count" = 1) */
  /* SIMD efficiency of the two walk kernels (only with "count" = 1): triangle tests executed by
   * active lanes / lane slots the wavefronts spent on them (64 per triangle iteration). */
  int64_t nearest_lane_tests, nearest_lane_slots, shadow_lane_tests, shadow_lane_slots;
  /* How far the staged triangle filter had to go (only with "count" = 1): wavefront-level triangle
   * iterations in which at least one lane reached the second slab pair, the distance stage, and the
   * exact Moeller-Trumbore test (src/intersection.c:58-86).  Compare with *_lane_slots / 64. */
  int64_t nearest_stage2, nearest_stage3, nearest_exact, shadow_stage2, shadow_stage3, shadow_exact;
  /* Cell visits of the wavefronts: iterations of the walk kernels' outer loop (only with "count" = 1). */
  int64_t nearest_visits, shadow_visits;
  /* Batches of 8 listed triangles the cell-mode walks looked at, and how many of them were skipped because
   * no ray of the wavefront could meet the batch's bounding box (only with "count" = 1). */
  int64_t nearest_batches, nearest_batches_culled, shadow_batches, shadow_batches_culled;
  /* Exact tests (src/intersection.c:58-86) that ran to their last line, counted per ray (lane), not per wavefront:
   * 45 flops each in the reference's accounting (only with "count" = 1). */
  int64_t nearest_full_tests, shadow_full_tests;
  /* Chunks of tiles the last rtgpu_render call cut this rank's share into, and how many of them it rendered a second
   * time because a queue of secondary rays, sized from earlier chunks, was too small (option "queue_sizing"). */
  int64_t chunks, chunks_rerendered;
  /* Shadow tickets (64 shadow queries of one light) answered from the light's lists, and tickets the lists were tried
   * for but which took the walk after all (option "light_lists"; counted without "count" too). */
  int64_t shadow_list_tickets, shadow_list_fallbacks;
  /* Tickets of primary rays (the 64 pixels of a tile) answered from the frame's view lists, and tickets the lists were
   * tried for but which took the walk after all (option "view_lists"; counted without "count" too). */
  int64_t view_list_tickets, view_list_fallbacks;
} rtgpu_counters;

/* Per-kernel device time since the previous rtgpu_timings_get, summed per kernel family, from HIP
 * events recorded around every launch on the context's stream (only with option "timing" = 1; 2 = around the two
 * walk kernels only, the other members stay 0: the events of a whole frame cost 0.07 ms of a 5.5 ms frame). */
typedef struct rtgpu_timings {
  float preprocess_ms, grid_ms, primary_ms, nearest_ms, hit_ms, shadow_ms, accumulate_ms, minmax_ms;
  int32_t nearest_launches, shadow_launches;
} rtgpu_timings;

const char *rtgpu_error_string(int code);
/* Text of the last failure on this context (HIP error text, unsupported combination, ...). */
const char *rtgpu_last_message(const rtgpu_ctx *ctx);

/* Creates a context on HIP device `device` and its stream.  Fails with RTGPU_E_NO_DEVICE when there
 * is no usable GPU: there is no CPU fallback. */
int rtgpu_create(int device, rtgpu_ctx **out);
void rtgpu_destroy(rtgpu_ctx *ctx);

/* Makes the context enqueue on the caller's HIP stream (a hipStream_t, e.g. torch's current stream)
 * instead of its own, so that rendering orders with the caller's collectives without host syncs.
 * NULL switches back to the context's own stream. */
int rtgpu_set_stream(rtgpu_ctx *ctx, void *hip_stream);
/* The same for the device's null stream (the legacy default stream, handle 0: what torch.cuda.current_stream() is
 * unless the caller made another one current), which rtgpu_set_stream cannot name because NULL means "own". */
int rtgpu_set_null_stream(rtgpu_ctx *ctx);

/* Options: "count" (0/1) S/T counters, "timing" (0/1/2) per-kernel events (2: walk kernels only), "chunk_bytes" (device
 * memory budget of the ray/node queues, default 8 GiB), "waves_per_cu" (persistent grid size; 0 = the default:
 * what each walk kernel is compiled for, 20, and 24 or 28), "pipeline" (default 1: rtgpu_frame_begin builds on its
 * own stream into the buffers the previous frame does not use, overlapping that frame's kernels; 0: on the context's stream),
 * "queue_sizing" (default 1: in scenes with reflective or transparent surfaces the queues of the secondary rays are sized from
 * what the scene's chunks produce, which costs one host synchronisation per chunk, and a chunk whose queues turn out too small
 * is rendered again with worst-case queues; 0: always worst case, 31 nodes per pixel at depth 5, no synchronisation),
 * "queue_margin" (default 1500: queues are planned for the largest ratio seen so far x 1.5; in permille),
 * "order" (default 1: the shaded hits of every recursion level below the first get their shadow walks and colours in the order
 * of the grid cells their hit points lie in, so that a wavefront's 64 shadow rays start close together; 0: in queue order;
 * 2: the first level too; same image), "order_parts" (default 8: that order distinguishes this many parts of a cell along each
 * axis, as far as 2^25 buckets allow; 1, 2, 4 or 8),
 * "record_test" (default 1; 1 / 0: the walk kernels test the records of a batch against the whole bundle of a wavefront's rays
 * before they stage the batch, or do not; -1: only unless half of the scene's triangles or more are transparent; same image),
 * "bundle_ends" (default -1: in scenes with reflective or transparent surfaces; 1 / 0: the shadow walk's culling knows, or does
 * not, that a wavefront's shadow rays end at their lights and come closer together on the way; same image),
 * "fuse_hit" (default 1: a recursion level that neither spawns rays nor meets a textured triangle shades its hits at the end
 * of the nearest-hit walk, so that the hit records do not go through memory; 0: always in a kernel of its own; same image),
 * "filter" (default 1; 0 switches off the conservative triangle filter and the batch culling of the
 * walk kernels, so that every listed triangle gets the exact test: same image, slower; for validation),
 * "light_lists" (default 1: in a scene without a transparent triangle, with point lights and at most 2^20 pairs of a triangle
 * and a light, the shadow queries towards each point light are answered
 * from lists over the directions from that light, built once per scene on the host, where the lists can tell; the others take
 * the walk; 0: every shadow query walks; same image), "light_lists_after" (default 1: the lists are built when the scene is
 * rendered a second time, so that a scene rendered once does not pay for them; 0: with the first frame; N: with frame N + 1),
 * "light_list_levels" (default 4: 1..4 levels of texels, each twice as coarse as the one before), "light_list_texels"
 * (default 256: texels per face edge of the finest level, a power of two from 8 to 256), "light_list_kernel" (default 1: the
 * lists get a kernel of their own where a launch gives every wavefront many tiles; 0: never, they are read inside the shadow
 * walk's kernel; 2: wherever that kernel applies),
 * "view_lists" (default 1: with the triangle filter on and "count" off, the nearest-hit queries of the primary rays are answered
 * from lists over the screen, built with every frame's set-up on the device, each answer checked against the ray's own walk
 * through the grid up to the hit; 0: every primary ray walks; same image), "view_list_texel_shift" (default 1: a texel of those
 * lists is 8 << shift pixels wide and high; 0..4), "view_list_capacity" (default 0: the references of those lists get
 * max(2^20, 64 per texel + 16 per triangle) places; otherwise that many, up to 2^30; a frame whose lists need more takes the
 * walk), "view_list_test_fallback" (default 0; 1 makes every other tile reject its list's answer and walk: for tests),
 * "contexts_on_gpu" (default 1: how many contexts render on this GPU at the same time, 1..16; with more than one a scene with a
 * plain grid gets 12 wavefronts per CU in its walk kernels (8 with three contexts or more) instead of what they are compiled
 * for, so that the frames share the GPU's compute units instead of queueing for them; same image.  Every context keeps two
 * streams busy, its caller's and one of its own for the next frame's set-up: a process with several should start HIP with
 * as many hardware queues, GPU_MAX_HW_QUEUES=16 in its environment, or streams that share a queue take turns). */
int rtgpu_set_option(rtgpu_ctx *ctx, const char *name, int64_t value);

/* Copies the scene to the GPU (replaces RT_Scene as the renderer's input).  A scene equal in every array and parameter to
 * the one the context already holds is recognised (the context keeps a host copy to compare with) and costs a comparison. */
int rtgpu_scene_upload(rtgpu_ctx *ctx, const rtgpu_scene_desc *scene);

/* Everything rtVisualizedSceneRaytrace does before its pixel loop (src/raytrace.c:341-366):
 * per-triangle preprocess for this camera (src/preprocess.c:17-38), light-driven domain growth,
 * grid dimensions (rtUddCreate) and the grid build (rtUddVoxelize), all from the pristine scene:
 * unlike the reference (which mutates RT_Scene.dmin/dmax) this call is idempotent. */
int rtgpu_frame_begin(rtgpu_ctx *ctx, const rtgpu_camera *cam);
int rtgpu_grid_info_get(const rtgpu_ctx *ctx, rtgpu_grid_info *out);

/* Image tiling.  The image is cut into 8x8-pixel tiles, numbered row-major; rank r of `world`
 * owns tiles r, r+world, r+2*world, ...  Tile-major buffers hold 64 pixels per owned tile (pixel
 * (x,y) of a tile at slot (y&7)*8 + (x&7)); slots outside the image are padding. */
int64_t rtgpu_tile_count(const rtgpu_ctx *ctx, int rank, int world);
/* No counterpart in the reference (it has no queues): the largest number of tiles rtgpu_render traces in one chunk,
 * whatever "chunk_bytes" allows, for a scene whose hits spawn up to `fan` secondary rays over `levels` recursion levels
 * with `nslots` shadow queries per hit: node indices are 32-bit and so are the tickets of the walk kernels.  Needs no GPU. */
int64_t rtgpu_chunk_tile_limit(int fan, int levels, int nslots);

/* The pixel loop (src/raytrace.c:371-404) for the tiles of `rank`.
 *   d_rgb    : tile_count*64*3 floats, raw colour (RT_VisualizedScenePixel.c), 0 where the reference leaves the pixel unwritten
 *   d_ids    : tile_count*64 int32, index of the primary-hit triangle (RT_VisualizedScenePixel.t - scene->t) or -1
 *   d_minmax : 6 floats, min r,g,b then max r,g,b over this rank's traced pixels (src/raytrace.c:396-399),
 *              initialised to FLT_MAX / FLT_MIN like src/raytrace.c:317-320
 * Asynchronous on the context's stream; rtgpu_sync waits. */
int rtgpu_render(rtgpu_ctx *ctx, int rank, int world, float *d_rgb, int32_t *d_ids, float *d_minmax);

/* No counterpart in the reference.  rtgpu_render for callers that cannot allocate device memory (tight serving loops,
 * harnesses that forbid hipMalloc, code that records HIP graphs): same arguments, same outputs bit for bit, but the queues of
 * the pixel loop (hits, shaded-hit nodes, secondary rays, shadow results, partial colour ranges) live in a static device
 * array of rtgpu_static_scratch_bytes() = 64 MiB that every GPU the library runs on holds until the process ends; the
 * call makes no hipMalloc, hipFree or other allocating call (per-kernel "timing" is off for it: it would create events).
 * This rank's whole share of the image is traced as one chunk, so its queues have to fit the array (temple: 145 bytes per
 * pixel, about 460 000 pixels; ulica, 64 lights and depth 5, worst-case queues: about 13 KB per pixel); a larger share, a scene with planar
 * lights (their random-number table is sized at run time) or a device index above 63 is refused with
 * RTGPU_E_UNSUPPORTED and a message that says which, before anything is enqueued.  All contexts of one GPU share that
 * GPU's array and it keeps the last call's contents (nothing in it has to be cleared: every entry is written before
 * it is read).  Calls that follow each other on the host are ordered on the device by the library, whichever contexts and
 * streams they use (a call from another context or stream first waits for an event recorded behind the previous call);
 * calls issued at the same time from several host threads race.  On a stream that is being captured the call enqueues its
 * kernels and memsets and neither records nor waits for an event, so the library knows nothing of the launches of the
 * resulting graph: the caller orders them against each other, against direct calls and against a later
 * rtgpu_frame_begin (the tests do not cover capture and replay). */
int rtgpu_render_static(rtgpu_ctx *ctx, int rank, int world, float *d_rgb, int32_t *d_ids, float *d_minmax);
int64_t rtgpu_static_scratch_bytes(void);

/* rtgpu_render in two halves, for planar lights on several GPUs.  The reference draws rand() per shaded hit in
 * pixel order (src/raytrace.c:241-262, each pixel's ray tree in post-order), so a rank has to know how many shaded
 * hits the pixels before its own have, on whichever rank they are:
 *   rtgpu_render_trace   primary rays, nearest-hit walks and hit kernels of this rank's tiles (one chunk);
 *   rtgpu_hit_counts     d_counts[rtgpu_tile_count(rank, world) * 64]: shaded hits per pixel of this rank, tile-major
 *                        (rtgpu_hit_total: their sum);
 *   (the host gathers every rank's counts: d_all_counts[world][per_rank], per_rank >= the largest share, and
 *    their sum total_hits)
 *   rtgpu_render_shade   rand() order, shadow walks, colours, colour range: the outputs of rtgpu_render.
 * Without planar lights, or with world = 1, d_all_counts may be NULL and the pair equals rtgpu_render. */
int rtgpu_render_trace(rtgpu_ctx *ctx, int rank, int world);
int rtgpu_hit_counts(rtgpu_ctx *ctx, int32_t *d_counts);
/* ... and their sum, this rank's shaded hits of all recursion levels, as one 64-bit integer in device memory: a host
 * adds the ranks' sums up with a collective instead of downloading the counts (total_hits of rtgpu_render_shade). */
int rtgpu_hit_total(rtgpu_ctx *ctx, int64_t *d_total);
int rtgpu_render_shade(rtgpu_ctx *ctx, const int32_t *d_all_counts, int64_t per_rank, int64_t total_hits, float *d_rgb,
                       int32_t *d_ids, float *d_minmax);

/* rtVisualizedSceneToBitmap's per-pixel map (src/raytrace.c:433-466) on n pixels in any order:
 * rgba = R<<24 | G<<16 | B<<8 (src/bitmap.h:70) of trunc(pow((c-min)*delta, gamma)*255).
 * d_minmax holds the GLOBAL min/max (after the exchange between ranks). */
int rtgpu_tonemap(rtgpu_ctx *ctx, const float *d_rgb, int64_t npixels, const float *d_minmax, float gamma,
                  uint32_t *d_rgba);

/* The same with the zero-terminated gamma list rtVisualizedSceneToBitmap takes as its param1 (src/raytrace.c:439-447),
 * given as ngammas gammas in HOST memory (read before the call returns): per channel the powers are summed in list order
 * as src/raytrace.c:456-460 sums them and divided by (float)ngammas (src/raytrace.c:462-466).  ngammas = 1 is
 * rtgpu_tonemap; ngammas = 0, the reference's empty list, gives 0 / 0 = NaN per channel, i.e. black pixels. */
int rtgpu_tonemap_multi(rtgpu_ctx *ctx, const float *d_rgb, int64_t npixels, const float *d_minmax, const float *gammas,
                        int32_t ngammas, uint32_t *d_rgba);

/* De-interleaves tile-major data gathered from all ranks (rank r's block first, each block
 * rtgpu_tile_count(r)*64 elements of elem_bytes) into a row-major w*h image. */
int rtgpu_untile(rtgpu_ctx *ctx, const void *d_gathered, int world, int elem_bytes, void *d_image);

/* The pixel rows of rtBitmapSave (src/bitmap.c:453-469) built on the device from a row-major RGBA8 image (the
 * output of rtgpu_untile): bottom-up, B G R per pixel (bpp 24) or B G R 0 (bpp 32), rows padded to four bytes.
 * d_rows needs rtgpu_bmp_row_bytes(w, bpp) * h bytes; a host prepends the 54-byte header (rtgpu_bmp_header). */
int64_t rtgpu_bmp_row_bytes(int32_t width, int32_t bpp);
int rtgpu_bmp_rows(rtgpu_ctx *ctx, const uint32_t *d_image, int32_t bpp, void *d_rows);
/* The 54-byte file header rtBitmapSave writes for a width x height image of bpp bits. */
int rtgpu_bmp_header(int32_t width, int32_t height, int32_t bpp, unsigned char header[54]);

int rtgpu_sync(rtgpu_ctx *ctx);
int rtgpu_counters_get(rtgpu_ctx *ctx, rtgpu_counters *out);
int rtgpu_timings_get(rtgpu_ctx *ctx, rtgpu_timings *out);
/* Diagnostic, no counterpart in the reference: shader-clock cycles the wavefronts of the two walk kernels spent
 * in each of 8 phases (out[0..7] nearest walk, out[8..15] shadow walk: set-up, choice of group and mode, header
 * staging, batch culling, record staging, triangle tests, stepping, lane/ray modes), summed over all wavefronts
 * since the last reset.  All zero unless the library was built with -DRT_PHASES (tools/phases.sh). */
int rtgpu_phase_cycles(rtgpu_ctx *ctx, int64_t out[16], int reset);
/* Diagnostic, no counterpart in the reference: per ticket (64 consecutive rays of the level-0 queue) of the nearest
 * walk (which = 0) or shadow walk (which = 1) of the last chunk: shader clocks, cell visits, batches tested ray by
 * ray, records tested; 4 uint32 per ticket.  *copied = the number of tickets written, at most max_tickets; 0 unless
 * the library was built with -DRT_TICKETS (tools/tickets.py). */
int rtgpu_ticket_log(rtgpu_ctx *ctx, int which, uint32_t *out, int max_tickets, int *copied);

/* Diagnostic, no counterpart in the reference: the reference normalises vectors with `1.0f / sqrt(x)` and measures
 * distances with `sqrt(x)` on float x, which C evaluates in double and rounds to float once (src/vectormath.h:45-48,
 * 60-66).  The kernels compute the same floats with fewer double-precision instructions; this call compares the two
 * forms on the device for all 2^32 float inputs and returns the number of inputs whose results differ
 * (mismatches[0]: reciprocal square root, mismatches[1]: square root).  Both must be 0.  About 0.1 s. */
int rtgpu_math_selfcheck(rtgpu_ctx *ctx, uint64_t mismatches[2]);

/* Diagnostics of the light lists, no counterpart in the reference, host code only (no GPU needed).  In a scene without
 * transparent triangles the shadow queries towards a point light are answered from a table around the light (option
 * "light_lists"): the faces of a cube around the light cut into texels, each listing the triangles the reference's test
 * (src/intersection.c:58-86) can accept for a shadow ray (src/raytrace.c:206-216, src/voxelize.c:615-738) seen from the
 * light in that texel.  These two calls return what the table is filled from, so that its completeness can be checked
 * against the test itself.
 * rtgpu_light_list_pairs: for n (triangle, light) pairs (verts n x 9: i, j, k; lights n x 3) and rays whose origin
 * has a norm of at most olim, out[n][6][5] = per cube face (+x, -x, +y, -y, +z, -z) found (0 / 1) and the rectangle
 * u0, u1, v0, v1 in face coordinates (the two other axes in ascending order, divided by the face axis' magnitude; the
 * faces overlap: coordinates run over [-1.25, 1.25]).
 * rtgpu_light_list_rects: the whole table for a scene as the library would build it with R x R texels per window
 * (olim <= 0: chosen from the scene): faces[nl * 6 * 2][8] = per light, face and window (0: where most triangles are,
 * 1: the whole face, for rays outside window 0) p0, q0, sp, sq, p1, q1, base, on (the last two as uint32 bits): the
 * window and the scale of its texels; rects[*n_rects][3] = window index ((light * 6 + face) * 2 + window), packed
 * texel rectangle (columns a0 | a1 << 8, rows b0 << 16 | b1 << 24), triangle.  cap: room in rects (in rectangles);
 * *n_rects is the number there would be.  *refs: the sum of the rectangles' areas. */
int rtgpu_light_list_pairs(int64_t n, const float *verts, const float *lights, float olim, double *out);
int rtgpu_light_list_rects(int32_t nt, const float *verts, int32_t nl, const float *lights, int32_t R, float olim, int64_t cap,
                           uint32_t *rects, float *faces, float *olim_out, int64_t *n_rects, int64_t *refs);

/* Diagnostics of the view lists, no counterpart in the reference, host code only (no GPU needed).  The nearest-hit
 * queries of the primary rays (src/raytrace.c:306-340, src/voxelize.c:536-613) are answered from a table over the screen
 * (option "view_lists", built with every frame's set-up): the screen cut into texels of 8 x 8 pixels or a power of two
 * times that, each listing the triangles the reference's test (src/intersection.c:58-86) can accept for the ray of one of
 * its pixels.  This call returns what the table is filled from: for n triangles (verts n x 9: i, j, k) and one camera for
 * all of them (ncams = 1) or one each (ncams = n), out[n][5] = found (0 / 1) and the rectangle of pixels, columns a0..a1 and
 * rows b0..b1, outside which no pixel's ray can be accepted for the triangle. */
int rtgpu_view_list_rects(int64_t n, const float *verts, const rtgpu_camera *cams, int64_t ncams, int32_t *out);

/* Single-GPU convenience for a C host: frame_begin must have been called; renders all tiles and
 * returns row-major host arrays (rgb w*h*3, ids w*h, minmax 6).  Any output may be NULL. */
int rtgpu_render_host(rtgpu_ctx *ctx, float *rgb, int32_t *ids, float *minmax);
/* Host-array form of rtgpu_tonemap (row-major in, row-major out). */
int rtgpu_tonemap_host(rtgpu_ctx *ctx, const float *rgb, int64_t npixels, const float *minmax, float gamma,
                       uint32_t *rgba);

/* Host-array form of rtgpu_tonemap_multi. */
int rtgpu_tonemap_multi_host(rtgpu_ctx *ctx, const float *rgb, int64_t npixels, const float *minmax, const float *gammas,
                             int32_t ngammas, uint32_t *rgba);

/* The drop-in form of the frame for a C host that keeps the reference's result type: the frame's pixels arrive in the
 * caller's own records, RT_VisualizedScenePixel of src/raytrace.h:23-27 (RT_Color c; RT_Triangle *t), as
 * rtVisualizedSceneSetPixel fills them (src/raytrace.c:402), row-major, map[y * w + x] (src/raytrace.h:41-50).
 * The layout is described, not assumed: stride = sizeof of a record, color_offset = offset of the four colour floats,
 * tri_offset = offset of the triangle pointer (-1: no such member), tri_base = the address of scene->t[0] and tri_stride
 * = sizeof(RT_Triangle) in the CALLER's address space: the pointer of a pixel is tri_base + id * tri_stride, NULL without
 * a primary hit; the library computes it as a number and never follows it.  The colour's fourth float and the bytes of
 * a record that neither member covers are 0; pixels whose ray never entered the grid, which the reference leaves
 * unwritten (src/raytrace.c:381-382), are all zero. */
typedef struct rtgpu_pixel_layout {
  int64_t stride, color_offset, tri_offset;
  uint64_t tri_base;
  int64_t tri_stride;
} rtgpu_pixel_layout;
/* rtgpu_frame_begin must have been called.  Renders all tiles on this context's GPU and writes width * height records
 * to `map` (host memory; page-locked memory from rtgpu_pinned_alloc is filled at the full rate of the host link, about
 * four times as fast as pageable memory) and the colour range to minmax[6] (may be NULL).  Returns when both are there. */
int rtgpu_render_pixelmap(rtgpu_ctx *ctx, void *map, const rtgpu_pixel_layout *layout, float *minmax);
/* rtVisualizedSceneToBitmap (src/raytrace.c:424-472) on such records: npixels records of `map`, the frame's colour range
 * (RT_VisualizedScene.min / .max), the gamma list as for rtgpu_tonemap_multi; rgba[npixels] in host memory. */
int rtgpu_tonemap_pixelmap(rtgpu_ctx *ctx, const void *map, const rtgpu_pixel_layout *layout, int64_t npixels, const float *minmax,
                           const float *gammas, int32_t ngammas, uint32_t *rgba);
/* No counterpart in the reference: page-locked host memory for the big host buffers of the two calls above (NULL if it
 * cannot be had; a caller then falls back to malloc).  Allocating and freeing it is slow: keep it from frame to frame. */
void *rtgpu_pinned_alloc(size_t bytes);
void rtgpu_pinned_free(void *p);

/* Introspection used by the parity tests: grid lists (cell_start: ncells+1, refs: nrefs, ascending
 * triangle index inside a cell as src/voxelize.c:252-283 inserts them) and the per-triangle
 * preprocess results (n[3], ij[3], ik[3], d : 10 floats per triangle, src/preprocess.c:19-32). */
int rtgpu_grid_download(rtgpu_ctx *ctx, int32_t *cell_start, int32_t *refs);
int rtgpu_triangles_download(rtgpu_ctx *ctx, float *out10);

#ifdef __cplusplus
}
#endif
#endif /* RTGPU_H */
