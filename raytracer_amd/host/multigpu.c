/* One process, N GPUs: the frame split by image tiles over the devices of one node, exchanged with RCCL.
 *
 * The reference renders a frame in one thread (src/raytrace.c:371-404) and tone-maps it with the frame's global
 * colour range (src/raytrace.c:396-399, 433-435).  Here device r of N traces the 8x8-pixel tiles r, r+N, ... of the
 * image with its own rtgpu context (the scene is replicated), and one exchange per frame puts the image together:
 *   1. ncclAllReduce MIN of the three channel minima and MAX of the three maxima (the tone map needs the global range),
 *   2. every device tone-maps its own tiles to RGBA8,
 *   3. the RGBA8 tiles (4 bytes per pixel) go to device 0 with grouped ncclSend / ncclRecv: every peer has its own
 *      direct xGMI link to the root, so this is not a ring,
 *   4. device 0 de-interleaves the tiles (rtgpu_untile) and the host downloads the image.
 * Scenes with planar lights add an all-gather of the per-pixel hit counts between the two halves of the frame, which
 * fixes the order of the reference's rand() draws (src/raytrace.c:245-251): rtgpu_render_trace, rtgpu_hit_counts,
 * ncclAllGather, rtgpu_render_shade.
 * With N = 1 the same calls are issued to a communicator of one rank.
 * With N > 1 two frames are in flight (RTGPU_FRAMES_IN_FLIGHT overrides, 1..2): a device's share of a frame leaves most of
 * it idle for most of a walk kernel, so frame k runs in slot k mod 2, and every slot has its own contexts, streams,
 * buffers and its own communicators (collectives of one communicator must execute in the order they were issued on
 * every device, which two streams do not guarantee).
 *
 * Built as its own library (librtmulti.so) because it is the only part of the host layer that needs RCCL; the command
 * line loads it when --devices is given.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "rt_host.h"

#define RT_MAX_DEVICES 64
#define RT_MAX_DEPTH 2

void rtTilePlan(int32_t width, int32_t height, int ndev, int64_t *tiles, int64_t *first_pixel); /* scene.c */

typedef struct {
  rtgpu_ctx *ctx;
  hipStream_t stream;
  float *rgb, *mm;
  int32_t *ids, *counts, *all_counts;
  uint32_t *rgba;
  int64_t npix;
} Rank;

static int hip_failed(hipError_t e, const char *what) {
  if (e != hipSuccess) { fprintf(stderr, "rtHostRenderDevices: %s: %s\n", what, hipGetErrorString(e)); errno = E_IO; return 1; }
  return 0;
}
static int nccl_failed(ncclResult_t r, const char *what) {
  if (r != ncclSuccess) { fprintf(stderr, "rtHostRenderDevices: %s: %s\n", what, ncclGetErrorString(r)); errno = E_IO; return 1; }
  return 0;
}
static int gpu_failed(rtgpu_ctx *ctx, int rc, const char *what) {
  if (rc) {
    fprintf(stderr, "rtHostRenderDevices: %s: %s: %s\n", what, rtgpu_error_string(rc), rtgpu_last_message(ctx));
    errno = rc <= E_INVALID_PARAM_VALUE ? rc : E_INVALID_PARAM_VALUE;
  }
  return rc;
}
#define HIPOK(x) do { if (hip_failed((x), #x)) goto out; } while (0)
#define NCCLOK(x) do { if (nccl_failed((x), #x)) goto out; } while (0)
#define GPUOK(c, x) do { if (gpu_failed((c), (x), #x)) goto out; } while (0)

/* Renders `frames` frames of the scene on devices 0..ndev-1 and returns the last one as a bitmap (NULL + errno on
 * failure).  seconds (optional): wall time per frame of the frames after the first, everything included from
 * rtgpu_frame_begin to the finished image on device 0. */
RT_Bitmap *rtHostRenderDevices(RT_Scene *scene, RT_Camera *camera, int ndev, int frames, double *seconds) {
  static Rank slots[RT_MAX_DEPTH][RT_MAX_DEVICES];
  static ncclComm_t comms[RT_MAX_DEPTH][RT_MAX_DEVICES];
  Rank *rk;
  ncclComm_t *comm;
  int devs[RT_MAX_DEVICES], have_comm = 0, r, f, k, avail = 0, depth, sl;
  int64_t tiles[RT_MAX_DEVICES], first[RT_MAX_DEVICES + 1], most = 0;
  const int64_t np = (int64_t)camera->sw * camera->sh;
  const float gamma = scene->cfg.gamma > 0.0f ? scene->cfg.gamma : 2.5f; /* src/raytrace.c:333-337 */
  RT_SceneFlat *flat = NULL;
  RT_Bitmap *bmp = NULL;
  uint32_t *gathered_s[RT_MAX_DEPTH] = {NULL, NULL}, *image_s[RT_MAX_DEPTH] = {NULL, NULL}, *gathered, *image = NULL;
  const char *env = getenv("RTGPU_FRAMES_IN_FLIGHT");
  rtgpu_camera cam;
  struct timespec t0, t1;
  int ok = 0;

  memset(slots, 0, sizeof slots);
  depth = env ? atoi(env) : (ndev > 1 ? 2 : 1);
  if (depth < 1) depth = 1;
  if (depth > RT_MAX_DEPTH) depth = RT_MAX_DEPTH;
  if (ndev < 1 || ndev > RT_MAX_DEVICES || frames < 1) { errno = E_INVALID_PARAM_VALUE; return NULL; }
  if (hip_failed(hipGetDeviceCount(&avail), "hipGetDeviceCount") || avail < ndev) {
    fprintf(stderr, "rtHostRenderDevices: %d devices asked for, %d present\n", ndev, avail);
    errno = E_INVALID_PARAM_VALUE;
    return NULL;
  }
  rtScenePreprocess(scene, camera); /* texture switch + texture coordinates (src/preprocess.c:13,41-79) */
  if (!(flat = rtSceneFlatten(scene))) goto out;
  rtCameraFlatten(camera, &cam);
  rtTilePlan(camera->sw, camera->sh, ndev, tiles, first);
  for (r = 0; r < ndev; r++) { devs[r] = r; if (tiles[r] > most) most = tiles[r]; }
  for (sl = 0; sl < depth; sl++) {
    rk = slots[sl]; comm = comms[sl];
    NCCLOK(ncclCommInitAll(comm, ndev, devs));
    have_comm = sl + 1;
    for (r = 0; r < ndev; r++) {
      Rank *q = &rk[r];
      q->npix = tiles[r] * 64;
      HIPOK(hipSetDevice(r));
      GPUOK(NULL, rtgpu_create(r, &q->ctx));
      HIPOK(hipStreamCreateWithFlags(&q->stream, hipStreamNonBlocking));
      GPUOK(q->ctx, rtgpu_set_stream(q->ctx, q->stream));
      GPUOK(q->ctx, rtgpu_scene_upload(q->ctx, &flat->desc));
      HIPOK(hipMalloc((void **)&q->rgb, sizeof(float) * 3 * (size_t)(q->npix + 1)));
      HIPOK(hipMalloc((void **)&q->ids, sizeof(int32_t) * (size_t)(q->npix + 1)));
      HIPOK(hipMalloc((void **)&q->rgba, sizeof(uint32_t) * (size_t)(most * 64 + 1)));
      HIPOK(hipMalloc((void **)&q->mm, sizeof(float) * 6));
      if (scene->npl > 0) { /* every rank's counts padded to the largest share */
        HIPOK(hipMalloc((void **)&q->counts, sizeof(int32_t) * (size_t)(most * 64 + 1)));
        HIPOK(hipMalloc((void **)&q->all_counts, sizeof(int32_t) * (size_t)(most * 64 * ndev + 1)));
        HIPOK(hipMemsetAsync(q->counts, 0, sizeof(int32_t) * (size_t)(most * 64 + 1), q->stream));
      }
    }
    HIPOK(hipSetDevice(0));
    HIPOK(hipMalloc((void **)&gathered_s[sl], sizeof(uint32_t) * (size_t)(first[ndev] + 1)));
    HIPOK(hipMalloc((void **)&image_s[sl], sizeof(uint32_t) * (size_t)(np + 1)));
  }

  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (f = 0; f < frames; f++) {
    if (f == depth) { /* the first frame of every slot pays for queues, code objects and the communicator's first use */
      for (sl = 0; sl < depth; sl++)
        for (r = 0; r < ndev; r++) { HIPOK(hipSetDevice(r)); HIPOK(hipStreamSynchronize(slots[sl][r].stream)); }
      clock_gettime(CLOCK_MONOTONIC, &t0);
    }
    rk = slots[f % depth]; comm = comms[f % depth]; gathered = gathered_s[f % depth]; image = image_s[f % depth];
    for (r = 0; r < ndev; r++) GPUOK(rk[r].ctx, rtgpu_frame_begin(rk[r].ctx, &cam));
    if (scene->npl > 0) {
      int64_t total = 0;
      int32_t *host_counts = malloc(sizeof(int32_t) * (size_t)(most * 64 * ndev + 1));
      if (!host_counts) { errno = E_MEMORY; goto out; }
      for (r = 0; r < ndev; r++) {
        GPUOK(rk[r].ctx, rtgpu_render_trace(rk[r].ctx, r, ndev));
        GPUOK(rk[r].ctx, rtgpu_hit_counts(rk[r].ctx, rk[r].counts));
      }
      NCCLOK(ncclGroupStart());
      for (r = 0; r < ndev; r++)
        NCCLOK(ncclAllGather(rk[r].counts, rk[r].all_counts, (size_t)(most * 64), ncclInt32, comm[r], rk[r].stream));
      NCCLOK(ncclGroupEnd());
      HIPOK(hipSetDevice(0));
      HIPOK(hipMemcpyAsync(host_counts, rk[0].all_counts, sizeof(int32_t) * (size_t)(most * 64 * ndev), hipMemcpyDeviceToHost, rk[0].stream));
      HIPOK(hipStreamSynchronize(rk[0].stream));
      for (k = 0; k < ndev; k++) { int64_t p; for (p = 0; p < tiles[k] * 64; p++) total += host_counts[(int64_t)k * most * 64 + p]; }
      free(host_counts);
      for (r = 0; r < ndev; r++)
        GPUOK(rk[r].ctx, rtgpu_render_shade(rk[r].ctx, rk[r].all_counts, most * 64, total, rk[r].rgb, rk[r].ids, rk[r].mm));
    } else {
      for (r = 0; r < ndev; r++) GPUOK(rk[r].ctx, rtgpu_render(rk[r].ctx, r, ndev, rk[r].rgb, rk[r].ids, rk[r].mm));
    }
    /* exchange 1: the global colour range (replaces the running min/max of src/raytrace.c:396-399) */
    NCCLOK(ncclGroupStart());
    for (r = 0; r < ndev; r++) {
      NCCLOK(ncclAllReduce(rk[r].mm, rk[r].mm, 3, ncclFloat, ncclMin, comm[r], rk[r].stream));
      NCCLOK(ncclAllReduce(rk[r].mm + 3, rk[r].mm + 3, 3, ncclFloat, ncclMax, comm[r], rk[r].stream));
    }
    NCCLOK(ncclGroupEnd());
    for (r = 0; r < ndev; r++) GPUOK(rk[r].ctx, rtgpu_tonemap(rk[r].ctx, rk[r].rgb, rk[r].npix, rk[r].mm, gamma, rk[r].rgba));
    /* exchange 2: RGBA8 tiles to device 0, rank-major (the layout rtgpu_untile reads) */
    NCCLOK(ncclGroupStart());
    for (r = 0; r < ndev; r++) {
      if (rk[r].npix == 0) continue;
      NCCLOK(ncclSend(rk[r].rgba, (size_t)rk[r].npix, ncclUint32, 0, comm[r], rk[r].stream));
      NCCLOK(ncclRecv(gathered + first[r], (size_t)rk[r].npix, ncclUint32, r, comm[0], rk[0].stream));
    }
    NCCLOK(ncclGroupEnd());
    GPUOK(rk[0].ctx, rtgpu_untile(rk[0].ctx, gathered, ndev, 4, image));
  }
  HIPOK(hipSetDevice(0));
  for (sl = 0; sl < depth; sl++) HIPOK(hipStreamSynchronize(slots[sl][0].stream));  /* (image = the last frame's) */
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (seconds) *seconds = ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec)) / (frames > depth ? frames - depth : 1);
  if (!(bmp = rtBitmapCreate(camera->sw, camera->sh, 0))) goto out;
  HIPOK(hipMemcpy(bmp->pixels, image, sizeof(uint32_t) * (size_t)np, hipMemcpyDeviceToHost));
  ok = 1;
out:
  for (sl = 0; sl < RT_MAX_DEPTH; sl++) {
    for (r = 0; r < ndev; r++) {
      Rank *q = &slots[sl][r];
      (void)hipSetDevice(r);
      if (q->stream) (void)hipStreamSynchronize(q->stream);
      if (q->ctx) { (void)rtgpu_set_stream(q->ctx, NULL); rtgpu_destroy(q->ctx); }
      (void)hipFree(q->rgb); (void)hipFree(q->ids); (void)hipFree(q->rgba); (void)hipFree(q->mm); (void)hipFree(q->counts); (void)hipFree(q->all_counts);
      if (q->stream) (void)hipStreamDestroy(q->stream);
    }
    (void)hipSetDevice(0);
    (void)hipFree(gathered_s[sl]); (void)hipFree(image_s[sl]);
    if (sl < have_comm) for (r = 0; r < ndev; r++) (void)ncclCommDestroy(comms[sl][r]);
  }
  rtSceneFlatFree(&flat);
  if (!ok) rtBitmapDestroy(&bmp);
  return bmp;
}
