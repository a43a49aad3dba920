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
 * Scenes with planar lights add an all-gather of the per-pixel hit counts, and an all-reduce of their sums, between the
 * two halves of the frame, which fixes the order of the reference's rand() draws (src/raytrace.c:245-251):
 * rtgpu_render_trace, rtgpu_hit_counts, rtgpu_hit_total, ncclAllGather + ncclAllReduce, rtgpu_render_shade.
 * With N = 1 the same calls are issued to a communicator of one rank.
 *
 * One host thread per device makes the per-device calls of a frame (rtgpu_frame_begin, rtgpu_render): in scenes with
 * reflective or transparent surfaces rtgpu_render looks at its queue lengths once per chunk of tiles, i.e. waits for its
 * device, and one thread calling it for device 0, 1, ... in turn would start every device only when the one before it
 * had finished its nearest-hit walks.  The threads live as long as the call; the host thread hands them a job, waits for
 * all of them and issues the collectives itself (grouped: all communicators in one thread).
 * With N > 1 two frames are in flight (RTGPU_FRAMES_IN_FLIGHT overrides, 1..2): a device's share of a frame leaves most of
 * it idle for most of a walk kernel, so frame k runs in slot k mod 2, and every slot has its own contexts, streams,
 * buffers and its own communicators (collectives of one communicator must execute in the order they were issued on
 * every device, which two streams do not guarantee).
 *
 * Rehearsal on one GPU: RTGPU_LOOPBACK=1 puts all N ranks (contexts, streams, threads) on device 0 and replaces the
 * collectives, which RCCL cannot run between ranks that share a device, by copies on that device made by the host thread
 * (the colour range and the hit totals go through the host).  Everything else, the threads included, is the code that
 * runs on N devices.  RTGPU_TRACE=1 prints when every per-device call of a frame started and returned.
 *
 * Built as its own library (librtmulti.so) because it is the only part of the host layer that needs RCCL; the command
 * line loads it when --devices is given.  Not re-entrant: one rtHostRenderDevices at a time per process.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <pthread.h>
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
  int64_t *total; /* this rank's shaded hits, then (after the all-reduce) the frame's */
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

static double now_ms(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

/* ---- one host thread per device ---- */
enum { JOB_RENDER = 1, JOB_TRACE, JOB_SHADE, JOB_QUIT };
static const char *job_name[] = {"", "frame_begin + render", "frame_begin + render_trace + hit_counts", "render_shade", "quit"};
static struct {
  int ndev, started;
  pthread_t th[RT_MAX_DEVICES];
  pthread_barrier_t go, done;
  /* the job, written by the host thread before it releases the workers */
  int job;
  Rank *rk;
  const rtgpu_camera *cam;
  int64_t per_rank, total_hits;
  /* what every worker reports */
  int rc[RT_MAX_DEVICES];
  double t_start[RT_MAX_DEVICES], t_end[RT_MAX_DEVICES];
} pool;

static void *worker(void *arg) {
  const int r = (int)(intptr_t)arg;
  for (;;) {
    Rank *q;
    int rc = 0;
    pthread_barrier_wait(&pool.go);
    if (pool.job == JOB_QUIT) break;
    q = &pool.rk[r];
    pool.t_start[r] = now_ms();
    if (pool.job == JOB_RENDER) {
      if (!(rc = rtgpu_frame_begin(q->ctx, pool.cam))) rc = rtgpu_render(q->ctx, r, pool.ndev, q->rgb, q->ids, q->mm);
    } else if (pool.job == JOB_TRACE) {
      if (!(rc = rtgpu_frame_begin(q->ctx, pool.cam)) && !(rc = rtgpu_render_trace(q->ctx, r, pool.ndev)) &&
          !(rc = rtgpu_hit_counts(q->ctx, q->counts)))
        rc = rtgpu_hit_total(q->ctx, q->total);
    } else if (pool.job == JOB_SHADE) {
      rc = rtgpu_render_shade(q->ctx, q->all_counts, pool.per_rank, pool.total_hits, q->rgb, q->ids, q->mm);
    }
    pool.t_end[r] = now_ms();
    pool.rc[r] = rc;
    pthread_barrier_wait(&pool.done);
  }
  return NULL;
}
static int pool_start(int ndev) {
  int r;
  memset(&pool, 0, sizeof pool);
  pool.ndev = ndev;
  if (pthread_barrier_init(&pool.go, NULL, (unsigned)ndev + 1) || pthread_barrier_init(&pool.done, NULL, (unsigned)ndev + 1)) { errno = E_MEMORY; return 1; }
  for (r = 0; r < ndev; r++) {
    if (pthread_create(&pool.th[r], NULL, worker, (void *)(intptr_t)r)) { errno = E_MEMORY; break; }
    pool.started++;
  }
  if (pool.started < ndev) { /* (the barriers count ndev + 1 threads, so the ones that exist stay where they wait; the call fails) */
    pool.started = 0;
    return 1;
  }
  return 0;
}
static void pool_stop(void) {
  int r;
  if (!pool.started) return;
  pool.job = JOB_QUIT;
  pthread_barrier_wait(&pool.go);
  for (r = 0; r < pool.started; r++) pthread_join(pool.th[r], NULL);
  pthread_barrier_destroy(&pool.go); pthread_barrier_destroy(&pool.done);
  pool.started = 0;
}
/* All devices do `job` for the frame of slot rk; returns when every call has returned.  1 = some call failed. */
static int pool_run(int job, Rank *rk, const rtgpu_camera *cam, int frame, double t0, int trace) {
  int r, bad = 0;
  pool.job = job; pool.rk = rk; pool.cam = cam;
  pthread_barrier_wait(&pool.go);
  pthread_barrier_wait(&pool.done);
  for (r = 0; r < pool.ndev; r++) {
    if (trace)
      fprintf(stderr, "trace: frame %d rank %d %s: started %.3f ms, returned %.3f ms\n", frame, r, job_name[job], pool.t_start[r] - t0, pool.t_end[r] - t0);
    if (pool.rc[r] && !bad) { gpu_failed(rk[r].ctx, pool.rc[r], job_name[job]); bad = 1; }
  }
  return bad;
}

/* ---- the exchanges: RCCL, or copies on the one device of a rehearsal (loop) ---- */
/* colour range: MIN of the minima, MAX of the maxima, on every rank (replaces the running min/max of src/raytrace.c:396-399) */
static int exchange_range(Rank *rk, ncclComm_t *comm, int ndev, int loop) {
  int r, k;
  if (!loop) {
    NCCLOK(ncclGroupStart());
    for (r = 0; r < ndev; r++) {
      NCCLOK(ncclAllReduce(rk[r].mm, rk[r].mm, 3, ncclFloat, ncclMin, comm[r], rk[r].stream));
      NCCLOK(ncclAllReduce(rk[r].mm + 3, rk[r].mm + 3, 3, ncclFloat, ncclMax, comm[r], rk[r].stream));
    }
    NCCLOK(ncclGroupEnd());
  } else {
    float h[6], g[6];
    for (r = 0; r < ndev; r++) {
      HIPOK(hipMemcpyAsync(h, rk[r].mm, sizeof h, hipMemcpyDeviceToHost, rk[r].stream));
      HIPOK(hipStreamSynchronize(rk[r].stream));
      for (k = 0; k < 6; k++) if (r == 0 || (k < 3 ? h[k] < g[k] : h[k] > g[k])) g[k] = h[k];
    }
    for (r = 0; r < ndev; r++) {
      HIPOK(hipMemcpyAsync(rk[r].mm, g, sizeof g, hipMemcpyHostToDevice, rk[r].stream));
      HIPOK(hipStreamSynchronize(rk[r].stream));
    }
  }
  return 0;
out:
  return 1;
}
/* RGBA8 tiles to rank 0, rank-major (the layout rtgpu_untile reads) */
static int exchange_tiles(Rank *rk, ncclComm_t *comm, int ndev, int loop, uint32_t *gathered, const int64_t *first) {
  int r;
  if (!loop) {
    NCCLOK(ncclGroupStart());
    for (r = 0; r < ndev; r++) {
      if (rk[r].npix == 0) continue;
      NCCLOK(ncclSend(rk[r].rgba, (size_t)rk[r].npix, ncclUint32, 0, comm[r], rk[r].stream));
      NCCLOK(ncclRecv(gathered + first[r], (size_t)rk[r].npix, ncclUint32, r, comm[0], rk[0].stream));
    }
    NCCLOK(ncclGroupEnd());
  } else {
    for (r = 0; r < ndev; r++) {
      if (rk[r].npix == 0) continue;
      if (r > 0) HIPOK(hipStreamSynchronize(rk[r].stream)); /* its tiles are tone-mapped */
      HIPOK(hipMemcpyAsync(gathered + first[r], rk[r].rgba, sizeof(uint32_t) * (size_t)rk[r].npix, hipMemcpyDeviceToDevice, rk[0].stream));
    }
  }
  return 0;
out:
  return 1;
}
/* planar lights: every rank's hit counts (padded to `most` tiles) on every rank, and the frame's number of shaded hits */
static int exchange_counts(Rank *rk, ncclComm_t *comm, int ndev, int loop, int64_t most, int64_t *total) {
  const size_t n = (size_t)(most * 64);
  int64_t one;
  int r, k;
  *total = 0;
  if (!loop) {
    NCCLOK(ncclGroupStart());
    for (r = 0; r < ndev; r++) {
      NCCLOK(ncclAllGather(rk[r].counts, rk[r].all_counts, n, ncclInt32, comm[r], rk[r].stream));
      NCCLOK(ncclAllReduce(rk[r].total, rk[r].total, 1, ncclInt64, ncclSum, comm[r], rk[r].stream));
    }
    NCCLOK(ncclGroupEnd());
    /* (the table of random numbers is sized on the host: eight bytes come down, not the counts) */
    HIPOK(hipMemcpyAsync(total, rk[0].total, sizeof *total, hipMemcpyDeviceToHost, rk[0].stream));
    HIPOK(hipStreamSynchronize(rk[0].stream));
  } else {
    for (r = 0; r < ndev; r++) {
      HIPOK(hipMemcpyAsync(&one, rk[r].total, sizeof one, hipMemcpyDeviceToHost, rk[r].stream));
      HIPOK(hipStreamSynchronize(rk[r].stream)); /* ... and its counts are complete */
      *total += one;
    }
    for (r = 0; r < ndev; r++)
      for (k = 0; k < ndev; k++)
        HIPOK(hipMemcpyAsync(rk[r].all_counts + (size_t)k * n, rk[k].counts, sizeof(int32_t) * n, hipMemcpyDeviceToDevice, rk[r].stream));
  }
  return 0;
out:
  return 1;
}

/* Renders `frames` frames of the scene on devices 0..ndev-1 and returns the last one as a bitmap (NULL + errno on
 * failure).  seconds (optional): wall time per frame of the frames after the first, everything included from
 * rtgpu_frame_begin to the finished image on device 0. */
RT_Bitmap *rtHostRenderDevices(RT_Scene *scene, RT_Camera *camera, int ndev, int frames, double *seconds) {
  static Rank slots[RT_MAX_DEPTH][RT_MAX_DEVICES];
  static ncclComm_t comms[RT_MAX_DEPTH][RT_MAX_DEVICES];
  Rank *rk;
  ncclComm_t *comm;
  int devs[RT_MAX_DEVICES], have_comm = 0, r, f, avail = 0, depth, sl;
  int64_t tiles[RT_MAX_DEVICES], first[RT_MAX_DEVICES + 1], most = 0;
  const int64_t np = (int64_t)camera->sw * camera->sh;
  const float gamma = scene->cfg.gamma > 0.0f ? scene->cfg.gamma : 2.5f; /* src/raytrace.c:333-337 */
  RT_SceneFlat *flat = NULL;
  RT_Bitmap *bmp = NULL;
  uint32_t *gathered_s[RT_MAX_DEPTH] = {NULL, NULL}, *image_s[RT_MAX_DEPTH] = {NULL, NULL}, *gathered, *image = NULL;
  const char *env = getenv("RTGPU_FRAMES_IN_FLIGHT"), *env_loop = getenv("RTGPU_LOOPBACK"), *env_trace = getenv("RTGPU_TRACE");
  const int loop = env_loop && atoi(env_loop) > 0, trace = env_trace && atoi(env_trace) > 0;
  rtgpu_camera cam;
  double t0, t_frame;
  int ok = 0;

  memset(slots, 0, sizeof slots);
  depth = env ? atoi(env) : (ndev > 1 ? 2 : 1);
  if (depth < 1) depth = 1;
  if (depth > RT_MAX_DEPTH) depth = RT_MAX_DEPTH;
  if (ndev < 1 || ndev > RT_MAX_DEVICES || frames < 1) { errno = E_INVALID_PARAM_VALUE; return NULL; }
  if (hip_failed(hipGetDeviceCount(&avail), "hipGetDeviceCount") || avail < (loop ? 1 : ndev)) {
    fprintf(stderr, "rtHostRenderDevices: %d devices asked for, %d present\n", ndev, avail);
    errno = E_INVALID_PARAM_VALUE;
    return NULL;
  }
  rtScenePreprocess(scene, camera); /* texture switch + texture coordinates (src/preprocess.c:13,41-79) */
  if (!(flat = rtSceneFlatten(scene))) goto out;
  rtCameraFlatten(camera, &cam);
  rtTilePlan(camera->sw, camera->sh, ndev, tiles, first);
  for (r = 0; r < ndev; r++) { devs[r] = loop ? 0 : r; if (tiles[r] > most) most = tiles[r]; }
  for (sl = 0; sl < depth; sl++) {
    rk = slots[sl]; comm = comms[sl];
    if (!loop) {
      NCCLOK(ncclCommInitAll(comm, ndev, devs));
      have_comm = sl + 1;
    }
    for (r = 0; r < ndev; r++) {
      Rank *q = &rk[r];
      q->npix = tiles[r] * 64;
      HIPOK(hipSetDevice(devs[r]));
      GPUOK(NULL, rtgpu_create(devs[r], &q->ctx));
      HIPOK(hipStreamCreateWithFlags(&q->stream, hipStreamNonBlocking));
      GPUOK(q->ctx, rtgpu_set_stream(q->ctx, q->stream));
      /* frames in flight share the device: the library sizes the walk kernels of scenes without secondary rays so that
       * the other frame's find room (one rank's share of eight, temple 4K, two in flight: 0.75 -> 0.64 ms per frame) */
      if (depth > 1) GPUOK(q->ctx, rtgpu_set_option(q->ctx, "contexts_on_gpu", depth));
      if (frames > 1) GPUOK(q->ctx, rtgpu_set_option(q->ctx, "light_lists_after", 0)); /* a sequence of frames: the lists from the first on */
      GPUOK(q->ctx, rtgpu_scene_upload(q->ctx, &flat->desc));
      HIPOK(hipMalloc((void **)&q->rgb, sizeof(float) * 3 * (size_t)(q->npix + 1)));
      HIPOK(hipMalloc((void **)&q->ids, sizeof(int32_t) * (size_t)(q->npix + 1)));
      HIPOK(hipMalloc((void **)&q->rgba, sizeof(uint32_t) * (size_t)(most * 64 + 1)));
      HIPOK(hipMalloc((void **)&q->mm, sizeof(float) * 6));
      if (scene->npl > 0) { /* every rank's counts padded to the largest share */
        HIPOK(hipMalloc((void **)&q->counts, sizeof(int32_t) * (size_t)(most * 64 + 1)));
        HIPOK(hipMalloc((void **)&q->all_counts, sizeof(int32_t) * (size_t)(most * 64 * ndev + 1)));
        HIPOK(hipMalloc((void **)&q->total, sizeof(int64_t)));
        HIPOK(hipMemsetAsync(q->counts, 0, sizeof(int32_t) * (size_t)(most * 64 + 1), q->stream));
      }
    }
    HIPOK(hipSetDevice(0));
    HIPOK(hipMalloc((void **)&gathered_s[sl], sizeof(uint32_t) * (size_t)(first[ndev] + 1)));
    HIPOK(hipMalloc((void **)&image_s[sl], sizeof(uint32_t) * (size_t)(np + 1)));
  }
  if (pool_start(ndev)) goto out;

  t0 = now_ms();
  for (f = 0; f < frames; f++) {
    if (f == depth) { /* the first frame of every slot pays for queues, code objects and the communicator's first use */
      for (sl = 0; sl < depth; sl++)
        for (r = 0; r < ndev; r++) { HIPOK(hipSetDevice(devs[r])); HIPOK(hipStreamSynchronize(slots[sl][r].stream)); }
      t0 = now_ms();
    }
    t_frame = now_ms();
    rk = slots[f % depth]; comm = comms[f % depth]; gathered = gathered_s[f % depth]; image = image_s[f % depth];
    if (scene->npl > 0) {
      if (pool_run(JOB_TRACE, rk, &cam, f, t_frame, trace)) goto out;
      if (exchange_counts(rk, comm, ndev, loop, most, &pool.total_hits)) goto out;
      pool.per_rank = most * 64;
      if (pool_run(JOB_SHADE, rk, &cam, f, t_frame, trace)) goto out;
    } else {
      if (pool_run(JOB_RENDER, rk, &cam, f, t_frame, trace)) goto out;
    }
    if (exchange_range(rk, comm, ndev, loop)) goto out;
    for (r = 0; r < ndev; r++) GPUOK(rk[r].ctx, rtgpu_tonemap(rk[r].ctx, rk[r].rgb, rk[r].npix, rk[r].mm, gamma, rk[r].rgba));
    if (exchange_tiles(rk, comm, ndev, loop, gathered, first)) goto out;
    GPUOK(rk[0].ctx, rtgpu_untile(rk[0].ctx, gathered, ndev, 4, image));
  }
  HIPOK(hipSetDevice(0));
  for (sl = 0; sl < depth; sl++) HIPOK(hipStreamSynchronize(slots[sl][0].stream));  /* (image = the last frame's) */
  if (seconds) *seconds = (now_ms() - t0) * 1e-3 / (frames > depth ? frames - depth : 1);
  if (!(bmp = rtBitmapCreate(camera->sw, camera->sh, 0))) goto out;
  HIPOK(hipMemcpy(bmp->pixels, image, sizeof(uint32_t) * (size_t)np, hipMemcpyDeviceToHost));
  ok = 1;
out:
  pool_stop();
  for (sl = 0; sl < RT_MAX_DEPTH; sl++) {
    for (r = 0; r < ndev; r++) {
      Rank *q = &slots[sl][r];
      (void)hipSetDevice(loop ? 0 : r);
      if (q->stream) (void)hipStreamSynchronize(q->stream);
      if (q->ctx) { (void)rtgpu_set_stream(q->ctx, NULL); rtgpu_destroy(q->ctx); }
      (void)hipFree(q->rgb); (void)hipFree(q->ids); (void)hipFree(q->rgba); (void)hipFree(q->mm); (void)hipFree(q->counts); (void)hipFree(q->all_counts);
      (void)hipFree(q->total);
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
