/* rtVisualizedSceneRaytrace / rtVisualizedSceneToBitmap of the host layer: the reference's frame
 * function and tone map (src/raytrace.h:57-64) with their bodies replaced by calls into the GPU
 * library.  This file is the drop-in stub: flatten RT_Scene / RT_Camera into plain arrays, hand them
 * to rtgpu, copy the result into the structures the reference's callers read.
 */
#include "rt_host.h"

#include <float.h>
#include <stdlib.h>
#include <string.h>

static rtgpu_ctx *g_ctx = NULL;
static int g_device = 0;

void rtHostSetDevice(int device) {
  if (device != g_device) rtHostShutdown();
  g_device = device;
}

void rtHostShutdown(void) {
  if (g_ctx) { rtgpu_destroy(g_ctx); g_ctx = NULL; }
}

static rtgpu_ctx *context(void) {
  if (!g_ctx) {
    int rc = rtgpu_create(g_device, &g_ctx);
    if (rc) {
      fprintf(stderr, "rtgpu: %s (the trace path has no CPU implementation)\n", rtgpu_error_string(rc));
      errno = rc <= E_INVALID_PARAM_VALUE ? rc : E_IO;
      return NULL;
    }
  }
  return g_ctx;
}

static int gpu_error(rtgpu_ctx *ctx, int rc) {
  if (rc) {
    fprintf(stderr, "rtgpu: %s: %s\n", rtgpu_error_string(rc), rtgpu_last_message(ctx));
    errno = rc <= E_INVALID_PARAM_VALUE ? rc : E_INVALID_PARAM_VALUE;
  }
  return rc;
}


/* The frame's big host buffers (colours, ids, the pixel map) are kept from call to call: fresh pages for 330 MB per 4K
 * frame cost more than the GPU frame itself.  grab() hands out the cached block if it is large enough (taking it out of
 * the cache), keep() puts a block back in place of a smaller or absent one. */
struct held { void *p; size_t bytes; };
static struct held held_rgb, held_ids, held_map;
static void *grab(struct held *h, size_t bytes, int zero) {
  void *p;
  if (h->p && h->bytes >= bytes) { p = h->p; h->p = NULL; h->bytes = 0; if (zero) memset(p, 0, bytes); return p; }
  return zero ? calloc(1, bytes ? bytes : 1) : malloc(bytes ? bytes : 1);
}
static void keep(struct held *h, void *p, size_t bytes) {
  if (!p) return;
  if (!h->p || h->bytes < bytes) { free(h->p); h->p = p; h->bytes = bytes; } else free(p);
}

RT_VisualizedScene *rtVisualizedSceneRaytrace(RT_Scene *scene, RT_Camera *camera) {
  const size_t np = (size_t)camera->sw * (size_t)camera->sh;
  RT_VisualizedScene *res = calloc(1, sizeof *res);
  RT_SceneFlat *flat = NULL;
  rtgpu_ctx *ctx;
  rtgpu_camera cam;
  rtgpu_grid_info gi;
  float *rgb = NULL, *tri = NULL, mm[6];
  int32_t *ids = NULL;
  size_t p;
  int k, ok = 0;
  if (!res || !(res->map = grab(&held_map, (np ? np : 1) * sizeof *res->map, np == 0))) { errno = E_MEMORY; goto out; }
  res->width = camera->sw; res->height = camera->sh;
  res->gamma = scene->cfg.gamma > 0.0f ? scene->cfg.gamma : 2.5f; /* src/raytrace.c:333-337 */
  rtScenePreprocess(scene, camera);                               /* texture switch + texture coordinates */
  if (!(ctx = context())) goto out;
  if (!(flat = rtSceneFlatten(scene))) goto out;
  rtCameraFlatten(camera, &cam);
  rgb = grab(&held_rgb, sizeof(float) * 3 * (np ? np : 1), 0);
  ids = grab(&held_ids, sizeof(int32_t) * (np ? np : 1), 0);
  tri = malloc(sizeof(float) * 10 * (size_t)(scene->nt > 0 ? scene->nt : 1));
  if (!rgb || !ids || !tri) { errno = E_MEMORY; goto out; }
  if (gpu_error(ctx, rtgpu_scene_upload(ctx, &flat->desc))) goto out;
  if (gpu_error(ctx, rtgpu_frame_begin(ctx, &cam))) goto out;
  if (gpu_error(ctx, rtgpu_render_host(ctx, rgb, ids, mm))) goto out;
  if (gpu_error(ctx, rtgpu_grid_info_get(ctx, &gi))) goto out;
  if (gpu_error(ctx, rtgpu_triangles_download(ctx, tri))) goto out;
  /* What the reference leaves behind in its arguments: preprocessed triangles (src/preprocess.c:19-32)
   * and the grown domain (src/raytrace.c:350-353, src/voxelize.c:140-141). */
  for (k = 0; k < scene->nt; k++) {
    RT_Triangle *t = &scene->t[k];
    memcpy(t->n, tri + 10 * (size_t)k, 12); memcpy(t->ij, tri + 10 * (size_t)k + 3, 12); memcpy(t->ik, tri + 10 * (size_t)k + 6, 12);
    t->d = tri[10 * (size_t)k + 9];
  }
  for (k = 0; k < 3; k++) { scene->dmin[k] = gi.dmin[k]; scene->dmax[k] = gi.dmax[k]; }
  res->total_flux = gi.total_flux;
  for (k = 0; k < 3; k++) { res->min.c[k] = mm[k]; res->max.c[k] = mm[3 + k]; }
  res->min.c[3] = FLT_MAX; res->max.c[3] = FLT_MIN;
  for (p = 0; p < np; p++) {
    res->map[p].c.c[0] = rgb[3 * p]; res->map[p].c.c[1] = rgb[3 * p + 1]; res->map[p].c.c[2] = rgb[3 * p + 2];
    res->map[p].c.c[3] = 0.0f; /* (the reference's map comes from calloc) */
    res->map[p].t = ids[p] >= 0 ? &scene->t[ids[p]] : NULL;
  }
  ok = 1;
out:
  rtSceneFlatFree(&flat);
  keep(&held_rgb, rgb, sizeof(float) * 3 * (np ? np : 1)); keep(&held_ids, ids, sizeof(int32_t) * (np ? np : 1)); free(tri);
  if (!ok) rtVisualizedSceneDestroy(&res);
  return res;
}

void rtVisualizedSceneDestroy(RT_VisualizedScene **self) {
  if (*self) {
    keep(&held_map, (*self)->map, (size_t)((*self)->width > 0 ? (*self)->width : 0) * (size_t)((*self)->height > 0 ? (*self)->height : 0) * sizeof *(*self)->map);
    free(*self); *self = NULL;
  }
}

RT_Bitmap *rtVisualizedSceneToBitmap(RT_VisualizedScene *s, int flags, void *param1) {
  const size_t np = (size_t)s->width * (size_t)s->height;
  RT_Bitmap *bmp = rtBitmapCreate(s->width, s->height, 0);
  rtgpu_ctx *ctx;
  float *rgb, mm[6], gamma = s->gamma;
  size_t p;
  int k;
  if (!bmp) return NULL;
  if (!(flags & F_HDR)) return bmp; /* src/raytrace.c:437: nothing is drawn without F_HDR */
  if (param1) { /* zero-terminated gamma list (src/raytrace.c:439-447); the GPU tone map takes one gamma */
    const float *g = param1;
    if (!(g[0] > 0.0f) || g[1] > 0.0f) { errno = E_INVALID_PARAM_VALUE; rtBitmapDestroy(&bmp); return NULL; }
    gamma = g[0];
  }
  if (!(ctx = context())) { rtBitmapDestroy(&bmp); return NULL; }
  rgb = grab(&held_rgb, sizeof(float) * 3 * (np ? np : 1), 0);
  if (!rgb) { errno = E_MEMORY; rtBitmapDestroy(&bmp); return NULL; }
  for (p = 0; p < np; p++) for (k = 0; k < 3; k++) rgb[3 * p + k] = s->map[p].c.c[k];
  for (k = 0; k < 3; k++) { mm[k] = s->min.c[k]; mm[3 + k] = s->max.c[k]; }
  if (gpu_error(ctx, rtgpu_tonemap_host(ctx, rgb, (int64_t)np, mm, gamma, bmp->pixels))) rtBitmapDestroy(&bmp);
  keep(&held_rgb, rgb, sizeof(float) * 3 * (np ? np : 1));
  return bmp;
}
