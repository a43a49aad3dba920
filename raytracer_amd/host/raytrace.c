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
  if (!res || !(res->map = calloc(np ? np : 1, sizeof *res->map))) { errno = E_MEMORY; goto out; }
  res->width = camera->sw; res->height = camera->sh;
  res->gamma = scene->cfg.gamma > 0.0f ? scene->cfg.gamma : 2.5f; /* src/raytrace.c:333-337 */
  rtScenePreprocess(scene, camera);                               /* texture switch + texture coordinates */
  if (!(ctx = context())) goto out;
  if (!(flat = rtSceneFlatten(scene))) goto out;
  rtCameraFlatten(camera, &cam);
  rgb = malloc(sizeof(float) * 3 * (np ? np : 1));
  ids = malloc(sizeof(int32_t) * (np ? np : 1));
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
    res->map[p].t = ids[p] >= 0 ? &scene->t[ids[p]] : NULL;
  }
  ok = 1;
out:
  rtSceneFlatFree(&flat);
  free(rgb); free(ids); free(tri);
  if (!ok) rtVisualizedSceneDestroy(&res);
  return res;
}

void rtVisualizedSceneDestroy(RT_VisualizedScene **self) {
  if (*self) { free((*self)->map); free(*self); *self = NULL; }
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
  rgb = malloc(sizeof(float) * 3 * (np ? np : 1));
  if (!rgb) { errno = E_MEMORY; rtBitmapDestroy(&bmp); return NULL; }
  for (p = 0; p < np; p++) for (k = 0; k < 3; k++) rgb[3 * p + k] = s->map[p].c.c[k];
  for (k = 0; k < 3; k++) { mm[k] = s->min.c[k]; mm[3 + k] = s->max.c[k]; }
  if (gpu_error(ctx, rtgpu_tonemap_host(ctx, rgb, (int64_t)np, mm, gamma, bmp->pixels))) rtBitmapDestroy(&bmp);
  free(rgb);
  return bmp;
}
