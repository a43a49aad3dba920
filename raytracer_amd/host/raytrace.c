/* rtVisualizedSceneRaytrace / rtVisualizedSceneToBitmap of the host layer: the reference's frame
 * function and tone map (src/raytrace.h:57-64) with their bodies replaced by calls into the GPU
 * library.  This file is the drop-in stub: flatten RT_Scene / RT_Camera into plain arrays, hand them
 * to rtgpu, copy the result into the structures the reference's callers read.
 */
#include "rt_host.h"

#include <float.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

static rtgpu_ctx *g_ctx = NULL;
static int g_device = 0;

void rtHostSetDevice(int device) {
  if (device != g_device) rtHostShutdown();
  g_device = device;
}

/* The pixel map (24 bytes per pixel: 199 MB for a 4K frame) is page-locked memory, which the GPU fills at the full rate of
 * the host link, and is kept from call to call: pinning fresh pages for every frame costs more than the frame itself.  A
 * block starts with a header that says how it was obtained; the map the caller sees lies 64 bytes behind it. */
struct map_hdr { uint64_t magic; size_t bytes; int pinned; };
#define MAP_MAGIC 0x52544d4150484452ull
#define MAP_HDR 64
static struct map_hdr *held_map; /* the block of the last destroyed frame, for the next one */
static void block_free(struct map_hdr *h) {
  if (!h) return;
  if (h->pinned) rtgpu_pinned_free(h); else free(h);
}
static void *map_grab(size_t bytes) {
  struct map_hdr *h = held_map;
  held_map = NULL;
  if (h && h->bytes < bytes) { block_free(h); h = NULL; }
  if (!h) {
    int pinned = 1;
    h = rtgpu_pinned_alloc(bytes + MAP_HDR);
    if (!h) { pinned = 0; h = malloc(bytes + MAP_HDR); }
    if (!h) return NULL;
    h->magic = MAP_MAGIC; h->bytes = bytes; h->pinned = pinned;
  }
  return (char *)h + MAP_HDR;
}
static void map_release(void *map) {
  struct map_hdr *h;
  if (!map) return;
  h = (struct map_hdr *)((char *)map - MAP_HDR);
  if (h->magic != MAP_MAGIC) return; /* not one of ours */
  if (!held_map || held_map->bytes < h->bytes) { block_free(held_map); held_map = h; } else block_free(h);
}

void rtHostShutdown(void) {
  block_free(held_map); held_map = NULL;
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

/* RT_VisualizedScenePixel as the GPU library is told about it (src/raytrace.h:23-27) */
static rtgpu_pixel_layout pixel_layout(const RT_Scene *scene) {
  rtgpu_pixel_layout l;
  l.stride = (int64_t)sizeof(RT_VisualizedScenePixel);
  l.color_offset = (int64_t)offsetof(RT_VisualizedScenePixel, c);
  l.tri_offset = (int64_t)offsetof(RT_VisualizedScenePixel, t);
  l.tri_base = scene ? (uint64_t)(uintptr_t)scene->t : 0;
  l.tri_stride = (int64_t)sizeof(RT_Triangle);
  return l;
}

RT_VisualizedScene *rtVisualizedSceneRaytrace(RT_Scene *scene, RT_Camera *camera) {
  const size_t np = (size_t)camera->sw * (size_t)camera->sh;
  RT_VisualizedScene *res = calloc(1, sizeof *res);
  RT_SceneFlat *flat = NULL;
  rtgpu_ctx *ctx;
  rtgpu_camera cam;
  rtgpu_grid_info gi;
  rtgpu_pixel_layout lay;
  float *tri = NULL, mm[6];
  int k, ok = 0;
  if (!res || !(res->map = map_grab((np ? np : 1) * sizeof *res->map))) { errno = E_MEMORY; goto out; }
  if (np == 0) memset(res->map, 0, sizeof *res->map);
  res->width = camera->sw; res->height = camera->sh;
  res->gamma = scene->cfg.gamma > 0.0f ? scene->cfg.gamma : 2.5f; /* src/raytrace.c:333-337 */
  rtScenePreprocess(scene, camera);                               /* texture switch + texture coordinates */
  if (!(ctx = context())) goto out;
  if (!(flat = rtSceneFlatten(scene))) goto out;
  rtCameraFlatten(camera, &cam);
  lay = pixel_layout(scene);
  tri = malloc(sizeof(float) * 10 * (size_t)(scene->nt > 0 ? scene->nt : 1));
  if (!tri) { errno = E_MEMORY; goto out; }
  /* (an upload of the scene the context already holds costs a comparison; the GPU writes the pixel map itself) */
  if (gpu_error(ctx, rtgpu_scene_upload(ctx, &flat->desc))) goto out;
  if (gpu_error(ctx, rtgpu_frame_begin(ctx, &cam))) goto out;
  if (np > 0 && gpu_error(ctx, rtgpu_render_pixelmap(ctx, res->map, &lay, mm))) goto out;
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
  /* (the reference leaves the pixels whose ray never enters the grid, and every pixel's fourth colour lane, as malloc
   * handed them out, src/raytrace.c:323,381-382; here they are defined: 0) */
  for (k = 0; k < 3; k++) { res->min.c[k] = np > 0 ? mm[k] : FLT_MAX; res->max.c[k] = np > 0 ? mm[3 + k] : FLT_MIN; }
  res->min.c[3] = FLT_MAX; res->max.c[3] = FLT_MIN;
  ok = 1;
out:
  rtSceneFlatFree(&flat);
  free(tri);
  if (!ok) rtVisualizedSceneDestroy(&res);
  return res;
}

void rtVisualizedSceneDestroy(RT_VisualizedScene **self) {
  if (self && *self) {
    map_release((*self)->map);
    free(*self); *self = NULL;
  }
}

RT_Bitmap *rtVisualizedSceneToBitmap(RT_VisualizedScene *s, int flags, void *param1) {
  const size_t np = (size_t)s->width * (size_t)s->height;
  RT_Bitmap *bmp = rtBitmapCreate(s->width, s->height, 0);
  const rtgpu_pixel_layout lay = pixel_layout(NULL);
  const float *gammas = param1 ? (const float *)param1 : &s->gamma; /* src/raytrace.c:439-440 */
  int32_t ng = 0;
  rtgpu_ctx *ctx;
  float mm[6];
  int k;
  if (!bmp) return NULL;
  if (!(flags & F_HDR)) return bmp; /* src/raytrace.c:437: nothing is drawn without F_HDR */
  if (param1) { while (gammas[ng] > 0.0f) ng++; } else ng = s->gamma > 0.0f ? 1 : 0; /* the zero-terminated list, src/raytrace.c:445-447 */
  if (!(ctx = context())) { rtBitmapDestroy(&bmp); return NULL; }
  for (k = 0; k < 3; k++) { mm[k] = s->min.c[k]; mm[3 + k] = s->max.c[k]; }
  if (gpu_error(ctx, rtgpu_tonemap_pixelmap(ctx, s->map, &lay, (int64_t)np, mm, gammas, ng, bmp->pixels))) rtBitmapDestroy(&bmp);
  return bmp;
}
