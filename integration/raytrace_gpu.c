/* Drop-in replacement for the reference's src/raytrace.c: the same three entry points
 *     rtVisualizedSceneRaytrace / rtVisualizedSceneDestroy / rtVisualizedSceneToBitmap   (src/raytrace.h:57-64)
 * with the per-pixel trace path and the tone map running on the GPU through librtgpu (include/rtgpu.h).
 *
 * This file is compiled against the reference's OWN headers (-I<reference>/src: scene.h, raytrace.h, bitmap.h,
 * preprocess.h, error.h) and linked with the reference's unmodified main.o, scene.o, bitmap.o, error.o,
 * stringtools.o, preprocess.o, intersection.o in place of raytrace.o, voxelize.o and texture.o:
 *     gcc -std=gnu11 -O2 -I<reference>/src -I<this repo>/include -c integration/raytrace_gpu.c
 *     gcc -o raytrace main.o scene.o bitmap.o error.o stringtools.o preprocess.o intersection.o raytrace_gpu.o \
 *         -L<this repo>/raytracer_amd -lrtgpu -lm
 * What stays on the host is what the reference does on the host before its pixel loop and what its callers read
 * afterwards: rtScenePreprocess (the texture on/off switch, src/preprocess.c:13,41-79; its per-triangle results are
 * left in RT_Scene as the reference leaves them), the grown scene domain, min/max and the pixel map.
 *
 * Environment: RTGPU_DEVICE selects the GPU (default 0).  Errors follow the reference's convention: errno is set to
 * one of its codes (src/error.h:19-42) and NULL is returned.
 */
#include <errno.h>
#include <float.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "raytrace.h"
#include "preprocess.h"
#include "error.h"
#include "rtgpu.h"

static rtgpu_ctx *g_ctx; /* one context per process and GPU, created at the first frame */

/* The reference's main() tests errno before anything has set it (src/main.c:119-122) and stops with "unable to parse
 * args" if it is positive.  The GPU runtime that librtgpu brings in initialises itself while the program is loaded and
 * leaves ENOENT (= 2, the reference's E_MEMORY) behind; the program's own constructors run after those of its
 * libraries, so this one restores the value main() expects. */
__attribute__((constructor)) static void clear_load_time_errno(void) { errno = 0; }

static void release_at_exit(void);
static rtgpu_ctx *context(void) {
  if (!g_ctx) {
    const char *dev = getenv("RTGPU_DEVICE");
    int rc = rtgpu_create(dev ? atoi(dev) : 0, &g_ctx);
    if (rc) {
      fprintf(stderr, "rtgpu: %s (the trace path has no CPU implementation)\n", rtgpu_error_string(rc));
      errno = rc <= E_INVALID_PARAM_VALUE ? rc : E_IO;
      return NULL;
    }
    atexit(release_at_exit);
  }
  return g_ctx;
}

static int gpu_failed(rtgpu_ctx *ctx, int rc) {
  if (rc) {
    fprintf(stderr, "rtgpu: %s: %s\n", rtgpu_error_string(rc), rtgpu_last_message(ctx));
    errno = rc <= E_INVALID_PARAM_VALUE ? rc : E_INVALID_PARAM_VALUE;
  }
  return rc;
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
/* At the end of the program: the kept block and the GPU context.  Registered with atexit when the context is created,
 * i.e. after the GPU runtime has registered its own exit handlers, so that it runs before them. */
static void release_at_exit(void) {
  block_free(held_map); held_map = NULL;
  if (g_ctx) { rtgpu_destroy(g_ctx); g_ctx = NULL; }
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
  const int32_t nt = scene->nt, ns = scene->ns, nl = scene->nl, npl = scene->npl;
  RT_VisualizedScene *res = calloc(1, sizeof *res);
  float *verts = NULL, mm[6];
  int32_t *sid = NULL;
  rtgpu_surface *surf = NULL;
  rtgpu_light *lgt = NULL;
  rtgpu_planar_light *pl = NULL;
  rtgpu_scene_desc d;
  rtgpu_camera cam;
  rtgpu_grid_info gi;
  rtgpu_pixel_layout lay;
  rtgpu_ctx *ctx;
  int32_t i;
  int k, ok = 0, saved;

  if (!res || !(res->map = map_grab((np ? np : 1) * sizeof *res->map))) { errno = E_MEMORY; goto out; }
  if (np == 0) memset(res->map, 0, sizeof *res->map);
  res->width = camera->sw;
  res->height = camera->sh;
  res->gamma = scene->cfg.gamma > 0.0f ? scene->cfg.gamma : 2.5f; /* src/raytrace.c:333-337 */

  /* host side of the frame: the reference's own preprocess (texture switch; n, ij, ik, d stay in RT_Scene) */
  rtScenePreprocess(scene, camera);

  /* RT_Scene / RT_Camera -> plain arrays */
  verts = malloc(sizeof(float) * 9 * (size_t)(nt > 0 ? nt : 1));
  sid = malloc(sizeof(int32_t) * (size_t)(nt > 0 ? nt : 1));
  surf = calloc((size_t)ns + 1, sizeof *surf);
  lgt = calloc((size_t)nl + 1, sizeof *lgt);
  pl = calloc((size_t)npl + 1, sizeof *pl);
  if (!verts || !sid || !surf || !lgt || !pl) { errno = E_MEMORY; goto out; }
  memset(&d, 0, sizeof d);
  for (i = 0; i < nt; i++) {
    const RT_Triangle *t = &scene->t[i];
    for (k = 0; k < 3; k++) { verts[9 * i + k] = t->i[k]; verts[9 * i + 3 + k] = t->j[k]; verts[9 * i + 6 + k] = t->k[k]; }
    sid[i] = t->sid;
    if (t->texture) d.texture_present = 1;
  }
  for (i = 0; i < ns; i++) {
    const RT_Surface *s = &scene->s[i];
    for (k = 0; k < 3; k++) surf[i].color[k] = s->color.c[k];
    surf[i].kd = s->kd; surf[i].ks = s->ks; surf[i].g = s->g; surf[i].ka = s->ka;
    surf[i].kt = s->kt; surf[i].eta = s->eta; surf[i].kr = s->kr;
  }
  for (i = 0; i < nl; i++) {
    const RT_Light *l = &scene->l[i];
    for (k = 0; k < 3; k++) { lgt[i].p[k] = l->p[k]; lgt[i].color[k] = l->color.c[k]; }
    lgt[i].flux = l->flux;
  }
  for (i = 0; i < npl; i++) {
    const RT_PlanarLight *q = &scene->pl[i];
    pl[i].flux = q->flux;
    for (k = 0; k < 3; k++) { pl[i].color[k] = q->color.c[k]; pl[i].a[k] = q->a[k]; pl[i].ab[k] = q->ab[k]; pl[i].ac[k] = q->ac[k]; }
  }
  d.nt = nt; d.ns = ns; d.nl = nl; d.npl = npl;
  d.vertices = verts; d.sid = sid; d.surfaces = surf; d.lights = lgt; d.planar = pl;
  for (k = 0; k < 3; k++) { d.dmin[k] = scene->dmin[k]; d.dmax[k] = scene->dmax[k]; d.vcoeff[k] = scene->cfg.vcoeff[k]; }
  d.gamma = scene->cfg.gamma; d.distmod = scene->cfg.distmod; d.vmode = (int32_t)scene->cfg.vmode;
  for (k = 0; k < 3; k++) { cam.ob[k] = camera->ob[k]; cam.ul[k] = camera->ul[k]; cam.bl[k] = camera->bl[k]; cam.ur[k] = camera->ur[k]; }
  cam.sw = camera->sw; cam.sh = camera->sh;
  lay = pixel_layout(scene);

  /* the GPU does what src/raytrace.c:346-404 does and writes the pixel map itself (an upload of the scene the context
   * already holds costs a comparison).  The GPU runtime may leave stray errno values (ENOENT, EAGAIN ...) on paths that
   * worked, and the caller reads errno (src/main.c:190): keep the value the host code had. */
  saved = errno;
  if (!(ctx = context())) goto out;
  if (gpu_failed(ctx, rtgpu_scene_upload(ctx, &d))) goto out;
  if (gpu_failed(ctx, rtgpu_frame_begin(ctx, &cam))) goto out;
  if (np > 0 && gpu_failed(ctx, rtgpu_render_pixelmap(ctx, res->map, &lay, mm))) goto out;
  if (gpu_failed(ctx, rtgpu_grid_info_get(ctx, &gi))) goto out;
  errno = saved;

  /* results into the reference's structures.  (The reference leaves the pixels whose ray never enters the grid, and every
   * pixel's fourth colour lane, as malloc handed them out, src/raytrace.c:323,381-382; here they are defined: 0.) */
  res->total_flux = gi.total_flux;
  for (k = 0; k < 3; k++) {
    res->min.c[k] = np > 0 ? mm[k] : FLT_MAX; res->max.c[k] = np > 0 ? mm[3 + k] : FLT_MIN;
    scene->dmin[k] = gi.dmin[k]; scene->dmax[k] = gi.dmax[k]; /* the reference grows the domain in place (src/raytrace.c:350-353, src/voxelize.c:140-141) */
  }
  res->min.c[3] = FLT_MAX; res->max.c[3] = FLT_MIN; /* src/raytrace.c:317-320: the alpha lane is never updated */
  ok = 1;
out:
  free(verts); free(sid); free(surf); free(lgt); free(pl);
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
  int k, saved;
  if (!bmp) return NULL;
  if (!(flags & F_HDR)) return bmp; /* src/raytrace.c:437: nothing is drawn without F_HDR */
  if (param1) { while (gammas[ng] > 0.0f) ng++; } else ng = s->gamma > 0.0f ? 1 : 0; /* the zero-terminated list, src/raytrace.c:445-447 */
  saved = errno;
  if (!(ctx = context())) { rtBitmapDestroy(&bmp); return NULL; }
  for (k = 0; k < 3; k++) { mm[k] = s->min.c[k]; mm[3 + k] = s->max.c[k]; }
  if (gpu_failed(ctx, rtgpu_tonemap_pixelmap(ctx, s->map, &lay, (int64_t)np, mm, gammas, ng, bmp->pixels))) rtBitmapDestroy(&bmp);
  else errno = saved;
  return bmp;
}
