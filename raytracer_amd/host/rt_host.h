/* Host layer of the MI355X ray tracer: the reference's C entry points, file formats and BMP output,
 * rebuilt from scratch around the rtgpu C ABI (include/rtgpu.h).
 *
 * Same function names, argument meaning and errno convention as the reference headers
 * (src/scene.h:121-177, src/raytrace.h:57-64, src/bitmap.h:102-119, src/error.h:19-47), so a program
 * written against the reference links against this layer unchanged.  The per-pixel trace path runs on
 * the GPU only; everything in this layer is loading, flattening and writing files.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include <errno.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/rtgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* errno codes (src/error.h:19-42) */
enum { E_IO = 1, E_MEMORY, E_INVALID_FILE, E_INVALID_FILE_FORMAT, E_INVALID_BPP, E_NOT_ENOUGH_SURFACES, E_INVALID_PARAM_VALUE };
char *rtGetErrorDesc(void);

typedef float RT_Vertex4f[4];
typedef float RT_Vertex2f[2];

typedef enum { VOX_DEFAULT, VOX_MODIFIED_DEFAULT, VOX_FIXED } RT_VoxelizationMode;

typedef struct _RT_Color { float c[4]; } RT_Color;

typedef struct _RT_Bitmap {
  int32_t width, height;
  uint32_t background;
  uint32_t *pixels; /* R<<24 | G<<16 | B<<8 | A */
} RT_Bitmap;

typedef struct _RT_Surface {
  RT_Color color;
  float kd, ks, g, ka;
  float kt, eta, kr;
} RT_Surface;

typedef struct _RT_Triangle {
  RT_Vertex4f i, j, k;
  RT_Vertex2f ti, tj, tk;
  RT_Bitmap *texture;
  RT_Surface *s;
  int32_t sid;
  RT_Vertex4f n, ij, ik; /* filled by rtScenePreprocess / rtVisualizedSceneRaytrace */
  float d;
} RT_Triangle;

typedef struct _RT_Light { RT_Vertex4f p; float flux; RT_Color color; } RT_Light;

typedef struct _RT_PlanarLight {
  float flux;
  RT_Color color;
  RT_Vertex4f a, b, c, ab, ac, n;
} RT_PlanarLight;

typedef struct _RT_SceneConfig {
  float epsilon, gamma, distmod;
  RT_VoxelizationMode vmode;
  float vcoeff[3];
} RT_SceneConfig;

typedef struct _RT_Scene {
  RT_SceneConfig cfg;
  float dmin[3], dmax[3];
  int32_t nt, nl, npl, ns;
  RT_Triangle *t;
  RT_Light *l;
  RT_PlanarLight *pl;
  RT_Surface *s;
  RT_Bitmap *texture; /* textures/brickwall.bmp if it could be loaded, owned by the scene */
} RT_Scene;

typedef struct _RT_Camera {
  RT_Vertex4f ob, ul, bl, ur;
  int32_t sw, sh;
} RT_Camera;

typedef struct _RT_VisualizedScenePixel { RT_Color c; RT_Triangle *t; } RT_VisualizedScenePixel;

typedef struct _RT_VisualizedScene {
  int32_t width, height;
  float total_flux, gamma;
  RT_Color min, max;
  RT_VisualizedScenePixel *map;
} RT_VisualizedScene;

#define F_GAMMA 1
#define F_HDR 2

/* scene.c */
RT_Scene *rtSceneLoad(const char *filename);
RT_Scene *rtSceneConfigureRenderer(RT_Scene *self, const char *filename);
RT_Scene *rtSceneSetSurfaces(RT_Scene *self, RT_Surface *s, uint32_t ns);
RT_Scene *rtSceneSetLights(RT_Scene *self, RT_Light *l, uint32_t nl);
RT_Scene *rtSceneSetPlanarLights(RT_Scene *self, RT_PlanarLight *l, uint32_t npl);
void rtSceneDestroy(RT_Scene **self);
RT_Light *rtLightLoad(const char *filename, uint32_t *n);
RT_PlanarLight *rtPlanarLightLoad(const char *filename, uint32_t *n);
RT_Surface *rtSurfaceLoad(const char *filename, uint32_t *n);
RT_Camera *rtCameraLoad(const char *filename);
void rtCameraDestroy(RT_Camera **self);
/* One call for what the CLI does between argument parsing and rendering (src/main.c:134-180).
 * Any of lgt/pnr/cfg may be NULL or missing (warnings in the reference); returns NULL with errno set
 * when geometry, attributes or camera cannot be loaded. */
RT_Scene *rtSceneLoadAll(const char *brs, const char *lgt, const char *pnr, const char *atr, const char *cfg,
                         float gamma, float epsilon, float distmod);

/* rtSceneLoad with a binary cache of the geometry file next to it (or anywhere): the cache is used when it was
 * written for a source of the same size and modification time, and (re)written otherwise.  cache = NULL: no cache. */
RT_Scene *rtSceneLoadCached(const char *filename, const char *cache);
RT_Scene *rtSceneLoadAllCached(const char *brs, const char *cache, const char *lgt, const char *pnr, const char *atr, const char *cfg,
                               float gamma, float epsilon, float distmod);

/* Flattening for the GPU boundary: fills a rtgpu_scene_desc whose arrays live in `storage`
 * (malloc'd, free with rtSceneFlatFree). */
typedef struct _RT_SceneFlat {
  rtgpu_scene_desc desc;
  float *vertices;
  int32_t *sid;
  rtgpu_surface *surfaces;
  rtgpu_light *lights;
  rtgpu_planar_light *planar;
} RT_SceneFlat;
RT_SceneFlat *rtSceneFlatten(const RT_Scene *scene);
void rtSceneFlatFree(RT_SceneFlat **flat);
void rtCameraFlatten(const RT_Camera *cam, rtgpu_camera *out);
/* Which tiles go to which device when a frame is split over ndev GPUs, and where each device's block starts in the
 * gathered buffer (tiles[ndev], first_pixel[ndev + 1]); see scene.c. */
void rtTilePlan(int32_t width, int32_t height, int ndev, int64_t *tiles, int64_t *first_pixel);
/* multigpu.c (librtmulti.so, needs RCCL): `frames` frames on devices 0..ndev-1 of this node, one process; returns the
 * last frame.  seconds (optional): wall time per frame after the first. */
RT_Bitmap *rtHostRenderDevices(RT_Scene *scene, RT_Camera *camera, int ndev, int frames, double *seconds);
/* The loaders' decimal parser (exact short path, strtof otherwise), exported for the tests. */
float rtParseFloat(const char *text, int *consumed);

/* preprocess.c: texture switch and texture coordinates (src/preprocess.c:13,41-79).  The geometric
 * part of the reference's preprocess (normals, edges) is computed on the GPU per frame and copied
 * back into the triangles by rtVisualizedSceneRaytrace. */
RT_Scene *rtScenePreprocess(RT_Scene *scene, RT_Camera *camera);

/* raytrace.c */
RT_VisualizedScene *rtVisualizedSceneRaytrace(RT_Scene *scene, RT_Camera *camera);
void rtVisualizedSceneDestroy(RT_VisualizedScene **self);
RT_Bitmap *rtVisualizedSceneToBitmap(RT_VisualizedScene *s, int flags, void *param1);
/* Device used by rtVisualizedSceneRaytrace (default 0) and release of the cached GPU context. */
void rtHostSetDevice(int device);
void rtHostShutdown(void);

/* bitmap.c */
RT_Bitmap *rtBitmapCreate(int32_t width, int32_t height, uint32_t background);
void rtBitmapDestroy(RT_Bitmap **self);
RT_Bitmap *rtBitmapLoad(const char *filename);
void rtBitmapSave(const RT_Bitmap *self, const char *filename, uint16_t bpp);
static inline uint32_t rtColorBuildRGBA(uint32_t r, uint32_t g, uint32_t b, uint32_t a) {
  return ((r & 0xff) << 24) | ((g & 0xff) << 16) | ((b & 0xff) << 8) | (a & 0xff);
}

#ifdef __cplusplus
}
#endif
#endif
