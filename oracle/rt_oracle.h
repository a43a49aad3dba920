/* TEST INFRASTRUCTURE — CPU oracle for the per-pixel trace path.
 *
 * Plain-C restatement of the reference renderer's frame function and of the
 * loaders/tone-map on either side of it.  Only tests/, the smoke check and the
 * benchmark's cpu_baseline leg may load this library; the product path never
 * does.  Parity is pinned: tests/test_oracle_vs_reference.py checks it
 * bit-for-bit against dumps of the unmodified reference (oracle/_ref/ref_dump)
 * committed under tests/golden/.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_scene ora_scene;

/* render flags */
#define ORA_SHADOW_CACHE 1 /* keep the per-(triangle, light) shadow cache (reference behaviour, scan-order dependent) */
#define ORA_COUNTERS 2     /* fill ora_counters */

typedef struct ora_counters {
  int64_t nearest_queries; /* calls of the nearest-hit walk (voxelize.c:536) */
  int64_t shadow_queries;  /* calls of the shadow walk (voxelize.c:615), early-outs included */
  int64_t shadow_backface; /* ... ended by the back-face test (voxelize.c:640-644) */
  int64_t shadow_cached;   /* ... ended by a shadow-cache hit (voxelize.c:647-652) */
  int64_t nearest_steps;   /* S: loop iterations at voxelize.c:565 */
  int64_t nearest_tests;   /* T: sum of voxel->nt over those iterations */
  int64_t shadow_steps;    /* S: loop iterations at voxelize.c:695 */
  int64_t shadow_tests;    /* T: sum of voxel->nt over those iterations */
  int64_t pixels_traced;   /* pixels whose primary ray entered the grid */
} ora_counters;

/* Any path may be NULL/missing except brs, atr, cam.  texture_present mirrors
 * "textures/brickwall.bmp could be loaded from the cwd" (preprocess.c:13). */
ora_scene *ora_load(const char *brs, const char *lgt, const char *pnr, const char *atr,
                    const char *cam, const char *cfg, int texture_present);
void ora_free(ora_scene *s);
void ora_set_resolution(ora_scene *s, int w, int h);
void ora_get_info(const ora_scene *s, int32_t out[8]); /* w,h,nt,nl,npl,ns,vmode,texture */
float ora_get_gamma(const ora_scene *s);

/* Frame function (raytrace.c:304-413).  rgb: w*h*3 floats, ids: w*h int32
 * (-1 = no primary hit), minmax: min r,g,b then max r,g,b, taint: w*h bytes,
 * 1 where the reference reads uninitialised texture samples (raytrace.c:90-120).
 * threads > 1 renders rows in parallel (needs the shadow cache off and no planar lights). */
int ora_render(ora_scene *s, int flags, int threads, float *rgb, int32_t *ids, float *minmax,
               uint8_t *taint, ora_counters *cnt);
/* Render only rows [y0, y1) (same outputs, full-size buffers; untouched rows stay as they were). */
int ora_render_rows(ora_scene *s, int flags, int threads, int y0, int y1, float *rgb, int32_t *ids,
                    float *minmax, uint8_t *taint, ora_counters *cnt);

/* Grid of the last render: nv[3], cell size s[3], grown domain dmin/dmax, CSR lists. */
int ora_grid_dims(const ora_scene *s, int32_t nv[3], float cs[3], float dmin[3], float dmax[3]);
int64_t ora_grid_refs(const ora_scene *s, int32_t *cell_start /* ncells+1 */, int32_t *refs /* may be NULL */);
/* Per-triangle preprocess results of the last render: n (3), ij (3), ik (3), d, texflag, ti,tj,tk (6) = 17 floats each */
void ora_triangles(const ora_scene *s, float *out17);
void ora_vertices(const ora_scene *s, float *out9, int32_t *sid);

/* Tone map (raytrace.c:424-472): rgba packed R<<24|G<<16|B<<8|A per pixel. */
void ora_tonemap(int w, int h, const float *rgb, const float *minmax, float gamma, uint32_t *rgba);
/* ... with the zero-terminated gamma list of raytrace.c:439-447 given as n gammas (averaged per channel, raytrace.c:456-466) */
void ora_tonemap_multi(int w, int h, const float *rgb, const float *minmax, const float *gammas, int n, uint32_t *rgba);
/* 24-bpp BMP file image (bitmap.c:249-479); returns byte count, writes into out if non-NULL. */
int64_t ora_bmp24(int w, int h, const uint32_t *rgba, uint8_t *out);

/* glibc rand() restated (TYPE_3 additive feedback, seed 1), for the planar-light replay. */
void ora_rand_fill(int32_t *out, int64_t n);

#ifdef __cplusplus
}
#endif
#endif
