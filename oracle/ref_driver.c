/* TEST INFRASTRUCTURE — not part of the product path.
 *
 * Dump driver for the *unmodified* reference renderer.  It is linked against
 * the reference's own object files, compiled in place from
 * /root/reference/src by oracle/Makefile (outputs only under oracle/_ref/).
 * It performs the same load sequence as the reference CLI
 * (/root/reference/src/main.c:134-189), optionally overrides the camera
 * resolution (equivalent to editing the last line of the .cam file,
 * /root/reference/src/scene.c:655-657), calls rtVisualizedSceneRaytrace and
 * writes, per pixel and row-major, {int32 id, float r, g, b} where
 * id = index of the primary-hit triangle or -1.
 *
 * usage: ref_dump -s PREFIX [-g brs] [-l lgt] [-L pnr] [-a atr] [-c cam] [-C cfg]
 *                 [-W width -H height] [-d dump.bin] [-o out.bmp] [-G g1,g2,...]
 * -G: a comma-separated gamma list, passed zero-terminated to the reference's rtVisualizedSceneToBitmap as its
 * param1 (/root/reference/src/raytrace.c:439-447); "-G ," is the empty list.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <errno.h>
#include <time.h>
#include <malloc.h>
#include "scene.h"
#include "raytrace.h"
#include "bitmap.h"

static char *cat(const char *a, const char *b) {
  char *r = malloc(strlen(a) + strlen(b) + 1);
  strcpy(r, a);
  strcat(r, b);
  return r;
}

int main(int argc, char **argv) {
  const char *g = NULL, *l = NULL, *L = NULL, *a = NULL, *c = NULL, *C = NULL, *s = NULL;
  const char *dump = NULL, *out = NULL, *glist = NULL;
  float gammas[65];
  int W = 0, H = 0, i, ng = 0;
  uint32_t n;

  /* Large blocks come from fresh (zeroed) mmaps, so that pixels the reference
   * never writes (rays that do not enter the grid, raytrace.c:381-382) read
   * back as zeros instead of recycled heap bytes. */
  mallopt(M_MMAP_THRESHOLD, 4096);
  mallopt(M_MMAP_MAX, 1 << 24);

  for (i = 1; i + 1 < argc; i += 2) {
    const char *k = argv[i], *v = argv[i + 1];
    if (!strcmp(k, "-g")) g = v; else if (!strcmp(k, "-l")) l = v;
    else if (!strcmp(k, "-L")) L = v; else if (!strcmp(k, "-a")) a = v;
    else if (!strcmp(k, "-c")) c = v; else if (!strcmp(k, "-C")) C = v;
    else if (!strcmp(k, "-s")) s = v; else if (!strcmp(k, "-d")) dump = v;
    else if (!strcmp(k, "-o")) out = v; else if (!strcmp(k, "-W")) W = atoi(v);
    else if (!strcmp(k, "-H")) H = atoi(v);
    else if (!strcmp(k, "-G")) glist = v;
    else { fprintf(stderr, "unknown option %s\n", k); return 2; }
  }
  if (s) {
    if (!g) g = cat(s, ".brs");
    if (!l) l = cat(s, ".lgt");
    if (!a) a = cat(s, ".atr");
    if (!c) c = cat(s, ".cam");
    if (!C) C = cat(s, ".cfg");
    if (!L) L = cat(s, ".pnr");
  }
  if (!g || !l || !a || !c) { fprintf(stderr, "missing inputs\n"); return 2; }
  { FILE *probe = fopen(g, "r"); if (!probe) { fprintf(stderr, "cannot open %s\n", g); return 2; } fclose(probe); }

  errno = 0;
  RT_Scene *scene = rtSceneLoad(g);
  if (errno > 0 || !scene) { fprintf(stderr, "geometry load failed\n"); return 1; }
  scene->cfg.epsilon = 0.0f;
  scene->cfg.gamma = 2.5f;
  scene->cfg.distmod = 2.0f;
  if (C) rtSceneConfigureRenderer(scene, C);
  errno = 0;
  RT_Light *lgt = rtLightLoad(l, &n);
  if (errno > 0) errno = 0; else rtSceneSetLights(scene, lgt, n);
  RT_PlanarLight *pl = L ? rtPlanarLightLoad(L, &n) : NULL;
  if (!L || errno > 0) errno = 0; else rtSceneSetPlanarLights(scene, pl, n);
  RT_Surface *surf = rtSurfaceLoad(a, &n);
  if (errno > 0) { fprintf(stderr, "attribute load failed\n"); return 1; }
  if (!rtSceneSetSurfaces(scene, surf, n)) { fprintf(stderr, "not enough surfaces\n"); return 1; }
  RT_Camera *cam = rtCameraLoad(c);
  if (errno > 0 || !cam) { fprintf(stderr, "camera load failed\n"); return 1; }
  if (W > 0 && H > 0) { cam->sw = W; cam->sh = H; }

  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  RT_VisualizedScene *vs = rtVisualizedSceneRaytrace(scene, cam);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  errno = 0;
  if (!vs) { fprintf(stderr, "raytrace failed\n"); return 1; }
  double secs = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);

  long hits = 0, p, np = (long)vs->width * vs->height;
  for (p = 0; p < np; p++) hits += vs->map[p].t != NULL;
  printf("{\"width\": %d, \"height\": %d, \"seconds\": %.6f, \"primary_hits\": %ld, "
         "\"min\": [%.9g, %.9g, %.9g], \"max\": [%.9g, %.9g, %.9g], \"gamma\": %.9g, \"nt\": %d, \"nl\": %d, \"npl\": %d}\n",
         vs->width, vs->height, secs, hits,
         vs->min.c[0], vs->min.c[1], vs->min.c[2], vs->max.c[0], vs->max.c[1], vs->max.c[2],
         vs->gamma, scene->nt, scene->nl, scene->npl);

  if (dump) {
    FILE *f = fopen(dump, "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", dump); return 1; }
    int32_t hdr[4] = {0x31445452 /* "RTD1" */, vs->width, vs->height, 0};
    float mm[8] = {vs->min.c[0], vs->min.c[1], vs->min.c[2], vs->max.c[0], vs->max.c[1], vs->max.c[2],
                   vs->gamma, vs->total_flux};
    fwrite(hdr, sizeof hdr, 1, f);
    fwrite(mm, sizeof mm, 1, f);
    for (p = 0; p < np; p++) {
      int32_t id = vs->map[p].t ? (int32_t)(vs->map[p].t - scene->t) : -1;
      fwrite(&id, 4, 1, f);
      fwrite(vs->map[p].c.c, 4, 3, f);
    }
    fclose(f);
  }
  if (out) {
    if (glist) {
      const char *q = glist;
      while (*q && ng < 64) {
        char *end;
        float gv = strtof(q, &end);
        if (end == q) break;
        gammas[ng++] = gv;
        q = *end == ',' ? end + 1 : end;
      }
      gammas[ng] = 0.0f;
    }
    RT_Bitmap *bmp = rtVisualizedSceneToBitmap(vs, F_HDR, glist ? gammas : NULL);
    rtBitmapSave(bmp, out, 24);
    rtBitmapDestroy(&bmp);
  }
  return 0;
}
