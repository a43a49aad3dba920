/* Scene, light, surface, camera and renderer-config loaders of the host layer.
 *
 * File grammars and parsing quirks follow the reference loaders (src/scene.c) closely because the
 * renderer's output depends on them: FLT_MIN as the initial domain maximum, the per-triangle vertex
 * nudge, "value <= 1 means already normalised" for colours, first-10-token surfaces, and so on.
 */
#include "rt_host.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

/* Next meaningful line: skips blank lines and lines starting with "//" in column 0 (src/scene.c:16-34). */
static char *read_line(FILE *f, char *buf, int cap) {
  while (fgets(buf, cap, f)) {
    const char *p = buf;
    while (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r') p++;
    if (*p == 0) continue;
    if (buf[0] == '/' && buf[1] == '/') continue;
    return buf;
  }
  return NULL;
}

/* normalised b - a with the reference's mixed precision (src/vectormath.h:60-66,122-127) */
static void unit_dir(float *out, const float *a, const float *b) {
  float s, inv;
  int k;
  for (k = 0; k < 3; k++) out[k] = b[k] - a[k];
  s = out[0] * out[0] + out[1] * out[1] + out[2] * out[2];
  inv = (float)(1.0 / sqrt((double)s));
  for (k = 0; k < 3; k++) out[k] *= inv;
}

static void grow(RT_Scene *s, const float *p) {
  int k;
  for (k = 0; k < 3; k++) {
    if (p[k] < s->dmin[k]) s->dmin[k] = p[k];
    if (p[k] > s->dmax[k]) s->dmax[k] = p[k];
  }
}

/* Pulls a vertex 1e-7 towards the triangle's centroid, per axis (src/scene.c:138-181). */
static void shrink_vertex(RT_Scene *s, float *p, const float *centroid) {
  float dir[3];
  int k;
  unit_dir(dir, centroid, p);
  for (k = 0; k < 3; k++) {
    if (dir[k] < 0.0f) p[k] += 0.0000001f;
    else if (dir[k] > 0.0f) p[k] += -0.0000001f;
  }
  grow(s, p);
}

/* sscanf("%f %f %f") and sscanf("%d %d %d") without the format interpreter (the geometry file is 1.7 MB of such
 * lines).  A decimal with at most 7 significant digits and at most 10 decimals, which is every number of the
 * bundled scenes, is converted on the exact path: its digits as an integer below 2^24 and the power of ten are
 * both exact binary32 values, so one correctly rounded division gives the correctly rounded result, the same
 * value strtof returns.  Anything else (more digits, exponents, inf, nan, hex) goes to strtof itself.  A field
 * that does not parse stops the line and leaves the remaining outputs as they were, as sscanf does. */
static float parse_float(const char *p, char **end) {
  static const float pow10[11] = {1.0f, 10.0f, 100.0f, 1000.0f, 10000.0f, 100000.0f, 1000000.0f, 10000000.0f,
                                  100000000.0f, 1000000000.0f, 10000000000.0f};
  const char *q = p;
  unsigned m = 0;
  int digits = 0, sig = 0, decimals = 0, neg = 0;
  while (*q == ' ' || (*q >= '\t' && *q <= '\r')) q++;
  if (*q == '-') { neg = 1; q++; } else if (*q == '+') q++;
  for (; *q >= '0' && *q <= '9'; q++, digits++) { if (m || *q != '0') sig++; m = m * 10u + (unsigned)(*q - '0'); if (sig > 7) goto slow; }
  if (*q == '.') {
    for (q++; *q >= '0' && *q <= '9'; q++, digits++, decimals++) { if (m || *q != '0') sig++; m = m * 10u + (unsigned)(*q - '0'); if (sig > 7) goto slow; }
  }
  if (digits == 0 || decimals > 10 || *q == 'e' || *q == 'E' || *q == 'x' || *q == 'X' || *q == 'p' || *q == 'P') goto slow;
  *end = (char *)q;
  {
    const float x = (float)m / pow10[decimals];
    return neg ? -x : x;
  }
slow:
  return strtof(p, end);
}
/* For the tests: parse_float and how many characters it consumed. */
float rtParseFloat(const char *text, int *consumed) {
  char *end;
  float x = parse_float(text, &end);
  if (consumed) *consumed = (int)(end - text);
  return x;
}
static void scan_floats(const char *p, float *out, int n) {
  int k;
  for (k = 0; k < n; k++) {
    char *end;
    float x = parse_float(p, &end);
    if (end == p) return;
    out[k] = x; p = end;
  }
}
static void scan_ints(const char *p, int *out, int n) {
  int k;
  for (k = 0; k < n; k++) {
    const char *q = p;
    long x = 0;
    int neg = 0, digits = 0;
    while (*q == ' ' || (*q >= '\t' && *q <= '\r')) q++;
    if (*q == '-') { neg = 1; q++; } else if (*q == '+') q++;
    for (; *q >= '0' && *q <= '9' && digits < 9; q++, digits++) x = x * 10 + (*q - '0');
    if (digits == 0 || (*q >= '0' && *q <= '9')) {  /* nothing, or too long for the short path */
      char *end;
      x = strtol(p, &end, 10);
      if (end == p) return;
      q = end; neg = 0;
    }
    out[k] = (int)(neg ? -x : x); p = q;
  }
}

RT_Scene *rtSceneLoad(const char *filename) {
  enum { WANT_NV, VERTS, WANT_NT, TRIS, SIDS, DONE } state = WANT_NV;
  FILE *f = fopen(filename, "r");
  RT_Scene *s = NULL;
  float (*v)[3] = NULL;
  char buf[1024], *line;
  int left = 0, at = 0, k;
  if (!f) { errno = E_IO; return NULL; }
  while (state != DONE && (line = read_line(f, buf, sizeof buf))) {
    switch (state) {
      case WANT_NV:
        s = calloc(1, sizeof *s);
        if (!s) { errno = E_MEMORY; goto out; }
        for (k = 0; k < 3; k++) { s->dmin[k] = FLT_MAX; s->dmax[k] = FLT_MIN; } /* src/scene.c:66-69 */
        left = 0; sscanf(line, "%d", &left);
        v = malloc(sizeof *v * (left > 0 ? left : 1));
        if (!v) { errno = E_MEMORY; goto out; }
        at = 0; state = left > 0 ? VERTS : WANT_NT;
        break;
      case VERTS:
        scan_floats(line, v[at], 3);
        grow(s, v[at]);
        at++;
        if (--left == 0) state = WANT_NT;
        break;
      case WANT_NT:
        left = 0; sscanf(line, "%d", &left);
        s->nt = left;
        s->t = calloc(left > 0 ? left : 1, sizeof *s->t);
        if (!s->t) { errno = E_MEMORY; goto out; }
        at = 0; state = left > 0 ? TRIS : SIDS;
        break;
      case TRIS: {
        int abc[3] = {0, 0, 0}, a, b, c;
        float cen[3];
        RT_Triangle *t = &s->t[at];
        scan_ints(line, abc, 3);
        a = abc[0]; b = abc[1]; c = abc[2];
        memcpy(t->i, v[a], 12); memcpy(t->j, v[b], 12); memcpy(t->k, v[c], 12);
        t->ti[0] = 0.0f; t->ti[1] = 0.0f; t->tj[0] = 1.0f; t->tj[1] = 0.0f; t->tk[0] = 0.0f; t->tk[1] = 1.0f;
        for (k = 0; k < 3; k++) cen[k] = (t->i[k] + t->j[k] + t->k[k]) / 3.0f;
        shrink_vertex(s, t->i, cen);
        shrink_vertex(s, t->j, cen);
        shrink_vertex(s, t->k, cen);
        at++;
        if (--left == 0) { state = SIDS; at = 0; }
        break;
      }
      case SIDS: { /* whitespace-separated surface ids, any number per line, until nt were read */
        char *tok;
        for (tok = strtok(line, " \t\n\r"); tok; tok = strtok(NULL, " \t\n\r")) {
          int sid = 0;
          scan_ints(tok, &sid, 1);
          if (at < s->nt) { s->t[at].sid = sid; s->t[at].s = NULL; }
          at++;
        }
        if (at >= s->nt) state = DONE;
        break;
      }
      default: break;
    }
  }
out:
  free(v);
  fclose(f);
  if (s) { /* defaults of src/scene.c:219-223 */
    s->cfg.epsilon = 0.0f; s->cfg.gamma = 2.5f; s->cfg.distmod = 2.0f; s->cfg.vmode = VOX_DEFAULT;
  }
  return s;
}

RT_Scene *rtSceneConfigureRenderer(RT_Scene *self, const char *filename) {
  FILE *f = fopen(filename, "r");
  char buf[1024], word[1024], *line, *tok;
  if (!f) { errno = E_IO; return NULL; }
  while ((line = read_line(f, buf, sizeof buf))) {
    tok = strtok(line, " \t");
    while (tok) {
      float *dst = !strcmp(tok, "epsilon") ? &self->cfg.epsilon : !strcmp(tok, "gamma") ? &self->cfg.gamma
                 : !strcmp(tok, "distmod") ? &self->cfg.distmod : NULL;
      if (dst) {
        if ((tok = strtok(NULL, " \t"))) sscanf(tok, "%f", dst);
      } else if (!strcmp(tok, "voxmode")) {
        word[0] = 0;
        if ((tok = strtok(NULL, " \t"))) sscanf(tok, "%s", word);
        self->cfg.vmode = !strcmp(word, "MODIFIED_DEFAULT") ? VOX_MODIFIED_DEFAULT : !strcmp(word, "FIXED") ? VOX_FIXED : VOX_DEFAULT;
      } else if (!strcmp(tok, "voxparams")) {
        int k;
        for (k = 0; k < 3; k++)
          if ((tok = strtok(NULL, " \t"))) sscanf(tok, "%f", &self->cfg.vcoeff[k]);
      } /* any other key is skipped one token at a time (src/scene.c:242-273) */
      if (!tok) break;
      tok = strtok(NULL, " \t");
    }
  }
  fclose(f);
  return self;
}

RT_Scene *rtSceneSetSurfaces(RT_Scene *self, RT_Surface *s, uint32_t ns) {
  int32_t i;
  self->ns = (int32_t)ns;
  self->s = s;
  for (i = 0; i < self->nt; i++) {
    if (self->t[i].sid < 0 || (uint32_t)self->t[i].sid >= ns) { errno = E_NOT_ENOUGH_SURFACES; return NULL; }
    self->t[i].s = &s[self->t[i].sid];
  }
  return self;
}

RT_Scene *rtSceneSetLights(RT_Scene *self, RT_Light *l, uint32_t nl) {
  self->l = l;
  self->nl = (int32_t)nl;
  return self;
}

RT_Scene *rtSceneSetPlanarLights(RT_Scene *self, RT_PlanarLight *pl, uint32_t npl) {
  self->pl = pl;
  self->npl = (int32_t)npl;
  return self;
}

void rtSceneDestroy(RT_Scene **self) {
  RT_Scene *s = *self;
  if (!s) return;
  free(s->t); free(s->l); free(s->pl); free(s->s);
  rtBitmapDestroy(&s->texture);
  free(s);
  *self = NULL;
}

RT_Light *rtLightLoad(const char *filename, uint32_t *n) {
  FILE *f = fopen(filename, "r");
  char buf[1024], *line;
  RT_Light *res = NULL;
  int count = -1, i = 0;
  if (!f) { errno = E_IO; return NULL; }
  while ((line = read_line(f, buf, sizeof buf))) {
    if (count < 0) {
      count = 0; sscanf(line, "%d", &count);
      *n = (uint32_t)count;
      res = calloc(count > 0 ? count : 1, sizeof *res);
      if (!res) { errno = E_MEMORY; break; }
    } else if (i < count) {
      RT_Light *l = &res[i++];
      sscanf(line, "%f %f %f %f %f %f %f", &l->p[0], &l->p[1], &l->p[2], &l->flux, &l->color.c[0], &l->color.c[1], &l->color.c[2]);
    }
  }
  fclose(f);
  return res;
}

RT_PlanarLight *rtPlanarLightLoad(const char *filename, uint32_t *n) {
  FILE *f = fopen(filename, "r");
  char buf[1024], *line;
  RT_PlanarLight *res = NULL;
  int count = -1, row = 0, i, k;
  if (!f) { errno = E_IO; return NULL; }
  while ((line = read_line(f, buf, sizeof buf))) {
    if (count < 0) {
      count = 0; sscanf(line, "%d", &count);
      *n = (uint32_t)count;
      res = calloc(count > 0 ? count : 1, sizeof *res);
      if (!res) { errno = E_MEMORY; break; }
    } else if (row / 4 < count) { /* four lines per light: "flux R G B", a, b, c */
      RT_PlanarLight *p = &res[row / 4];
      float *dst = row % 4 == 1 ? p->a : row % 4 == 2 ? p->b : p->c;
      if (row % 4 == 0) sscanf(line, "%f %f %f %f", &p->flux, &p->color.c[0], &p->color.c[1], &p->color.c[2]);
      else sscanf(line, "%f %f %f", &dst[0], &dst[1], &dst[2]);
      row++;
    }
  }
  for (i = 0; res && i < count; i++) { /* src/scene.c:496-501 */
    RT_PlanarLight *p = &res[i];
    float s, inv;
    for (k = 0; k < 3; k++) { p->ab[k] = p->b[k] - p->a[k]; p->ac[k] = p->c[k] - p->a[k]; }
    p->n[0] = p->ab[1] * p->ac[2] - p->ab[2] * p->ac[1];
    p->n[1] = p->ab[2] * p->ac[0] - p->ab[0] * p->ac[2];
    p->n[2] = p->ab[0] * p->ac[1] - p->ab[1] * p->ac[0];
    s = p->n[0] * p->n[0] + p->n[1] * p->n[1] + p->n[2] * p->n[2];
    inv = (float)(1.0 / sqrt((double)s));
    for (k = 0; k < 3; k++) p->n[k] *= inv;
  }
  fclose(f);
  return res;
}

RT_Surface *rtSurfaceLoad(const char *filename, uint32_t *n) {
  FILE *f = fopen(filename, "r");
  char buf[1024], *line, *tok;
  RT_Surface *res = NULL;
  int count = -1, i = 0, j;
  float val = 0.0f;
  if (!f) { errno = E_IO; return NULL; }
  while ((line = read_line(f, buf, sizeof buf))) {
    if (count < 0) {
      count = 0; sscanf(line, "%d", &count);
      *n = (uint32_t)count;
      res = calloc(count > 0 ? count : 1, sizeof *res);
      if (!res) { errno = E_MEMORY; break; }
    } else if (i < count) { /* kd ks g ka R G B kt eta kr: first ten space/tab separated tokens */
      RT_Surface *q = &res[i++];
      float *slot[10] = {&q->kd, &q->ks, &q->g, &q->ka, &q->color.c[0], &q->color.c[1], &q->color.c[2], &q->kt, &q->eta, &q->kr};
      for (j = 0, tok = strtok(line, " \t"); tok && j < 10; tok = strtok(NULL, " \t"), j++) {
        sscanf(tok, "%f", &val); /* an unparsable token repeats the previous value (src/scene.c:550) */
        *slot[j] = (j >= 4 && j <= 6) ? (val <= 1.0f ? val : val / 255.0f) : val;
      }
    }
  }
  fclose(f);
  return res;
}

RT_Camera *rtCameraLoad(const char *filename) {
  FILE *f = fopen(filename, "r");
  char buf[1024], *line;
  RT_Camera *cam = NULL;
  int stage = 0;
  if (!f) { errno = E_IO; return NULL; }
  while ((line = read_line(f, buf, sizeof buf))) {
    if (stage == 0) {
      cam = calloc(1, sizeof *cam);
      if (!cam) { errno = E_MEMORY; break; }
    }
    if (stage < 4) { /* observer, upper-left, bottom-left, upper-right (src/scene.c:617-650) */
      float *dst = stage == 0 ? cam->ob : stage == 1 ? cam->ul : stage == 2 ? cam->bl : cam->ur;
      sscanf(line, "%f %f %f", &dst[0], &dst[1], &dst[2]);
    } else if (stage == 4) {
      sscanf(line, "%d %d", &cam->sw, &cam->sh);
    }
    stage++;
  }
  fclose(f);
  return cam;
}

void rtCameraDestroy(RT_Camera **self) {
  free(*self);
  *self = NULL;
}

/* Binary cache of a geometry file: what rtSceneLoad leaves in the triangles (nudged vertices, surface ids) and the
 * scene's bounds, keyed by the size and modification time of the text file.  A cache that does not match its source,
 * or cannot be read, is ignored and rewritten. */
typedef struct { char magic[8]; int64_t src_size, src_mtime_ns; int32_t nt, pad; float dmin[3], dmax[3]; } GeomCacheHeader;
static int source_stamp(const char *path, int64_t *size, int64_t *mtime_ns) {
  struct stat st;
  if (stat(path, &st)) return 0;
  *size = (int64_t)st.st_size;
  *mtime_ns = (int64_t)st.st_mtim.tv_sec * 1000000000ll + st.st_mtim.tv_nsec;
  return 1;
}
RT_Scene *rtSceneLoadCached(const char *filename, const char *cache) {
  GeomCacheHeader h;
  int64_t size = 0, mtime = 0;
  RT_Scene *s = NULL;
  FILE *f;
  int32_t i;
  if (!cache || !source_stamp(filename, &size, &mtime)) return rtSceneLoad(filename);
  f = fopen(cache, "rb");
  if (f) {
    float *rec = NULL;
    if (fread(&h, sizeof h, 1, f) == 1 && !memcmp(h.magic, "RTGEOM1", 8) && h.src_size == size && h.src_mtime_ns == mtime && h.nt >= 0 &&
        (s = calloc(1, sizeof *s)) && (s->t = calloc(h.nt > 0 ? h.nt : 1, sizeof *s->t)) && (rec = malloc(sizeof(float) * 10 * (h.nt > 0 ? h.nt : 1))) &&
        fread(rec, sizeof(float) * 10, (size_t)h.nt, f) == (size_t)h.nt) {
      s->nt = h.nt;
      memcpy(s->dmin, h.dmin, 12); memcpy(s->dmax, h.dmax, 12);
      for (i = 0; i < h.nt; i++) {
        RT_Triangle *t = &s->t[i];
        memcpy(t->i, rec + 10 * i, 12); memcpy(t->j, rec + 10 * i + 3, 12); memcpy(t->k, rec + 10 * i + 6, 12);
        memcpy(&t->sid, rec + 10 * i + 9, 4);
        t->tj[0] = 1.0f; t->tk[1] = 1.0f;
      }
      s->cfg.epsilon = 0.0f; s->cfg.gamma = 2.5f; s->cfg.distmod = 2.0f; s->cfg.vmode = VOX_DEFAULT;
      free(rec); fclose(f);
      errno = 0;
      return s;
    }
    free(rec);
    if (s) { free(s->t); free(s); s = NULL; }
    fclose(f);
  }
  errno = 0;  /* (a missing cache is not an error) */
  s = rtSceneLoad(filename);
  {
    const int load_errno = errno;  /* what the caller sees: whether the cache can be written does not matter */
    if (s && load_errno == 0 && (f = fopen(cache, "wb"))) {
      float *rec = malloc(sizeof(float) * 10 * (s->nt > 0 ? s->nt : 1));
      int ok = rec != NULL;
      memset(&h, 0, sizeof h);
      memcpy(h.magic, "RTGEOM1", 8);
      h.src_size = size; h.src_mtime_ns = mtime; h.nt = s->nt;
      memcpy(h.dmin, s->dmin, 12); memcpy(h.dmax, s->dmax, 12);
      for (i = 0; ok && i < s->nt; i++) {
        const RT_Triangle *t = &s->t[i];
        memcpy(rec + 10 * i, t->i, 12); memcpy(rec + 10 * i + 3, t->j, 12); memcpy(rec + 10 * i + 6, t->k, 12);
        memcpy(rec + 10 * i + 9, &t->sid, 4);
      }
      ok = ok && fwrite(&h, sizeof h, 1, f) == 1 && fwrite(rec, sizeof(float) * 10, (size_t)s->nt, f) == (size_t)s->nt;
      free(rec);
      if (fclose(f) != 0) ok = 0;
      if (!ok) remove(cache);
    }
    errno = load_errno;
  }
  return s;
}

static RT_Scene *load_all(const char *brs, const char *cache, const char *lgt, const char *pnr, const char *atr, const char *cfg,
                          float gamma, float epsilon, float distmod);
RT_Scene *rtSceneLoadAll(const char *brs, const char *lgt, const char *pnr, const char *atr, const char *cfg,
                         float gamma, float epsilon, float distmod) {
  return load_all(brs, NULL, lgt, pnr, atr, cfg, gamma, epsilon, distmod);
}
RT_Scene *rtSceneLoadAllCached(const char *brs, const char *cache, const char *lgt, const char *pnr, const char *atr, const char *cfg,
                               float gamma, float epsilon, float distmod) {
  return load_all(brs, cache, lgt, pnr, atr, cfg, gamma, epsilon, distmod);
}
static RT_Scene *load_all(const char *brs, const char *cache, const char *lgt, const char *pnr, const char *atr, const char *cfg,
                         float gamma, float epsilon, float distmod) {
  RT_Scene *scene;
  RT_Surface *surf;
  uint32_t n = 0;
  FILE *probe = fopen(brs, "r");
  if (!probe) { errno = E_IO; return NULL; } /* the reference dereferences NULL here (src/scene.c:220) */
  fclose(probe);
  errno = 0;
  scene = cache ? rtSceneLoadCached(brs, cache) : rtSceneLoad(brs);
  if (!scene || errno > 0) { rtSceneDestroy(&scene); if (!errno) errno = E_IO; return NULL; }
  scene->cfg.epsilon = epsilon; scene->cfg.gamma = gamma; scene->cfg.distmod = distmod; /* src/main.c:139-141 */
  if (cfg) rtSceneConfigureRenderer(scene, cfg);
  errno = 0; /* a missing config is only a warning (src/main.c:144-147) */
  if (lgt) {
    RT_Light *l = rtLightLoad(lgt, &n);
    if (errno > 0) errno = 0; else rtSceneSetLights(scene, l, n);
  }
  if (pnr) {
    RT_PlanarLight *pl = rtPlanarLightLoad(pnr, &n);
    if (errno > 0) errno = 0; else rtSceneSetPlanarLights(scene, pl, n);
  }
  surf = rtSurfaceLoad(atr, &n);
  if (errno > 0 || !surf) { rtSceneDestroy(&scene); if (!errno) errno = E_IO; return NULL; }
  if (!rtSceneSetSurfaces(scene, surf, n)) { int e = errno; rtSceneDestroy(&scene); errno = e; return NULL; }
  return scene;
}

RT_SceneFlat *rtSceneFlatten(const RT_Scene *s) {
  RT_SceneFlat *f = calloc(1, sizeof *f);
  int i, k;
  if (!f) { errno = E_MEMORY; return NULL; }
  f->vertices = malloc(sizeof(float) * 9 * (size_t)(s->nt > 0 ? s->nt : 1));
  f->sid = malloc(sizeof(int32_t) * (size_t)(s->nt > 0 ? s->nt : 1));
  f->surfaces = calloc(s->ns > 0 ? s->ns : 1, sizeof *f->surfaces);
  f->lights = calloc(s->nl > 0 ? s->nl : 1, sizeof *f->lights);
  f->planar = calloc(s->npl > 0 ? s->npl : 1, sizeof *f->planar);
  if (!f->vertices || !f->sid || !f->surfaces || !f->lights || !f->planar) { rtSceneFlatFree(&f); errno = E_MEMORY; return NULL; }
  for (i = 0; i < s->nt; i++) {
    const RT_Triangle *t = &s->t[i];
    for (k = 0; k < 3; k++) { f->vertices[9 * i + k] = t->i[k]; f->vertices[9 * i + 3 + k] = t->j[k]; f->vertices[9 * i + 6 + k] = t->k[k]; }
    f->sid[i] = t->sid;
  }
  for (i = 0; i < s->ns; i++) {
    const RT_Surface *q = &s->s[i];
    rtgpu_surface *o = &f->surfaces[i];
    for (k = 0; k < 3; k++) o->color[k] = q->color.c[k];
    o->kd = q->kd; o->ks = q->ks; o->g = q->g; o->ka = q->ka; o->kt = q->kt; o->eta = q->eta; o->kr = q->kr;
  }
  for (i = 0; i < s->nl; i++) {
    for (k = 0; k < 3; k++) { f->lights[i].p[k] = s->l[i].p[k]; f->lights[i].color[k] = s->l[i].color.c[k]; }
    f->lights[i].flux = s->l[i].flux;
  }
  for (i = 0; i < s->npl; i++) {
    const RT_PlanarLight *p = &s->pl[i];
    f->planar[i].flux = p->flux;
    for (k = 0; k < 3; k++) { f->planar[i].color[k] = p->color.c[k]; f->planar[i].a[k] = p->a[k]; f->planar[i].ab[k] = p->ab[k]; f->planar[i].ac[k] = p->ac[k]; }
  }
  f->desc.nt = s->nt; f->desc.ns = s->ns; f->desc.nl = s->nl; f->desc.npl = s->npl;
  f->desc.vertices = f->vertices; f->desc.sid = f->sid; f->desc.surfaces = f->surfaces;
  f->desc.lights = f->lights; f->desc.planar = f->planar;
  for (k = 0; k < 3; k++) { f->desc.dmin[k] = s->dmin[k]; f->desc.dmax[k] = s->dmax[k]; f->desc.vcoeff[k] = s->cfg.vcoeff[k]; }
  f->desc.gamma = s->cfg.gamma; f->desc.distmod = s->cfg.distmod; f->desc.vmode = (int32_t)s->cfg.vmode;
  f->desc.texture_present = s->texture != NULL;
  return f;
}

void rtSceneFlatFree(RT_SceneFlat **flat) {
  RT_SceneFlat *f = *flat;
  if (!f) return;
  free(f->vertices); free(f->sid); free(f->surfaces); free(f->lights); free(f->planar);
  free(f);
  *flat = NULL;
}

/* Tile bookkeeping of a frame split over ndev devices: the image is cut into 8x8-pixel tiles numbered row-major, tile t
 * belongs to device t % ndev.  tiles[r] = number of tiles of device r; first_pixel[r] = offset, in pixels, of device r's
 * block in the rank-major gathered buffer (first_pixel[ndev] = its total length): the layout rtgpu_untile reads. */
void rtTilePlan(int32_t width, int32_t height, int ndev, int64_t *tiles, int64_t *first_pixel) {
  const int64_t ntiles = (int64_t)((width + 7) / 8) * ((height + 7) / 8);
  int64_t at = 0;
  int r;
  for (r = 0; r < ndev; r++) {
    tiles[r] = r < ntiles ? (ntiles - r + ndev - 1) / ndev : 0;
    first_pixel[r] = at;
    at += tiles[r] * 64;
  }
  first_pixel[ndev] = at;
}

void rtCameraFlatten(const RT_Camera *cam, rtgpu_camera *out) {
  int k;
  for (k = 0; k < 3; k++) { out->ob[k] = cam->ob[k]; out->ul[k] = cam->ul[k]; out->bl[k] = cam->bl[k]; out->ur[k] = cam->ur[k]; }
  out->sw = cam->sw; out->sh = cam->sh;
}

/* src/preprocess.c:13 and :41-79: the cwd-relative texture file acts as an on/off switch; the 1st, 2nd,
 * 7th and 8th triangle with surface id 7 receive it together with fixed texture coordinates. */
RT_Scene *rtScenePreprocess(RT_Scene *scene, RT_Camera *camera) {
  static const float uv[4][6] = {{0, 1, 1, 0, 0, 0}, {0, 1, 1, 1, 1, 0}, {0, 1, 0.8f, 0, 0, 0}, {0, 1, 0.8f, 1, 0.8f, 0}};
  int i, seen = 0, saved = errno;
  (void)camera;
  if (!scene->texture) {
    scene->texture = rtBitmapLoad("textures/brickwall.bmp");
    if (!scene->texture && saved == 0) errno = E_IO; /* the reference leaves E_IO set; its caller reports a warning (src/main.c:190-193) */
  }
  for (i = 0; i < scene->nt; i++) {
    RT_Triangle *t = &scene->t[i];
    t->texture = NULL;
    if (t->sid == 7) {
      int slot = seen == 0 ? 0 : seen == 1 ? 1 : seen == 6 ? 2 : seen == 7 ? 3 : -1;
      if (slot >= 0) {
        t->ti[0] = uv[slot][0]; t->ti[1] = uv[slot][1]; t->tj[0] = uv[slot][2]; t->tj[1] = uv[slot][3];
        t->tk[0] = uv[slot][4]; t->tk[1] = uv[slot][5];
        t->texture = scene->texture;
      }
      seen++;
    }
  }
  return scene;
}
