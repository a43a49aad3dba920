/* TEST INFRASTRUCTURE — CPU oracle for the per-pixel trace path (see rt_oracle.h).
 *
 * Every function cites the reference file:line (under /root/reference/src) whose
 * arithmetic it restates.  Build with -O2 -ffp-contract=off: the reference is
 * strict binary32 with no fused multiply-add, plus the double-precision spots
 * called out below.
 */
#include "rt_oracle.h"
#include <errno.h>
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef float v3[3];

typedef struct {
  float kd, ks, g, ka, col[3], kt, eta, kr; /* scene.h:32-36 */
} Surf;

typedef struct {
  v3 a, b, c;        /* vertices i, j, k (scene.h:41) */
  float ta[2], tb[2], tc[2]; /* texture coordinates ti, tj, tk */
  int tex;           /* texture pointer != NULL */
  int sid;
  v3 n, ab, ac;      /* n, ij, ik */
  float d;
} Tri;

typedef struct { v3 p; float flux; float col[3]; } Light;       /* scene.h:59-63 */
typedef struct { float flux; float col[3]; v3 a, b, c, ab, ac, n; } PLight; /* scene.h:67-73 */

struct ora_scene {
  /* as loaded (pristine) */
  float eps, gamma, distmod; int vmode; float vcoeff[3];
  v3 dmin0, dmax0;
  int nt, nl, npl, ns;
  Tri *t; Light *l; PLight *pl; Surf *s;
  v3 ob, ul, bl, ur; int sw, sh;
  int texture_present;
  /* per render */
  v3 dmin, dmax;
  int nv[3]; float cs[3];
  int *cell_start; int *refs;  /* CSR of the grid, refs in insertion (= triangle) order */
  int *shadow_cache;           /* [thread][nt*nl], -1 = empty */
  float total_flux;
};

/* ---------- text input (scene.c:16-34) ---------- */
static char *next_line(FILE *f, char *buf) {
  while (fgets(buf, 1024, f)) {
    size_t n = strlen(buf), i;
    for (i = 0; i < n; i++)
      if (buf[i] != ' ' && buf[i] != '\t' && buf[i] != '\n' && buf[i] != '\r') break;
    if (i == n) continue;
    if (buf[0] == '/' && buf[1] == '/') continue; /* comment only at column 0 (scene.c:29) */
    return buf;
  }
  return NULL;
}

/* ---------- vector helpers (vectormath.h) ---------- */
static inline float dot3(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; } /* :95 */
static inline void cross3(float *o, const float *a, const float *b) { /* :87-92 */
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
static inline void norm3(float *v) { /* :60-66 — 1.0f/sqrt(float) is evaluated in double, then rounded */
  float s = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  float inv = (float)(1.0 / sqrt((double)s));
  v[0] *= inv; v[1] *= inv; v[2] *= inv;
}
static inline void ray3(float *o, const float *a, const float *b) { /* :122-127 */
  o[0] = b[0] - a[0]; o[1] = b[1] - a[1]; o[2] = b[2] - a[2];
  norm3(o);
}
static inline float dist3(const float *a, const float *b) { /* :45-48 */
  float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  return (float)sqrt((double)(dx * dx + dy * dy + dz * dz));
}
static inline float absf(float x) { return x >= 0.0f ? x : -x; } /* common.h:27 */
#define MIN3(a, b, c) ((c) < ((a) < (b) ? (a) : (b)) ? (c) : ((a) < (b) ? (a) : (b))) /* common.h:35 */
#define MAX3(a, b, c) ((c) > ((a) > (b) ? (a) : (b)) ? (c) : ((a) > (b) ? (a) : (b))) /* common.h:38 */

/* float -> int32 as x86-64 cvttss2si does it (what the reference binary executes
 * for every `int = float` assignment): out-of-range and NaN give INT_MIN. */
static inline int f2i(float x) {
  if (!(x > -2147483904.0f && x < 2147483648.0f)) return INT_MIN;
  return (int)x;
}

/* ---------- loaders ---------- */
static void nudge(float *p, const float *cent, float *dmin, float *dmax) { /* scene.c:147-181 */
  v3 dir; int k;
  ray3(dir, cent, p);
  for (k = 0; k < 3; k++) {
    if (dir[k] < 0.0f) p[k] += 0.0000001f;       /* += -delta, delta = -1e-7f */
    else if (dir[k] > 0.0f) p[k] += -0.0000001f; /* += delta */
    if (p[k] < dmin[k]) dmin[k] = p[k];
    if (p[k] > dmax[k]) dmax[k] = p[k];
  }
}

static int load_brs(ora_scene *s, const char *path) { /* scene.c:38-226 */
  FILE *f = fopen(path, "r");
  char buf[1024], *line;
  int vcount = -1, tcount = -1, pcount = -1, i = 0, k;
  v3 *v = NULL;
  if (!f) return -1;
  for (k = 0; k < 3; k++) { s->dmin0[k] = FLT_MAX; s->dmax0[k] = FLT_MIN; } /* :66-69 (FLT_MIN, not -FLT_MAX) */
  while ((line = next_line(f, buf))) {
    if (vcount == -1) {
      sscanf(line, "%d", &vcount);
      v = malloc(sizeof(v3) * (vcount > 0 ? vcount : 1));
      i = 0;
    } else if (vcount > 0) {
      sscanf(line, "%f %f %f", &v[i][0], &v[i][1], &v[i][2]);
      for (k = 0; k < 3; k++) {
        if (v[i][k] < s->dmin0[k]) s->dmin0[k] = v[i][k];
        if (v[i][k] > s->dmax0[k]) s->dmax0[k] = v[i][k];
      }
      i++; vcount--;
    } else if (tcount == -1) {
      sscanf(line, "%d", &tcount);
      s->nt = tcount;
      s->t = calloc(tcount > 0 ? tcount : 1, sizeof(Tri));
      i = 0;
    } else if (tcount > 0) {
      int a, b, c; Tri *t = &s->t[i]; v3 cent;
      sscanf(line, "%d %d %d", &a, &b, &c);
      memcpy(t->a, v[a], sizeof(v3)); memcpy(t->b, v[b], sizeof(v3)); memcpy(t->c, v[c], sizeof(v3));
      t->ta[0] = 0.0f; t->ta[1] = 0.0f; t->tb[0] = 1.0f; t->tb[1] = 0.0f; t->tc[0] = 0.0f; t->tc[1] = 1.0f; /* :129-134 */
      for (k = 0; k < 3; k++) cent[k] = (t->a[k] + t->b[k] + t->c[k]) / 3.0f; /* :143-145 */
      nudge(t->a, cent, s->dmin0, s->dmax0);
      nudge(t->b, cent, s->dmin0, s->dmax0);
      nudge(t->c, cent, s->dmin0, s->dmax0);
      i++; tcount--;
    } else if (pcount == -1 || pcount > 0) { /* :188-212 */
      char *tok;
      if (pcount == -1) { i = 0; pcount = s->nt; }
      for (tok = strtok(line, " \t\n\r"); tok; tok = strtok(NULL, " \t\n\r")) {
        int sid = 0;
        sscanf(tok, "%d", &sid);
        if (i < s->nt) s->t[i].sid = sid;
        i++; pcount--;
      }
    }
  }
  free(v);
  fclose(f);
  return s->t ? 0 : -1;
}

static void load_cfg(ora_scene *s, const char *path) { /* scene.c:230-278 */
  FILE *f = path ? fopen(path, "r") : NULL;
  char buf[1024], *line, *tok, word[1024];
  if (!f) return;
  while ((line = next_line(f, buf))) {
    for (tok = strtok(line, " \t"); tok; tok = strtok(NULL, " \t")) {
      if (!strcmp(tok, "epsilon")) { tok = strtok(NULL, " \t"); if (tok) sscanf(tok, "%f", &s->eps); }
      else if (!strcmp(tok, "gamma")) { tok = strtok(NULL, " \t"); if (tok) sscanf(tok, "%f", &s->gamma); }
      else if (!strcmp(tok, "distmod")) { tok = strtok(NULL, " \t"); if (tok) sscanf(tok, "%f", &s->distmod); }
      else if (!strcmp(tok, "voxmode")) {
        tok = strtok(NULL, " \t"); word[0] = 0; if (tok) sscanf(tok, "%s", word);
        s->vmode = !strcmp(word, "MODIFIED_DEFAULT") ? 1 : !strcmp(word, "FIXED") ? 2 : 0;
      } else if (!strcmp(tok, "voxparams")) {
        int k;
        for (k = 0; k < 3; k++) { tok = strtok(NULL, " \t"); if (tok) sscanf(tok, "%f", &s->vcoeff[k]); }
      }
      if (!tok) break;
    }
  }
  fclose(f);
}

static void load_lgt(ora_scene *s, const char *path) { /* scene.c:381-429 */
  FILE *f = path ? fopen(path, "r") : NULL;
  char buf[1024], *line; int count = -1, i = 0;
  if (!f) return;
  while ((line = next_line(f, buf))) {
    if (count == -1) {
      sscanf(line, "%d", &count);
      s->l = calloc(count > 0 ? count : 1, sizeof(Light));
    } else if (i < count) {
      Light *l = &s->l[i++];
      sscanf(line, "%f %f %f %f %f %f %f", &l->p[0], &l->p[1], &l->p[2], &l->flux, &l->col[0], &l->col[1], &l->col[2]);
    }
  }
  s->nl = count > 0 ? count : 0;
  fclose(f);
}

static void load_pnr(ora_scene *s, const char *path) { /* scene.c:432-508 */
  FILE *f = path ? fopen(path, "r") : NULL;
  char buf[1024], *line; int count = -1, i = 0, k;
  if (!f) return;
  while ((line = next_line(f, buf))) {
    if (count == -1) {
      sscanf(line, "%d", &count);
      s->pl = calloc(count > 0 ? count : 1, sizeof(PLight));
    } else if (i / 4 < count) {
      PLight *p = &s->pl[i / 4];
      switch (i % 4) {
        case 0: sscanf(line, "%f %f %f %f", &p->flux, &p->col[0], &p->col[1], &p->col[2]); break;
        case 1: sscanf(line, "%f %f %f", &p->a[0], &p->a[1], &p->a[2]); break;
        case 2: sscanf(line, "%f %f %f", &p->b[0], &p->b[1], &p->b[2]); break;
        case 3: sscanf(line, "%f %f %f", &p->c[0], &p->c[1], &p->c[2]); break;
      }
      i++;
    }
  }
  s->npl = count > 0 ? count : 0;
  for (i = 0; i < s->npl; i++) { /* :496-501 */
    PLight *p = &s->pl[i];
    for (k = 0; k < 3; k++) { p->ab[k] = p->b[k] - p->a[k]; p->ac[k] = p->c[k] - p->a[k]; }
    cross3(p->n, p->ab, p->ac);
    norm3(p->n);
  }
  fclose(f);
}

static int load_atr(ora_scene *s, const char *path) { /* scene.c:511-596 */
  FILE *f = fopen(path, "r");
  char buf[1024], *line, *tok; int count = -1, i = 0, j; float tmp = 0.0f;
  if (!f) return -1;
  while ((line = next_line(f, buf))) {
    if (count == -1) {
      sscanf(line, "%d", &count);
      s->s = calloc(count > 0 ? count : 1, sizeof(Surf));
    } else if (i < count) {
      Surf *q = &s->s[i++];
      j = 0;
      for (tok = strtok(line, " \t"); tok; tok = strtok(NULL, " \t")) {
        sscanf(tok, "%f", &tmp); /* a failed conversion keeps the previous value (:550) */
        switch (j) {
          case 0: q->kd = tmp; break;
          case 1: q->ks = tmp; break;
          case 2: q->g = tmp; break;
          case 3: q->ka = tmp; break;
          case 4: case 5: case 6: q->col[j - 4] = tmp <= 1.0f ? tmp : tmp / 255.0f; break; /* :566-572 */
          case 7: q->kt = tmp; break;
          case 8: q->eta = tmp; break;
          case 9: q->kr = tmp; break;
        }
        if (++j >= 10) break;
      }
    }
  }
  s->ns = count > 0 ? count : 0;
  fclose(f);
  return 0;
}

static int load_cam(ora_scene *s, const char *path) { /* scene.c:600-666 */
  FILE *f = fopen(path, "r");
  char buf[1024], *line; int stage = 0;
  if (!f) return -1;
  while ((line = next_line(f, buf))) {
    float *dst = stage == 0 ? s->ob : stage == 1 ? s->ul : stage == 2 ? s->bl : stage == 3 ? s->ur : NULL;
    if (dst) sscanf(line, "%f %f %f", &dst[0], &dst[1], &dst[2]);
    else if (stage == 4) sscanf(line, "%d %d", &s->sw, &s->sh);
    stage++;
  }
  fclose(f);
  return stage >= 5 ? 0 : -1;
}

ora_scene *ora_load(const char *brs, const char *lgt, const char *pnr, const char *atr, const char *cam,
                    const char *cfg, int texture_present) {
  ora_scene *s = calloc(1, sizeof *s);
  int i;
  if (load_brs(s, brs)) { ora_free(s); return NULL; }
  s->eps = 0.0f; s->gamma = 2.5f; s->distmod = 2.0f; s->vmode = 0; /* main.c:110,139-141 */
  load_cfg(s, cfg);
  load_lgt(s, lgt);
  load_pnr(s, pnr);
  if (load_atr(s, atr) || load_cam(s, cam)) { ora_free(s); return NULL; }
  for (i = 0; i < s->nt; i++)
    if (s->t[i].sid < 0 || s->t[i].sid >= s->ns) { ora_free(s); return NULL; } /* scene.c:291-294 */
  s->texture_present = texture_present;
  return s;
}

void ora_free(ora_scene *s) {
  if (!s) return;
  free(s->t); free(s->l); free(s->pl); free(s->s); free(s->cell_start); free(s->refs); free(s->shadow_cache);
  free(s);
}
void ora_set_resolution(ora_scene *s, int w, int h) { s->sw = w; s->sh = h; }
void ora_get_info(const ora_scene *s, int32_t out[8]) {
  out[0] = s->sw; out[1] = s->sh; out[2] = s->nt; out[3] = s->nl; out[4] = s->npl; out[5] = s->ns;
  out[6] = s->vmode; out[7] = s->texture_present;
}
float ora_get_gamma(const ora_scene *s) { return s->gamma > 0.0f ? s->gamma : 2.5f; } /* raytrace.c:333-337 */

/* ---------- preprocess (preprocess.c:8-83) ---------- */
static void preprocess(ora_scene *s) {
  int i, k, id = 0;
  for (i = 0; i < s->nt; i++) {
    Tri *t = &s->t[i]; v3 io;
    for (k = 0; k < 3; k++) { t->ab[k] = t->b[k] - t->a[k]; t->ac[k] = t->c[k] - t->a[k]; }
    for (k = 0; k < 3; k++) io[k] = s->ob[k] - t->a[k];
    norm3(io);
    cross3(t->n, t->ab, t->ac);
    norm3(t->n);
    if (dot3(t->n, io) < 0.0f) { t->n[0] = -t->n[0]; t->n[1] = -t->n[1]; t->n[2] = -t->n[2]; }
    t->d = -dot3(t->a, t->n);
  }
  for (i = 0; i < s->nt; i++) { /* :41-79 */
    Tri *t = &s->t[i];
    t->tex = 0;
    if (t->sid == 7) {
      static const float uv[4][6] = {{0, 1, 1, 0, 0, 0}, {0, 1, 1, 1, 1, 0}, {0, 1, 0.8f, 0, 0, 0}, {0, 1, 0.8f, 1, 0.8f, 0}};
      int slot = id == 0 ? 0 : id == 1 ? 1 : id == 6 ? 2 : id == 7 ? 3 : -1;
      if (slot >= 0) {
        t->ta[0] = uv[slot][0]; t->ta[1] = uv[slot][1]; t->tb[0] = uv[slot][2]; t->tb[1] = uv[slot][3];
        t->tc[0] = uv[slot][4]; t->tc[1] = uv[slot][5];
        t->tex = s->texture_present;
      }
      id++;
    }
  }
}

/* ---------- grid (voxelize.c:126-219, 236-284) ---------- */
static int grid_create(ora_scene *s) {
  float ds[3], v, tmp; int k;
  for (k = 0; k < 3; k++) {
    s->dmin[k] -= 0.001f;
    s->dmax[k] += 0.001f;
    ds[k] = s->dmax[k] - s->dmin[k] + 0.001; /* double constant: float diff promoted, sum rounded back (:142) */
  }
  if (s->vmode != 0)
    for (k = 0; k < 3; k++) if (s->vcoeff[k] <= 0.0f) return -1; /* E_INVALID_PARAM_VALUE (:169-174) */
  v = pow(s->nt / (ds[0] * ds[1] * ds[2]), 0.33333f); /* :153 */
  for (k = 0; k < 3; k++) {
    if (s->vmode == 0) tmp = ceil(ds[k] * v);
    else if (s->vmode == 1) tmp = ceil(ds[k] * v * s->vcoeff[k]);
    else tmp = ceil(s->vcoeff[k]);
    s->nv[k] = f2i(tmp);
    s->cs[k] = ds[k] / tmp;
  }
  return 0;
}
static inline int cell_of(const ora_scene *s, int i, int j, int k) { return (i * s->nv[1] + j) * s->nv[2] + k; } /* voxelize.h:47 */

static void grid_fill(ora_scene *s) {
  int nc = s->nv[0] * s->nv[1] * s->nv[2], pass, t, i, j, k;
  int *cnt;
  free(s->cell_start); free(s->refs);
  s->cell_start = calloc(nc + 1, sizeof(int));
  cnt = calloc(nc, sizeof(int));
  s->refs = NULL;
  for (pass = 0; pass < 2; pass++) {
    for (t = 0; t < s->nt; t++) {
      const Tri *q = &s->t[t];
      int lo[3], hi[3];
      for (k = 0; k < 3; k++) { /* :256-267 */
        int a = f2i((q->a[k] - s->dmin[k]) / s->cs[k]);
        int b = f2i((q->b[k] - s->dmin[k]) / s->cs[k]);
        int c = f2i((q->c[k] - s->dmin[k]) / s->cs[k]);
        lo[k] = MIN3(a, b, c); hi[k] = MAX3(a, b, c);
      }
      for (i = lo[0]; i <= hi[0]; i++)
        for (j = lo[1]; j <= hi[1]; j++)
          for (k = lo[2]; k <= hi[2]; k++) { /* every cell of the index box (:278-284) */
            int c = cell_of(s, i, j, k);
            if (pass == 0) s->cell_start[c + 1]++;
            else s->refs[s->cell_start[c] + cnt[c]++] = t;
          }
    }
    if (pass == 0) {
      for (i = 0; i < nc; i++) s->cell_start[i + 1] += s->cell_start[i];
      s->refs = malloc(sizeof(int) * (s->cell_start[nc] > 0 ? s->cell_start[nc] : 1));
    }
  }
  free(cnt);
}

static inline int voxel_of(const ora_scene *s, const float *p, int *i, int *j, int *k) { /* voxelize.h:32-43 */
  *i = f2i((p[0] - s->dmin[0]) / s->cs[0]);
  *j = f2i((p[1] - s->dmin[1]) / s->cs[1]);
  *k = f2i((p[2] - s->dmin[2]) / s->cs[2]);
  if (*i < 0 || *i >= s->nv[0]) return 0;
  if (*j < 0 || *j >= s->nv[1]) return 0;
  if (*k < 0 || *k >= s->nv[2]) return 0;
  return 1;
}

static int startup_voxel(const ora_scene *s, const float *o, const float *r, int *i, int *j, int *k) { /* voxelize.c:417-471 */
  float d1 = FLT_MAX, d2 = FLT_MAX, d; v3 p; int a, side;
  if (voxel_of(s, o, i, j, k)) return 1;
  for (a = 0; a < 3; a++) {
    if (r[a] != 0.0f) {
      for (side = 0; side < 2; side++) {
        d = ((side ? s->dmax[a] : s->dmin[a]) - o[a]) / r[a];
        if (d > 0.0f) {
          if (d < d1) { d2 = d1; d1 = d; }
          else if (d < d2) d2 = d;
        }
      }
    }
  }
  for (a = 0; a < 3; a++) p[a] = o[a] + d1 * r[a];
  if (voxel_of(s, p, i, j, k)) return 1;
  for (a = 0; a < 3; a++) p[a] = o[a] + d2 * r[a];
  if (voxel_of(s, p, i, j, k)) return 1;
  return 0;
}

/* ---------- Möller–Trumbore (intersection.c:58-86) ---------- */
static inline int hit_test(const Tri *t, const float *o, const float *r, float *d, float *u, float *v) {
  v3 p, tv, q; float det, inv;
  cross3(p, r, t->ac);
  det = dot3(t->ab, p);
  if (det > -0.000001f && det < 0.000001f) return 0;
  inv = 1.0f / det;
  tv[0] = o[0] - t->a[0]; tv[1] = o[1] - t->a[1]; tv[2] = o[2] - t->a[2];
  *u = dot3(tv, p) * inv;
  if (*u < 0.0f || *u > 1.0f) return 0;
  cross3(q, tv, t->ab);
  *v = dot3(r, q) * inv;
  if (*v < 0.0f || *u + *v > 1.0f) return 0;
  *d = dot3(t->ac, q) * inv;
  if (*d < 0.0f) return 0;
  return 1;
}

typedef struct { float dt[3], t[3]; int step[3]; } Dda;

static inline void dda_init(const ora_scene *s, const float *o, const float *r, int i, int j, int k, Dda *w) { /* voxelize.c:56-121 */
  int idx[3] = {i, j, k}, a;
  for (a = 0; a < 3; a++) {
    float p1 = s->dmin[a] + idx[a] * s->cs[a];
    float p2 = p1 + s->cs[a];
    if (r[a] == 0.0f) { w->dt[a] = FLT_MAX; w->t[a] = 0.0f; }
    else {
      float t1 = (p1 - o[a]) / r[a], t2 = (p2 - o[a]) / r[a];
      w->dt[a] = absf(t2 - t1);
      w->t[a] = t1 < t2 ? t1 : t2;
    }
    w->step[a] = r[a] > 0.0f ? 1 : -1;
  }
}
static inline void dda_step(Dda *w, int *i, int *j, int *k) { /* voxelize.c:594-606 */
  float xn = w->t[0] + w->dt[0], yn = w->t[1] + w->dt[1], zn;
  if (xn < yn) {
    zn = w->t[2] + w->dt[2];
    if (xn < zn) { *i += w->step[0]; w->t[0] = xn; } else { *k += w->step[2]; w->t[2] = zn; }
  } else {
    zn = w->t[2] + w->dt[2];
    if (yn < zn) { *j += w->step[1]; w->t[1] = yn; } else { *k += w->step[2]; w->t[2] = zn; }
  }
}

/* voxelize.c:536-613; returns triangle index or -1 */
static int find_nearest(const ora_scene *s, int current, float *ipoint, const float *o, const float *r,
                        int *i_, int *j_, int *k_, float *u, float *v, ora_counters *cnt) {
  Dda w; int i = *i_, j = *j_, k = *k_, c;
  dda_init(s, o, r, i, j, k, &w);
  if (cnt) cnt->nearest_queries++;
  for (;;) {
    int cell = cell_of(s, i, j, k), beg = s->cell_start[cell], end = s->cell_start[cell + 1];
    if (cnt) { cnt->nearest_steps++; cnt->nearest_tests += end - beg; }
    if (end > beg) {
      float a = w.t[0] + w.dt[0], b = w.t[1] + w.dt[1], cc = w.t[2] + w.dt[2];
      float dmin = MIN3(a, b, cc), d, ut, vt; int nearest = -1;
      for (c = beg; c < end; c++) {
        int ti = s->refs[c];
        if (hit_test(&s->t[ti], o, r, &d, &ut, &vt)) {
          if (ti != current && d < dmin) { dmin = d; nearest = ti; *u = ut; *v = vt; }
        }
      }
      if (nearest >= 0) {
        ipoint[0] = o[0] + dmin * r[0]; ipoint[1] = o[1] + dmin * r[1]; ipoint[2] = o[2] + dmin * r[2]; /* vectormath.h:131 */
        *i_ = i; *j_ = j; *k_ = k;
        return nearest;
      }
    }
    dda_step(&w, &i, &j, &k);
    if (i < 0 || i >= s->nv[0]) return -1;
    if (j < 0 || j >= s->nv[1]) return -1;
    if (k < 0 || k >= s->nv[2]) return -1;
  }
}

/* voxelize.c:615-738; returns 1 if the point is in shadow, *ts = transparency product otherwise */
static int find_shadow(const ora_scene *s, int current, const float *a, const float *b, int lindex, float *ts,
                       int *cache, ora_counters *cnt) {
  int ai[3], bi[3], lo[3], hi[3], i, j, k, c; float d, u, v, dmax; v3 r; Dda w;
  const Tri *cur = &s->t[current];
  *ts = 1.0f;
  if (cnt) cnt->shadow_queries++;
  ray3(r, a, b);
  if (s->s[cur->sid].kt == 0.0f) { /* :640-644 */
    if (dot3(r, cur->n) <= 0.0f) { if (cnt) cnt->shadow_backface++; return 1; }
  }
  if (lindex >= 0 && cache) { /* :647-655 */
    int *slot = &cache[(size_t)current * s->nl + lindex];
    if (*slot >= 0) {
      if (hit_test(&s->t[*slot], a, r, &d, &u, &v)) { if (cnt) cnt->shadow_cached++; return 1; }
      *slot = -1;
    }
  }
  dmax = dist3(a, b);
  if (!voxel_of(s, a, &ai[0], &ai[1], &ai[2])) return 0; /* :661-664 */
  if (!voxel_of(s, b, &bi[0], &bi[1], &bi[2])) return 0; /* :667-670 */
  for (c = 0; c < 3; c++) { lo[c] = ai[c] < bi[c] ? ai[c] : bi[c]; hi[c] = ai[c] < bi[c] ? bi[c] : ai[c]; }
  i = ai[0]; j = ai[1]; k = ai[2];
  dda_init(s, a, r, i, j, k, &w);
  for (;;) {
    int cell = cell_of(s, i, j, k), beg = s->cell_start[cell], end = s->cell_start[cell + 1];
    if (cnt) { cnt->shadow_steps++; cnt->shadow_tests += end - beg; }
    for (c = beg; c < end; c++) {
      int ti = s->refs[c];
      if (hit_test(&s->t[ti], a, r, &d, &u, &v)) {
        if (ti != current) {
          float kt = s->s[s->t[ti].sid].kt;
          if (kt > 0.0f) { *ts *= kt; continue; } /* :703-706 */
          if (d > 0.00001f && d < dmax) {
            if (lindex >= 0 && cache) cache[(size_t)current * s->nl + lindex] = ti;
            return 1;
          }
        }
      }
    }
    dda_step(&w, &i, &j, &k);
    if (i < lo[0] || i > hi[0]) return 0;
    if (j < lo[1] || j > hi[1]) return 0;
    if (k < lo[2] || k > hi[2]) return 0;
  }
}

/* ---------- procedural texture (texture.c) ---------- */
static const unsigned char perm256[256] = { /* texture.c:13-25, Perlin's reference permutation */
  151,160,137,91,90,15,131,13,201,95,96,53,194,233,7,225,140,36,103,30,69,142,8,99,37,240,21,10,23,190,6,148,
  247,120,234,75,0,26,197,62,94,252,219,203,117,35,11,32,57,177,33,88,237,149,56,87,174,20,125,136,171,168,68,175,
  74,165,71,134,139,48,27,166,77,146,158,231,83,111,229,122,60,211,133,230,220,105,92,41,55,46,245,40,244,102,143,54,
  65,25,63,161,1,216,80,73,209,76,132,187,208,89,18,169,200,196,135,130,116,188,159,86,164,100,109,198,173,186,3,64,
  52,217,226,250,124,123,5,202,38,147,118,126,255,82,85,212,207,206,59,227,47,16,58,17,182,189,28,42,23,183,170,213,
  119,248,152,2,44,154,163,70,221,153,101,155,167,43,172,9,129,22,39,253,19,98,108,110,79,113,224,232,178,185,112,104,
  218,246,97,228,251,34,242,193,238,210,144,12,191,179,162,241,81,51,145,235,249,14,239,107,49,192,214,31,181,199,106,157,
  184,84,204,176,115,121,50,45,127,4,150,254,138,236,205,93,222,114,67,29,24,72,243,141,128,195,78,66,215,61,156,180};
static inline int P(int i) { return perm256[i & 255]; } /* the table is stored twice (512 entries, :26-38) */
static inline double fade(double t) { return t * t * t * (t * (t * 6 - 15) + 10); } /* :41-44 */
static inline double lerpd(double t, double a, double b) { return a + t * (b - a); } /* :46-49 */
static inline double grad(int hash, double x, double y, double z) { /* :51-58 */
  int h = hash & 15;
  double u = h < 8 || h == 12 || h == 13 ? x : y, v = h < 4 || h == 12 || h == 13 ? y : z;
  return ((h & 1) == 0 ? u : -u) + ((h & 2) == 0 ? v : -v);
}
static double noise3(double x, double y, double z) { /* :60-87 */
  int X, Y, Z, A, AA, AB, B, BA, BB; double u, v, w;
  X = (int)floor(x) & 255; Y = (int)floor(y) & 255; Z = (int)floor(z) & 255;
  x -= floor(x); y -= floor(y); z -= floor(z);
  u = fade(x); v = fade(y); w = fade(z);
  A = P(X) + Y; AA = P(A) + Z; AB = P(A + 1) + Z;
  B = P(X + 1) + Y; BA = P(B) + Z; BB = P(B + 1) + Z;
  return lerpd(w, lerpd(v, lerpd(u, grad(P(AA), x, y, z), grad(P(BA), x - 1, y, z)),
                        lerpd(u, grad(P(AB), x, y - 1, z), grad(P(BB), x - 1, y - 1, z))),
               lerpd(v, lerpd(u, grad(P(AA + 1), x, y, z - 1), grad(P(BA + 1), x - 1, y, z - 1)),
                        lerpd(u, grad(P(AB + 1), x, y - 1, z - 1), grad(P(BB + 1), x - 1, y - 1, z - 1))));
}
/* texture.c:89-155 with the literal arguments of raytrace.c:84-91; the cos() bump terms feed an unused array */
static void bricks(float x, float y, float *color) {
  const float bheight = 0.04f, bwidth = 0.10f, filling = 0.005f, rfactor = 2160.0f, gfactor = 0.0f, bfactor = 0.0f, brickpos = 33;
  float w = 2 * filling + bwidth, h = 2 * filling + bheight;
  const float brick[3] = {173 / 255.0f, 106 / 255.0f, 64 / 255.0f}, fill[3] = {215 / 255.0f, 205 / 255.0f, 178 / 255.0f};
  float basef = 0.7f, derf = 0.4f, posmod[4], bl, br, bt, bb; const float *base;
  double ay = y / h; int row = floor(ay);
  double ax = (x / w) + ((row & 1) ? 0.5 : 0); int col = floor(ax);
  ax = ax - col; ay = ay - row;
  posmod[0] = 0.2f * noise3(brickpos * row, brickpos * col, 0.435);
  posmod[1] = 0.2f * noise3(brickpos * row, brickpos * col, 0.645);
  posmod[2] = 0.2f * noise3(brickpos * row, brickpos * col, 0.354);
  posmod[3] = 0.2f * noise3(brickpos * row, brickpos * col, 0.768);
  bl = filling / w + posmod[0] * filling / w;
  br = (w - filling) / w + posmod[1] * (w - filling) / w;
  bt = filling / h + posmod[2] * filling / h;
  bb = (h - filling) / h + posmod[3] * (h - filling) / h;
  if (ax < bl || ax > br || ay < bt || ay > bb) base = fill; else base = brick;
  color[0] = base[0]; color[1] = base[1]; color[2] = base[2];
  if (base == brick) {
    float nz = (float)noise3(row * x, col * y, row * col);
    color[0] += basef * nz; color[1] += basef * nz; color[2] += basef * nz;
  }
  color[0] += derf * (float)noise3(rfactor * x, rfactor * y, row * col);
  color[1] += derf * (float)noise3(gfactor * x, gfactor * y, row * col);
  color[2] += derf * (float)noise3(bfactor * x, bfactor * y, row * col);
}
/* raytrace.c:80-128.  Returns 1 if the reference would read an uninitialised sample here. */
static int apply_texture(const Tri *t, float *norm, float *nc, float u, float v) {
  float px = t->ta[0] + (t->tb[0] - t->ta[0]) * u + (t->tc[0] - t->ta[0]) * v;
  float py = t->ta[1] + (t->tb[1] - t->ta[1]) * u + (t->tc[1] - t->ta[1]) * v;
  const float delta = 0.002f; float tmp, ugrad, vgrad; int k, taint = 0;
  float avg[3], cx1[3] = {0, 0, 0}, cx2[3] = {0, 0, 0}, cy1[3] = {0, 0, 0}, cy2[3] = {0, 0, 0};
  bricks(px, py, avg);
  if ((tmp = px - delta) >= 0.0f) bricks(tmp, py, cx1); else taint = 1;
  if ((tmp = px + delta) <= 1.0f) bricks(tmp, py, cx2); else taint = 1;
  if ((tmp = py - delta) >= 0.0f) bricks(px, tmp, cy1); else taint = 1;
  if ((tmp = py + delta) <= 1.0f) bricks(px, tmp, cy2); else taint = 1;
  nc[0] = avg[0]; nc[1] = avg[1]; nc[2] = avg[2];
  ugrad = (cx2[0] + cx2[1] + cx2[2]) * 0.333f - (cx1[0] + cx1[1] + cx1[2]) * 0.333f;
  vgrad = (cy2[0] + cy2[1] + cy2[2]) * 0.333f - (cy1[0] + cy1[1] + cy1[2]) * 0.333f;
  for (k = 0; k < 3; k++) norm[k] = norm[k] + t->ab[k] * ugrad;
  for (k = 0; k < 3; k++) norm[k] = norm[k] + t->ac[k] * (-vgrad);
  norm3(norm);
  return taint;
}

/* ---------- shading recursion (raytrace.c:144-299) ---------- */
typedef struct { ora_counters *cnt; int *cache; int primary; int taint; } TraceCtx;

static void reflect_dir(float *out, const float *n, const float *l, float ndl) { /* vectormath.h:144-159 */
  out[0] = 2.0f * n[0] * ndl - l[0];
  out[1] = 2.0f * n[1] * ndl - l[1];
  out[2] = 2.0f * n[2] * ndl - l[2];
  norm3(out);
}

static void light_term(const ora_scene *s, const Surf *sf, const float *norm, const float *r, const float *rnew,
                       const float *onew, const float *lp, float lflux, const float *lcol, const float *nc, float ts,
                       float *out) { /* raytrace.c:217-235 / 271-289 */
  float ndl = dot3(norm, rnew), df, rf = 0.0f, scale; v3 tmpv; int k;
  df = sf->kd * ndl;
  if (df < 0.0f && sf->kt > 0.0f) df = -df;
  if (sf->ks > 0.0f) {
    reflect_dir(tmpv, norm, rnew, ndl);
    rf = sf->ks * pow(dot3(r, tmpv), sf->g); /* float * double, rounded on assignment */
    if (rf < 0.0f && sf->kt > 0.0f) rf = -rf;
  }
  scale = ts * lflux * (df + rf) / (dist3(onew, lp) + s->distmod);
  for (k = 0; k < 3; k++) out[k] = (lcol[k] + nc[k]) * scale;
}

static void trace(const ora_scene *s, TraceCtx *cx, int current, const float *o, const float *r, int level,
                  int i, int j, int k, float *res) {
  v3 onew, rnew, rray, inv, norm, nc, rcolor, tmp = {0, 0, 0}; float u = 0, v = 0, ts; int nearest, c, d, q;
  const Tri *t; const Surf *sf;
  res[0] = res[1] = res[2] = 0.0f;
  if (level == 0) return;
  nearest = find_nearest(s, current, onew, o, r, &i, &j, &k, &u, &v, cx->cnt);
  if (nearest < 0) return;
  if (cx->primary < 0) cx->primary = nearest;
  t = &s->t[nearest]; sf = &s->s[t->sid];
  for (q = 0; q < 3; q++) norm[q] = t->n[q];
  if (dot3(r, norm) > 0.0f) for (q = 0; q < 3; q++) norm[q] = -norm[q];
  for (q = 0; q < 3; q++) nc[q] = sf->col[q];
  if (t->sid == 7 && t->tex) cx->taint |= apply_texture(t, norm, nc, u, v);
  if (sf->ka > 0.0f) { float f = sf->ka * s->total_flux; for (q = 0; q < 3; q++) res[q] = nc[q] * f; }
  for (q = 0; q < 3; q++) inv[q] = -r[q];
  if (sf->kr > 0.0f) { /* :193-197 */
    reflect_dir(rray, norm, inv, dot3(norm, inv));
    trace(s, cx, nearest, onew, rray, level - 1, i, j, k, rcolor);
    for (q = 0; q < 3; q++) { rcolor[q] = rcolor[q] * sf->kr; res[q] = res[q] + rcolor[q]; }
  }
  if (sf->kt > 0.0f) { /* :200-204, vectormath.h:162-169 — the sqrt and the subtraction are done in double */
    float ndl = dot3(norm, inv), eta = sf->eta;
    float f = eta * ndl - sqrt(1.0f - (eta * eta) * (1.0f - ndl * ndl));
    for (q = 0; q < 3; q++) rray[q] = f * norm[q] - eta * inv[q];
    norm3(rray);
    trace(s, cx, nearest, onew, rray, level - 1, i, j, k, rcolor);
    for (q = 0; q < 3; q++) { rcolor[q] = rcolor[q] * sf->kt; res[q] = res[q] + rcolor[q]; }
  }
  for (c = 0; c < s->nl; c++) { /* :211-238 */
    const Light *l = &s->l[c];
    ray3(rnew, onew, l->p);
    if (!find_shadow(s, nearest, onew, l->p, c, &ts, cx->cache, cx->cnt)) {
      light_term(s, sf, norm, r, rnew, onew, l->p, l->flux, l->col, nc, ts, tmp);
      for (q = 0; q < 3; q++) res[q] = res[q] + tmp[q];
    }
  }
  for (c = 0; c < s->npl; c++) { /* :241-296, accumulation reproduced as written */
    const PLight *pl = &s->pl[c]; v3 sum = {0, 0, 0}; const int nsamples = 16;
    for (d = 0; d < nsamples; d++) {
      float eta = rand() / (float)RAND_MAX, psi = rand() / (float)RAND_MAX, flux = pl->flux / nsamples; v3 p;
      for (q = 0; q < 3; q++) p[q] = pl->a[q];
      for (q = 0; q < 3; q++) p[q] = p[q] + pl->ab[q] * eta;
      for (q = 0; q < 3; q++) p[q] = p[q] + pl->ac[q] * psi;
      ray3(rnew, onew, p);
      if (!find_shadow(s, nearest, onew, p, -1, &ts, NULL, cx->cnt))
        light_term(s, sf, norm, r, rnew, onew, p, flux, pl->col, nc, ts, sum);
      for (q = 0; q < 3; q++) sum[q] = sum[q] * (1.0f / nsamples);
      for (q = 0; q < 3; q++) tmp[q] = tmp[q] + sum[q];
      for (q = 0; q < 3; q++) res[q] = res[q] + tmp[q];
    }
  }
}

/* ---------- frame function (raytrace.c:304-413) ---------- */
static int frame_setup(ora_scene *s, int nthreads, int use_cache) {
  int k, i;
  memcpy(s->dmin, s->dmin0, sizeof(v3)); memcpy(s->dmax, s->dmax0, sizeof(v3));
  preprocess(s);
  s->total_flux = 0.0f;
  for (k = 0; k < s->nl; k++) { /* :346-355 */
    s->total_flux += s->l[k].flux;
    for (i = 0; i < 3; i++) {
      if (s->l[k].p[i] < s->dmin[i]) s->dmin[i] = s->l[k].p[i] - 0.001f;
      if (s->l[k].p[i] > s->dmax[i]) s->dmax[i] = s->l[k].p[i] + 0.001f;
    }
  }
  if (grid_create(s)) return 7; /* E_INVALID_PARAM_VALUE */
  grid_fill(s);
  free(s->shadow_cache); s->shadow_cache = NULL;
  if (use_cache && s->nl > 0) {
    size_t n = (size_t)nthreads * s->nt * s->nl, q;
    s->shadow_cache = malloc(n * sizeof(int));
    for (q = 0; q < n; q++) s->shadow_cache[q] = -1;
  }
  return 0;
}

int ora_render_rows(ora_scene *s, int flags, int threads, int y0, int y1, float *rgb, int32_t *ids, float *minmax,
                    uint8_t *taint, ora_counters *cnt_out) {
  int w = s->sw, h = s->sh, y, k, rc;
  float w_inv = 1.0f / w, h_inv = 1.0f / h;
  int use_cache = (flags & ORA_SHADOW_CACHE) != 0;
  ora_counters total; float gmin[3], gmax[3];
  if (threads < 1) threads = 1;
  if (threads > 1 && (use_cache || s->npl > 0)) return -1;
  if ((rc = frame_setup(s, threads, use_cache))) return rc;
  memset(&total, 0, sizeof total);
  for (k = 0; k < 3; k++) { gmin[k] = FLT_MAX; gmax[k] = FLT_MIN; } /* :317-320 */
  srand(1); /* a fresh reference process starts rand() from its default seed */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) if (threads > 1)
#endif
  for (y = y0; y < y1; y++) {
    ora_counters lc; float lmin[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, lmax[3] = {FLT_MIN, FLT_MIN, FLT_MIN}; int x, q;
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    memset(&lc, 0, sizeof lc);
    for (x = 0; x < w; x++) {
      float xc = x * w_inv, yc = y * h_inv; v3 ray, color; int i, j, kk; TraceCtx cx; size_t p = (size_t)y * w + x;
      for (q = 0; q < 3; q++) /* vectormath.h:113-117 */
        ray[q] = xc * (s->ur[q] - s->ul[q]) + yc * (s->bl[q] - s->ul[q]) + s->ul[q] - s->ob[q];
      norm3(ray);
      if (!startup_voxel(s, s->ob, ray, &i, &j, &kk)) continue; /* pixel stays as allocated (:381-382) */
      cx.cnt = (flags & ORA_COUNTERS) ? &lc : NULL;
      cx.cache = s->shadow_cache ? s->shadow_cache + (size_t)tid * s->nt * s->nl : NULL;
      cx.primary = -1; cx.taint = 0;
      lc.pixels_traced++;
      trace(s, &cx, -1, s->ob, ray, 5, i, j, kk, color);
      for (q = 0; q < 3; q++) {
        if (color[q] > lmax[q]) lmax[q] = color[q];
        if (color[q] < lmin[q]) lmin[q] = color[q];
      }
      rgb[p * 3] = color[0]; rgb[p * 3 + 1] = color[1]; rgb[p * 3 + 2] = color[2];
      ids[p] = cx.primary;
      if (taint) taint[p] = (uint8_t)cx.taint;
    }
#ifdef _OPENMP
#pragma omp critical
#endif
    {
      int64_t *a = (int64_t *)&total, *b = (int64_t *)&lc; size_t n = sizeof total / sizeof(int64_t), z;
      for (z = 0; z < n; z++) a[z] += b[z];
      for (q = 0; q < 3; q++) {
        if (lmax[q] > gmax[q]) gmax[q] = lmax[q];
        if (lmin[q] < gmin[q]) gmin[q] = lmin[q];
      }
    }
  }
  for (k = 0; k < 3; k++) { minmax[k] = gmin[k]; minmax[3 + k] = gmax[k]; }
  if (cnt_out) *cnt_out = total;
  return 0;
}

int ora_render(ora_scene *s, int flags, int threads, float *rgb, int32_t *ids, float *minmax, uint8_t *taint,
               ora_counters *cnt) {
  size_t np = (size_t)s->sw * s->sh, p;
  for (p = 0; p < np; p++) { rgb[p * 3] = rgb[p * 3 + 1] = rgb[p * 3 + 2] = 0.0f; ids[p] = -1; }
  if (taint) memset(taint, 0, np);
  return ora_render_rows(s, flags, threads, 0, s->sh, rgb, ids, minmax, taint, cnt);
}

int ora_grid_dims(const ora_scene *s, int32_t nv[3], float cs[3], float dmin[3], float dmax[3]) {
  int k;
  if (!s->cell_start) return -1;
  for (k = 0; k < 3; k++) { nv[k] = s->nv[k]; cs[k] = s->cs[k]; dmin[k] = s->dmin[k]; dmax[k] = s->dmax[k]; }
  return 0;
}
int64_t ora_grid_refs(const ora_scene *s, int32_t *cell_start, int32_t *refs) {
  int nc = s->nv[0] * s->nv[1] * s->nv[2];
  memcpy(cell_start, s->cell_start, sizeof(int) * (nc + 1));
  if (refs) memcpy(refs, s->refs, sizeof(int) * s->cell_start[nc]);
  return s->cell_start[nc];
}
void ora_triangles(const ora_scene *s, float *o) {
  int i, k;
  for (i = 0; i < s->nt; i++, o += 17) {
    const Tri *t = &s->t[i];
    for (k = 0; k < 3; k++) { o[k] = t->n[k]; o[3 + k] = t->ab[k]; o[6 + k] = t->ac[k]; }
    o[9] = t->d; o[10] = (float)t->tex;
    o[11] = t->ta[0]; o[12] = t->ta[1]; o[13] = t->tb[0]; o[14] = t->tb[1]; o[15] = t->tc[0]; o[16] = t->tc[1];
  }
}
void ora_vertices(const ora_scene *s, float *o, int32_t *sid) {
  int i, k;
  for (i = 0; i < s->nt; i++, o += 9) {
    for (k = 0; k < 3; k++) { o[k] = s->t[i].a[k]; o[3 + k] = s->t[i].b[k]; o[6 + k] = s->t[i].c[k]; }
    sid[i] = s->t[i].sid;
  }
}

/* ---------- tone map + BMP (raytrace.c:424-472, bitmap.h:70, bitmap.c:249-479) ---------- */
static inline uint32_t f2u_x86(float x) { /* float -> uint32_t: 64-bit cvttss2si, low half kept */
  if (!(x > -9223373136366403584.0f && x < 9223372036854775808.0f)) return 0u;
  return (uint32_t)(uint64_t)(int64_t)x;
}
/* raytrace.c:439-466: the zero-terminated gamma list (param1) as n gammas; per channel `r += pow(x, gammas[k]) * 255.0f`
 * (double product, double sum, rounded to float at every +=), then r / (float)n, converted to uint32_t and masked
 * (bitmap.h:70).  n = 0 (an empty list): 0.0f / 0.0f = NaN, which the conversion turns into 0. */
void ora_tonemap_multi(int w, int h, const float *rgb, const float *mm, const float *gammas, int n, uint32_t *rgba) {
  float delta[3]; size_t p, np = (size_t)w * h; int k, q;
  for (k = 0; k < 3; k++) delta[k] = 1.0f / (mm[3 + k] - mm[k]);
  for (p = 0; p < np; p++) {
    uint32_t ch[3];
    for (k = 0; k < 3; k++) {
      float acc = 0.0f;
      for (q = 0; q < n; q++) acc += pow((rgb[p * 3 + k] - mm[k]) * delta[k], gammas[q]) * 255.0f;
      ch[k] = f2u_x86(acc / (float)n);
    }
    rgba[p] = ((ch[0] & 0xff) << 24) | ((ch[1] & 0xff) << 16) | ((ch[2] & 0xff) << 8);
  }
}
void ora_tonemap(int w, int h, const float *rgb, const float *mm, float gamma, uint32_t *rgba) {
  ora_tonemap_multi(w, h, rgb, mm, &gamma, 1, rgba); /* raytrace.c:439: the default list is {s->gamma, 0} */
}
int64_t ora_bmp24(int w, int h, const uint32_t *rgba, uint8_t *out) {
  uint32_t row = (uint32_t)(4 * ceil(24 * w / 32.0)); /* bitmap.c:307 */
  int64_t total = 54 + (int64_t)row * h; int x, y;
  if (!out) return total;
  memset(out, 0, total);
  out[0] = 'B'; out[1] = 'M';
  { uint32_t v = (uint32_t)total; memcpy(out + 2, &v, 4); v = 54; memcpy(out + 10, &v, 4); v = 40; memcpy(out + 14, &v, 4);
    memcpy(out + 18, &w, 4); memcpy(out + 22, &h, 4); out[26] = 1; out[28] = 24; v = (uint32_t)(total - 54); memcpy(out + 34, &v, 4); }
  for (y = h - 1; y >= 0; y--) { /* bottom-up, B G R (bitmap.c:458-467) */
    uint8_t *dst = out + 54 + (size_t)(h - 1 - y) * row;
    for (x = 0; x < w; x++) {
      uint32_t p = rgba[(size_t)y * w + x];
      dst[3 * x] = (p >> 8) & 0xff; dst[3 * x + 1] = (p >> 16) & 0xff; dst[3 * x + 2] = (p >> 24) & 0xff;
    }
  }
  return total;
}

/* glibc random_r TYPE_3 (x^31 + x^3 + 1) with srandom_r(1) seeding, as rand() runs it */
void ora_rand_fill(int32_t *out, int64_t n) {
  int32_t r[34]; uint32_t ring[31]; int i, f = 3, b = 0; int64_t q;
  r[0] = 1;
  for (i = 1; i < 31; i++) {
    long hi = r[i - 1] / 127773, lo = r[i - 1] % 127773, word = 16807 * lo - 2836 * hi;
    if (word < 0) word += 2147483647;
    r[i] = (int32_t)word;
  }
  for (i = 0; i < 31; i++) ring[i] = (uint32_t)r[i];
  for (q = -310; q < n; q++) {
    uint32_t val = ring[f] += ring[b];
    if (q >= 0) out[q] = (int32_t)(val >> 1);
    if (++f >= 31) f = 0;
    if (++b >= 31) b = 0;
  }
}
