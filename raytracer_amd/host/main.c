/* Command line of the ray tracer: the reference's options (src/main.c:36-104) plus -W/-H to override
 * the camera resolution, -d to pick the GPU, -I to dump primary-hit triangle ids and --devices N to split the
 * frame by image tiles over N GPUs of this node (one process, RCCL: host/multigpu.c).
 */
#define _GNU_SOURCE
#include "rt_host.h"

#include <dlfcn.h>
#include <libgen.h>
#include <limits.h>
#include <unistd.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static void usage(const char *exe) {
  printf("usage: %s [options]\n\n"
         "    -g PATH  geometry (.brs)      -l PATH  point lights (.lgt)   -L PATH  planar lights (.pnr)\n"
         "    -a PATH  attributes (.atr)    -c PATH  camera (.cam)         -C PATH  renderer config (.cfg)\n"
         "    -s PREFIX  PREFIX.brs/.lgt/.atr/.cam/.cfg/.pnr for everything not given explicitly\n"
         "    -o PATH  output image (24-bpp BMP)\n"
         "    -G gamma   -E epsilon   -D distmod   (a .cfg file overrides them)\n"
         "    -W width -H height  override the camera resolution    -d N  GPU index\n"
         "    -I PATH  also write primary-hit triangle ids (int32, row-major, -1 = none)\n"
         "    -B PATH  binary cache of the geometry file (written if missing or older than its source)\n"
         "    --devices N  split the frame by 8x8 tiles over GPUs 0..N-1 (RCCL: colour range all-reduce, RGBA8 tile gather)\n"
         "    --frames K   render K frames and report the time of a frame after the first\n"
         "    --gammas G1,G2,...  tone map averaged over a list of gammas (rtVisualizedSceneToBitmap's param1)\n", exe);
}

static char *join(const char *a, const char *b) {
  char *r = malloc(strlen(a) + strlen(b) + 1);
  strcpy(r, a); strcat(r, b);
  return r;
}

int main(int argc, char **argv) {
  const char *opt[128] = {0};
  const char *letters = "glLacCsoGEDWHdIB";
  char *g, *l, *L, *a, *c, *C;
  float gamma = 2.5f, epsilon = 0.0f, distmod = 2.0f;
  int i, devices = 0, frames = 1, ngammas = -1;
  float gammas[65];
  if (argc <= 1) { usage(argv[0]); return 0; }
  for (i = 1; i < argc; i++) { /* value glued to the option ("-sPREFIX") or in the next argument */
    const char *arg = argv[i];
    if (!strcmp(arg, "--devices") && i + 1 < argc) { devices = atoi(argv[++i]); if (devices < 1) { fprintf(stderr, "--devices needs a positive number\n"); return 1; } continue; }
    if (!strcmp(arg, "--frames") && i + 1 < argc) { frames = atoi(argv[++i]); if (frames < 1) frames = 1; continue; }
    if (!strcmp(arg, "--gammas") && i + 1 < argc) { /* zero-terminated, as src/raytrace.c:439-447 reads it; "," is the empty list */
      const char *q = argv[++i];
      ngammas = 0;
      while (*q && ngammas < 64) {
        char *end;
        float v = strtof(q, &end);
        if (end == q) break;
        gammas[ngammas++] = v;
        q = *end == ',' ? end + 1 : end;
      }
      gammas[ngammas] = 0.0f;
      continue;
    }
    if (arg[0] != '-' || !arg[1] || !strchr(letters, arg[1])) { fprintf(stderr, "unknown option %s\n", arg); return 1; }
    if (arg[2]) opt[(int)arg[1]] = arg + 2;
    else if (i + 1 < argc) opt[(int)arg[1]] = argv[++i];
  }
  if (opt['G']) sscanf(opt['G'], "%f", &gamma);
  if (opt['E']) sscanf(opt['E'], "%f", &epsilon);
  if (opt['D']) sscanf(opt['D'], "%f", &distmod);
  if ((!opt['s'] && (!opt['g'] || (!opt['l'] && !opt['L']) || !opt['a'] || !opt['c'])) || !opt['o']) {
    fprintf(stderr, "some of required options are missing\n");
    return 0;
  }
  g = opt['g'] ? strdup(opt['g']) : join(opt['s'], ".brs");
  l = opt['l'] ? strdup(opt['l']) : opt['s'] ? join(opt['s'], ".lgt") : NULL;
  a = opt['a'] ? strdup(opt['a']) : join(opt['s'], ".atr");
  c = opt['c'] ? strdup(opt['c']) : join(opt['s'], ".cam");
  C = opt['C'] ? strdup(opt['C']) : opt['s'] ? join(opt['s'], ".cfg") : NULL;
  L = opt['L'] ? strdup(opt['L']) : opt['s'] ? join(opt['s'], ".pnr") : NULL;
  if (opt['d']) rtHostSetDevice(atoi(opt['d']));

  errno = 0;
  RT_Scene *scene = rtSceneLoadAllCached(g, opt['B'], l, L, a, C, gamma, epsilon, distmod);  /* -B file: binary cache of the geometry */
  if (!scene) { fprintf(stderr, "unable to load scene: %s\n", rtGetErrorDesc()); return 1; }
  RT_Camera *cam = rtCameraLoad(c);
  if (!cam || errno > 0) { fprintf(stderr, "unable to load camera: %s\n", rtGetErrorDesc()); return 1; }
  if (opt['W'] && opt['H']) { cam->sw = atoi(opt['W']); cam->sh = atoi(opt['H']); }

  if (devices > 0) { /* one process, `devices` GPUs: librtmulti.so next to this program */
    char exe[PATH_MAX], lib[PATH_MAX + 32];
    ssize_t n = readlink("/proc/self/exe", exe, sizeof exe - 1);
    RT_Bitmap *(*render)(RT_Scene *, RT_Camera *, int, int, double *);
    void *h;
    double secs = 0.0;
    exe[n > 0 ? n : 0] = 0;
    snprintf(lib, sizeof lib, "%s/librtmulti.so", dirname(exe));
    h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
    if (!h) { fprintf(stderr, "--devices: cannot load %s: %s\n", lib, dlerror()); return 1; }
    *(void **)&render = dlsym(h, "rtHostRenderDevices");
    if (!render) { fprintf(stderr, "--devices: %s\n", dlerror()); return 1; }
    errno = 0;
    RT_Bitmap *out = render(scene, cam, devices, frames, &secs);
    if (!out) { fprintf(stderr, "rendering on %d device(s) failed: %d, %s\n", devices, errno, rtGetErrorDesc()); return 1; }
    printf("...ray-tracing on %d device(s) done. Time per frame: %.6f seconds\n", devices, secs);
    errno = 0;
    rtBitmapSave(out, opt['o'], 24);
    if (errno == E_IO || errno == E_MEMORY || errno == E_INVALID_BPP) { fprintf(stderr, "problem while saving result image: %d, %s\n", errno, rtGetErrorDesc()); return 1; }
    rtBitmapDestroy(&out);
    rtCameraDestroy(&cam);
    rtSceneDestroy(&scene);
    return 0;
  }

  struct timespec t0, t1;
  RT_VisualizedScene *vs = NULL;
  for (i = 0; i < frames; i++) { /* --frames K: the time reported is the last call's (the first pays for the GPU context and the queues) */
    if (vs) rtVisualizedSceneDestroy(&vs);
    clock_gettime(CLOCK_MONOTONIC, &t0);
    vs = rtVisualizedSceneRaytrace(scene, cam);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (!vs) break;
  }
  if (errno > 0) { fprintf(stderr, "errno set by ray-trace process: %d, %s\n", errno, rtGetErrorDesc()); errno = 0; }
  if (!vs) return 1;
  printf("...ray-tracing done. Time taken: %.3f seconds\n", (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec));
  if (opt['I']) {
    FILE *f = fopen(opt['I'], "wb");
    size_t p, np = (size_t)vs->width * vs->height;
    for (p = 0; f && p < np; p++) {
      int32_t id = vs->map[p].t ? (int32_t)(vs->map[p].t - scene->t) : -1;
      fwrite(&id, 4, 1, f);
    }
    if (f) fclose(f);
  }
  /* Success is decided on what the calls return: the GPU runtime and stdio below may leave a stray errno on a
   * path that worked, and the host layer's own codes are small positive numbers too. */
  int failed = 0;
  errno = 0;
  RT_Bitmap *bmp = rtVisualizedSceneToBitmap(vs, F_HDR, ngammas >= 0 ? gammas : NULL);
  if (!bmp) {
    fprintf(stderr, "problem while creating result image: %d, %s\n", errno, rtGetErrorDesc());
    failed = 1;
  } else {
    errno = 0;
    rtBitmapSave(bmp, opt['o'], 24);
    if (errno == E_IO || errno == E_MEMORY || errno == E_INVALID_BPP) {
      fprintf(stderr, "problem while saving result image: %d, %s\n", errno, rtGetErrorDesc());
      failed = 1;
    }
  }
  rtBitmapDestroy(&bmp);
  rtVisualizedSceneDestroy(&vs);
  rtCameraDestroy(&cam);
  rtSceneDestroy(&scene);
  rtHostShutdown();
  return failed;
}
