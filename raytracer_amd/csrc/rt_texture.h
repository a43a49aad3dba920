// Procedural brick texture of the reference, evaluated on the GPU inside the hit kernel.
// texture.c:13-155 (Perlin "improved noise" in binary64, brick pattern) and raytrace.c:80-128
// (five evaluations per hit: colour at the hit plus +-0.002 in u and v for the bump gradient).
#pragma once
#include "rt_math.h"

namespace rt {

// Perlin's reference permutation (texture.c:13-25; the reference stores it twice back to back).
__device__ const unsigned char kPerm[256] = {
    151, 160, 137, 91,  90,  15,  131, 13,  201, 95,  96,  53,  194, 233, 7,   225, 140, 36,  103, 30,  69,  142,
    8,   99,  37,  240, 21,  10,  23,  190, 6,   148, 247, 120, 234, 75,  0,   26,  197, 62,  94,  252, 219, 203,
    117, 35,  11,  32,  57,  177, 33,  88,  237, 149, 56,  87,  174, 20,  125, 136, 171, 168, 68,  175, 74,  165,
    71,  134, 139, 48,  27,  166, 77,  146, 158, 231, 83,  111, 229, 122, 60,  211, 133, 230, 220, 105, 92,  41,
    55,  46,  245, 40,  244, 102, 143, 54,  65,  25,  63,  161, 1,   216, 80,  73,  209, 76,  132, 187, 208, 89,
    18,  169, 200, 196, 135, 130, 116, 188, 159, 86,  164, 100, 109, 198, 173, 186, 3,   64,  52,  217, 226, 250,
    124, 123, 5,   202, 38,  147, 118, 126, 255, 82,  85,  212, 207, 206, 59,  227, 47,  16,  58,  17,  182, 189,
    28,  42,  23,  183, 170, 213, 119, 248, 152, 2,   44,  154, 163, 70,  221, 153, 101, 155, 167, 43,  172, 9,
    129, 22,  39,  253, 19,  98,  108, 110, 79,  113, 224, 232, 178, 185, 112, 104, 218, 246, 97,  228, 251, 34,
    242, 193, 238, 210, 144, 12,  191, 179, 162, 241, 81,  51,  145, 235, 249, 14,  239, 107, 49,  192, 214, 31,
    181, 199, 106, 157, 184, 84,  204, 176, 115, 121, 50,  45,  127, 4,   150, 254, 138, 236, 205, 93,  222, 114,
    67,  29,  24,  72,  243, 141, 128, 195, 78,  66,  215, 61,  156, 180};

RT_D int perm(int i) { return kPerm[i & 255]; }
RT_D double fade(double t) { return t * t * t * (t * (t * 6 - 15) + 10); }  // texture.c:41-44
RT_D double lerpd(double t, double a, double b) { return a + t * (b - a); }  // texture.c:46-49
RT_D double grad(int hash, double x, double y, double z) {                   // texture.c:51-58
  int h = hash & 15;
  double u = (h < 8 || h == 12 || h == 13) ? x : y;
  double v = (h < 4 || h == 12 || h == 13) ? y : z;
  return ((h & 1) == 0 ? u : -u) + ((h & 2) == 0 ? v : -v);
}
// texture.c:60-87
__device__ __noinline__ double noise3(double x, double y, double z) {
  double fx = floor(x), fy = floor(y), fz = floor(z);
  int X = (int)fx & 255, Y = (int)fy & 255, Z = (int)fz & 255;
  x -= fx; y -= fy; z -= fz;
  double u = fade(x), v = fade(y), w = fade(z);
  int A = perm(X) + Y, AA = perm(A) + Z, AB = perm(A + 1) + Z;
  int B = perm(X + 1) + Y, BA = perm(B) + Z, BB = perm(B + 1) + Z;
  return lerpd(w,
               lerpd(v, lerpd(u, grad(perm(AA), x, y, z), grad(perm(BA), x - 1, y, z)),
                     lerpd(u, grad(perm(AB), x, y - 1, z), grad(perm(BB), x - 1, y - 1, z))),
               lerpd(v, lerpd(u, grad(perm(AA + 1), x, y, z - 1), grad(perm(BA + 1), x - 1, y, z - 1)),
                     lerpd(u, grad(perm(AB + 1), x, y - 1, z - 1), grad(perm(BB + 1), x - 1, y - 1, z - 1))));
}

// texture.c:89-155 with the literal arguments of raytrace.c:84-91.  The cos() terms of the original
// only feed an array nobody reads, so they are not evaluated.
__device__ __noinline__ f3 bricks(float x, float y) {
  const float bheight = 0.04f, bwidth = 0.10f, filling = 0.005f, brickpos = 33.0f;
  const float rfactor = 2160.0f, gfactor = 0.0f, bfactor = 0.0f, basef = 0.7f, derf = 0.4f;
  float w = 2 * filling + bwidth, h = 2 * filling + bheight;
  f3 brick = mk3(173 / 255.0f, 106 / 255.0f, 64 / 255.0f), fill = mk3(215 / 255.0f, 205 / 255.0f, 178 / 255.0f);
  double ay = y / h;
  int row = (int)floor(ay);
  double ax = (x / w) + ((row & 1) ? 0.5 : 0.0);
  int col = (int)floor(ax);
  ax = ax - col;
  ay = ay - row;
  float pm0 = (float)(0.2f * noise3(brickpos * row, brickpos * col, 0.435));
  float pm1 = (float)(0.2f * noise3(brickpos * row, brickpos * col, 0.645));
  float pm2 = (float)(0.2f * noise3(brickpos * row, brickpos * col, 0.354));
  float pm3 = (float)(0.2f * noise3(brickpos * row, brickpos * col, 0.768));
  float bl = filling / w + pm0 * filling / w;
  float br = (w - filling) / w + pm1 * (w - filling) / w;
  float bt = filling / h + pm2 * filling / h;
  float bb = (h - filling) / h + pm3 * (h - filling) / h;
  f3 c;
  if (ax < bl || ax > br || ay < bt || ay > bb) {
    c = fill;
  } else {
    float nz = (float)noise3(row * x, col * y, row * col);
    c = mk3(brick.x + basef * nz, brick.y + basef * nz, brick.z + basef * nz);
  }
  c.x += derf * (float)noise3(rfactor * x, rfactor * y, row * col);
  c.y += derf * (float)noise3(gfactor * x, gfactor * y, row * col);
  c.z += derf * (float)noise3(bfactor * x, bfactor * y, row * col);
  return c;
}

// Fixed texture coordinates of the four textured triangles (preprocess.c:45-71): ti, tj, tk.
__device__ const float kTexUV[4][6] = {{0.0f, 1.0f, 1.0f, 0.0f, 0.0f, 0.0f},
                                       {0.0f, 1.0f, 1.0f, 1.0f, 1.0f, 0.0f},
                                       {0.0f, 1.0f, 0.8f, 0.0f, 0.0f, 0.0f},
                                       {0.0f, 1.0f, 0.8f, 1.0f, 0.8f, 0.0f}};

// raytrace.c:80-128.  Where the reference reads an uninitialised neighbour sample (px or py within
// 0.002 of the [0,1] border, raytrace.c:94-105) the sample is taken as black here; those pixels are
// not comparable with the reference, which returns whatever its stack held.
RT_D void apply_texture(int slot, f3 ij, f3 ik, float u, float v, f3 &norm, f3 &nc) {
  const float *t = kTexUV[slot];
  float px = t[0] + (t[2] - t[0]) * u + (t[4] - t[0]) * v;
  float py = t[1] + (t[3] - t[1]) * u + (t[5] - t[1]) * v;
  const float delta = 0.002f;
  f3 zero = mk3(0.0f, 0.0f, 0.0f), cx1 = zero, cx2 = zero, cy1 = zero, cy2 = zero;
  f3 avg = bricks(px, py);
  float tmp;
  if ((tmp = px - delta) >= 0.0f) cx1 = bricks(tmp, py);
  if ((tmp = px + delta) <= 1.0f) cx2 = bricks(tmp, py);
  if ((tmp = py - delta) >= 0.0f) cy1 = bricks(px, tmp);
  if ((tmp = py + delta) <= 1.0f) cy2 = bricks(px, tmp);
  nc = avg;
  float ugrad = (cx2.x + cx2.y + cx2.z) * 0.333f - (cx1.x + cx1.y + cx1.z) * 0.333f;
  float vgrad = (cy2.x + cy2.y + cy2.z) * 0.333f - (cy1.x + cy1.y + cy1.z) * 0.333f;
  norm = add3(norm, mul3(ij, ugrad));
  norm = add3(norm, mul3(ik, -vgrad));
  norm = normalize3(norm);
}

}  // namespace rt
