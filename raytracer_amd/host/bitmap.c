/* RT_Bitmap and the BMP file reader/writer of the host layer (interface of src/bitmap.h:102-119).
 *
 * The renderer only ever writes 24-bpp files (src/main.c:202) and only ever asks whether
 * "textures/brickwall.bmp" can be loaded, so the writer covers 24 and 32 bpp and the reader decodes
 * uncompressed 24/32-bpp images; other depths are accepted with a blank pixel array (their pixels are
 * never sampled by the trace path).  The byte layout of what is written matches src/bitmap.c:274-304
 * and :453-469: 54-byte header written field by field, rows bottom-up, B G R, rows padded to 4 bytes.
 */
#include "rt_host.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

RT_Bitmap *rtBitmapCreate(int32_t width, int32_t height, uint32_t background) {
  RT_Bitmap *b = malloc(sizeof *b);
  size_t n = (size_t)width * (size_t)height, i;
  if (!b) return NULL;
  b->width = width; b->height = height; b->background = background;
  b->pixels = malloc((n ? n : 1) * sizeof(uint32_t));
  if (!b->pixels) { free(b); return NULL; }
  for (i = 0; i < n; i++) b->pixels[i] = background;
  return b;
}

void rtBitmapDestroy(RT_Bitmap **self) {
  if (*self) { free((*self)->pixels); free(*self); *self = NULL; }
}

static uint32_t rd32(const unsigned char *p) { return p[0] | p[1] << 8 | p[2] << 16 | (uint32_t)p[3] << 24; }
static void wr32(unsigned char *p, uint32_t v) { p[0] = v; p[1] = v >> 8; p[2] = v >> 16; p[3] = v >> 24; }

RT_Bitmap *rtBitmapLoad(const char *filename) {
  unsigned char h[54];
  FILE *f = fopen(filename, "rb");
  RT_Bitmap *b = NULL;
  int32_t w, ht, bpp, y, x;
  if (!f) { errno = E_IO; return NULL; }
  if (fread(h, 1, 54, f) != 54 || h[0] != 'B' || h[1] != 'M') { errno = E_INVALID_FILE; goto out; }
  if (rd32(h + 14) == 40 && rd32(h + 30) != 0) { errno = E_INVALID_FILE_FORMAT; goto out; } /* compressed */
  w = (int32_t)rd32(h + 18); ht = (int32_t)rd32(h + 22); bpp = h[28] | h[29] << 8;
  if (w <= 0 || ht == 0) { errno = E_INVALID_FILE; goto out; }
  b = rtBitmapCreate(w, ht > 0 ? ht : -ht, 0);
  if (!b) { errno = E_MEMORY; goto out; }
  if (bpp == 24 || bpp == 32) {
    size_t row = (size_t)(4 * ceil(bpp * w / 32.0));
    unsigned char *line = malloc(row);
    if (!line) { rtBitmapDestroy(&b); errno = E_MEMORY; goto out; }
    fseek(f, (long)rd32(h + 10), SEEK_SET);
    for (y = 0; y < b->height; y++) {
      int32_t dst = ht > 0 ? b->height - 1 - y : y;
      if (fread(line, 1, row, f) != row) break;
      for (x = 0; x < w; x++) {
        const unsigned char *p = line + (size_t)x * (bpp / 8);
        b->pixels[(size_t)dst * w + x] = rtColorBuildRGBA(p[2], p[1], p[0], 0);
      }
    }
    free(line);
  }
out:
  fclose(f);
  return b;
}

void rtBitmapSave(const RT_Bitmap *self, const char *filename, uint16_t bpp) {
  unsigned char h[54];
  uint32_t row, total;
  unsigned char *line;
  FILE *f;
  int32_t x, y;
  if (bpp != 24 && bpp != 32) { errno = E_INVALID_BPP; return; }
  f = fopen(filename, "wb");
  if (!f) { errno = E_IO; return; }
  row = (uint32_t)(4 * ceil(bpp * self->width / 32.0));
  total = (uint32_t)(54 + 4 * ceil(bpp * self->width / 32.0) * self->height);
  memset(h, 0, sizeof h);
  h[0] = 'B'; h[1] = 'M';
  wr32(h + 2, total); wr32(h + 10, 54); wr32(h + 14, 40);
  wr32(h + 18, (uint32_t)self->width); wr32(h + 22, (uint32_t)self->height);
  h[26] = 1; h[28] = (unsigned char)bpp;
  wr32(h + 34, total - 54);
  fwrite(h, 1, 54, f);
  line = malloc(row ? row : 1);
  if (!line) { errno = E_MEMORY; fclose(f); return; }
  for (y = self->height - 1; y >= 0; y--) {
    memset(line, 0, row);
    for (x = 0; x < self->width; x++) {
      uint32_t p = self->pixels[(size_t)y * self->width + x];
      unsigned char *o = line + (size_t)x * (bpp / 8);
      o[0] = (p >> 8) & 0xff; o[1] = (p >> 16) & 0xff; o[2] = (p >> 24) & 0xff;
    }
    fwrite(line, 1, row, f);
  }
  free(line);
  fclose(f);
}
