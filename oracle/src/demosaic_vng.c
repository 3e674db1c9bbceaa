/* oracle/src/demosaic_vng.c -- TEST INFRASTRUCTURE: CPU restatement of the VNG4 interpolation of a Bayer mosaic and of the
 * dual demosaic that blends it with RCD / AMaZE by a detail mask.
 *   lin_interpolate()   src/iop/demosaic/basic.c:22-111
 *   vng_interpolate()   src/iop/demosaic/vng.c:34-221     (Bayer: four colours, the two greens apart, mixed at the end)
 *   dual_demosaic()     src/iop/demosaic/dual.c:40-110    (VNG4 of the UN-equilibrated mosaic, two passes of colour
 *                                                          smoothing, the blurred sigmoid of the raw detail mask of the
 *                                                          high-frequency image, out = mask * (high - vng) + vng)
 * Sums run in the reference's operand order; the lookup tables are the reference's, built the same way. */
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

void oracle_color_smoothing(float *out, int width, int height, int passes); /* demosaic_rcd.c */
void oracle_detail_mask_threshold(const float *rm, float *out, int width, int height, float threshold, int detail); /* detailmask.c */

/* the dcraw filter word with the second green of a 2 x 2 cell as colour 3 (vng.c:61-68) */
static uint32_t filters4_of(const uint32_t filters)
{
  return (filters & 3) == 1 ? filters | 0x03030303u : filters | 0x0c0c0c0cu;
}

/* basic.c:22-111; `filters` = the word of the frame (roi origin added to every coordinate, as the reference does) */
static void lin_interpolate(float *out, const float *in, const int width, const int height, const int rx, const int ry,
                            const uint32_t filters)
{
  const int colors = 4;
  /* the frame's outermost ring: mean of the adjoining photosites of each colour, :28-58 */
  for(int row = 0; row < height; row++)
    for(int col = 0; col < width; col++)
    {
      if(col == 1 && row >= 1 && row < height - 1) col = width - 1;
      if(col >= width) break;
      float sum[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
      int count[4] = { 0, 0, 0, 0 };
      for(int y = row - 1; y != row + 2; y++)
        for(int x = col - 1; x != col + 2; x++)
          if(y >= 0 && x >= 0 && y < height && x < width)
          {
            const int f = oracle_fc(y + ry, x + rx, filters);
            sum[f] += in[(size_t)y * width + x];
            count[f]++;
          }
      const int f = oracle_fc(row + ry, col + rx, filters);
      for(int c = 0; c < colors; c++)
        out[4 * ((size_t)row * width + col) + c] = (c != f && count[c] != 0) ? sum[c] / count[c] : in[(size_t)row * width + col];
    }
  /* the interior: weighted 3 x 3 sums per colour, neighbours in row-major order, :72-109 */
  for(int row = 1; row < height - 1; row++)
    for(int col = 1; col < width - 1; col++)
    {
      float sum[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
      int wsum[4] = { 0, 0, 0, 0 };
      /* the table is indexed by row % 16, col % 16 of the BUFFER and built from fcol(row + roi.y, col + roi.x): the same
       * thing, the pattern has periods 8 and 2 */
      const int f = oracle_fc(row + ry, col + rx, filters);
      for(int y = -1; y <= 1; y++)
        for(int x = -1; x <= 1; x++)
        {
          const int weight = 1 << ((y == 0) + (x == 0));
          const int color = oracle_fc(row + y + ry, col + x + rx, filters);
          if(color == f) continue;
          sum[color] += in[(size_t)(row + y) * width + col + x] * weight;
          wsum[color] += weight;
        }
      float *px = out + 4 * ((size_t)row * width + col);
      for(int c = 0; c < colors; c++)
        if(c != f) px[c] = sum[c] / wsum[c];
      px[f] = in[(size_t)row * width + col];
    }
}

static const signed char k_terms[] = {
  -2, -2, +0, -1, 1, 0x01, -2, -2, +0, +0, 2, 0x01, -2, -1, -1, +0, 1, 0x01, -2, -1, +0, -1, 1, 0x02, -2, -1, +0, +0, 1, 0x03,
  -2, -1, +0, +1, 2, 0x01, -2, +0, +0, -1, 1, 0x06, -2, +0, +0, +0, 2, 0x02, -2, +0, +0, +1, 1, 0x03, -2, +1, -1, +0, 1, 0x04,
  -2, +1, +0, -1, 2, 0x04, -2, +1, +0, +0, 1, 0x06, -2, +1, +0, +1, 1, 0x02, -2, +2, +0, +0, 2, 0x04, -2, +2, +0, +1, 1, 0x04,
  -1, -2, -1, +0, 1, 0x80, -1, -2, +0, -1, 1, 0x01, -1, -2, +1, -1, 1, 0x01, -1, -2, +1, +0, 2, 0x01, -1, -1, -1, +1, 1, 0x88,
  -1, -1, +1, -2, 1, 0x40, -1, -1, +1, -1, 1, 0x22, -1, -1, +1, +0, 1, 0x33, -1, -1, +1, +1, 2, 0x11, -1, +0, -1, +2, 1, 0x08,
  -1, +0, +0, -1, 1, 0x44, -1, +0, +0, +1, 1, 0x11, -1, +0, +1, -2, 2, 0x40, -1, +0, +1, -1, 1, 0x66, -1, +0, +1, +0, 2, 0x22,
  -1, +0, +1, +1, 1, 0x33, -1, +0, +1, +2, 2, 0x10, -1, +1, +1, -1, 2, 0x44, -1, +1, +1, +0, 1, 0x66, -1, +1, +1, +1, 1, 0x22,
  -1, +1, +1, +2, 1, 0x10, -1, +2, +0, +1, 1, 0x04, -1, +2, +1, +0, 2, 0x04, -1, +2, +1, +1, 1, 0x04, +0, -2, +0, +0, 2, 0x80,
  +0, -1, +0, +1, 2, 0x88, +0, -1, +1, -2, 1, 0x40, +0, -1, +1, +0, 1, 0x11, +0, -1, +2, -2, 1, 0x40, +0, -1, +2, -1, 1, 0x20,
  +0, -1, +2, +0, 1, 0x30, +0, -1, +2, +1, 2, 0x10, +0, +0, +0, +2, 2, 0x08, +0, +0, +2, -2, 2, 0x40, +0, +0, +2, -1, 1, 0x60,
  +0, +0, +2, +0, 2, 0x20, +0, +0, +2, +1, 1, 0x30, +0, +0, +2, +2, 2, 0x10, +0, +1, +1, +0, 1, 0x44, +0, +1, +1, +2, 1, 0x10,
  +0, +1, +2, -1, 2, 0x40, +0, +1, +2, +0, 1, 0x60, +0, +1, +2, +1, 1, 0x20, +0, +1, +2, +2, 1, 0x10, +1, -2, +1, +0, 1, 0x80,
  +1, -1, +1, +1, 1, 0x88, +1, +0, +1, +2, 1, 0x08, +1, +0, +2, -1, 1, 0x40, +1, +0, +2, +1, 1, 0x10
};
static const signed char k_chood[] = { -1, -1, -1, 0, -1, +1, 0, +1, +1, +1, +1, 0, +1, -1, 0, -1 };

/* vng.c:34-221 for a Bayer mosaic.  The ring buffer of three rows in the reference only delays the stores, so that every
 * gradient and average reads the LINEAR interpolation: here the interpolation is kept and the result goes to `out` */
int oracle_demosaic_vng4(float *out, const float *in, const int width, const int height, const int rx, const int ry,
                         const uint32_t filters)
{
  const uint32_t filters4 = filters4_of(filters);
  float *lin = (float *)malloc(sizeof(float) * 4 * (size_t)width * height);
  if(!lin) return 1;
  lin_interpolate(lin, in, width, height, rx, ry, filters4);
  memcpy(out, lin, sizeof(float) * 4 * (size_t)width * height);
  const int prow = 8, pcol = 2, colors = 4;
  int *codes = (int *)malloc(sizeof(int) * prow * pcol * 320), *ip = codes, *code[8][2];
  for(int row = 0; row < prow; row++)
    for(int col = 0; col < pcol; col++)
    {
      code[row][col] = ip;
      const signed char *cp = k_terms;
      for(int t = 0; t < 64; t++)
      {
        const int y1 = *cp++, x1 = *cp++, y2 = *cp++, x2 = *cp++, weight = *cp++, grads = *cp++;
        const int color = oracle_fc(row + y1, col + x1, filters4);
        if(oracle_fc(row + y2, col + x2, filters4) != color) continue;
        const int diag = (oracle_fc(row, col + 1, filters4) == color && oracle_fc(row + 1, col, filters4) == color) ? 2 : 1;
        if(abs(y1 - y2) == diag && abs(x1 - x2) == diag) continue;
        *ip++ = (y1 * width + x1) * 4 + color;
        *ip++ = (y2 * width + x2) * 4 + color;
        *ip++ = weight;
        for(int g = 0; g < 8; g++)
          if(grads & 1 << g) *ip++ = g;
        *ip++ = -1;
      }
      *ip++ = INT_MAX;
      cp = k_chood;
      for(int g = 0; g < 8; g++)
      {
        const int y = *cp++, x = *cp++;
        *ip++ = (y * width + x) * 4;
        const int color = oracle_fc(row, col, filters4);
        if(oracle_fc(row + y, col + x, filters4) != color && oracle_fc(row + y * 2, col + x * 2, filters4) == color)
          *ip++ = (y * width + x) * 8 + color;
        else
          *ip++ = 0;
      }
    }
#pragma omp parallel for schedule(static) private(ip)
  for(int row = 2; row < height - 2; row++)
    for(int col = 2; col < width - 2; col++)
    {
      int g;
      float gval[8] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
      const float *pix = lin + 4 * ((size_t)row * width + col);
      ip = code[(row + ry) % prow][(col + rx) % pcol];
      while((g = ip[0]) != INT_MAX) /* the gradients, vng.c:135-143 */
      {
        const float diff = fabsf(pix[g] - pix[ip[1]]) * ip[2];
        gval[ip[3]] += diff;
        ip += 5;
        if((g = ip[-1]) == -1) continue;
        gval[g] += diff;
        while((g = *ip++) != -1) gval[g] += diff;
      }
      ip++;
      float gmin = gval[0], gmax = gval[0];
      for(g = 1; g < 8; g++)
      {
        if(gmin > gval[g]) gmin = gval[g];
        if(gmax < gval[g]) gmax = gval[g];
      }
      float *o = out + 4 * ((size_t)row * width + col);
      if(gmax == 0) continue; /* the pixel keeps its linear interpolation */
      const float thold = gmin + (gmax * 0.5f);
      float sum[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
      const int color = oracle_fc(row + ry, col + rx, filters4);
      int num = 0;
      for(g = 0; g < 8; g++, ip += 2) /* the neighbours below the threshold, :160-170 */
        if(gval[g] <= thold)
        {
          for(int c = 0; c < colors; c++)
            if(c == color && ip[1]) sum[c] += (pix[c] + pix[ip[1]]) * 0.5f;
            else sum[c] += pix[ip[0] + c];
          num++;
        }
      for(int c = 0; c < colors; c++)
      {
        float tot = pix[color];
        if(c != color) tot += (sum[c] - sum[color]) / num;
        o[c] = tot;
      }
    }
  /* VNG4: the two greens become one, :205-208 */
  for(size_t i = 0; i < (size_t)width * height; i++) out[i * 4 + 1] = (out[i * 4 + 1] + out[i * 4 + 3]) / 2.0f;
  free(codes);
  free(lin);
  return 0;
}

/* dual.c:40-110; rgb: the high-frequency demosaic (RCD / AMaZE) of the frame going in, the blend coming out;
 * raw: the mosaic as the module received it */
int oracle_dual_demosaic(float *rgb, const float *raw, const int width, const int height, const int rx, const int ry,
                         const uint32_t filters, const float dual_threshold, const float wb[4])
{
  if(width < 16 || height < 16) return 0;
  if(dual_threshold <= 0.0f) return 0;
  const size_t np = (size_t)width * height;
  float *blend = (float *)malloc(sizeof(float) * np), *vng = (float *)malloc(sizeof(float) * 4 * np);
  if(!blend || !vng) return 1;
  if(oracle_demosaic_vng4(vng, raw, width, height, rx, ry, filters)) return 1;
  oracle_color_smoothing(vng, width, height, 2);
  const float contrastf = 0.005f * powf(dual_threshold, 1.1f); /* slider2contrast(), dual.c:35-38 */
  oracle_rawdetail_mask(rgb, blend, width, height, wb);
  oracle_detail_mask_threshold(blend, blend, width, height, contrastf, 1);
  for(size_t idx = 0; idx < np; idx++)
    for(int c = 0; c < 4; c++)
    {
      const float a = blend[idx], b = rgb[4 * idx + c], cc = vng[4 * idx + c];
      rgb[4 * idx + c] = a * (b - cc) + cc; /* intp(), demosaic.c:250-257 */
    }
  free(blend);
  free(vng);
  return 0;
}
