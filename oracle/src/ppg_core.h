/* oracle/src/ppg_core.h -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Per-pixel form of Ansel's PPG interpolation.  The reference has it twice:
 *   demosaic_ppg()     src/iop/demosaic/ppg.c:20-217   whole image, raw samples as they come
 *   rcd_ppg_border()   src/iop/demosaic/rcd.c:92-272   only the outer RCD_MARGIN ring, every sample
 *                                                       read through fmaxf(0, .)
 * Both run three in-place passes over the output: (1) the outer 3-px ring gets per-colour
 * averages of the in-bounds 3x3 neighbours, (2) red/blue sites with ring >= 3 get a green by a
 * gradient-selected, clamped guess, (3) every pixel with ring >= 1 gets its remaining colours
 * from colour differences of neighbours.  Pass 3 only ever reads a neighbour's native sample
 * and its green, both final after passes 1-2, and only writes the other channels -- so each
 * output pixel is a pure function of the input, which is what this header evaluates.  Channels
 * that pass 2 leaves uninitialised in the reference (`dt_aligned_pixel_t color;`, ppg.c:78 /
 * rcd.c:150) are always overwritten by pass 3 and never read before.
 */
#ifndef ORACLE_PPG_CORE_H
#define ORACLE_PPG_CORE_H
#include <math.h>
#include "oracle.h"

typedef struct
{
  const float *in;
  int iw, ih; /* input buffer geometry */
  int ow, oh; /* output geometry */
  int ox, oy; /* position of the output window inside the input */
  uint32_t filters;
  int clamp;  /* 1: samples go through fmaxf(0, .)  (rcd_ppg_border) */
  const float *in1; /* what pass 1 reads: the mosaic BEFORE the median pre-filter (ppg.c:30-57 run on `in`, the later
                       passes on `input`, :60-67); NULL = in */
} ppg_ctx_t;

static inline float ppg_s(const ppg_ctx_t *k, const int j, const int i)
{
  const float v = k->in[(size_t)(j + k->oy) * k->iw + i + k->ox];
  return k->clamp ? fmaxf(0.0f, v) : v;
}

static inline int ppg_ring_lt(const ppg_ctx_t *k, const int j, const int i, const int r)
{
  return j < r || i < r || j >= k->oh - r || i >= k->ow - r;
}

/* pass 1: ppg.c:30-57 / rcd.c:96-127 */
static inline void ppg_pass1(const ppg_ctx_t *k, const int j, const int i, float rgb[3])
{
  float sum[8] = { 0.f };
  for(int y = j - 1; y != j + 2; y++)
    for(int x = i - 1; x != i + 2; x++)
    {
      const int yy = y + k->oy, xx = x + k->ox;
      if(yy >= 0 && xx >= 0 && yy < k->ih && xx < k->iw)
      {
        const int f = oracle_fc(y, x, k->filters);
        {
          const float v1 = (k->in1 ? k->in1 : k->in)[(size_t)(y + k->oy) * k->iw + x + k->ox];
          sum[f] += k->clamp ? fmaxf(0.0f, v1) : v1;
        }
        sum[f + 4]++;
      }
    }
  const int f = oracle_fc(j, i, k->filters);
  const float own1 = (k->in1 ? k->in1 : k->in)[(size_t)(j + k->oy) * k->iw + i + k->ox];
  const float own = k->clamp ? fmaxf(0.0f, own1) : own1;
  for(int c = 0; c < 3; c++) rgb[c] = (c != f && sum[c + 4] > 0.0f) ? sum[c] / sum[c + 4] : own;
}

/* pass 2 green at a red/blue site: ppg.c:83-115 / rcd.c:146-187 */
static inline float ppg_pass2_green(const ppg_ctx_t *k, const int j, const int i)
{
  const float pc = ppg_s(k, j, i);
  const float pym = ppg_s(k, j - 1, i), pym2 = ppg_s(k, j - 2, i), pym3 = ppg_s(k, j - 3, i);
  const float pyM = ppg_s(k, j + 1, i), pyM2 = ppg_s(k, j + 2, i), pyM3 = ppg_s(k, j + 3, i);
  const float pxm = ppg_s(k, j, i - 1), pxm2 = ppg_s(k, j, i - 2), pxm3 = ppg_s(k, j, i - 3);
  const float pxM = ppg_s(k, j, i + 1), pxM2 = ppg_s(k, j, i + 2), pxM3 = ppg_s(k, j, i + 3);
  const float guessx = (pxm + pc + pxM) * 2.0f - pxM2 - pxm2;
  const float diffx = (fabsf(pxm2 - pc) + fabsf(pxM2 - pc) + fabsf(pxm - pxM)) * 3.0f
                      + (fabsf(pxM3 - pxM) + fabsf(pxm3 - pxm)) * 2.0f;
  const float guessy = (pym + pc + pyM) * 2.0f - pyM2 - pym2;
  const float diffy = (fabsf(pym2 - pc) + fabsf(pyM2 - pc) + fabsf(pym - pyM)) * 3.0f
                      + (fabsf(pyM3 - pyM) + fabsf(pym3 - pym)) * 2.0f;
  if(diffx > diffy) return fmaxf(fminf(guessy * .25f, fmaxf(pym, pyM)), fminf(pym, pyM));
  return fmaxf(fminf(guessx * .25f, fmaxf(pxm, pxM)), fminf(pxm, pxM));
}

/* channel c (native colour or green only) of pixel (j,i) after passes 1-2 */
static inline float ppg_pre3(const ppg_ctx_t *k, const int j, const int i, const int c)
{
  if(ppg_ring_lt(k, j, i, 3))
  {
    float rgb[3];
    ppg_pass1(k, j, i, rgb);
    return rgb[c];
  }
  const int f = oracle_fc(j, i, k->filters);
  if(c == 1 && !(f & 1)) return ppg_pass2_green(k, j, i);
  return ppg_s(k, j, i);
}

/* the finished pixel; rgb[] only, alpha is the caller's business */
static inline void ppg_pixel(const ppg_ctx_t *k, const int j, const int i, float color[3])
{
  const int c = oracle_fc(j, i, k->filters);
  if(ppg_ring_lt(k, j, i, 3))
    ppg_pass1(k, j, i, color);
  else
  {
    color[0] = color[2] = 0.f;
    if(!(c & 1)) color[c] = ppg_s(k, j, i);
    color[1] = ppg_pre3(k, j, i, 1);
  }
  if(ppg_ring_lt(k, j, i, 1)) return;
  /* pass 3: ppg.c:130-205 / rcd.c:191-269 */
#define PRE(dj, di, ch) ppg_pre3(k, j + (dj), i + (di), ch)
  if(c & 1)
  {
    const int h = (oracle_fc(j, i + 1, k->filters) == 0) ? 0 : 2; /* colour of the row neighbours */
    const int v = 2 - h;
    color[v] = (PRE(-1, 0, v) + PRE(1, 0, v) + 2.0f * color[1] - PRE(-1, 0, 1) - PRE(1, 0, 1)) * .5f;
    color[h] = (PRE(0, -1, h) + PRE(0, 1, h) + 2.0f * color[1] - PRE(0, -1, 1) - PRE(0, 1, 1)) * .5f;
  }
  else
  {
    const int o = 2 - c;
    const float tl = PRE(-1, -1, o), br = PRE(1, 1, o), tr = PRE(-1, 1, o), bl = PRE(1, -1, o);
    const float tlg = PRE(-1, -1, 1), brg = PRE(1, 1, 1), trg = PRE(-1, 1, 1), blg = PRE(1, -1, 1);
    const float diff1 = fabsf(tl - br) + fabsf(tlg - color[1]) + fabsf(brg - color[1]);
    const float guess1 = tl + br + 2.0f * color[1] - tlg - brg;
    const float diff2 = fabsf(tr - bl) + fabsf(trg - color[1]) + fabsf(blg - color[1]);
    const float guess2 = tr + bl + 2.0f * color[1] - trg - blg;
    color[o] = (diff1 > diff2) ? guess2 * .5f : ((diff1 < diff2) ? guess1 * .5f : (guess1 + guess2) * .25f);
  }
#undef PRE
}
#endif
