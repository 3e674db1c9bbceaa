// ppg_device.h -- per-pixel PPG interpolation, shared by the RCD border ring and the PPG demosaic.
//
// Reference: demosaic_ppg(), src/iop/demosaic/ppg.c:20-217 (raw samples) and rcd_ppg_border(),
// src/iop/demosaic/rcd.c:92-272 (samples clamped with fmaxf(0, .), outer RCD_MARGIN ring only).
// Both are three in-place passes on the CPU; pass 3 only reads a neighbour's native sample and
// its green, which are final after passes 1-2, so every output pixel is a pure function of the
// mosaic.  That is the form used here: one thread = one finished float4, no inter-pass buffer,
// no second launch.  The 4..8 neighbour greens a pixel needs are recomputed (13 taps each, all
// L1/L2 hits) instead of being exchanged through HBM.
#pragma once
#include "hip_common.h"

namespace ansel
{

__device__ __forceinline__ int ppg_fc(const int row, const int col, const uint32_t filters)
{
  return filters >> ((((row << 1) & 14) + (col & 1)) << 1) & 3;
}

struct ppg_ctx
{
  const float *in;
  int iw, ih; // input geometry
  int w, h;   // output geometry
  int ox, oy; // output window origin inside the input
  uint32_t filters;
  const float *in1; // what pass 1 reads: the mosaic before PPG's median pre-filter (ppg.c:30-57 vs :60-67); nullptr = in
};

template <bool CLAMP> __device__ __forceinline__ float bs1(const ppg_ctx &k, const int j, const int i)
{
  const float v = (k.in1 ? k.in1 : k.in)[(size_t)(j + k.oy) * k.iw + i + k.ox];
  return CLAMP ? fmaxf(0.0f, v) : v;
}

template <bool CLAMP> __device__ __forceinline__ float bs(const ppg_ctx &k, const int j, const int i)
{
  const float v = k.in[(size_t)(j + k.oy) * k.iw + i + k.ox];
  return CLAMP ? fmaxf(0.0f, v) : v;
}
__device__ __forceinline__ bool ring_lt(const ppg_ctx &k, const int j, const int i, const int r)
{
  return j < r || i < r || j >= k.h - r || i >= k.w - r;
}

// pass 1: ppg.c:30-57 / rcd.c:96-127 -- per-colour average of the in-bounds 3x3 neighbours
template <bool CLAMP> __device__ void ppg_pass1(const ppg_ctx &k, const int j, const int i, float rgb[3])
{
  float sum[4] = { 0.f, 0.f, 0.f, 0.f }, cnt[4] = { 0.f, 0.f, 0.f, 0.f };
  for(int dy = -1; dy <= 1; dy++)
    for(int dx = -1; dx <= 1; dx++)
    {
      const int y = j + dy, x = i + dx;
      const int yy = y + k.oy, xx = x + k.ox;
      if(yy >= 0 && xx >= 0 && yy < k.ih && xx < k.iw)
      {
        const int f = ppg_fc(y, x, k.filters);
        const float v = bs1<CLAMP>(k, y, x);
#pragma unroll
        for(int c = 0; c < 4; c++)
          if(c == f)
          {
            sum[c] += v;
            cnt[c] += 1.0f;
          }
      }
    }
  const int f = ppg_fc(j, i, k.filters);
  const float self = bs1<CLAMP>(k, j, i);
#pragma unroll
  for(int c = 0; c < 3; c++) rgb[c] = (c != f && cnt[c] > 0.0f) ? sum[c] / cnt[c] : self;
}

// pass 2 green at a red/blue site: ppg.c:83-115 / rcd.c:146-187
template <bool CLAMP> __device__ float ppg_pass2_green(const ppg_ctx &k, const int j, const int i)
{
  // per axis: the second-difference estimate of green from the three samples either side of the site, and how busy the axis is;
  // the quieter axis' estimate (a quarter of it), held between its two nearest greens
  const float centre = bs<CLAMP>(k, j, i);
  struct axis_t
  {
    float estimate, activity, lo, hi;
  };
  auto along = [&](const int dj, const int di) {
    const float b1 = bs<CLAMP>(k, j - dj, i - di), b2 = bs<CLAMP>(k, j - 2 * dj, i - 2 * di), b3 = bs<CLAMP>(k, j - 3 * dj, i - 3 * di);
    const float a1 = bs<CLAMP>(k, j + dj, i + di), a2 = bs<CLAMP>(k, j + 2 * dj, i + 2 * di), a3 = bs<CLAMP>(k, j + 3 * dj, i + 3 * di);
    axis_t r;
    r.estimate = (b1 + centre + a1) * 2.0f - a2 - b2;
    r.activity = (fabsf(b2 - centre) + fabsf(a2 - centre) + fabsf(b1 - a1)) * 3.0f + (fabsf(a3 - a1) + fabsf(b3 - b1)) * 2.0f;
    r.lo = fminf(b1, a1);
    r.hi = fmaxf(b1, a1);
    return r;
  };
  const axis_t rows = along(1, 0), cols = along(0, 1);
  const axis_t &quiet = cols.activity > rows.activity ? rows : cols;
  return fmaxf(fminf(quiet.estimate * .25f, quiet.hi), quiet.lo);
}

// channel c (native colour or green) of pixel (j,i) after passes 1-2
template <bool CLAMP> __device__ float ppg_pre3(const ppg_ctx &k, const int j, const int i, const int c)
{
  if(ring_lt(k, j, i, 3))
  {
    float rgb[3];
    ppg_pass1<CLAMP>(k, j, i, rgb);
    return c == 0 ? rgb[0] : (c == 1 ? rgb[1] : rgb[2]);
  }
  const int f = ppg_fc(j, i, k.filters);
  if(c == 1 && !(f & 1)) return ppg_pass2_green<CLAMP>(k, j, i);
  return bs<CLAMP>(k, j, i);
}

// the finished pixel (pass 3: ppg.c:130-205 / rcd.c:191-269)
template <bool CLAMP> __device__ float4 ppg_pixel(const ppg_ctx &k, const int j, const int i)
{
  const int c = ppg_fc(j, i, k.filters);
  float color[3];
  if(ring_lt(k, j, i, 3))
    ppg_pass1<CLAMP>(k, j, i, color);
  else
  {
    color[0] = color[2] = 0.0f;
    const float self = bs<CLAMP>(k, j, i);
    if(c == 0) color[0] = self;
    if(c == 2) color[2] = self;
    color[1] = ppg_pre3<CLAMP>(k, j, i, 1);
  }
  if(!ring_lt(k, j, i, 1))
  {
    if(c & 1)
    {
      const int hcol = (ppg_fc(j, i + 1, k.filters) == 0) ? 0 : 2;
      const int vcol = 2 - hcol;
      const float vv = (ppg_pre3<CLAMP>(k, j - 1, i, vcol) + ppg_pre3<CLAMP>(k, j + 1, i, vcol) + 2.0f * color[1]
                        - ppg_pre3<CLAMP>(k, j - 1, i, 1) - ppg_pre3<CLAMP>(k, j + 1, i, 1)) * .5f;
      const float hv = (ppg_pre3<CLAMP>(k, j, i - 1, hcol) + ppg_pre3<CLAMP>(k, j, i + 1, hcol) + 2.0f * color[1]
                        - ppg_pre3<CLAMP>(k, j, i - 1, 1) - ppg_pre3<CLAMP>(k, j, i + 1, 1)) * .5f;
      color[0] = (hcol == 0) ? hv : vv;
      color[2] = (hcol == 0) ? vv : hv;
    }
    else
    {
      const int o = 2 - c;
      const float tl = ppg_pre3<CLAMP>(k, j - 1, i - 1, o), br = ppg_pre3<CLAMP>(k, j + 1, i + 1, o);
      const float tr = ppg_pre3<CLAMP>(k, j - 1, i + 1, o), bl = ppg_pre3<CLAMP>(k, j + 1, i - 1, o);
      const float tlg = ppg_pre3<CLAMP>(k, j - 1, i - 1, 1), brg = ppg_pre3<CLAMP>(k, j + 1, i + 1, 1);
      const float trg = ppg_pre3<CLAMP>(k, j - 1, i + 1, 1), blg = ppg_pre3<CLAMP>(k, j + 1, i - 1, 1);
      const float diff1 = fabsf(tl - br) + fabsf(tlg - color[1]) + fabsf(brg - color[1]);
      const float guess1 = tl + br + 2.0f * color[1] - tlg - brg;
      const float diff2 = fabsf(tr - bl) + fabsf(trg - color[1]) + fabsf(blg - color[1]);
      const float guess2 = tr + bl + 2.0f * color[1] - trg - blg;
      const float v = (diff1 > diff2) ? guess2 * .5f : ((diff1 < diff2) ? guess1 * .5f : (guess1 + guess2) * .25f);
      if(o == 0) color[0] = v; else color[2] = v;
    }
  }
  // alpha: the reference writes 0 from ring 3 inwards and leaves the caller's buffer alone on the
  // outer 3 px (the callers store three channels there)
  return make_float4(color[0], color[1], color[2], 0.0f);
}

} // namespace ansel
