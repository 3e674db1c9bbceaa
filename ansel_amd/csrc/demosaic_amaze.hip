// demosaic_amaze.hip -- AMaZE demosaic on gfx950.
//
// Reference: amaze_demosaic_RT(), src/iop/demosaic/amaze.cc:181-1419 (CPU only in the reference; there
// is no OpenCL AMaZE).  160 x 160 tiles overlapping by 32; per tile a fixed sequence of stencil stages
// over tile-sized planes.  One workgroup (512 threads) owns one tile at a time and walks the stages
// with a barrier between them; the planes live in a per-workgroup slab of global memory (1.5 MB, L2 /
// Infinity-Cache resident while the tile is in flight) laid out EXACTLY as the reference lays out its
// buffer, because a few stencils read words their logical plane did not write in this tile and see
// what the plane sharing that memory left there, and two border fills run past their plane into the
// next one (oracle/src/demosaic_amaze.c explains and pins this).  The slab is zeroed per tile, which
// makes every pixel a function of the frame alone (the reference keeps it per OpenMP thread).
//
// Three stages are Gauss-Seidel sweeps whose order is part of the result:
//   S3  colour-difference choice + saturation bound: hcd chains run along rows (step 2), vcd chains
//       along columns (step 2) -- one thread per chain, 2 x 2 x 160 chains;
//   S7 / S10  diagonal-neighbour votes on the H/V and +/- weights: row r reads row r-1 already
//       voted -- the plane is staged in LDS and the rows are walked with one barrier per row.
//
// Round 3: amaze_stream keeps every plane of a tile in LDS (amaze_stream_body.h); the body below stays for the few kinds of
// cut tile that kernel refuses (amz::stream_tile_ok()), drawn from the same queue in the same launch (amaze_frame).
#include "hip_common.h"

#include <math.h>
#include <stdlib.h>
#include <atomic>

#define AMZ_FN __device__ __forceinline__
#define AMZ_MEMBER static __device__ __forceinline__
#define AMZ_HD __host__ __device__ __forceinline__
#include "amaze_stream_body.h"

using namespace ansel;

namespace
{

#define TS 160
#define TSH 80
#define V1 TS
#define V2 (2 * TS)
#define V3 (3 * TS)
#define P1 (-TS + 1)
#define P2 (-2 * TS + 2)
#define P3 (-3 * TS + 3)
#define M1 (TS + 1)
#define M2 (2 * TS + 2)
#define M3 (3 * TS + 3)
#define PAD 32
#define NT ((int)blockDim.x) // the slab tiles run in workgroups of 512 (amaze_tiles) or 640 (amaze_frame) threads
#define SLAB_THREADS 512
#define CHAIN_U 8 // sites of a stage-3 chain fetched ahead

// plane offsets in floats: the reference's buffer layout, amaze.cc:274-327 (128 bytes between planes)
enum
{
  O_GREEN = 0,
  O_DELHVSQSUM = O_GREEN + TS * TS + PAD,
  O_DIRWTS0 = O_DELHVSQSUM + TS * TS + PAD,
  O_DIRWTS1 = O_DIRWTS0 + TS * TS + PAD,
  O_VCD = O_DIRWTS1 + TS * TS + PAD,
  O_HCD = O_VCD + TS * TS + PAD,
  O_VCDALT = O_HCD + TS * TS + PAD,
  O_HCDALT = O_VCDALT + TS * TS + PAD,
  O_CDDIFFSQ = O_HCDALT + TS * TS + PAD,
  O_HVWT = O_CDDIFFSQ + TS * TS + 2 * PAD,
  O_DGRB2 = O_HVWT + TS * TSH + PAD,
  O_DGINTH = O_DGRB2 + TS * TS + PAD,
  O_DSQ1M = O_DGINTH + TS * TS + PAD,
  O_DSQ1P = O_DSQ1M + TS * TSH + PAD,
  O_CFA = O_DSQ1P + TS * TSH + PAD,
  O_NYQUIST = O_CFA + TS * TS + PAD,
  O_NYQUTEST = O_NYQUIST + TS * TSH / 4 + PAD,
  O_END = O_NYQUTEST + TS * TSH + PAD + 16 * TS
};

__device__ __forceinline__ float sqr(const float x) { return x * x; }
__device__ __forceinline__ float fmin2(const float a, const float b) { return b < a ? b : a; } // std::min
__device__ __forceinline__ float fmax2(const float a, const float b) { return a < b ? b : a; } // std::max
__device__ __forceinline__ float lim(const float a, const float b, const float c) { return fmax2(b, fmin2(a, c)); }
__device__ __forceinline__ float ulim(const float a, const float b, const float c) { return (b < c) ? lim(a, b, c) : lim(a, c, b); }
__device__ __forceinline__ float intp(const float a, const float b, const float c) { return a * (b - c) + c; }
// xmul2f / xdiv2f / xdivf, amaze.cc:77-121: exponent arithmetic unless the value is +-0
__device__ __forceinline__ float expo(const float d, const int n)
{
  const unsigned u = __float_as_uint(d);
  return (u & 0x7FFFFFFFu) ? __uint_as_float(u + ((unsigned)n << 23)) : d;
}
__device__ __forceinline__ float xmul2f(const float d) { return expo(d, 1); }
__device__ __forceinline__ float xdiv2f(const float d) { return expo(d, -1); }
__device__ __forceinline__ float xdivf(const float d, const int n) { return expo(d, -n); }
// clampnan(), amaze.cc:61-75: only infinities are clamped (the NaN branch is shadowed)
__device__ __forceinline__ float clampnan(const float x, const float m, const float M)
{
  return isfinite(x) ? x : (x < m ? m : (x > M ? M : x));
}
__device__ __forceinline__ int fct(const int r, const int c, const uint32_t filters)
{
  return filters >> ((((r << 1) & 14) + (c & 1)) << 1) & 3;
}

struct amaze_args
{
  int width, height, ntx, ntiles;
  uint32_t filters;
  int ex, ey;
  float clip_pt;
  int slab_all; // the first kernel's body for every tile (ANSEL_HIP_AMAZE_SLAB); else only for those amz::stream_tile_ok() refuses
};

#define EPS 1e-5f
#define EPSSQ 1e-10f
#define ARTHRESH 0.75f

// one site of a stage-3 chain (amaze.cc:585-705): `prev` is the already updated neighbour, c0 / c1 the site's and the next
// neighbour's colour difference, a0..a2 the Hamilton-Adams alternative around the site, before / here / after the mosaic
__device__ __forceinline__ float chain_site(const float prev, const float c0, const float c1, const float a0, const float a1,
                                            const float a2, const float before, const float here, const float after,
                                            const bool gsite, const float clip_pt)
{
  const float cdvar = 3.f * (sqr(prev) + sqr(c0) + sqr(c1)) - sqr(prev + c0 + c1);
  const float altvar = 3.f * (sqr(a0) + sqr(a1) + sqr(a2)) - sqr(a0 + a1 + a2);
  float h = c0;
  if(altvar < cdvar) h = a1;
  if(gsite)
  {
    const float Gint = -h + here;
    if(h > 0)
    {
      if(3.f * h > (Gint + here))
        h = -ulim(Gint, before, after) + here;
      else
      {
        const float wt = 1.f - 3.f * h / (EPS + Gint + here);
        h = wt * h + (1.f - wt) * (-ulim(Gint, before, after) + here);
      }
    }
    if(Gint > clip_pt) h = -ulim(Gint, before, after) + here;
  }
  else
  {
    const float Gint = h + here;
    if(h < 0)
    {
      if(3.f * h < -(Gint + here))
        h = ulim(Gint, before, after) - here;
      else
      {
        const float wt = 1.f + 3.f * h / (EPS + Gint + here);
        h = wt * h + (1.f - wt) * (ulim(Gint, before, after) - here);
      }
    }
    if(Gint > clip_pt) h = ulim(Gint, before, after) - here;
  }
  return h;
}


// all threads of the workgroup over a rows x cols rectangle of the tile: whole tile rows are dealt out (a division by the
// constant TS instead of one by the rectangle's width), the lanes left and right of the rectangle idle
#define FOR_RECT(r0, r1, c0, c1)                                            \
  for(int _k = tid, _n = ((r1) - (r0)) * TS; _k < _n; _k += NT)             \
    for(int rr = (r0) + _k / TS, cc = _k % TS, _once = 1; _once && cc >= (c0) && cc < (c1); _once = 0)
// ... a division by the rectangle's width, no idle lanes (the output stage: 128 of 160 columns)
#define FOR_RECT_TIGHT(r0, r1, c0, c1)                                                            \
  for(int _k = tid, _w = (c1) - (c0), _n = _w > 0 ? ((r1) - (r0)) * _w : 0; _k < _n; _k += NT) \
    for(int rr = (r0) + _k / _w, cc = (c0) + _k % _w, _once = 1; _once; _once = 0)
// ... over rows x the TSH columns of a half-width plane
#define FOR_HALF(r0, r1)                                                    \
  for(int _k = tid, _n = ((r1) - (r0)) * TSH; _k < _n; _k += NT)            \
    for(int rr = (r0) + _k / TSH, cc = _k % TSH, _once = 1; _once; _once = 0)
// all threads over the R/B sites (every other column, phase from the CFA) of rows r0..r1, columns from c0
#define FOR_RB(r0, r1, c0, c1)                                                                       \
  for(int _k = tid, _n = ((r1) - (r0)) * TSH; _k < _n; _k += NT)                                     \
    for(int rr = (r0) + _k / TSH, cc = (c0) + (fct(rr, 2, filters) & 1) + 2 * (_k % TSH), _once = 1; _once && cc < (c1); _once = 0)

// TIMED: the measuring build (ANSEL_HIP_AMAZE_TIMED, tools/amaze_stage_clocks.py) adds the cycles between the STAMPs of
// every tile into stamps[]
#define N_STAMPS 20
template <bool TIMED>
__device__ __forceinline__ void slab_tile(const float *__restrict__ in, float *__restrict__ out, float *const B, const amaze_args &a,
                                          const int top, const int left, float *const vote, int *const nyq,
                                          unsigned long long *__restrict__ stamps)
{
  long long t_prev = 0;
#define STAMP(k)                                                           \
  if(TIMED && threadIdx.x == 0)                                            \
  {                                                                        \
    const long long _t = (long long)__builtin_readcyclecounter();          \
    atomicAdd(&stamps[k], (unsigned long long)(_t - t_prev));              \
    t_prev = _t;                                                           \
  }
  const int tid = threadIdx.x;
  const int width = a.width, height = a.height;
  const uint32_t filters = a.filters;
  const float clip_pt = a.clip_pt, clip_pt8 = 0.8f * a.clip_pt;
  float *const cfa = B + O_CFA, *const green = B + O_GREEN, *const delhvsqsum = B + O_DELHVSQSUM;
  float *const dirwts0 = B + O_DIRWTS0, *const dirwts1 = B + O_DIRWTS1, *const vcd = B + O_VCD, *const hcd = B + O_HCD;
  float *const vcdalt = B + O_VCDALT, *const hcdalt = B + O_HCDALT, *const cddiffsq = B + O_CDDIFFSQ;
  float *const hvwt = B + O_HVWT, *const dgintv = B + O_DGRB2, *const dginth = B + O_DGINTH, *const dgrb2 = B + O_DGRB2;
  float *const dsq1m = B + O_DSQ1M, *const dsq1p = B + O_DSQ1P, *const nyqutest = B + O_NYQUTEST;
  unsigned char *const nyquist = (unsigned char *)(B + O_NYQUIST);
  // shared storage, amaze.cc:300-327
  float *const dgrb0 = vcdalt, *const dgrb1 = vcdalt + TS * TSH;
  float *const delp = cddiffsq, *const delm = cddiffsq + TS * TSH + PAD, *const rbint = delm;
  float *const pmwt = delhvsqsum, *const rbm = vcd, *const rbp = vcd + TS * TSH + PAD;
  unsigned char *const nyquist2 = (unsigned char *)cddiffsq;
  const float *const d0 = dirwts0, *const d1 = dirwts1;

  {
    const int bottom = min(top + TS, height + 16), right = min(left + TS, width + 16);
    const int rr1 = bottom - top, cc1 = right - left;
    const int rrmin = top < 0 ? 16 : 0, ccmin = left < 0 ? 16 : 0;
    const int rrmax = bottom > height ? height - top : rr1, ccmax = right > width ? width - left : cc1;

    if(TIMED) t_prev = (long long)__builtin_readcyclecounter();
    for(int k = tid; k < O_END; k += NT) B[k] = 0.0f;
    __syncthreads();
    STAMP(0)

    // ---- S0 tile load, amaze.cc:352-460: the nine fills in the reference's order (later ones overwrite
    //      earlier ones; the right strip wraps into the next tile row, the bottom strip may run past the
    //      plane into the flag bytes behind it -- both as in the reference)
#define PUT(idx, v)       \
  {                       \
    const int _i = (idx); \
    const float _v = (v); \
    cfa[_i] = _v;         \
    green[_i] = _v;       \
  }
    // (the conditions are uniform: a barrier only behind a fill that ran)
#define FILL(cond, loop, idx, v) \
  if(cond)                       \
  {                              \
    loop PUT(idx, v);            \
    __syncthreads();             \
  }
    FILL(rrmin > 0, FOR_RECT(0, 16, ccmin, ccmax), rr * TS + cc, in[(size_t)(32 - rr + top) * width + (cc + left)])
    FILL(true, FOR_RECT(rrmin, rrmax, ccmin, ccmax), rr * TS + cc, in[(size_t)(rr + top) * width + (cc + left)])
    FILL(rrmax < rr1, FOR_RECT(0, 16, ccmin, ccmax), (rrmax + rr) * TS + cc, in[(size_t)(height - rr - 2) * width + (left + cc)])
    FILL(ccmin > 0, FOR_RECT(rrmin, rrmax, 0, 16), rr * TS + cc, in[(size_t)(rr + top) * width + (32 - cc + left)])
    FILL(ccmax < cc1, FOR_RECT(rrmin, rrmax, 0, 16), rr * TS + ccmax + cc, in[(size_t)(top + rr) * width + (width - cc - 2)])
    FILL(rrmin > 0 && ccmin > 0, FOR_RECT(0, 16, 0, 16), rr * TS + cc, in[(size_t)(32 - rr) * width + (32 - cc)])
    FILL(rrmax < rr1 && ccmax < cc1, FOR_RECT(0, 16, 0, 16), (rrmax + rr) * TS + ccmax + cc,
         in[(size_t)(height - rr - 2) * width + (width - cc - 2)])
    FILL(rrmin > 0 && ccmax < cc1, FOR_RECT(0, 16, 0, 16), rr * TS + ccmax + cc, in[(size_t)(32 - rr) * width + (width - cc - 2)])
    FILL(rrmax < rr1 && ccmin > 0, FOR_RECT(0, 16, 0, 16), (rrmax + rr) * TS + cc, in[(size_t)(height - rr - 2) * width + (32 - cc)])
#undef FILL
#undef PUT
    STAMP(1)

    // ---- S1 gradients, :463-473
    FOR_RECT(2, rr1 - 2, 2, cc1 - 2)
    {
      const int i = rr * TS + cc;
      const float delh = fabsf(cfa[i + 1] - cfa[i - 1]);
      const float delv = fabsf(cfa[i + V1] - cfa[i - V1]);
      dirwts0[i] = EPS + fabsf(cfa[i + V2] - cfa[i]) + fabsf(cfa[i] - cfa[i - V2]) + delv;
      dirwts1[i] = EPS + fabsf(cfa[i + 2] - cfa[i]) + fabsf(cfa[i] - cfa[i - 2]) + delh;
      delhvsqsum[i] = sqr(delh) + sqr(delv);
    }
    __syncthreads();

    STAMP(2)
    // ---- S2 colour differences by adaptive ratios and by Hamilton-Adams, :478-582
    FOR_RECT(4, rr1 - 4, 4, cc1 - 4)
    {
      const int i = rr * TS + cc;
      const bool gsite = fct(rr, cc, filters) & 1;
      const float cru = cfa[i - V1] * (d0[i - V2] + d0[i]) / (d0[i - V2] * (EPS + cfa[i]) + d0[i] * (EPS + cfa[i - V2]));
      const float crd = cfa[i + V1] * (d0[i + V2] + d0[i]) / (d0[i + V2] * (EPS + cfa[i]) + d0[i] * (EPS + cfa[i + V2]));
      const float crl = cfa[i - 1] * (d1[i - 2] + d1[i]) / (d1[i - 2] * (EPS + cfa[i]) + d1[i] * (EPS + cfa[i - 2]));
      const float crr = cfa[i + 1] * (d1[i + 2] + d1[i]) / (d1[i + 2] * (EPS + cfa[i]) + d1[i] * (EPS + cfa[i + 2]));
      const float guha = cfa[i - V1] + xdiv2f(cfa[i] - cfa[i - V2]);
      const float gdha = cfa[i + V1] + xdiv2f(cfa[i] - cfa[i + V2]);
      const float glha = cfa[i - 1] + xdiv2f(cfa[i] - cfa[i - 2]);
      const float grha = cfa[i + 1] + xdiv2f(cfa[i] - cfa[i + 2]);
      float guar = fabsf(1.f - cru) < ARTHRESH ? cfa[i] * cru : guha;
      float gdar = fabsf(1.f - crd) < ARTHRESH ? cfa[i] * crd : gdha;
      float glar = fabsf(1.f - crl) < ARTHRESH ? cfa[i] * crl : glha;
      float grar = fabsf(1.f - crr) < ARTHRESH ? cfa[i] * crr : grha;
      const float hwt = d1[i - 1] / (d1[i - 1] + d1[i + 1]);
      const float vwt = d0[i - V1] / (d0[i + V1] + d0[i - V1]);
      const float Gintvha = vwt * gdha + (1.f - vwt) * guha;
      const float Ginthha = hwt * grha + (1.f - hwt) * glha;
      float v, h, va, ha;
      if(gsite)
      {
        v = cfa[i] - (vwt * gdar + (1.f - vwt) * guar);
        h = cfa[i] - (hwt * grar + (1.f - hwt) * glar);
        va = cfa[i] - Gintvha;
        ha = cfa[i] - Ginthha;
      }
      else
      {
        v = (vwt * gdar + (1.f - vwt) * guar) - cfa[i];
        h = (hwt * grar + (1.f - hwt) * glar) - cfa[i];
        va = Gintvha - cfa[i];
        ha = Ginthha - cfa[i];
      }
      if(cfa[i] > clip_pt8 || Gintvha > clip_pt8 || Ginthha > clip_pt8)
      {
        guar = guha;
        gdar = gdha;
        glar = glha;
        grar = grha;
        v = va;
        h = ha;
      }
      vcd[i] = v;
      hcd[i] = h;
      vcdalt[i] = va;
      hcdalt[i] = ha;
      dgintv[i] = fmin2(sqr(guha - gdha), sqr(guar - gdar));
      dginth[i] = fmin2(sqr(glha - grha), sqr(glar - grar));
    }
    __syncthreads();

    STAMP(3)
    // ---- S3 choose the smoother estimate and bound it, in place (:585-705).  hcd only depends on hcd two
    //      columns to the left in the same row, vcd on vcd two rows up in the same column: chains, one lane each,
    //      the updated neighbour carried in a register.  Nothing else a site reads depends on the chain, so a lane
    //      fetches what CHAIN_U sites read in one go: one memory round trip per CHAIN_U sites.
    //      (Staging column blocks of the horizontal chains in LDS was measured and is slower: the stage waits for the
    //      slab traffic of the other workgroups either way.)
    for(int chain = tid; chain < 4 * TS; chain += NT)
    {
      const bool horizontal = chain < 2 * TS;
      const int line = (horizontal ? chain : chain - 2 * TS) >> 1, par = chain & 1; // neighbouring lanes share cache lines
      const int len = horizontal ? cc1 : rr1, lines = horizontal ? rr1 : cc1; // extent along / across the chain
      if(line < 4 || line >= lines - 4) continue;
      const int step = horizontal ? 1 : TS;
      float *const cd = horizontal ? hcd : vcd;
      const float *const alt = horizontal ? hcdalt : vcdalt;
      const int origin = horizontal ? line * TS : line; // index of position 0 of the chain's line
      const bool gsite = (horizontal ? fct(line, 4 + par, filters) : fct(4 + par, line, filters)) & 1;
      float carried = cd[origin + (2 + par) * step]; // the neighbour in front of the first site: never written
      for(int p0 = 4 + par; p0 < len - 4; p0 += 2 * CHAIN_U)
      {
        float C[CHAIN_U + 1], A[CHAIN_U + 2], F[2 * CHAIN_U + 1];
#pragma unroll
        for(int k = 0; k <= CHAIN_U; k++) C[k] = (p0 + 2 * k < TS) ? cd[origin + (p0 + 2 * k) * step] : 0.f;
#pragma unroll
        for(int k = 0; k <= CHAIN_U + 1; k++) A[k] = (p0 - 2 + 2 * k < TS) ? alt[origin + (p0 - 2 + 2 * k) * step] : 0.f;
#pragma unroll
        for(int k = 0; k <= 2 * CHAIN_U; k++) F[k] = (p0 - 1 + k < TS) ? cfa[origin + (p0 - 1 + k) * step] : 0.f;
#pragma unroll
        for(int k = 0; k < CHAIN_U; k++)
          if(p0 + 2 * k < len - 4)
          {
            carried = chain_site(carried, C[k], C[k + 1], A[k], A[k + 1], A[k + 2], F[2 * k], F[2 * k + 1], F[2 * k + 2], gsite, clip_pt);
            cd[origin + (p0 + 2 * k) * step] = carried;
          }
      }
    }
    __syncthreads();
    FOR_RECT(4, rr1 - 4, 4, cc1 - 4)
    {
      const int i = rr * TS + cc;
      if(!(fct(rr, cc, filters) & 1)) cddiffsq[i] = sqr(vcd[i] - hcd[i]);
    }
    __syncthreads();

    STAMP(4)
    // ---- S4 H/V weight at R/B sites from colour-difference variances, :707-760
    FOR_RB(6, rr1 - 6, 6, cc1 - 6)
    {
      const int i = rr * TS + cc;
      const float uave = vcd[i] + vcd[i - V1] + vcd[i - V2] + vcd[i - V3];
      const float dave = vcd[i] + vcd[i + V1] + vcd[i + V2] + vcd[i + V3];
      const float lave = hcd[i] + hcd[i - 1] + hcd[i - 2] + hcd[i - 3];
      const float rave = hcd[i] + hcd[i + 1] + hcd[i + 2] + hcd[i + 3];
      float vu = sqr(vcd[i] - uave) + sqr(vcd[i - V1] - uave) + sqr(vcd[i - V2] - uave) + sqr(vcd[i - V3] - uave);
      float vd = sqr(vcd[i] - dave) + sqr(vcd[i + V1] - dave) + sqr(vcd[i + V2] - dave) + sqr(vcd[i + V3] - dave);
      float hl = sqr(hcd[i] - lave) + sqr(hcd[i - 1] - lave) + sqr(hcd[i - 2] - lave) + sqr(hcd[i - 3] - lave);
      float hr = sqr(hcd[i] - rave) + sqr(hcd[i + 1] - rave) + sqr(hcd[i + 2] - rave) + sqr(hcd[i + 3] - rave);
      const float hwt = d1[i - 1] / (d1[i - 1] + d1[i + 1]);
      const float vwt = d0[i - V1] / (d0[i + V1] + d0[i - V1]);
      const float vcdvar = EPSSQ + vwt * vd + (1.f - vwt) * vu;
      const float hcdvar = EPSSQ + hwt * hr + (1.f - hwt) * hl;
      vu = (dgintv[i]) + (dgintv[i - V1]) + (dgintv[i - V2]);
      vd = (dgintv[i]) + (dgintv[i + V1]) + (dgintv[i + V2]);
      hl = (dginth[i]) + (dginth[i - 1]) + (dginth[i - 2]);
      hr = (dginth[i]) + (dginth[i + 1]) + (dginth[i + 2]);
      const float vcdvar1 = EPSSQ + vwt * vd + (1.f - vwt) * vu;
      const float hcdvar1 = EPSSQ + hwt * hr + (1.f - hwt) * hl;
      const float varwt = hcdvar / (vcdvar + hcdvar);
      const float diffwt = hcdvar1 / (vcdvar1 + hcdvar1);
      // the product is formed in binary64 in the reference (0.5 is a double literal there)
      if((0.5 - (double)varwt) * (0.5 - (double)diffwt) > 0 && fabsf(0.5f - diffwt) < fabsf(0.5f - varwt))
        hvwt[i >> 1] = varwt;
      else
        hvwt[i >> 1] = diffwt;
    }
    if(tid == 0)
    {
      nyq[0] = 0x7fffffff; // first flagged row
      nyq[1] = 0;          // last flagged row
      nyq[2] = TS + 1;     // min flagged column
      nyq[3] = 0;          // max flagged column
    }
    __syncthreads();

    STAMP(5)
    // ---- S5 Nyquist texture test, :763-820; bounding box of the flagged sites by LDS atomics
    {
      const float gg0 = 0.5f * 0.07384411893421103f, gg1 = 0.5f * 0.06207511968171489f, gg2 = 0.5f * 0.0521818194747806f;
      const float gg3 = 0.5f * 0.03687419286733595f, gg4 = 0.5f * 0.03099732204057846f, gg5 = 0.5f * 0.018413194161458882f;
      const float go0 = 0.14659727707323927f, go1 = 0.103592713382435f, go2 = 0.0732036125103057f, go3 = 0.0365543548389495f;
      const float *const cd = cddiffsq, *const dq = delhvsqsum;
      FOR_RB(6, rr1 - 6, 6, cc1 - 6)
      {
        const int i = rr * TS + cc;
        const float test
            = (go0 * cd[i] + go1 * (cd[i - M1] + cd[i + P1] + cd[i - P1] + cd[i + M1])
               + go2 * (cd[i - V2] + cd[i - 2] + cd[i + 2] + cd[i + V2]) + go3 * (cd[i - M2] + cd[i + P2] + cd[i - P2] + cd[i + M2]))
              - (gg0 * dq[i] + gg1 * (dq[i - V1] + dq[i + 1] + dq[i - 1] + dq[i + V1])
                 + gg2 * (dq[i - M1] + dq[i + P1] + dq[i - P1] + dq[i + M1])
                 + gg3 * (dq[i - V2] + dq[i - 2] + dq[i + 2] + dq[i + V2])
                 + gg4 * (dq[i - V2 - 1] + dq[i - V2 + 1] + dq[i - TS - 2] + dq[i - TS + 2] + dq[i + TS - 2] + dq[i + TS + 2]
                          + dq[i + V2 - 1] + dq[i + V2 + 1])
                 + gg5 * (dq[i - M2] + dq[i + P2] + dq[i - P2] + dq[i + M2]));
        nyqutest[i >> 1] = test;
        if(test > 0.f)
        {
          nyquist[i >> 1] = 1;
          atomicMin(&nyq[0], rr);
          atomicMax(&nyq[1], rr);
          atomicMin(&nyq[2], cc);
          atomicMax(&nyq[3], cc);
        }
      }
    }
    __syncthreads();
    // the reference tracks "first flagged row" with 0 meaning none (rows start at 6, so 0 is never a row)
    int nystartrow = nyq[0] == 0x7fffffff ? 0 : nyq[0], nyendrow = nyq[1], nystartcol = nyq[2], nyendcol = nyq[3];
    const bool do_nyquist = nystartrow != nyendrow && nystartcol != nyendcol;
    if(do_nyquist)
    {
      nyendrow++;
      nyendcol++;
      nystartcol -= (nystartcol & 1);
      nystartrow = max(8, nystartrow);
      nyendrow = min(rr1 - 8, nyendrow);
      nystartcol = max(8, nystartcol);
      nyendcol = min(cc1 - 8, nyendcol);
      // amaze.cc:830: the second flag plane shares its bytes with cddiffsq
      for(int k = tid; k < (TS - 8) * TSH; k += NT) nyquist2[4 * TSH + k] = 0;
      __syncthreads();
      // ---- S6 majority vote on the flags, :832-845
      FOR_RB(nystartrow, nyendrow, nystartcol, nyendcol)
      {
        const int i = rr * TS + cc;
        const unsigned char *ny = nyquist;
        const unsigned n = ny[(i - V2) >> 1] + ny[(i - M1) >> 1] + ny[(i + P1) >> 1] + ny[(i - 2) >> 1] + ny[(i + 2) >> 1]
                           + ny[(i - P1) >> 1] + ny[(i + M1) >> 1] + ny[(i + V2) >> 1];
        nyquist2[i >> 1] = n > 4 ? 1 : (n < 4 ? 0 : ny[i >> 1]);
      }
      __syncthreads();
      // ---- area interpolation of the weight in flagged regions, :850-890
      FOR_RB(nystartrow, nyendrow, nystartcol, nyendcol)
      {
        const int i = rr * TS + cc;
        if(nyquist2[i >> 1])
        {
          float sumcfa = 0.f, sumh = 0.f, sumv = 0.f, sumsqh = 0.f, sumsqv = 0.f, areawt = 0.f;
          for(int p = -6; p < 7; p += 2)
          {
            int i1 = i + (p * TS) - 6;
            for(int q = -6; q < 7; q += 2, i1 += 2)
              if(nyquist2[i1 >> 1])
              {
                const float c = cfa[i1];
                sumcfa += c;
                sumh += (cfa[i1 - 1] + cfa[i1 + 1]);
                sumv += (cfa[i1 - V1] + cfa[i1 + V1]);
                sumsqh += sqr(c - cfa[i1 - 1]) + sqr(c - cfa[i1 + 1]);
                sumsqv += sqr(c - cfa[i1 - V1]) + sqr(c - cfa[i1 + V1]);
                areawt += 1;
              }
          }
          sumh = sumcfa - xdiv2f(sumh);
          sumv = sumcfa - xdiv2f(sumv);
          areawt = xdiv2f(areawt);
          const float hcdvar = EPSSQ + fabsf(areawt * sumsqh - sumh * sumh);
          const float vcdvar = EPSSQ + fabsf(areawt * sumsqv - sumv * sumv);
          hvwt[i >> 1] = hcdvar / (vcdvar + hcdvar);
        }
      }
    }
    __syncthreads();

    STAMP(6)
    // ---- S7 the weight vote, in place, row r sees row r-1 voted (:894-905): staged in LDS, one barrier per row
    for(int k = tid; k < TS * TSH; k += NT) vote[k] = hvwt[k];
    __syncthreads();
    for(int rr = 8; rr < rr1 - 8; rr++)
    {
      const int cc = 8 + (fct(rr, 2, filters) & 1) + 2 * tid;
      if(cc < cc1 - 8)
      {
        const int i = rr * TS + cc;
        const float alt = xdivf(vote[(i - M1) >> 1] + vote[(i + P1) >> 1] + vote[(i - P1) >> 1] + vote[(i + M1) >> 1], 2);
        const float w = vote[i >> 1];
        vote[i >> 1] = fabsf(0.5f - w) < fabsf(0.5f - alt) ? alt : w;
      }
      __syncthreads();
    }
    for(int k = tid; k < TS * TSH; k += NT) hvwt[k] = vote[k];
    __syncthreads();
    STAMP(7)
    // green at R/B sites and its curvature, :907-917
    FOR_RB(8, rr1 - 8, 8, cc1 - 8)
    {
      const int i = rr * TS + cc;
      dgrb0[i >> 1] = intp(hvwt[i >> 1], vcd[i], hcd[i]);
      green[i] = cfa[i] + dgrb0[i >> 1];
    }
    __syncthreads();
    FOR_RB(8, rr1 - 8, 8, cc1 - 8)
    {
      const int i = rr * TS + cc;
      // the reference reads the flag byte whether or not the flags were built; unbuilt flags only feed
      // planes that are not used in that case
      const bool flagged = nyquist2[i >> 1] != 0;
      dgrb2[2 * (i >> 1)] = flagged ? sqr(green[i] - xdiv2f(green[i - 1] + green[i + 1])) : 0.f;
      dgrb2[2 * (i >> 1) + 1] = flagged ? sqr(green[i] - xdiv2f(green[i - V1] + green[i + V1])) : 0.f;
    }
    __syncthreads();

    // ---- S8 refine flagged regions with the curvature of green, :923-956
    if(do_nyquist)
    {
#define GH(k) dgrb2[2 * (k)]
#define GV(k) dgrb2[2 * (k) + 1]
      const float q0 = 0.169917f, q1 = 0.108947f, q2 = 0.069855f, q3 = 0.0287182f;
      FOR_RB(nystartrow, nyendrow, nystartcol, nyendcol)
      {
        const int i = rr * TS + cc;
        if(nyquist2[i >> 1])
        {
          const float gvarh
              = EPSSQ + (q0 * GH(i >> 1) + q1 * (GH((i - M1) >> 1) + GH((i + P1) >> 1) + GH((i - P1) >> 1) + GH((i + M1) >> 1))
                         + q2 * (GH((i - V2) >> 1) + GH((i - 2) >> 1) + GH((i + 2) >> 1) + GH((i + V2) >> 1))
                         + q3 * (GH((i - M2) >> 1) + GH((i + P2) >> 1) + GH((i - P2) >> 1) + GH((i + M2) >> 1)));
          const float gvarv
              = EPSSQ + (q0 * GV(i >> 1) + q1 * (GV((i - M1) >> 1) + GV((i + P1) >> 1) + GV((i - P1) >> 1) + GV((i + M1) >> 1))
                         + q2 * (GV((i - V2) >> 1) + GV((i - 2) >> 1) + GV((i + 2) >> 1) + GV((i + V2) >> 1))
                         + q3 * (GV((i - M2) >> 1) + GV((i + P2) >> 1) + GV((i - P2) >> 1) + GV((i + M2) >> 1)));
          const float dg = (hcd[i] * gvarv + vcd[i] * gvarh) / (gvarv + gvarh);
          dgrb0[i >> 1] = dg;
          green[i] = cfa[i] + dg;
        }
      }
#undef GH
#undef GV
    }
    __syncthreads();

    STAMP(8)
    // ---- S9 diagonal gradients and squared diagonal differences, :958-983 (delp/delm reuse cddiffsq)
    FOR_HALF(6, rr1 - 6)
    {
      const int c2 = 6 + 2 * cc;
      if(c2 < cc1 - 6)
      {
        const int i = rr * TS + c2;
        const bool odd = fct(rr, 2, filters) & 1;
        const int ga = odd ? i + 1 : i, sb = odd ? i : i + 1;
        delp[i >> 1] = fabsf(cfa[ga + P1] - cfa[ga - P1]);
        delm[i >> 1] = fabsf(cfa[ga + M1] - cfa[ga - M1]);
        dsq1p[i >> 1] = (sqr(cfa[sb] - cfa[sb - P1]) + sqr(cfa[sb] - cfa[sb + P1]));
        dsq1m[i >> 1] = (sqr(cfa[sb] - cfa[sb - M1]) + sqr(cfa[sb] - cfa[sb + M1]));
      }
    }
    __syncthreads();
    // diagonal R/B estimates, :986-1107 (rbm/rbp reuse vcd, pmwt reuses delhvsqsum)
    FOR_RB(8, rr1 - 8, 8, cc1 - 8)
    {
      const int i = rr * TS + cc, h = i >> 1;
      const float *const sm = dsq1m, *const sp = dsq1p;
      const float ge0 = 0.13719494435797422f, ge1 = 0.05640252782101291f;
      const float crse = xmul2f(cfa[i + M1]) / (EPS + cfa[i] + (cfa[i + M2]));
      const float crnw = xmul2f(cfa[i - M1]) / (EPS + cfa[i] + (cfa[i - M2]));
      const float crne = xmul2f(cfa[i + P1]) / (EPS + cfa[i] + (cfa[i + P2]));
      const float crsw = xmul2f(cfa[i - P1]) / (EPS + cfa[i] + (cfa[i - P2]));
      const float rbse = fabsf(1.f - crse) < ARTHRESH ? cfa[i] * crse : (cfa[i + M1]) + xdiv2f(cfa[i] - cfa[i + M2]);
      const float rbnw = fabsf(1.f - crnw) < ARTHRESH ? cfa[i] * crnw : (cfa[i - M1]) + xdiv2f(cfa[i] - cfa[i - M2]);
      const float rbne = fabsf(1.f - crne) < ARTHRESH ? cfa[i] * crne : (cfa[i + P1]) + xdiv2f(cfa[i] - cfa[i + P2]);
      const float rbsw = fabsf(1.f - crsw) < ARTHRESH ? cfa[i] * crsw : (cfa[i - P1]) + xdiv2f(cfa[i] - cfa[i - P2]);
      const float wtse = EPS + delm[h] + delm[(i + M1) >> 1] + delm[(i + M2) >> 1];
      const float wtnw = EPS + delm[h] + delm[(i - M1) >> 1] + delm[(i - M2) >> 1];
      const float wtne = EPS + delp[h] + delp[(i + P1) >> 1] + delp[(i + P2) >> 1];
      const float wtsw = EPS + delp[h] + delp[(i - P1) >> 1] + delp[(i - P2) >> 1];
      float vm = (wtse * rbnw + wtnw * rbse) / (wtse + wtnw);
      float vp = (wtne * rbsw + wtsw * rbne) / (wtne + wtsw);
      const float rbvarm
          = EPSSQ + (ge0 * (sm[(i - V1) >> 1] + sm[(i - 1) >> 1] + sm[(i + 1) >> 1] + sm[(i + V1) >> 1])
                     + ge1 * (sm[(i - V2 - 1) >> 1] + sm[(i - V2 + 1) >> 1] + sm[(i - 2 - V1) >> 1] + sm[(i + 2 - V1) >> 1]
                              + sm[(i - 2 + V1) >> 1] + sm[(i + 2 + V1) >> 1] + sm[(i + V2 - 1) >> 1] + sm[(i + V2 + 1) >> 1]));
      const float pw
          = rbvarm / ((EPSSQ + (ge0 * (sp[(i - V1) >> 1] + sp[(i - 1) >> 1] + sp[(i + 1) >> 1] + sp[(i + V1) >> 1])
                                + ge1 * (sp[(i - V2 - 1) >> 1] + sp[(i - V2 + 1) >> 1] + sp[(i - 2 - V1) >> 1] + sp[(i + 2 - V1) >> 1]
                                         + sp[(i - 2 + V1) >> 1] + sp[(i + 2 + V1) >> 1] + sp[(i + V2 - 1) >> 1] + sp[(i + V2 + 1) >> 1])))
                      + rbvarm);
      if(vp < cfa[i])
      {
        if(xmul2f(vp) < cfa[i])
          vp = ulim(vp, cfa[i - P1], cfa[i + P1]);
        else
        {
          const float pwt = xmul2f(cfa[i] - vp) / (EPS + vp + cfa[i]);
          vp = pwt * vp + (1.f - pwt) * ulim(vp, cfa[i - P1], cfa[i + P1]);
        }
      }
      if(vm < cfa[i])
      {
        if(xmul2f(vm) < cfa[i])
          vm = ulim(vm, cfa[i - M1], cfa[i + M1]);
        else
        {
          const float mwt = xmul2f(cfa[i] - vm) / (EPS + vm + cfa[i]);
          vm = mwt * vm + (1.f - mwt) * ulim(vm, cfa[i - M1], cfa[i + M1]);
        }
      }
      if(vp > clip_pt) vp = ulim(vp, cfa[i - P1], cfa[i + P1]);
      if(vm > clip_pt) vm = ulim(vm, cfa[i - M1], cfa[i + M1]);
      // pmwt shares delhvsqsum and rbm/rbp share vcd: neither is read any more by this stage's neighbours
      pmwt[h] = pw;
      rbp[h] = vp;
      rbm[h] = vm;
    }
    __syncthreads();

    STAMP(9)
    // ---- S10 vote on the diagonal weight, in place, row by row (:1109-1126), in LDS like S7
    for(int k = tid; k < TS * TSH; k += NT) vote[k] = pmwt[k];
    __syncthreads();
    for(int rr = 10; rr < rr1 - 10; rr++)
    {
      const int cc = 10 + (fct(rr, 2, filters) & 1) + 2 * tid;
      if(cc < cc1 - 10)
      {
        const int i = rr * TS + cc;
        const float alt = xdivf(vote[(i - M1) >> 1] + vote[(i + P1) >> 1] + vote[(i - P1) >> 1] + vote[(i + M1) >> 1], 2);
        const float w = vote[i >> 1];
        if(fabsf(0.5f - w) < fabsf(0.5f - alt)) vote[i >> 1] = alt;
      }
      __syncthreads();
    }
    for(int k = tid; k < TS * TSH; k += NT) pmwt[k] = vote[k];
    __syncthreads();
    FOR_RB(10, rr1 - 10, 10, cc1 - 10)
    {
      const int i = rr * TS + cc, h = i >> 1;
      rbint[h] = xdiv2f(cfa[i] + rbm[h] * (1.f - pmwt[h]) + rbp[h] * pmwt[h]);
    }
    __syncthreads();

    STAMP(10)
    // ---- S11 where the diagonal estimate discriminates better, redo green from R+B, :1129-1236
    FOR_RB(12, rr1 - 12, 12, cc1 - 12)
    {
      const int i = rr * TS + cc, h = i >> 1;
      if(!(fabsf(0.5f - pmwt[h]) < fabsf(0.5f - hvwt[h])))
      {
        // binary64 where the reference has double literals
        const float cru = (float)((double)cfa[i - V1] * 2.0 / (double)(EPS + rbint[h] + rbint[(h - V1)]));
        const float crd = (float)((double)cfa[i + V1] * 2.0 / (double)(EPS + rbint[h] + rbint[(h + V1)]));
        const float crl = (float)((double)cfa[i - 1] * 2.0 / (double)(EPS + rbint[h] + rbint[(h - 1)]));
        const float crr = (float)((double)cfa[i + 1] * 2.0 / (double)(EPS + rbint[h] + rbint[(h + 1)]));
        const float gu = fabsf(1.f - cru) < ARTHRESH ? rbint[h] * cru : cfa[i - V1] + xdiv2f(rbint[h] - rbint[(h - V1)]);
        const float gd = fabsf(1.f - crd) < ARTHRESH ? rbint[h] * crd : cfa[i + V1] + xdiv2f(rbint[h] - rbint[(h + V1)]);
        const float gl = fabsf(1.f - crl) < ARTHRESH ? rbint[h] * crl : cfa[i - 1] + xdiv2f(rbint[h] - rbint[(h - 1)]);
        const float gr = fabsf(1.f - crr) < ARTHRESH ? rbint[h] * crr : cfa[i + 1] + xdiv2f(rbint[h] - rbint[(h + 1)]);
        float Gintv = (d0[i - V1] * gd + d0[i + V1] * gu) / (d0[i + V1] + d0[i - V1]);
        float Ginth = (d1[i - 1] * gr + d1[i + 1] * gl) / (d1[i - 1] + d1[i + 1]);
        if(Gintv < rbint[h])
        {
          if(2 * Gintv < rbint[h])
            Gintv = ulim(Gintv, cfa[i - V1], cfa[i + V1]);
          else
          {
            const float vwt = (float)(2.0 * (double)(rbint[h] - Gintv) / (double)(EPS + Gintv + rbint[h]));
            Gintv = vwt * Gintv + (1.f - vwt) * ulim(Gintv, cfa[i - V1], cfa[i + V1]);
          }
        }
        if(Ginth < rbint[h])
        {
          if(2 * Ginth < rbint[h])
            Ginth = ulim(Ginth, cfa[i - 1], cfa[i + 1]);
          else
          {
            const float hwt = (float)(2.0 * (double)(rbint[h] - Ginth) / (double)(EPS + Ginth + rbint[h]));
            Ginth = hwt * Ginth + (1.f - hwt) * ulim(Ginth, cfa[i - 1], cfa[i + 1]);
          }
        }
        if(Ginth > clip_pt) Ginth = ulim(Ginth, cfa[i - 1], cfa[i + 1]);
        if(Gintv > clip_pt) Gintv = ulim(Gintv, cfa[i - V1], cfa[i + V1]);
        const float g = Ginth * (1.f - hvwt[h]) + Gintv * hvwt[h];
        green[i] = g;
        dgrb0[h] = g - cfa[i];
      }
    }
    __syncthreads();

    STAMP(11)
    // ---- S12 split G-B from G-R: the B coset moves to the second plane, :1239-1244
    FOR_HALF(0, TS)
    {
      // rr, cc enumerate (tile row, half-plane column)
      if(rr >= 13 - a.ey && rr < rr1 - 12 && ((rr - (13 - a.ey)) & 1) == 0)
      {
        const int h = rr * TSH + cc;
        if(h >= ((rr * TS + 13 - a.ex) >> 1) && h < ((rr * TS + cc1 - 12) >> 1))
        {
          dgrb1[h] = dgrb0[h];
          dgrb0[h] = 0;
        }
      }
    }
    __syncthreads();

    // ---- S13 chrominance at the opposite R/B sites from the four diagonal neighbours, :1246-1276
    FOR_RB(14, rr1 - 14, 14, cc1 - 14)
    {
      const int i = rr * TS + cc;
      float *const D = (1 - fct(rr, cc, filters) / 2) ? dgrb1 : dgrb0;
      const float wtnw = 1.f / (EPS + fabsf(D[(i - M1) >> 1] - D[(i + M1) >> 1]) + fabsf(D[(i - M1) >> 1] - D[(i - M3) >> 1])
                                + fabsf(D[(i + M1) >> 1] - D[(i - M3) >> 1]));
      const float wtne = 1.f / (EPS + fabsf(D[(i + P1) >> 1] - D[(i - P1) >> 1]) + fabsf(D[(i + P1) >> 1] - D[(i + P3) >> 1])
                                + fabsf(D[(i - P1) >> 1] - D[(i + P3) >> 1]));
      const float wtsw = 1.f / (EPS + fabsf(D[(i - P1) >> 1] - D[(i + P1) >> 1]) + fabsf(D[(i - P1) >> 1] - D[(i + M3) >> 1])
                                + fabsf(D[(i + P1) >> 1] - D[(i - P3) >> 1]));
      const float wtse = 1.f / (EPS + fabsf(D[(i + M1) >> 1] - D[(i - M1) >> 1]) + fabsf(D[(i + M1) >> 1] - D[(i - P3) >> 1])
                                + fabsf(D[(i - M1) >> 1] - D[(i + M3) >> 1]));
      const float v
          = (wtnw * (1.325f * D[(i - M1) >> 1] - 0.175f * D[(i - M3) >> 1] - 0.075f * D[(i - M1 - 2) >> 1] - 0.075f * D[(i - M1 - V2) >> 1])
             + wtne * (1.325f * D[(i + P1) >> 1] - 0.175f * D[(i + P3) >> 1] - 0.075f * D[(i + P1 + 2) >> 1] - 0.075f * D[(i + P1 + V2) >> 1])
             + wtsw * (1.325f * D[(i - P1) >> 1] - 0.175f * D[(i - P3) >> 1] - 0.075f * D[(i - P1 - 2) >> 1] - 0.075f * D[(i - P1 - V2) >> 1])
             + wtse * (1.325f * D[(i + M1) >> 1] - 0.175f * D[(i + M3) >> 1] - 0.075f * D[(i + M1 + 2) >> 1] - 0.075f * D[(i + M1 + V2) >> 1]))
            / (wtnw + wtne + wtsw + wtse);
      // every value read above sits at a site of the other R/B colour: nothing written here is read here
      D[i >> 1] = v;
    }
    __syncthreads();

    STAMP(12)
    // ---- S14 output, :1278-1411 (alpha is left as it is)
    FOR_RECT_TIGHT(16, rr1 - 16, 16, cc1 - 16)
    {
      const int row = rr + top, col = cc + left, i = rr * TS + cc;
      if(col < width && row < height)
      {
        float *const o = out + 4 * ((size_t)row * width + col);
        const bool gfirst = (fct(rr, 2, filters) & 1) == 1;
        const bool at_green = gfirst ? !(cc & 1) : (cc & 1);
        if(at_green)
        {
          const float temp = 1.f / (hvwt[(i - V1) >> 1] + 2.f - hvwt[(i + 1) >> 1] - hvwt[(i - 1) >> 1] + hvwt[(i + V1) >> 1]);
          o[0] = clampnan(green[i] - ((hvwt[(i - V1) >> 1]) * dgrb0[(i - V1) >> 1] + (1.f - hvwt[(i + 1) >> 1]) * dgrb0[(i + 1) >> 1]
                                      + (1.f - hvwt[(i - 1) >> 1]) * dgrb0[(i - 1) >> 1] + (hvwt[(i + V1) >> 1]) * dgrb0[(i + V1) >> 1])
                                         * temp, 0.0f, 1.0f);
          o[2] = clampnan(green[i] - ((hvwt[(i - V1) >> 1]) * dgrb1[(i - V1) >> 1] + (1.f - hvwt[(i + 1) >> 1]) * dgrb1[(i + 1) >> 1]
                                      + (1.f - hvwt[(i - 1) >> 1]) * dgrb1[(i - 1) >> 1] + (hvwt[(i + V1) >> 1]) * dgrb1[(i + V1) >> 1])
                                         * temp, 0.0f, 1.0f);
        }
        else
        {
          o[0] = clampnan(green[i] - dgrb0[i >> 1], 0.0f, 1.0f);
          o[2] = clampnan(green[i] - dgrb1[i >> 1], 0.0f, 1.0f);
        }
        o[1] = clampnan(green[i], 0.0f, 1.0f);
      }
    }
    __syncthreads();
    STAMP(13)
  }
#undef STAMP
}

template <bool TIMED>
__global__ __launch_bounds__(SLAB_THREADS) void amaze_tiles(const float *__restrict__ in, float *__restrict__ out,
                                                            float *__restrict__ slabs, const amaze_args a,
                                                            unsigned long long *__restrict__ stamps)
{
  __shared__ float vote[TS * TSH];
  __shared__ int nyq[4];
  float *const B = slabs + (size_t)blockIdx.x * O_END;
  for(int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x)
  {
    const int top = -16 + (tile / a.ntx) * (TS - 32), left = -16 + (tile % a.ntx) * (TS - 32);
    if(a.slab_all || !amz::stream_tile_ok(a.width, a.height, top, left)) slab_tile<TIMED>(in, out, B, a, top, left, vote, nyq, stamps);
  }
}

// ---- the full tiles, on chip
struct stream_env
{
  float *lds;
  __device__ __forceinline__ int tid() const { return (int)threadIdx.x; }
  __device__ __forceinline__ float ldf(const int idx, const int) const { return lds[idx]; }
  __device__ __forceinline__ void stf(const int idx, const int, const float v) const { lds[idx] = v; }
  __device__ __forceinline__ unsigned char ldb(const int bidx, const int) const { return ((const unsigned char *)lds)[bidx]; }
  __device__ __forceinline__ void stb(const int bidx, const int, const unsigned char v) const { ((unsigned char *)lds)[bidx] = v; }
  __device__ __forceinline__ void zero(const int word) const { lds[word] = 0.0f; }
  // a workgroup barrier that orders LDS only: __syncthreads() also waits for the global stores in flight (the output rows),
  // a round trip to memory per step that nothing in the tile depends on
  __device__ __forceinline__ void sync() const
  {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  }
  // between two stretches of ONE wave's LDS accesses: the hardware executes them in program order, the compiler must too
  __device__ __forceinline__ void wave_sync() const
  {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
  }
  // the value of lane - 1 / lane + 1 of the wave (the first / last lane keeps its own): one DPP move (wave_shr:1 / wave_shl:1)
  // instead of a ds_bpermute round trip
  __device__ __forceinline__ float shfl_up1(const float v) const
  {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x138, 0xf, 0xf, false));
  }
  __device__ __forceinline__ float shfl_down1(const float v) const
  {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x130, 0xf, 0xf, false));
  }
  // a value that is the same in every lane of the wave, as the compiler cannot know: in a scalar register
  __device__ __forceinline__ int uniform(const int v) const { return __builtin_amdgcn_readfirstlane(v); }
  // the value of lane ^ 1 (both lanes of the pair call it): one DPP move (quad_perm [1, 0, 3, 2])
  __device__ __forceinline__ float swap1(const float v) const
  {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, false));
  }
  // x + y with the operands in this order (the compiler is free to swap them, and of two NaNs the sum is the first)
  __device__ __forceinline__ float add(const float x, const float y) const
  {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
  }
  __device__ __forceinline__ void stamp(const int) const {}
  // the three colour values of a pixel as one 12-byte store (alpha is left as it is)
  __device__ __forceinline__ void store_rgb(float *const o, const float r, const float g, const float b) const
  {
    typedef float v3f_t __attribute__((ext_vector_type(3)));
    const v3f_t v = { r, g, b };
    __builtin_memcpy(o, &v, 12);
  }
};
// the measuring build (ANSEL_HIP_AMAZE_TIMED): cycles between the barriers of a step, summed per phase over all tiles
struct stream_env_timed : stream_env
{
  long long acc[16]; // (indexed by literals only: registers)
  long long t_prev;
  __device__ __forceinline__ void stamp(const int k)
  {
    const long long t = (long long)__builtin_readcyclecounter();
    acc[k] += t - t_prev;
    t_prev = t;
  }
};

template <bool TIMED>
__global__ __launch_bounds__(amz::STREAM_THREADS) void amaze_stream(const float *__restrict__ in, float *__restrict__ out, const amz::args a,
                                                        const int ntx, const int ntiles, const int ty0,
                                                        unsigned long long *__restrict__ stamps)
{
  extern __shared__ __attribute__((aligned(16))) float amz_lds[];
  if(TIMED)
  {
    stream_env_timed env;
    env.lds = amz_lds;
#pragma unroll
    for(int k = 0; k < 16; k++) env.acc[k] = 0;
    for(int tile = blockIdx.x; tile < ntiles; tile += gridDim.x)
    {
      const int top = -16 + (ty0 + tile / ntx) * (TS - 32), left = -16 + (tile % ntx) * (TS - 32);
      if(!amz::stream_tile_ok(a.width, a.height, top, left)) continue;
      env.t_prev = (long long)__builtin_readcyclecounter();
      amz::tile(env, in, out, a, top, left);
    }
    if(threadIdx.x == 0)
#pragma unroll
      for(int k = 0; k < 16; k++) atomicAdd(&stamps[k], (unsigned long long)env.acc[k]);
  }
  else
  {
    stream_env env{ amz_lds };
    for(int tile = blockIdx.x; tile < ntiles; tile += gridDim.x)
    {
      const int top = -16 + (ty0 + tile / ntx) * (TS - 32), left = -16 + (tile % ntx) * (TS - 32);
      if(amz::stream_tile_ok(a.width, a.height, top, left)) amz::tile(env, in, out, a, top, left);
    }
  }
}

// ---- one launch for the frame: the workgroups draw tiles from a queue, last tile first -- the tiles amz::stream_tile_ok()
//      refuses lie in the last tile row / column and take the first kernel's body (slab_tile: the vote plane in the first
//      50 KB of the LDS block the other tiles use whole).  A workgroup per CU either way; drawn, not dealt, because the two
//      kinds of tile do not take the same time
__global__ __launch_bounds__(amz::STREAM_THREADS) void amaze_frame(const float *__restrict__ in, float *__restrict__ out,
                                                                   float *__restrict__ slabs, const amaze_args a, const amz::args sa,
                                                                   unsigned int *__restrict__ queue)
{
  extern __shared__ __attribute__((aligned(16))) float amz_lds[];
  __shared__ int nyq[4];
  __shared__ unsigned int drawn;
  stream_env env{ amz_lds };
  float *const B = slabs + (size_t)blockIdx.x * O_END;
  for(;;)
  {
    if(threadIdx.x == 0) drawn = atomicAdd(queue, 1u);
    __syncthreads();
    const unsigned int item = drawn;
    __syncthreads();
    if(item >= (unsigned int)a.ntiles) break;
    const int tile = a.ntiles - 1 - (int)item;
    const int top = -16 + (tile / a.ntx) * (TS - 32), left = -16 + (tile % a.ntx) * (TS - 32);
    if(amz::stream_tile_ok(a.width, a.height, top, left))
      amz::tile(env, in, out, sa, top, left);
    else
      slab_tile<false>(in, out, B, a, top, left, amz_lds, nyq, nullptr);
  }
}

} // namespace

namespace ansel
{

// band: a row band of the frame -- `in` holds the mosaic rows [in_row0, in_row0 + in_rows), `out` the rows
// [out_row0, out_row0 + out_rows), the tile rows [tv0, tv1) of the frame's own 128-row tile grid are run; nullptr: the frame
int amaze_demosaic_launch(int devid, const dt_hip_piece_t *piece, uint32_t filters, const float *in, float4 *out, const rcd_band_t *band)
{
  // the test hooks' overrides (testhooks.hip dt_hip_test_dispatch(): tests only, not in include/ansel_hip.h) are read ONCE per
  // launch: a concurrent toggle cannot be seen as two values by one launch (grid size against allocation)
  const int ov_slab = dispatch_override(DISPATCH_AMAZE_SLAB), ov_unfused = dispatch_override(DISPATCH_AMAZE_UNFUSED),
            ov_blocks = dispatch_override(DISPATCH_AMAZE_BLOCKS);
  const int width = piece->roi_in.width, height = piece->roi_in.height;
  if(width <= 0 || height <= 0) return DT_HIP_SUCCESS;
  if(width < 34 || height < 34)
  {
    // the mirrored border reads row / column 32 of the frame (amaze.cc:361-459)
    set_last_error("amaze: frames smaller than 34 x 34 read outside the mosaic in the reference");
    return DT_HIP_INVALID_ARG;
  }
  amaze_args a;
  a.width = width;
  a.height = height;
  a.filters = filters;
  const int f00 = filters & 3, f01 = (filters >> 2) & 3; // FC(0,0), FC(0,1)
  if(f00 == 1)
  {
    if(f01 == 0) { a.ey = 0; a.ex = 1; }
    else { a.ey = 1; a.ex = 0; }
  }
  else
  {
    if(f00 == 0) { a.ey = 0; a.ex = 0; }
    else { a.ey = 1; a.ex = 1; }
  }
  a.clip_pt = fminf(piece->processed_maximum[0], fminf(piece->processed_maximum[1], piece->processed_maximum[2]));
  a.ntx = (width + 16 + (TS - 32) - 1) / (TS - 32);
  const int nty = (height + 16 + (TS - 32) - 1) / (TS - 32);
  a.ntiles = a.ntx * nty;
  a.slab_all = (ov_slab || measuring_env("ANSEL_HIP_AMAZE_SLAB") != nullptr) && !band;
  const int ty_first = band ? band->tv0 : 0, ty_end = band ? (band->tv1 < nty ? band->tv1 : nty) : nty;
  if(band)
  {
    // what the band's tile rows read of the mosaic and write of the output must be in the buffers
    const int need0 = ty_first * (TS - 32) - 16 > 0 ? ty_first * (TS - 32) - 16 : 0;
    const int need1 = ty_end * (TS - 32) + 16 < height ? ty_end * (TS - 32) + 16 : height;
    if(ty_first < 0 || ty_end <= ty_first || band->in_row0 > need0 || band->in_row0 + band->in_rows < need1
       || band->out_row0 != ty_first * (TS - 32)
       || band->out_row0 + band->out_rows != (ty_end * (TS - 32) < height ? ty_end * (TS - 32) : height))
    {
      set_last_error("amaze band: tile rows [%d,%d) need frame rows outside the band buffers", ty_first, ty_end);
      return DT_HIP_INVALID_ARG;
    }
    a.ntiles = a.ntx * (ty_end - ty_first);
  }
  // which tiles keep every plane on chip (amaze_stream_body.h): all but some of the last tile row / column
  int stream_tiles = 0;
  if(!a.slab_all)
    for(int ty = ty_first; ty < ty_end; ty++)
      for(int tx = 0; tx < a.ntx; tx++)
        if(amz::stream_tile_ok(width, height, -16 + ty * (TS - 32), -16 + tx * (TS - 32))) stream_tiles++;
  const int slab_tiles = a.ntiles - stream_tiles;
  if(band && slab_tiles > 0)
  {
    set_last_error("amaze band: %d tiles of these tile rows take the first kernel's body (a last tile column of odd width, a mirrored "
                   "strip past its plane), which has no row-band mode", slab_tiles);
    return DT_HIP_INVALID_ARG;
  }
  hipStream_t s = stream_of(devid);
  const bool timed = measuring_env("ANSEL_HIP_AMAZE_TIMED") != nullptr;
  const bool unfused = timed || ov_unfused || measuring_env("ANSEL_HIP_AMAZE_UNFUSED") != nullptr; // one kernel per kind of tile
  amz::args sa;
  sa.width = width;
  sa.height = height;
  sa.filters = filters;
  sa.ex = a.ex;
  sa.ey = a.ey;
  sa.clip_pt = a.clip_pt;
  sa.in_row0 = band ? band->in_row0 : 0;
  sa.out_row0 = band ? band->out_row0 : 0;
  sa.out_row1 = band ? band->out_row0 + band->out_rows : height;
  // (honoured by the measuring launch only: with a part of a phase switched off the output is wrong)
  sa.variant = timed && measuring_env("ANSEL_HIP_AMAZE_VARIANT") ? atoi(measuring_env("ANSEL_HIP_AMAZE_VARIANT")) : 0;
  if(stream_tiles > 0)
  {
    // the opt-in to more than 64 KB of LDS is per device
    static std::atomic<unsigned long long> attr_set{ 0ull };
    const int hd = hip_device_of(devid);
    if(hd < 0 || hd >= 64) return DT_HIP_INVALID_ARG;
    if(!(attr_set.load() >> hd & 1ull))
    {
      ANSEL_HIP_CHECK(hipFuncSetAttribute((const void *)amaze_stream<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)amz::LDS_BYTES));
#ifdef ANSEL_HIP_MEASURING
      ANSEL_HIP_CHECK(hipFuncSetAttribute((const void *)amaze_stream<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)amz::LDS_BYTES));
#endif
      ANSEL_HIP_CHECK(hipFuncSetAttribute((const void *)amaze_frame, hipFuncAttributeMaxDynamicSharedMemorySize, (int)amz::LDS_BYTES));
      attr_set.fetch_or(1ull << hd);
    }
  }
  // one workgroup per CU: the LDS of a CU each (ANSEL_HIP_AMAZE_BLOCKS: fewer, so that the tests see a workgroup walk many tiles)
  const char *const sb_env = measuring_env("ANSEL_HIP_AMAZE_STREAM_BLOCKS") ? measuring_env("ANSEL_HIP_AMAZE_STREAM_BLOCKS") : measuring_env("ANSEL_HIP_AMAZE_BLOCKS");
  const int sb_max = ov_blocks > 0 ? ov_blocks : (sb_env && atoi(sb_env) > 0 ? atoi(sb_env) : 256);
  if(stream_tiles > 0 && slab_tiles > 0 && !unfused)
  {
    const int blocks = a.ntiles < sb_max ? a.ntiles : sb_max;
    // a slab per workgroup: any of them may draw a tile of the first kind
    float *slabs = (float *)dt_hip_alloc_device_buffer(devid, (size_t)blocks * O_END * sizeof(float));
    unsigned int *queue = (unsigned int *)dt_hip_alloc_device_buffer(devid, 256);
    int rc = DT_HIP_SUCCESS;
    if(!slabs || !queue)
      rc = DT_HIP_SYSMEM_ALLOCATION;
    else if(hipMemsetAsync(queue, 0, sizeof(unsigned int), s) != hipSuccess)
      rc = DT_HIP_DEFAULT_ERROR;
    else
    {
      launch_scope ls(devid, "amaze_frame");
      amaze_frame<<<blocks, amz::STREAM_THREADS, amz::LDS_BYTES, s>>>(in, (float *)out, slabs, a, sa, queue);
    }
    if(rc == DT_HIP_SUCCESS) rc = check_launch("amaze_frame");
    if(queue) dt_hip_release_mem_object(queue);
    if(slabs) dt_hip_release_mem_object(slabs);
    return rc;
  }
  if(stream_tiles > 0)
  {
    const int sblocks = a.ntiles < sb_max ? a.ntiles : sb_max;
#ifdef ANSEL_HIP_MEASURING
    // the phase-clock instantiation exists in the measuring library only (tools/amaze_stage_clocks.py)
    if(timed)
    {
      unsigned long long *stamps = (unsigned long long *)dt_hip_alloc_device_buffer(devid, sizeof(unsigned long long) * 32);
      if(!stamps) return DT_HIP_SYSMEM_ALLOCATION;
      unsigned long long host[32];
      if(hipMemsetAsync(stamps, 0, sizeof(host), s) == hipSuccess)
      {
        amaze_stream<true><<<sblocks, amz::STREAM_THREADS, amz::LDS_BYTES, s>>>(in, (float *)out, sa, a.ntx, a.ntiles, ty_first, stamps);
        if(hipMemcpyAsync(host, stamps, sizeof(host), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess)
          for(int k = 0; k < 16; k++)
            fprintf(stderr, "[amaze_stream_timed] phase %d cycles_per_tile %llu\n", k, host[k] / (unsigned long long)stream_tiles);
      }
      dt_hip_release_mem_object(stamps);
    }
    else
#endif
    {
      launch_scope ls(devid, "amaze_stream");
      amaze_stream<false><<<sblocks, amz::STREAM_THREADS, amz::LDS_BYTES, s>>>(in, (float *)out, sa, a.ntx, a.ntiles, ty_first, nullptr);
    }
    const int rc = check_launch("amaze_stream");
    if(rc != DT_HIP_SUCCESS) return rc;
    if(slab_tiles == 0) return DT_HIP_SUCCESS;
  }
  // two 512-thread workgroups per CU (the vote plane is 50 KiB of LDS each)
  const char *const blocks_env = measuring_env("ANSEL_HIP_AMAZE_BLOCKS");
  const int max_blocks = ov_blocks > 0 ? ov_blocks : (blocks_env ? atoi(blocks_env) : 512);
  const int blocks = a.ntiles < max_blocks ? a.ntiles : max_blocks;
  float *slabs = (float *)dt_hip_alloc_device_buffer(devid, (size_t)blocks * O_END * sizeof(float));
  if(!slabs) return DT_HIP_SYSMEM_ALLOCATION;
  if(timed)
  {
    // the measuring build: cycles per stage summed over the tiles, printed (tools/amaze_stage_clocks.py)
    unsigned long long *stamps = (unsigned long long *)dt_hip_alloc_device_buffer(devid, sizeof(unsigned long long) * N_STAMPS);
    if(!stamps)
    {
      dt_hip_release_mem_object(slabs);
      return DT_HIP_SYSMEM_ALLOCATION;
    }
    unsigned long long host[N_STAMPS];
    if(hipMemsetAsync(stamps, 0, sizeof(host), s) == hipSuccess)
    {
      amaze_tiles<true><<<blocks, SLAB_THREADS, 0, s>>>(in, (float *)out, slabs, a, stamps);
      if(hipMemcpyAsync(host, stamps, sizeof(host), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess)
        for(int k = 0; k < 14; k++) fprintf(stderr, "[amaze_timed] stamp %d cycles_per_tile %llu\n", k, host[k] / (unsigned long long)(slab_tiles > 0 ? slab_tiles : 1));
    }
    dt_hip_release_mem_object(stamps);
  }
  else
  {
    launch_scope ls(devid, "amaze_tiles");
    amaze_tiles<false><<<blocks, SLAB_THREADS, 0, s>>>(in, (float *)out, slabs, a, nullptr);
  }
  dt_hip_release_mem_object(slabs);
  return check_launch("amaze_tiles");
}

} // namespace ansel
