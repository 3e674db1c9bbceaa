// nlm2_body.h -- the non-local-means chunk kernel for INTERIOR chunks (no patch leaves the frame), written once and
// compiled twice: as the body of nlm_chunks_v2<...> (nlmeans.hip, gfx950) and, with a host environment whose
// workgroup is 1024 OS threads on a barrier, by tests/native/nlm2_host.cpp -- so that every index of the schedule
// below is checked against the oracle on a CPU before the kernel ever meets a GPU.
//
// Same arithmetic as nlm_chunks_pipelined (nlmeans.hip; reference src/pixel/nlmeans_core.c:315-532): per chunk and
// patch offset the terms of the column-sum recurrence (A1), the recurrence down the rows (A2), the row-sum recurrence
// across the columns (B), the weight 2^-x and the accumulation (C).  What differs:
//
//   * the window keeps (x, y) of a pixel as one 8-byte word and z in a plane of its own: a pixel is two LDS reads
//     (ds_read_b64 + ds_read_b32) instead of three;
//   * A1: the term of table row t is D(b) - D(b - S) with b = the row entering the patch, S = 2 P + 1 and D(r) the
//     squared differences of row r -- so the terms t, t + S, t + 2 S, ... of one column form a chain in which every D
//     is an "entering" row once and a "leaving" row once.  A work item is such a chain (<= 10 terms): it squares each
//     row ONCE and keeps it in registers for its second use.  Per term 4.7 LDS reads and 15 VALU instructions instead
//     of 12 and ~21.  (a - b)^2 is computed by the same two operations on the same operands as in the reference's
//     diff_of_pixels_diff(), so the term is the same binary32 value;
//   * B reads every table entry once: the value that leaves the sliding window is the one that entered it S steps
//     earlier and is still in a register;
//   * C reads 3 words per pixel and offset instead of 4; the offsets' window shifts come from an LDS table, not from a
//     scalar load at the top of every iteration (whose latency every wave's first LDS wait would pay);
//   * no per-offset geometry: an interior chunk's row / column ranges are the chunk itself;
//   * DEEP (when four tables fit LDS beside the window -- chunks up to 58 rows, e.g. the 100 MP frame's 72 x 56): the
//     four steps run on four DIFFERENT offsets at once, one barrier per offset: A1(i+3) and C(i) on the twelve
//     parallel waves, A2(i+2) and B(i+1) -- the two latency chains, measured at 60 % of a two-table iteration -- on
//     serial waves beside them instead of in front of them.  Otherwise two tables, two barriers per offset
//     (C(i-1) || A2(i), then A1(i+1) || B(i)).
//   Measured on a 24 MP frame (profiles/r02_nlm_variants.json): steps switched off one at a time put A2 + B at 11.4 ms
//   of a 16.1 ms two-table run and A1 + C at 7.3 ms; 16-byte / 8-byte table accesses for the recurrences (which need a
//   16-byte aligned, even pitch) were slower than 4-byte ones on an odd pitch and are not used.
#pragma once

#ifdef __HIPCC__
#define NLM2_FN __device__ __forceinline__
#else
#define NLM2_FN inline
#endif

#define NL2_THREADS 1024
#define NL2_SERIAL 256                     // threads of the serial group (the two recurrences, the first table row)
#define NL2_PAR (NL2_THREADS - NL2_SERIAL) // threads of the parallel group (terms, weights)
#define NL2_PX 7                           // accumulators per parallel thread: ceil(72 * 69 / 768)
#define NL2_MSEG 10                        // most terms of one A1 work item (patch radius 3, 69 rows: one chain of 10)
// two layouts: window pitch (pixels) / table pitch (floats, odd: B walks the table one row per lane).  The tight one
// is exactly what a 72-column chunk with patch radius 2 and shifts up to 7 needs, and lets four tables fit
#define NL2_WP_TIGHT 92
#define NL2_TP_TIGHT 77
#define NL2_WP_LOOSE 96
#define NL2_TP_LOOSE 81

namespace nlm2
{

struct alignas(8) f2 // one 8-byte LDS word (ds_read_b64); without the alignment the compiler reads it as two dwords
{
  float x, y;
};

NLM2_FN int imin(const int a, const int b) { return a < b ? a : b; }

// LDS floats of one workgroup: `ntab` tables, 16 spare table rows (the recurrences read whole batches), the offsets'
// window shifts, the window
inline size_t lds_floats(const int ntab, const int chk_h, const int reach, const int npatch, const int WP, const int TP)
{
  return (size_t)ntab * chk_h * TP + 16 * TP + ((npatch + 3) & ~3) + (size_t)(chk_h + 2 * reach) * 3 * WP;
}

// dt_fast_mexp2f(), src/math/math.h:290-301.  The reference's target converts float -> int with cvttss2si: out of
// range and NaN give INT_MIN.  Env::cvt_i32_sat() is the device's v_cvt_i32_f32 (saturating, NaN -> 0).  Saturation
// and cvttss2si only differ above 2^31 (INT_MAX against INT_MIN), where both sums 0x3f800000 + cv wrap to a negative
// k0 and the result is 0 either way; so only NaN needs a correction -- one compare instead of two and an AND.
template <class Env> NLM2_FN float mexp2(const float x)
{
  const float v = x * -8388608.0f;
  const int cs = Env::cvt_i32_sat(v);
  const int cv = (v != v) ? (int)0x80000000 : cs;
  const int k0 = (int)(0x3f800000u + (unsigned)cv);
  return Env::int_as_float(k0 >= 0x800000 ? k0 : 0);
}

// The same for x = distortion * sharpness, given the distortion and sharpness * -2^23 (exact: a power of two).  Scaling
// by a power of two commutes with the product's rounding -- (d * s) * -2^23 == d * (s * -2^23) bit for bit -- except where
// d * s is subnormal or overflows, and there both forms convert to the same integer (0; INT_MIN): one product instead
// of two.  And the NaN correction as ONE instruction: maxNum(v, -inf) is v for every number and -inf, which converts
// to INT_MIN, for a NaN (v is the result of a product, so a quiet one).
template <class Env> NLM2_FN float mexp2_scaled(const float distortion, const float sharp_m23)
{
  const float v = Env::max_num(distortion * sharp_m23, -__builtin_inff());
  const int k0 = (int)(0x3f800000u + (unsigned)Env::cvt_i32_sat(v));
  return Env::int_as_float(k0 >= 0x800000 ? k0 : 0);
}

// Env: tid(), bid(), lds(), sync(), prio_high(), cvt_i32_sat(), int_as_float(); TIMED + clock() for the measuring build.
// Args: nlm_args of nlmeans.hip (W, H, chk_w, chk_h, nchx, npatch, sharpness, norm[3], luma, chroma, skip_blend,
// reach, cy0, out_row0, out_row1, variant).  F4 / I2: float4 / int2.
// CENTER (round 5): the weight of denoise (profiled)'s non-local-means mode (center_weight >= 0, nlmeans_core.c:416-424): the
// squared difference of the two CENTRE pixels, scaled by center_weight x the patch's area (a.cpn: compute_center_pixel_norm(),
// :147-153), joins the patch's distortion, the sum is divided by 1 + center_weight, and the weight is 2^-max(0, that x sharpness
// - 2).  Only the C role changes; the module's default patch radius there is 1, which the third version's bodies do not take.
template <int P, int WP, int TP, bool DEEP, bool CENTER = false, class Env, class Args, class F4, class I2>
NLM2_FN void body(const Env &env, const F4 *__restrict__ in, F4 *__restrict__ out, const Args &a, const I2 *__restrict__ patches)
{
  constexpr int S = 2 * P + 1;
  constexpr int pitch = TP;
  constexpr int NTAB = DEEP ? 4 : 2;
  const int tid = env.tid();
  const int W = a.W, H = a.H;
  const int cy_launch = env.bid() / a.nchx, cx = env.bid() - cy_launch * a.nchx;
  const int cy = cy_launch + a.cy0;
  const int top = cy * a.chk_h, left = cx * a.chk_w;
  const int bot = imin(top + a.chk_h, H), right = imin(left + a.chk_w, W);
  const int ch = bot - top, cw = right - left;
  const int reach = a.reach;
  // interior: the chunk is whole and no patch of any offset reaches past the frame -- everything else is
  // nlm_chunks_pipelined's (uniform over the workgroup, before any barrier)
  if(!(top >= reach && bot + reach <= H && left >= reach && right + reach <= W && ch == a.chk_h && cw == a.chk_w)) return;

  float *const lds = env.lds();
  const int csw = cw + 2 * P + 1; // table columns: frame columns left - P - 1 .. right + P - 1
  const int tabsz = ch * pitch;
  const int wh = ch + 2 * reach;
  const int n = a.npatch;
  int *const dsv = (int *)(lds + NTAB * tabsz + 16 * pitch); // window shift of every offset
  float *const winf = lds + NTAB * tabsz + 16 * pitch + ((n + 3) & ~3);
  f2 *const XY = (f2 *)winf;                  // [wh][WP]
  float *const Z = winf + 2 * wh * WP;        // [wh][WP]
  const int r0 = top - reach, c0 = left - reach;
  const float n0 = a.norm[0], n1 = a.norm[1], n2 = a.norm[2];

  // (four fetches of a thread in flight, then their stores: nlm3_body.h)
  for(int i0 = tid; i0 < wh * WP; i0 += 4 * NL2_THREADS)
  {
    F4 v[4];
#pragma unroll
    for(int u = 0; u < 4; u++)
    {
      const int i = i0 + u * NL2_THREADS;
      const int wy = i / WP, wx = i - wy * WP;
      const int r = r0 + wy, c = c0 + wx; // r is inside the frame for an interior chunk; the pitch may run past its right edge
      v[u].x = v[u].y = v[u].z = v[u].w = 0.0f;
      if(i < wh * WP && c < W) v[u] = in[(long)r * W + c];
    }
#pragma unroll
    for(int u = 0; u < 4; u++)
    {
      const int i = i0 + u * NL2_THREADS;
      if(i < wh * WP)
      {
        f2 xy;
        xy.x = v[u].x;
        xy.y = v[u].y;
        XY[i] = xy;
        Z[i] = v[u].z;
      }
    }
  }
  for(int i = tid; i < n; i += NL2_THREADS) dsv[i] = patches[i].x * WP + patches[i].y;

  // DEEP: waves 0-1 run A2 and the first table row, wave 2 runs B, waves 3-15 are the parallel group -- THIRTEEN waves:
  // a workgroup's waves go to the four SIMDs round-robin, so the SIMD of waves 3, 7, 11, 15 hosts no serial wave, and
  // with wave 3 idle (as it was) its three parallel waves finished a stage in 2 850 cycles while those that share a
  // SIMD with a recurrence needed 4 370 (tools/nlm_phase_clocks.py, profiles/r02_nlm_phase_clocks_*.json)
  constexpr int SER = DEEP ? 192 : NL2_SERIAL;
  constexpr int PAR = NL2_THREADS - SER;
  const bool par = tid >= SER;
  if(!par) env.prio_high(); // the recurrences are latency chains: let them issue ahead of the parallel waves
  const int u = tid - SER;

  // ---- parallel group: the pixels a thread accumulates for the whole chunk: window offset | table offset << 16
  float accx[NL2_PX], accy[NL2_PX], accz[NL2_PX], accw[NL2_PX];
  int pix[NL2_PX];
#pragma unroll
  for(int k = 0; k < NL2_PX; k++)
  {
    accx[k] = accy[k] = accz[k] = accw[k] = 0.0f;
    const int idx = u + PAR * k;
    const int r = idx / cw, c = idx - r * cw;
    pix[k] = (par && idx < ch * cw) ? (((reach + r) * WP + (reach + c)) | ((r * pitch + c) << 16)) : -1;
  }
  // ---- ... and its A1 work item: column ci (1 .. csw-1; column 0 is outside every patch and stays 0), class k of
  //      the table rows modulo S, segment seg of that class's chain: terms t = 1 + k + (j0 + jj) S, jj < jn
  int a1w = 0, a1t = 0, jn = 0;
  {
    const int ncol = csw - 1;
    const int nseg = PAR / (ncol * S);
    const int m0 = (ch - 2) / S + 1;
    const int mseg = (m0 + nseg - 1) / nseg;
    if(par && u < ncol * S * nseg)
    {
      const int ci = 1 + u % ncol, q = u / ncol;
      const int k = q % S, seg = q / S;
      const int mk = (ch - 2 - k >= 0) ? (ch - 2 - k) / S + 1 : 0;
      const int j0 = seg * mseg;
      const int left_over = mk - j0;
      jn = left_over < 0 ? 0 : (left_over > mseg ? mseg : left_over);
      a1w = (reach + k + j0 * S - P) * WP + (reach - P - 1 + ci); // the first row of the chain: the one leaving at t
      a1t = (1 + k + j0 * S) * pitch + ci;
    }
  }
  env.sync();

  // ---- A1, rows 1.. of the table: the terms of the column recurrence (nlmeans_core.c:437-488), parallel group
  auto A1 = [&](float *const T, const int dS) {
    if(u < ch - 1) T[(u + 1) * pitch] = 0.0f; // column 0 (B leaves row sums in it)
    if(jn > 0)
    {
      const f2 *const pxy = XY + a1w;
      const float *const pz = Z + a1w;
      f2 o = pxy[0], s = pxy[dS];
      float oz = pz[0], sz = pz[dS];
      float dx = o.x - s.x, dy = o.y - s.y, dz = oz - sz;
      float px2 = dx * dx, py2 = dy * dy, pz2 = dz * dz;
#pragma unroll
      for(int jj = 0; jj < NL2_MSEG; jj++)
      {
        if(jj < jn)
        {
          const int d = (jj + 1) * S * WP;
          o = pxy[d];
          s = pxy[d + dS];
          oz = pz[d];
          sz = pz[d + dS];
          dx = o.x - s.x;
          dy = o.y - s.y;
          dz = oz - sz;
          const float nx2 = dx * dx, ny2 = dy * dy, nz2 = dz * dz;
          T[a1t + jj * S * pitch] = ((nx2 - px2) * n0 + (ny2 - py2) * n1) + (nz2 - pz2) * n2;
          px2 = nx2;
          py2 = ny2;
          pz2 = nz2;
        }
      }
    }
  };
  // ---- row 0 of the table: the from-scratch sums at the chunk's first row (init_column_sums(), :208-262)
  auto A1_first = [&](float *const T, const int dS) {
    float v = 0.0f;
    if(tid >= 1)
    {
      const int w0 = (reach - P) * WP + (reach - P - 1 + tid);
#pragma unroll
      for(int r = 0; r < S; r++)
      {
        const f2 o = XY[w0 + r * WP], s = XY[w0 + r * WP + dS];
        const float oz = Z[w0 + r * WP], sz = Z[w0 + r * WP + dS];
        const float dx = o.x - s.x, dy = o.y - s.y, dz = oz - sz;
        v += dx * dx * n0 + dy * dy * n1 + dz * dz * n2;
      }
    }
    T[tid] = v;
  };
  // ---- A2: the column recurrence, one thread per table column, 16 rows of LDS traffic in flight, one dependent
  //      addition per row
  auto A2 = [&](float *const T) {
    float v = T[tid];
    int t0 = 1;
    for(; t0 + 16 <= ch; t0 += 16)
    {
      float term[16];
      float *const col = T + t0 * pitch + tid;
#pragma unroll
      for(int k = 0; k < 16; k++) term[k] = col[k * pitch];
      term[0] = v + term[0];
#pragma unroll
      for(int k = 1; k < 16; k++) term[k] = term[k - 1] + term[k];
      v = term[15];
#pragma unroll
      for(int k = 0; k < 16; k++) col[k * pitch] = term[k];
    }
    if(t0 < ch)
    {
      float term[16];
      float *const col = T + t0 * pitch + tid;
      const int live = ch - t0;
#pragma unroll
      for(int k = 0; k < 16; k++) term[k] = col[k * pitch];
#pragma unroll
      for(int k = 0; k < 16; k++)
      {
        const float next = v + term[k];
        v = k < live ? next : v;
        term[k] = v;
      }
#pragma unroll
      for(int k = 0; k < 16; k++)
        if(k < live) col[k * pitch] = term[k];
    }
  };
  // ---- B: the sliding row sum (:405-415), one thread per table row, in place: the distortion of chunk column j goes
  //      into slot j, whose own value (the one leaving the window at j) was read S steps earlier and is in `carry`
  auto B = [&](float *const T, const int rr) {
    float *const rowp = T + rr * pitch; // slot x = frame column left - P - 1 + x
    float carry[S];
#pragma unroll
    for(int k = 0; k < S; k++) carry[k] = rowp[k];
    float distortion = 0.0f;
#pragma unroll
    for(int k = 1; k < S; k++) distortion += carry[k]; // columns left - P .. left + P - 1
    int jb = 0;
    for(; jb + 16 <= cw; jb += 16)
    {
      float hi[16], d[16];
#pragma unroll
      for(int k = 0; k < 16; k++) hi[k] = rowp[jb + k + S];
#pragma unroll
      for(int k = 0; k < 16; k++) d[k] = hi[k] - (k < S ? carry[k < S ? k : 0] : hi[k >= S ? k - S : 0]);
#pragma unroll
      for(int k = 0; k < S; k++) carry[k] = hi[16 - S + k];
      d[0] = distortion + d[0];
#pragma unroll
      for(int k = 1; k < 16; k++) d[k] = d[k - 1] + d[k];
      distortion = d[15];
#pragma unroll
      for(int k = 0; k < 16; k++) rowp[jb + k] = d[k];
    }
    if(jb < cw)
    {
      float hi[16], d[16];
      const int live = cw - jb;
#pragma unroll
      for(int k = 0; k < 16; k++) hi[k] = rowp[jb + k + S];
#pragma unroll
      for(int k = 0; k < 16; k++)
      {
        const float next = distortion + (hi[k] - (k < S ? carry[k < S ? k : 0] : hi[k >= S ? k - S : 0]));
        distortion = k < live ? next : distortion;
        d[k] = distortion;
      }
#pragma unroll
      for(int k = 0; k < 16; k++)
        if(k < live) rowp[jb + k] = d[k];
    }
  };
  // ---- C: weights and accumulation (:416-436), parallel group; center_weight < 0: w = 2^-(distortion * sharpness)
  // A lane without a pixel in a slot accumulates harmless garbage from window / table offset 0 -- its accumulators are
  // never stored -- which saves four selects per pixel and offset.  (Skipping a slot that is empty in every lane of a
  // wave with a uniform branch was measured: the branches split the LDS reads of the slots into blocks that wait one
  // after the other, 12.3 against 11.7 ms on 24 MP.)
#pragma unroll
  for(int k = 0; k < NL2_PX; k++) pix[k] = pix[k] < 0 ? 0 : pix[k];
  auto C = [&](const float *const T, const int dS) {
    float dist[NL2_PX], qx[NL2_PX], qy[NL2_PX], qz[NL2_PX];
#pragma unroll
    for(int k = 0; k < NL2_PX; k++)
    {
      const int wo = (pix[k] & 0xffff) + dS;
      dist[k] = T[pix[k] >> 16];
      const f2 q = XY[wo];
      qx[k] = q.x;
      qy[k] = q.y;
      qz[k] = Z[wo];
    }
#pragma unroll
    for(int k = 0; k < NL2_PX; k++)
    {
      float w;
      if constexpr(CENTER)
      {
        // pixel_difference(own, shifted, center_norm), :156-165: (diff * diff) * norm per channel, summed (x + y) + z
        const int own = pix[k] & 0xffff;
        const f2 o = XY[own];
        const float dx = o.x - qx[k], dy = o.y - qy[k], dz = Z[own] - qz[k];
        const float dis = (dist[k] + (dx * dx * a.cpn + dy * dy * a.cpn + dz * dz * a.cpn)) / (1.0f + a.center_weight);
        w = mexp2<Env>(Env::max_num(0.0f, dis * a.sharpness - 2.0f));
      }
      else
        w = mexp2<Env>(dist[k] * a.sharpness);
      accx[k] = accx[k] + qx[k] * w;
      accy[k] = accy[k] + qy[k] * w;
      accz[k] = accz[k] + qz[k] * w;
      accw[k] = accw[k] + 1.0f * w;
    }
  };


  // a.variant: 0 = the shipped schedule; bits 4..8 switch a step off (timing experiments only: the result is then
  // wrong) -- A1, A2, B, C, first row
  long long tm_out[6] = { 0, 0, 0, 0, 0, 0 };
  const int var = a.variant;
  const bool do_a1 = !(var & 16), do_a2 = !(var & 32), do_b = !(var & 64), do_c = !(var & 128), do_first = !(var & 256);
  if(DEEP)
  {
    // stage s: A1(s) and C(s - 3) on the parallel waves; A2(s - 1), then the first table row of offset s, on waves
    // 0-1; B(s - 2) on wave 2.  Offset p lives in table p & 3 from its A1 (stage p) to its C (stage p + 3); the table
    // is written again in stage p + 4.
    // Env::TIMED (a measuring build, never the product's): cycles of each step of this wave, summed over the offsets;
    // reading the clock also drains the wave's LDS queue at every step boundary, so the sum is an upper bound
    long long tm[6] = { 0, 0, 0, 0, 0, 0 }; // A1, C, A2, first row, B, barrier wait
    for(int s = 0; s < n + 3; s++)
    {
      long long t0 = 0, t1 = 0, t2 = 0;
      if constexpr(Env::TIMED) t0 = env.clock();
      if(par)
      {
        if(s < n && do_a1) A1(lds + (s & 3) * tabsz, dsv[s]);
        if constexpr(Env::TIMED) t1 = env.clock();
        if(s >= 3 && do_c) C(lds + ((s - 3) & 3) * tabsz, dsv[s - 3]);
        if constexpr(Env::TIMED)
        {
          t2 = env.clock();
          tm[0] += t1 - t0;
          tm[1] += t2 - t1;
        }
      }
      else if(tid < 128)
      {
        if(tid < csw)
        {
          if(s >= 1 && s <= n && do_a2) A2(lds + ((s - 1) & 3) * tabsz);
          if constexpr(Env::TIMED) t1 = env.clock();
          if(s < n && do_first) A1_first(lds + (s & 3) * tabsz, dsv[s]);
          if constexpr(Env::TIMED)
          {
            t2 = env.clock();
            tm[2] += t1 - t0;
            tm[3] += t2 - t1;
          }
        }
        else if constexpr(Env::TIMED) t2 = t0;
      }
      else
      {
        if(s >= 2 && s <= n + 1 && do_b)
        {
          if(tid - 128 < ch) B(lds + ((s - 2) & 3) * tabsz, tid - 128);
        }
        if constexpr(Env::TIMED)
        {
          t2 = env.clock();
          tm[4] += t2 - t0;
        }
      }
      env.sync();
      if constexpr(Env::TIMED) tm[5] += env.clock() - t2;
    }
    if constexpr(Env::TIMED)
    {
      for(int k = 0; k < 6; k++) tm_out[k] = tm[k];
    }
  }
  else
  {
    if(par) A1(lds, dsv[0]);
    else if(tid < csw) A1_first(lds, dsv[0]);
    env.sync();
    for(int i = 0; i <= n; i++)
    {
      float *const Ti = lds + (i & 1) * tabsz;       // offset i, and i + 2
      float *const To = lds + ((i + 1) & 1) * tabsz; // offsets i - 1 and i + 1
      // phase 1: C(i - 1) beside A2(i)
      if(par)
      {
        if(i >= 1 && do_c) C(To, dsv[i - 1]);
      }
      else if(tid < csw && i < n && do_a2)
        A2(Ti);
      env.sync();
      // phase 2: A1(i + 1) beside B(i) and the first row of A1(i + 1)
      if(par)
      {
        if(i + 1 < n && do_a1) A1(To, dsv[i + 1]);
      }
      else if(tid < 128)
      {
        if(tid < csw && i + 1 < n && do_first) A1_first(To, dsv[i + 1]);
      }
      else if(i < n && do_b)
      {
        if(tid - 128 < ch) B(Ti, tid - 128);
      }
      env.sync();
    }
  }

  // ---- normalise, blend (:490-521)
#pragma unroll
  for(int k = 0; k < NL2_PX; k++)
  {
    const int idx = u + PAR * k;
    if(!par || idx >= ch * cw) continue;
    const int rr = idx / cw;
    const int row = top + rr, col = left + (idx - rr * cw);
    if(row < a.out_row0 || row >= a.out_row1) continue;
    const long o = (long)row * W + col;
    F4 r;
    if(a.skip_blend)
    {
      r.x = accx[k] / accw[k];
      r.y = accy[k] / accw[k];
      r.z = accz[k] / accw[k];
      r.w = accw[k] / accw[k];
    }
    else
    {
      const F4 ip = in[o];
      r.x = (ip.x * (1.0f - a.luma)) + (accx[k] / accw[k] * a.luma);
      r.y = (ip.y * (1.0f - a.chroma)) + (accy[k] / accw[k] * a.chroma);
      r.z = (ip.z * (1.0f - a.chroma)) + (accz[k] / accw[k] * a.chroma);
      r.w = (ip.w * 0.0f) + (accw[k] / accw[k] * 1.0f);
    }
    out[o] = r;
    env.store_cell(a, o, r.x);
  }
  if constexpr(Env::TIMED)
  {
    // over the chunk's first row (the pixels above stay alive for the compiler, the measurement replaces them): pixel
    // 2 w holds {A1, C, A2, first}, pixel 2 w + 1 {B, barrier wait, 0, 0} of wave w
    env.sync();
    if((tid & 63) == 0)
    {
      const int wv = tid >> 6;
      F4 r0_, r1_;
      r0_.x = (float)tm_out[0];
      r0_.y = (float)tm_out[1];
      r0_.z = (float)tm_out[2];
      r0_.w = (float)tm_out[3];
      r1_.x = (float)tm_out[4];
      r1_.y = (float)tm_out[5];
      r1_.z = r1_.w = 0.0f;
      out[(long)top * W + left + 2 * wv] = r0_;
      out[(long)top * W + left + 2 * wv + 1] = r1_;
    }
  }
}

} // namespace nlm2
