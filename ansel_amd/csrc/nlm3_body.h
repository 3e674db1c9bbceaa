// nlm3_body.h -- the non-local-means chunk kernel for interior chunks, third version: every wave of the workgroup has ONE
// role for the whole chunk, and what a role reads again for the next patch offset stays in its registers.
//
// Same arithmetic as nlm2_body.h (reference src/pixel/nlmeans_core.c:315-532): per chunk and patch offset the terms of the
// column-sum recurrence (A1), the recurrence down the rows (A2), the row-sum recurrence across the columns (B), the weight
// 2^-x and the accumulation (C); offset p lives in table p & 3 from its A1 (stage p) to its C (stage p + 3), one
// workgroup barrier per stage.  What the second version's counters said (profiles/r02_pmc_sq_100MP.txt):
// the LDS pipe is the bound -- 2 cycles per 4- or 8-byte read of a wave, 4 per 4-byte write, and a wave on its own issues
// them at a fifth of that rate -- with 262 KB through it per offset; the two recurrences, on one and two waves, waited for
// 149 and 112 LDS instructions each.  Here:
//
//   * offsets come in rows of equal dy and consecutive dx (the reference's order whenever `scattering` is 0), so the
//     pixel a lane compares with / accumulates for offset (dy, dx + 1) is the right-hand neighbour of the one for (dy, dx):
//       - an A1 lane owns TWO adjacent table columns and a chain of <= MSEG + 1 rows one patch height apart.  The chain's
//         own pixels never change: registers for the whole chunk.  The shifted pixels slide: one new pixel per row and
//         offset (a ring of two, the loop unrolled by two), instead of four pixel reads per row;
//       - a C lane owns NPXL adjacent pixels of a row: one new pixel per offset (a ring of NPXL, the loop unrolled by
//         NPXL) instead of NPXL, and no accumulator slot is ever empty (7 waves x 64 lanes = 56 rows x 8 lanes);
//     the window is stored de-interleaved by column parity so that lanes two columns apart read adjacent words;
//   * A2 fetches its whole column (<= 56 terms + the 5 first-row terms the A1 lanes of the chain heads leave in a side
//     table instead of recomputing them) in ONE round trip, B its whole row as 19 ds_read_b128 and stores it as 18
//     ds_write_b128 (pitch 84: rows 16-byte aligned, sixteen lanes one row apart hit 64 different banks);
//   * waves are dealt to roles so that every SIMD (waves w, w + 4, w + 8, w + 12 share one) carries the same load:
//     A2 A2 B A1 | A1 A1 A1 A1 | C C A1 C | C C C C.
//
// Written against the same environment as nlm2_body.h and compiled for the host by tests/native/nlm2_host.cpp.
#pragma once

#include "nlm2_body.h"

#include <type_traits>

#define NL3_THREADS 1024
#define NL3_TP 84  // table pitch in floats
#define NL3_XO 3   // table slot x (frame column left - P - 1 + x) is stored at x + 3: slot 1, the first that is read, is 16-byte aligned
#define NL3_FP 80  // pitch of the first-row side table (one row per row of the patch)
#define NL3_WPH 46 // half the window pitch: pixels per column-parity plane of a window row (window <= 92 columns)
#define NL3_A1_LANES 384

namespace nlm3
{
using nlm2::f2;
using nlm2::imin;

struct alignas(16) f4
{
  float x, y, z, w;
};

// 16- and 8-byte LDS accesses: on the device through a pointer to the aligned struct (what makes the compiler emit
// ds_read_b128 / ds_read_b64 instead of pairs of dwords), on the host as copies
#ifdef __HIPCC__
NLM2_FN f4 ld4(const float *const p) { return *reinterpret_cast<const f4 *>(p); }
NLM2_FN void st4(float *const p, const f4 v) { *reinterpret_cast<f4 *>(p) = v; }
NLM2_FN f2 ld2(const float *const p) { return *reinterpret_cast<const f2 *>(p); }
NLM2_FN void st2(float *const p, const f2 v) { *reinterpret_cast<f2 *>(p) = v; }
#else
NLM2_FN f4 ld4(const float *const p)
{
  f4 v;
  __builtin_memcpy(&v, p, 16);
  return v;
}
NLM2_FN void st4(float *const p, const f4 v) { __builtin_memcpy(p, &v, 16); }
NLM2_FN f2 ld2(const float *const p)
{
  f2 v;
  __builtin_memcpy(&v, p, 8);
  return v;
}
NLM2_FN void st2(float *const p, const f2 v) { __builtin_memcpy(p, &v, 8); }
#endif

// rows a variant takes: NPXL pixels per C lane on 72 / NPXL... lanes per row, 448 C lanes
template <int NPXL> constexpr int max_rows() { return NPXL == 9 ? 56 : 74; }

// Patch radius P (round 6: 1 beside the module's default 2 -- denoise (profiled)'s non-local-means mode defaults to 1): the chains of
// the A1 role are S = 2 P + 1 rows apart, cut into a1_segments<P>() segments so that S x segments chains of 32 column pairs fill
// the ten half-waves of the role's five main waves (P = 2: 5 x 2 = 10; P = 1: 3 x 3 = 9, the tenth half-wave repeats the ninth).
// Radius 3 would need 7 x 2 = 14 half-waves or chains of 9 rows in registers: it keeps the second version's body.
template <int P> constexpr int a1_segments() { return P == 1 ? 3 : 2; }

// LDS floats of one workgroup: four tables -- always of the variant's most rows, so that the column recurrence runs a
// compile-time number of rows: those beyond the chunk hold garbage nobody reads --, two first-row side tables, the
// window (x, y as 8-byte words + z)
template <int NPXL, int P = 2> inline size_t lds_floats(const int chk_h, const int reach)
{
  return (size_t)4 * max_rows<NPXL>() * NL3_TP + 2 * (2 * P + 1) * NL3_FP + (size_t)(chk_h + 2 * reach) * 2 * NL3_WPH * 3;
}

// ---- the fused variant (round 4; launched as nlm_chunks_v4): THREE tables and no B role.  Chunks of 57 - 64 rows (the 45 MP
// and 60 MP frames' grids: 64) do not fit four tables beside the window (179 KB), nor 2 + 1 + 6 + 8 = 17 waves a workgroup.
// The row recurrence moves into the C role: a C lane owns NPXL = 9 adjacent pixels of a row, eight lanes a row; it reads
// the NPXL + 5 column sums its pixels' patches span, and the recurrence runs along the row in eight phases -- in phase k the
// lanes k of every row hold their final sums, the carry travels to lane k + 1 by a DPP row shift (lanes k' < k just
// recompute what they had: no lane mask).  72 dependent additions per wave and offset instead of per workgroup, but no
// table that lives for a fourth stage, no 37 b128 accesses per row, and sixteen waves: A2 A2 A1 A1 | A1 A1 A1 A1 | C x 8.
constexpr int FUSED_MAXCH = 64;
template <int NPXL, int P = 2> inline size_t lds_floats_fused(const int chk_h, const int reach)
{
  return (size_t)3 * FUSED_MAXCH * NL3_TP + 2 * (2 * P + 1) * NL3_FP + (size_t)(chk_h + 2 * reach) * 2 * NL3_WPH * 3;
}
// the A1 role's dealing (body(), A1): five waves of 2 x 32 column pairs, one chain per half-wave, + one wave of the pairs that are left
template <int MSEG, int P> inline bool a1_fits(const int chk_w, const int chk_h)
{
  constexpr int S = 2 * P + 1, NSEG = a1_segments<P>(), NCH = S * NSEG;
  static_assert(NCH <= 10, "one chain per half-wave of the five main A1 waves");
  const int ncp = (chk_w + 2 * P + 1) / 2;
  if(ncp < 32 || NCH * (ncp - 32) > 64) return false;
  const int m0 = (chk_h - 2) / S + 1;
  return (m0 + NSEG - 1) / NSEG <= MSEG;
}
template <int NPXL, int MSEG, int P = 2> inline bool fits_fused(const int chk_w, const int chk_h, const int radius, const int reach)
{
  constexpr int S = 2 * P + 1, LPR = (72 + NPXL - 1) / NPXL;
  if(radius != P || chk_w > 72 || (chk_w & 1) || chk_w + 2 * reach > 2 * NL3_WPH || chk_h > FUSED_MAXCH || chk_h < 2 * S) return false;
  if(LPR != 8) return false; // two pixel rows per 16-lane DPP row
  return a1_fits<MSEG, P>(chk_w, chk_h);
}

// can this body take the chunk grid?  (the launch and the host harness ask the same question)
template <int NPXL, int MSEG, int P = 2> inline bool fits(const int chk_w, const int chk_h, const int radius, const int reach)
{
  constexpr int S = 2 * P + 1, LPR = (72 + NPXL - 1) / NPXL;
  if(radius != P || chk_w > 72 || (chk_w & 1) || chk_w + 2 * reach > 2 * NL3_WPH || chk_h > max_rows<NPXL>() || chk_h < 2 * S) return false;
  if(LPR * chk_h > 448) return false;
  return a1_fits<MSEG, P>(chk_w, chk_h);
}

// a chunk of the border ring the BORDER body takes (the others keep the first version's body): rows for every chain
inline constexpr bool border_fits(const int cw, const int ch) { return ch >= 10 && cw >= 1; }

// the offsets as rows of consecutive column shifts: ndx per row, every row starting at the same column shift
template <class I2> inline bool regular_grid(const I2 *const patches, const int n, int *const ndx_out)
{
  if(n < 1) return false;
  int ndx = 1;
  while(ndx < n && patches[ndx].x == patches[0].x) ndx++;
  if(n % ndx) return false;
  for(int p = 0; p < n; p++)
  {
    const int row = p / ndx, j = p - row * ndx;
    if(patches[p].x != patches[row * ndx].x || patches[p].y != patches[0].y + j) return false;
  }
  *ndx_out = ndx;
  return true;
}

// the column recurrence over table rows T0 .. T1 - 1, every sum stored with ds_write_addtid_b32 (address = M0 + offset +
// 4 * lane: no address register, 2 LDS cycles per store instead of 4); the row offset must be an immediate
// BLOCKS (round 6; the third version): eight rows per Env::chain8() -- on the device ONE statement that writes M0 once for its eight
// stores.  A store of its own (st_addtid()) is three issue slots, s_mov_b32 m0 / s_nop / ds_write_addtid_b32, beside the row's one
// addition: 224 of the ~260 instructions an A2 wave issued per offset.  Measured same box, three runs each, with the C role's fetches
// behind its accumulation: 21.07 -> 20.86 ms at 100 MP; on the fused variant (64 rows) 14.01 -> 14.11 at 60 MP and at patch radius 1
// 21.28 -> 21.61: not used there.
template <int T0, int T1, int ROWBYTES, class Env, bool BLOCKS = false> struct column_chain
{
  static NLM2_FN void run(const Env &env, float *const wave_base, const int lane, float &v, const float *const term)
  {
    if constexpr(BLOCKS && T0 + 8 <= T1)
    {
      env.template chain8<T0, ROWBYTES>(wave_base, lane, v, term);
      if constexpr(T0 + 8 < T1) column_chain<T0 + 8, T1, ROWBYTES, Env, BLOCKS>::run(env, wave_base, lane, v, term);
    }
    else
    {
      v = v + term[T0];
      env.template st_addtid<T0 * ROWBYTES>(wave_base, lane, v);
      if constexpr(T0 + 1 < T1) column_chain<T0 + 1, T1, ROWBYTES, Env, BLOCKS>::run(env, wave_base, lane, v, term);
    }
  }
};

// Env: tid(), bid(), lds(), sync(), prio_high(), st_addtid<>(), lane_shr1(), cvt_i32_sat(), int_as_float().  Args: nlm_args of nlmeans.hip.
// BORDER: a chunk of the outermost ring, where patches and shifted pixels leave the frame and the reference clips, per
// offset, the rows and columns a patch sums (init_column_sums(), :208-262; the three branches of :437-488) and the
// pixels it weighs (:398-404).  All of that clipping is "this squared difference is not there", and the interior's
// arithmetic reproduces it with the squared difference set to +0 where one of its two pixels lies outside the frame:
//   * a term (entering - leaving) with one side +0 is the reference's add-only / subtract-only branch bit for bit
//     ((x - 0) n == x n, (0 - x) n == -(x n), and s + (-t) == s - t), with both sides +0 it is +0, its "no change";
//   * a column sum that starts below the chunk's first row (rows whose shifted pixel is above the frame) is the sum, in
//     ascending rows, of what has entered by then -- the from-scratch sum of the reference; the rows above hold
//     partial sums nobody reads.  Columns outside the patch range stay +0, as the reference's table does;
//   * the row recurrence needs no clipping at all: in front of the first weighed column it adds zeros;
//   * C gives weight +0 to the pixels the offset does not reach (their shifted pixel is a zero of the window).
// The chunk may be narrower / lower than the grid's (the frame's last column and row of chunks).
// TALL (round 5; FUSED, interior chunks): the chunk grid's rows are higher than the body holds (65 - 69: the 24 / 42 / 150 MP
// frames).  The body runs the chunk's first TALL_HEAD rows exactly as it runs a chunk of that height -- window, tables,
// chains: nothing below row TALL_HEAD - 1 enters them -- and its A2 lanes store, per offset, the column sum they hold behind
// the last row to seeds_out[offset][slot]: nlm_tail_body.h continues the recurrence from there through the rows that are left.
constexpr int TALL_HEAD = 64;
constexpr int TALL_SEED_PITCH = 80; // floats per offset (NLT_SEED_PITCH)
// P, CENTER (round 6): patch radius 1 or 2; the weight of denoise (profiled)'s non-local-means mode (center_weight >= 0,
// nlmeans_core.c:416-424) in the C role -- the squared difference of the two CENTRE pixels x a.cpn joins the patch's distortion, the
// sum is divided by 1 + center_weight (Env::div_uniform(): the correctly rounded quotient by a wave-uniform divisor), and the weight
// is 2^-max(0, that x sharpness - 2).  A C lane keeps its NPXL own pixels in registers beside the ring of shifted ones.
template <int NPXL, int MSEG, bool BORDER = false, bool FUSED = false, bool TALL = false, int P = 2, bool CENTER = false, class Env,
          class Args, class F4, class I2>
NLM2_FN void body(const Env &env, const F4 *__restrict__ in, F4 *__restrict__ out, const Args &a, const I2 *__restrict__ patches,
                  const int ndx, float *__restrict__ seeds_out = nullptr)
{
  static_assert(!TALL || FUSED, "the head of a tall chunk runs on the fused body");
  constexpr int S = 2 * P + 1, TP = NL3_TP, XO = NL3_XO, FP = NL3_FP, WPH = NL3_WPH;
  constexpr int NSEG = a1_segments<P>(), NCH = S * NSEG;
  constexpr int MAXCH = FUSED ? FUSED_MAXCH : max_rows<NPXL>();
  constexpr int NT = FUSED ? 3 : 4; // tables: offset p lives in table tslot(p) from its A1 to its C
  auto tslot = [](const int p) { return FUSED ? p % 3 : (p & 3); };
  constexpr int LPR = (72 + NPXL - 1) / NPXL; // C lanes per chunk row
  const int tid = env.tid();
  const int W = a.W, H = a.H;
  const int cy_launch = env.bid() / a.nchx, cx = env.bid() - cy_launch * a.nchx;
  const int cy = cy_launch + a.cy0;
  const int top = cy * a.chk_h, left = cx * a.chk_w;
  const int bot = imin(top + a.chk_h, H), right = imin(left + a.chk_w, W);
  const int ch_grid = bot - top, cw = right - left;
  const int reach = a.reach;
  // interior: the chunk is whole and no patch of any offset reaches past the frame (uniform, before any barrier)
  const bool interior = top >= reach && bot + reach <= H && left >= reach && right + reach <= W && ch_grid == a.chk_h && cw == a.chk_w;
  const int ch = TALL ? imin(TALL_HEAD, ch_grid) : ch_grid; // the rows this body computes
  if(BORDER ? (interior || !border_fits(cw, ch)) : !interior) return;

  float *const lds = env.lds();
  constexpr int tabsz = MAXCH * TP;
  const int wh = ch + 2 * reach;
  const int n = a.npatch, ndy = n / ndx;
  // the window first, the tables behind it: an A1 lane whose chain is shorter than MSEG reads "rows" beyond its chain --
  // beyond the window for the last chains -- and those words must exist (what they hold does not matter)
  float *const XY = lds;                        // [wh][2][WPH] 8-byte words
  float *const Z = XY + 2 * (wh * 2 * WPH);     // [wh][2][WPH]
  float *const tab = Z + wh * 2 * WPH;          // [NT][MAXCH][TP]
  float *const Fb = tab + NT * tabsz;           // [2][S][FP]
  const int r0 = top - reach, c0 = left - reach;
  const float n0 = a.norm[0], n1 = a.norm[1], n2 = a.norm[2];
  const int ww = cw + 2 * reach; // window columns that exist

  // four fetches of a thread in flight, then their stores (round 6: the loop had ONE -- `s_waitcnt vmcnt(0)` in front of every store,
  // seven memory round trips one after the other at the head of every chunk with nothing else running on the CU)
  for(int i0 = tid; i0 < wh * 2 * WPH; i0 += 4 * NL3_THREADS)
  {
    F4 v[4];
#pragma unroll
    for(int u = 0; u < 4; u++)
    {
      const int i = i0 + u * NL3_THREADS;
      const int wy = i / (2 * WPH), rem = i - wy * (2 * WPH);
      const int par = rem / WPH, h = rem - par * WPH;
      const int wx = 2 * h + par;
      const int r = r0 + wy, c = c0 + wx; // r is inside the frame for an interior chunk
      v[u].x = v[u].y = v[u].z = v[u].w = 0.0f;
      if(i < wh * 2 * WPH && wx < ww && c < W && (!BORDER || (r >= 0 && r < H && c >= 0))) v[u] = in[(long)r * W + c];
    }
#pragma unroll
    for(int u = 0; u < 4; u++)
    {
      const int i = i0 + u * NL3_THREADS;
      if(i < wh * 2 * WPH)
      {
        f2 xy;
        xy.x = v[u].x;
        xy.y = v[u].y;
        st2(XY + 2 * i, xy);
        Z[i] = v[u].z;
      }
    }
  }
  env.sync();
  // window pixel (wy, wx): index of its words
  auto widx = [&](const int wy, const int wx) { return (wy * 2 + (wx & 1)) * WPH + (wx >> 1); };

  const int w = tid >> 6, lane = tid & 63;
  // 0; timing experiments (ANSEL_NLM2_VARIANT, measuring builds): 16 / 32 / 64 / 128 switch A1 / A2 / B / C off (wrong results).
  // A constant in the product's device build: the switches' uniform branches would cut a stage into basic blocks that the
  // scheduler cannot interleave
#if defined(__HIPCC__) && !defined(ANSEL_HIP_MEASURING)
  constexpr int var = 0;
#else
  const int var = a.variant;
#endif
  // ---------------------------------------------------------------------------------------------------------------------
  // which wave plays what (waves w, w + 4, w + 8, w + 12 share a SIMD).  Four roles: layout 0 is round 3's -- A2 A2 B A1 | A1 A1 A1 A1 |
  // C C A1 C | C C C C --, layout 1 (measuring builds, ANSEL_NLM2_VARIANT bit 4096) deals the VALU instructions evenly over the SIMDs
  // instead of the VALU + LDS instructions: A2 A2 B A1 | A1 A1 A1 C | A1 A1 C C | C C C C
  // CENTER (four roles): a C wave carries ~2.4 times the instructions of the plain weight -- A2 C C A1 | A2 C C A1 | B C C A1 | C A1 A1 A1
  // puts two of them on three SIMDs and one beside three A1 waves on the fourth
  const bool lay1 = !FUSED && !CENTER && (var & 4096);
  const int a1_index = FUSED ? ((w >= 2 && w <= 7) ? w - 2 : -1)
                       : CENTER ? (w == 7 ? 0 : (w == 11 ? 1 : (w >= 12 ? w - 10 : -1)))
                                : (lay1 ? ((w >= 3 && w <= 6) ? w - 3 : ((w == 8 || w == 9) ? w - 4 : -1))
                                        : (w == 3 ? 0 : ((w >= 4 && w <= 7) ? w - 3 : (w == 10 ? 5 : -1))));
  const int c_index = FUSED ? w - 8
                      : CENTER ? (w <= 6 ? w - 3 : w - 4) // waves 3 - 6 and 8 - 10
                               : (lay1 ? (w == 7 ? 0 : w - 9) : (w == 8 ? 0 : (w == 9 ? 1 : w - 9)));
  if(a1_index >= 0)
  {
    // ---- A1: the terms of the column recurrence (nlmeans_core.c:437-488) for table rows 1.., and the five squared
    //      differences the first table row sums (init_column_sums(), :208-262)
    const int ai = a1_index;
    const int ncp = (cw + 2 * P + 1) / 2; // (an odd last slot has a partner nobody reads)
    constexpr int nseg = NSEG;            // fits(); a narrower border chunk keeps the layout and repeats items
    const int m0 = (ch - 2) / S + 1;
    const int mseg = (m0 + nseg - 1) / nseg;
    // Work items = (column pair g < ncp, chain q < 10).  Lanes 0-31 and 32-63 of a wave are served by the LDS in separate
    // passes, and a pass is conflict-free when its 32 lanes read 32 consecutive words: waves 0-4 hold column pairs 0-31
    // of one chain per half (with the items in plain order, 38 to a chain, every half straddled two window rows whose
    // words share banks: 3-4 LDS cycles per fetch instead of 2, profiles/r03_nlm_bank_model.txt), the last wave the
    // <= 6 pairs per chain that are left.  Lanes beyond the last item repeat it (same reads, the same values stored to the
    // same words): no lane of the role is ever masked.
    const int extra = ncp - 32; // 0 .. 6 (fits())
    int g, q;
    if(ai < 5)
    {
      q = imin(2 * ai + (lane >> 5), NCH - 1); // (patch radius 1: nine chains, the tenth half-wave repeats the ninth)
      g = lane & 31;
    }
    else
    {
      const int l = extra > 0 ? imin(lane, S * nseg * extra - 1) : 0;
      q = extra > 0 ? l / extra : 0;
      g = extra > 0 ? 32 + l - q * extra : 0;
    }
    const int k = q % S, seg = q / S;
    const int mk = (ch - 2 - k >= 0) ? (ch - 2 - k) / S + 1 : 0;
    const int j0 = seg * mseg;
    const int left_over = mk - j0;
    const int jn = left_over < 0 ? 0 : (left_over > mseg ? mseg : left_over); // terms of the chain; its rows are 0 .. jn
    const bool head = seg == 0; // the chain starts at row k of the patch around the chunk's first row
    const int wr0 = reach - P + k + j0 * S;  // window row of the chain's first row (the one leaving at its first term)
    const int x0 = 1 + 2 * g;                // table slots x0, x0 + 1
    const int wc0 = reach - P - 1 + x0;      // window column of slot x0
    // Every lane computes MSEG + 1 rows and MSEG terms whatever its chain's length -- no branch, no lane mask in the loop:
    // a row beyond the chain reads whatever lies a patch height further down (the tables, behind the window's last rows:
    // every row of every lane is base + a constant), a term beyond it is not stored; the first-row value of a lane that is
    // no chain head goes to a word nobody reads (the first two words of the side table, in front of the slots).
    constexpr int RSTEP = S * 2 * WPH; // words between two rows of a chain
    const int rowbase = wr0 * 2 * WPH;
    constexpr int DUMMY = 0;
    const int toff0 = (1 + k + j0 * S) * TP + XO + x0; // the chain's first term in a table
    const int foff = k * FP + XO + x0;
    // the chain's own pixels
    float ox[MSEG + 1][2], oy[MSEG + 1][2], oz[MSEG + 1][2];
#pragma unroll
    for(int i = 0; i <= MSEG; i++)
#pragma unroll
      for(int c = 0; c < 2; c++)
      {
        const int wx = wc0 + c;
        const int wi = rowbase + i * RSTEP + (wx & 1) * WPH + (wx >> 1);
        const f2 v = ld2(XY + 2 * wi);
        ox[i][c] = v.x;
        oy[i][c] = v.y;
        oz[i][c] = Z[wi];
      }
    // BORDER: which of the chain's rows and columns are in the frame, and (per row of offsets / per offset) whose shifted
    // pixel is
    bool own_row[MSEG + 1], row_ok[MSEG + 1], own_col[2];
#pragma unroll
    for(int i = 0; i <= MSEG; i++)
    {
      const int R = r0 + wr0 + i * S;
      own_row[i] = row_ok[i] = !BORDER || (R >= 0 && R < H);
    }
#pragma unroll
    for(int c = 0; c < 2; c++) own_col[c] = !BORDER || (c0 + wc0 + c >= 0 && c0 + wc0 + c < W);
    // the shifted pixels: ring slot of column c at step j of a row of offsets is (c + j) & 1
    float sx[MSEG + 1][2], sy[MSEG + 1][2], sz[MSEG + 1][2];
#pragma unroll
    for(int i = 0; i <= MSEG; i++) sx[i][0] = sx[i][1] = sy[i][0] = sy[i][1] = sz[i][0] = sz[i][1] = 0.0f;

    // both columns of every row of the chain at the head of a row of offsets
    auto load_row_head = [&](const int dy, const int dx) {
      const int dyo = rowbase + dy * 2 * WPH;
#pragma unroll
      for(int c = 0; c < 2; c++)
      {
        const int wx = wc0 + c + dx;
        const int base = dyo + (wx & 1) * WPH + (wx >> 1);
#pragma unroll
        for(int i = 0; i <= MSEG; i++)
        {
          const int wi = base + i * RSTEP;
          const f2 v = ld2(XY + 2 * wi);
          sx[i][c] = v.x;
          sy[i][c] = v.y;
          sz[i][c] = Z[wi];
        }
      }
    };
    // A step computes from the ring; the fetch of what the next step needs follows it (slide()), in flight while the wave
    // stores its terms and waits at the barrier
    auto step = [&](auto ph_tag, const int p, const int dx) {
      constexpr int PH = decltype(ph_tag)::value;
      float *const T = tab + tslot(p) * tabsz;
      float *const F = Fb + (p & 1) * S * FP;
      const int fo = head ? foff : DUMMY;
      bool col_ok[2] = { true, true };
      if(BORDER)
      {
#pragma unroll
        for(int c = 0; c < 2; c++) col_ok[c] = own_col[c] && (unsigned)(c0 + wc0 + c + dx) < (unsigned)W;
      }
      float px2[2], py2[2], pz2[2];
#pragma unroll
      for(int c = 0; c < 2; c++)
      {
        const int sl = (c + PH) & 1;
        const float dx_ = ox[0][c] - sx[0][sl], dy_ = oy[0][c] - sy[0][sl], dz_ = oz[0][c] - sz[0][sl];
        px2[c] = dx_ * dx_;
        py2[c] = dy_ * dy_;
        pz2[c] = dz_ * dz_;
        if(BORDER && !(row_ok[0] && col_ok[c])) px2[c] = py2[c] = pz2[c] = 0.0f;
      }
      {
        f2 d;
        d.x = px2[0] * n0 + py2[0] * n1 + pz2[0] * n2;
        d.y = px2[1] * n0 + py2[1] * n1 + pz2[1] * n2;
        st2(F + fo, d);
      }
#pragma unroll
      for(int i = 1; i <= MSEG; i++)
      {
        f2 t;
        float nx2[2], ny2[2], nz2[2];
#pragma unroll
        for(int c = 0; c < 2; c++)
        {
          const int sl = (c + PH) & 1;
          const float dx_ = ox[i][c] - sx[i][sl], dy_ = oy[i][c] - sy[i][sl], dz_ = oz[i][c] - sz[i][sl];
          nx2[c] = dx_ * dx_;
          ny2[c] = dy_ * dy_;
          nz2[c] = dz_ * dz_;
          if(BORDER && !(row_ok[i] && col_ok[c])) nx2[c] = ny2[c] = nz2[c] = 0.0f;
        }
        t.x = ((nx2[0] - px2[0]) * n0 + (ny2[0] - py2[0]) * n1) + (nz2[0] - pz2[0]) * n2;
        t.y = ((nx2[1] - px2[1]) * n0 + (ny2[1] - py2[1]) * n1) + (nz2[1] - pz2[1]) * n2;
        if(i - 1 < jn) st2(T + toff0 + (i - 1) * S * TP, t); // a lane mask per term (scalar registers), one address register
        env.sched_fence(); // row by row: the scheduler must not square all seven rows first (42 more registers: spills)
#pragma unroll
        for(int c = 0; c < 2; c++)
        {
          px2[c] = nx2[c];
          py2[c] = ny2[c];
          pz2[c] = nz2[c];
        }
      }
    };
    // the column that slides in for step j + 1, fetched once step j has taken its squared differences: column c = 1 of
    // step j + 1 goes to ring slot (1 + j + 1) & 1 = PH, the slot column c = 0 of step j has just left.  Unconditional --
    // behind the last step of a row it fetches a column nobody uses (it exists: window column <= 2 reach + cw - 1) -- so
    // that the ring's registers are written on one path only (a conditional fetch made the register allocator keep both
    // versions of the ring: 19 spills); the head of a row is fetched at the top of the row, 15 exposed fetches per chunk.
    auto slide = [&](auto ph_tag, const int dy, const int dx) {
      constexpr int PH = decltype(ph_tag)::value;
      const int wx = wc0 + 1 + dx + 1;
      const int base = rowbase + dy * 2 * WPH + (wx & 1) * WPH + (wx >> 1);
#pragma unroll
      for(int i = 0; i <= MSEG; i++)
      {
        const int wi = base + i * RSTEP;
        const f2 v = ld2(XY + 2 * wi);
        sx[i][PH] = v.x;
        sy[i][PH] = v.y;
        sz[i][PH] = Z[wi];
      }
    };
    for(int dyi = 0; dyi < ndy; dyi++)
    {
      const int dy = patches[dyi * ndx].x, dx0 = patches[dyi * ndx].y;
      if(BORDER)
      {
#pragma unroll
        for(int i = 0; i <= MSEG; i++) row_ok[i] = own_row[i] && (unsigned)(r0 + wr0 + i * S + dy) < (unsigned)H;
      }
      if(!(var & 16)) load_row_head(dy, dx0);
      for(int jb = 0; jb < ndx; jb += 2)
      {
        if(!(var & 16))
        {
          step(std::integral_constant<int, 0>(), dyi * ndx + jb, dx0 + jb);
          slide(std::integral_constant<int, 0>(), dy, dx0 + jb);
        }
        env.sync();
        if(jb + 1 < ndx)
        {
          if(!(var & 16))
          {
            step(std::integral_constant<int, 1>(), dyi * ndx + jb + 1, dx0 + jb + 1);
            slide(std::integral_constant<int, 1>(), dy, dx0 + jb + 1);
          }
          env.sync();
        }
      }
    }
    env.sync();
    env.sync();
    if(!FUSED) env.sync();
    return;
  }
  // ---------------------------------------------------------------------------------------------------------------------
  if(w < 2)
  {
    // ---- A2: the column recurrence, one lane per table column, the whole column fetched at once
    const int x = 1 + w * 64 + lane;
    const bool active = x <= cw + 2 * P && !(var & 32);
    if(!(var & 512)) env.prio_high(); // the two recurrences are the stage's latency chains
    env.sync();
    for(int p = 0; p < n; p++)
    {
      if(active)
      {
        float *const col = tab + tslot(p) * tabsz + XO + x;
        const float *const F = Fb + (p & 1) * S * FP + XO + x;
        // all MAXCH rows, whatever the chunk's height: rows beyond it hold garbage that nobody reads (a guard per row
        // costs a lane mask per row: 55 pairs of scalar registers, spilled)
        float f[S], term[MAXCH];
#pragma unroll
        for(int r = 0; r < S; r++) f[r] = F[r * FP];
#pragma unroll
        for(int t = 1; t < MAXCH; t++) term[t] = col[t * TP];
        float v = 0.0f;
#pragma unroll
        for(int r = 0; r < S; r++) v += f[r];
        float *const wave_base = tab + tslot(p) * tabsz + XO + 1 + w * 64; // lane 0's column
        env.template st_addtid<0>(wave_base, lane, v);
        column_chain<1, MAXCH, TP * 4, Env, !FUSED && P == 2 && !CENTER>::run(env, wave_base, lane, v, term); // (where it measured a gain)
        if constexpr(TALL)
        {
          static_assert(!TALL || MAXCH == TALL_HEAD, "the exported sum is the one behind the head's last row");
          seeds_out[(size_t)p * TALL_SEED_PITCH + x] = v;
        }
      }
      env.sync();
    }
    env.sync();
    if(!FUSED) env.sync();
    return;
  }
  // ---------------------------------------------------------------------------------------------------------------------
  if(!FUSED && w == 2)
  {
    // ---- B: the sliding row sum (:405-415), one lane per table row: the row fetched as 19 x 16 bytes, the distortion of
    //      chunk column c stored at slot c + 1 (18 x 16 bytes)
    const bool active = lane < ch && !(var & 64);
    if(!(var & 512)) env.prio_high();
    env.sync();
    env.sync();
    for(int p = 0; p < n; p++)
    {
      if(active)
      {
        float *const rowp = tab + (p & 3) * tabsz + lane * TP;
        float cs[80]; // cs[x + XO] = slot x
#pragma unroll
        for(int b = 4; b < 80; b += 4)
        {
          const f4 v = ld4(rowp + b);
          cs[b] = v.x;
          cs[b + 1] = v.y;
          cs[b + 2] = v.z;
          cs[b + 3] = v.w;
        }
        cs[XO] = 0.0f; // slot 0: the column in front of every patch is never summed (init_column_sums())
        float distortion = 0.0f;
#pragma unroll
        for(int kk = 1; kk < S; kk++) distortion += cs[XO + kk]; // columns left - P .. left + P - 1
        float d[72];
#pragma unroll
        for(int b = 0; b < 72; b += 4)
        {
#pragma unroll
          for(int c = b; c < b + 4; c++)
          {
            distortion = distortion + (cs[XO + c + S] - cs[XO + c]);
            d[c] = distortion;
          }
          f4 v;
          v.x = d[b];
          v.y = d[b + 1];
          v.z = d[b + 2];
          v.w = d[b + 3];
          st4(rowp + 4 + b, v);
          if(!(var & 1024)) env.sched_fence(); // four sums, their store, the next four: the stores' transfer overlaps the chain
        }
      }
      env.sync();
    }
    env.sync();
    return;
  }
  // ---------------------------------------------------------------------------------------------------------------------
  {
    // ---- C: weights and accumulation (:416-436); center_weight < 0: w = 2^-(distortion * sharpness).
    // Software-pipelined by one offset: a stage FETCHES the distortions of its offset (ready since the barrier) and the
    // one pixel that slides into the ring, and ACCUMULATES the offset before it, whose operands arrived a stage ago --
    // nothing a stage computes with was fetched in that stage (the last offset of a row is the exception: it is
    // accumulated in its own stage, so that a new row starts with an empty ring).  Ring of NPXL + 1 pixels: pixel i of
    // step j of a row sits in slot (i + j) % (NPXL + 1); the slot the incoming pixel takes was last read two steps ago.
    // CENTER: the lane's own pixels take 3 NPXL more registers (128 is all a sixteen-wave workgroup has), so a stage ACCUMULATES the
    // offset before FIRST and fetches behind it: one set of distortions, a ring of NPXL pixels (the incoming pixel takes the slot the
    // accumulation has just left), the weights one pixel at a time; the fetches are still in flight across the barrier.
    constexpr int NR = CENTER ? NPXL : NPXL + 1;
    constexpr int DB = CENTER ? 1 : 2; // sets of distortions
    constexpr int AB = CENTER ? 1 : 3; // pixels abreast in accumulate() (CENTER: three abreast measured 29.2 against 28.3 ms at 100 MP)
    static_assert(CENTER || NR % 2 == 0, "the distortions alternate between two register sets with the ring's period");
    const int ci = c_index;
    // a wave holds 8 chunk rows x 8 lanes; its lanes 0-31 take the even rows, 32-63 the odd ones: the nine distortions a
    // lane reads sit 9 words apart within a row and 84 apart between rows, and four rows TWO apart put the 32 lanes of an
    // LDS pass on 32 different banks (four consecutive rows: 2-way conflicts on every read)
    static_assert(LPR == 8, "the lane layout of the C role");
    const int r = 8 * ci + 2 * ((lane & 31) >> 3) + (lane >> 5), j8 = lane & 7;
    const int cb = NPXL * j8; // first chunk column of the lane
    const bool active = r < ch && cb < cw && !(var & 128);
    float accx[NPXL], accy[NPXL], accz[NPXL], accw[NPXL];
    float qx[NR], qy[NR], qz[NR];
    float dist[DB][NPXL];
    unsigned reached[2] = { ~0u, ~0u }; // BORDER: bit i = the offset's shifted pixel of pixel i is in the frame
#pragma unroll
    for(int i = 0; i < NPXL; i++) accx[i] = accy[i] = accz[i] = accw[i] = dist[0][i] = dist[DB - 1][i] = 0.0f;
#pragma unroll
    for(int i = 0; i < NR; i++) qx[i] = qy[i] = qz[i] = 0.0f;
    const int doff = r * TP + 4 + cb;
    const float sharp_m23 = a.sharpness * -8388608.0f;
    // CENTER: the lane's own pixels (they never change) and the divisor 1 + center_weight with its refined reciprocal
    [[maybe_unused]] float ox[CENTER ? NPXL : 1], oy[CENTER ? NPXL : 1], oz[CENTER ? NPXL : 1];
    [[maybe_unused]] const float cden = 1.0f + a.center_weight;
    [[maybe_unused]] const float crcp = Env::rcp_refined(cden);
    if constexpr(CENTER)
    {
#pragma unroll
      for(int i = 0; i < NPXL; i++)
      {
        // (a lane without pixels -- beyond the chunk's rows or columns -- reads words of the window or the tables behind it)
        const int wi = widx(reach + r, reach + cb + i);
        const f2 v = ld2(XY + 2 * wi);
        ox[i] = v.x;
        oy[i] = v.y;
        oz[i] = Z[wi];
      }
    }
    // FUSED: the column sums of slots cb .. cb + NPXL + S - 1 (slot x at x + XO), fetched by every lane of the wave -- rows
    // beyond the chunk are table rows nobody wrote, and nobody uses what comes of them
    [[maybe_unused]] float cs[NPXL + S];
    [[maybe_unused]] const bool row_head = j8 == 0;
    [[maybe_unused]] auto fetch_sums = [&](const int p) {
      const float *const T = tab + tslot(p) * tabsz + r * TP + XO + cb;
#pragma unroll
      for(int i = 0; i < NPXL + S; i++) cs[i] = T[i];
    };
    // FUSED: the sliding row sum (:405-415) of the offset whose column sums are in cs[], into dist[M & (DB - 1)].  The chain of a
    // row runs through its eight lanes: phase k completes lane k, whose last sum is the carry of lane k + 1 (a DPP shift
    // by one lane; lane 0 of a row starts from the sum of the first 2 P columns); every lane recomputes its nine sums in
    // every phase from the carry it sees -- from its own phase on that is the final one.
    [[maybe_unused]] auto row_chain = [&](auto m_tag) {
      constexpr int M = decltype(m_tag)::value;
      float e[NPXL];
      // slot 0: the column in front of every patch is never summed (init_column_sums())
      e[0] = cs[S] - (row_head ? 0.0f : cs[0]);
#pragma unroll
      for(int i = 1; i < NPXL; i++) e[i] = cs[i + S] - cs[i];
      float first = 0.0f;
#pragma unroll
      for(int kk = 1; kk < S; kk++) first += cs[kk]; // columns left - P .. left + P - 1 (the row's lane 0 only)
      float carry = first;
#pragma unroll
      for(int ph = 0; ph < LPR; ph++)
      {
        float d = carry;
#pragma unroll
        for(int i = 0; i < NPXL; i++)
        {
          d = d + e[i];
          dist[M & (DB - 1)][i] = d;
        }
        if(ph + 1 < LPR)
        {
          const float from_left = env.lane_shr1(d);
          carry = row_head ? first : from_left;
        }
      }
    };

    auto fetch = [&](auto m_tag, const int p, const bool first, const int dy, const int dx) {
      constexpr int M = decltype(m_tag)::value;
      if constexpr(!FUSED)
      {
        const float *const T = tab + (p & 3) * tabsz + doff;
#pragma unroll
        for(int i = 0; i < NPXL; i++) dist[M & (DB - 1)][i] = T[i];
      }
      if(BORDER)
      {
        unsigned m = 0;
        if((unsigned)(top + r + dy) < (unsigned)H)
        {
#pragma unroll
          for(int i = 0; i < NPXL; i++) m |= ((unsigned)(left + cb + i + dx) < (unsigned)W ? 1u : 0u) << i;
        }
        reached[M & (DB - 1)] = m;
      }
      const int wy = reach + r + dy;
      if(first)
      {
#pragma unroll
        for(int i = 0; i < NPXL; i++)
        {
          const int wi = widx(wy, reach + cb + i + dx);
          const f2 v = ld2(XY + 2 * wi);
          qx[i] = v.x;
          qy[i] = v.y;
          qz[i] = Z[wi];
        }
      }
      else
      {
        constexpr int SL = (NPXL - 1 + M) % NR;
        const int wi = widx(wy, reach + cb + NPXL - 1 + dx);
        const f2 v = ld2(XY + 2 * wi);
        qx[SL] = v.x;
        qy[SL] = v.y;
        qz[SL] = Z[wi];
      }
    };
    // The weights and accumulations of an offset, THREE pixels abreast (round 5).  The compiler had emitted the nine pixels one
    // after the other through two scratch registers: 13 instructions a pixel, each waiting for the one before -- and a wave
    // issues a dependent instruction every ~8 cycles, an independent one every ~5 (profiles/r05_valu_issue_cycles.json): ~950
    // cycles of a ~2 300-cycle stage in every one of the role's waves.  Here each step of the weight (dt_fast_mexp2f(): product,
    // NaN guard, conversion, exponent add, range select) and each of the four accumulations is written for three pixels at a time
    // and fenced, so that three independent instructions follow each other; same operations on the same operands per pixel.
    auto accumulate = [&](auto m_tag) {
      constexpr int M = decltype(m_tag)::value;
      static_assert(NPXL % AB == 0, "accumulate() takes the pixels of a lane AB at a time");
#pragma unroll
      for(int g = 0; g < NPXL; g += AB)
      {
        float v[AB], wgt[AB], t[AB];
        int k0[AB];
        if constexpr(CENTER)
        {
          // pixel_difference(own, shifted, center_norm), :156-165: (diff * diff) * norm per channel, summed (x + y) + z; then
          // (distortion + that) / (1 + center_weight), x sharpness - 2, floored at 0 (:421-423), and dt_fast_mexp2f()'s product.
          // No NaN guard behind it: fmaxf() has turned a NaN into 0, and -inf converts to INT_MIN on either target
          float cx2[AB], cy2[AB], cz2[AB];
#pragma unroll
          for(int j = 0; j < AB; j++)
          {
            const float dx_ = ox[g + j] - qx[(g + j + M) % NR], dy_ = oy[g + j] - qy[(g + j + M) % NR], dz_ = oz[g + j] - qz[(g + j + M) % NR];
            cx2[j] = dx_ * dx_ * a.cpn;
            cy2[j] = dy_ * dy_ * a.cpn;
            cz2[j] = dz_ * dz_ * a.cpn;
          }
          env.sched_fence();
#pragma unroll
          for(int j = 0; j < AB; j++) v[j] = dist[M & (DB - 1)][g + j] + (cx2[j] + cy2[j] + cz2[j]);
          env.sched_fence();
#pragma unroll
          for(int j = 0; j < AB; j++) v[j] = Env::div_uniform(v[j], cden, crcp);
          env.sched_fence();
          // max(0, q s - 2) x -2^23 as min(0, q s' + 2^24) with s' = s x -2^23: scaling by a power of two is exact and commutes
          // with both roundings (the product's and the difference's: q s - 2 is never subnormal, a subnormal q s leaves -2 either
          // way, an overflow is -inf either way), max x negative = min, and a NaN gives 0 through either -- one product fewer
#pragma unroll
          for(int j = 0; j < AB; j++) v[j] = Env::min_num(0.0f, v[j] * sharp_m23 + 16777216.0f);
          env.sched_fence();
        }
        else
        {
#pragma unroll
          for(int j = 0; j < AB; j++) v[j] = dist[M & (DB - 1)][g + j] * sharp_m23;
          env.sched_fence();
#pragma unroll
          for(int j = 0; j < AB; j++) v[j] = Env::max_num(v[j], -__builtin_inff());
          env.sched_fence();
        }
#pragma unroll
        for(int j = 0; j < AB; j++) k0[j] = Env::cvt_i32_sat(v[j]);
        env.sched_fence();
#pragma unroll
        for(int j = 0; j < AB; j++) k0[j] = (int)(0x3f800000u + (unsigned)k0[j]);
        env.sched_fence();
#pragma unroll
        for(int j = 0; j < AB; j++)
        {
          wgt[j] = Env::int_as_float(k0[j] >= 0x800000 ? k0[j] : 0);
          if(BORDER && !(reached[M & (DB - 1)] >> (g + j) & 1u)) wgt[j] = 0.0f;
        }
        env.sched_fence();
#pragma unroll
        for(int j = 0; j < AB; j++) t[j] = qx[(g + j + M) % NR] * wgt[j];
        env.sched_fence();
#pragma unroll
        for(int j = 0; j < AB; j++) accx[g + j] = accx[g + j] + t[j];
#pragma unroll
        for(int j = 0; j < AB; j++) t[j] = qy[(g + j + M) % NR] * wgt[j];
        env.sched_fence();
#pragma unroll
        for(int j = 0; j < AB; j++) accy[g + j] = accy[g + j] + t[j];
#pragma unroll
        for(int j = 0; j < AB; j++) t[j] = qz[(g + j + M) % NR] * wgt[j];
        env.sched_fence();
#pragma unroll
        for(int j = 0; j < AB; j++) accz[g + j] = accz[g + j] + t[j];
#pragma unroll
        for(int j = 0; j < AB; j++) accw[g + j] = accw[g + j] + 1.0f * wgt[j];
        env.sched_fence();
      }
    };
    env.sync();
    env.sync();
    if(!FUSED) env.sync();
    for(int dyi = 0; dyi < ndy; dyi++)
    {
      const int dy = patches[dyi * ndx].x, dx0 = patches[dyi * ndx].y;
      for(int jb = 0; jb < ndx; jb += NR)
      {
        auto run = [&](auto m_tag) {
          constexpr int M = decltype(m_tag)::value;
          if constexpr(M < NR)
          {
            if(jb + M < ndx)
            {
              if constexpr(CENTER)
              {
                // the offset before first, then this offset's fetches behind it (one set of distortions, a ring of NPXL)
                if(active && jb + M > 0) accumulate(std::integral_constant<int, (M + NR - 1) % NR>());
                if constexpr(FUSED) fetch_sums(dyi * ndx + jb + M);
                if(active) fetch(m_tag, dyi * ndx + jb + M, jb + M == 0, dy, dx0 + jb + M);
                if constexpr(FUSED)
                {
                  if(!(var & 64)) row_chain(m_tag);
                }
                if(active && jb + M + 1 == ndx) accumulate(m_tag);
              }
              else if constexpr(FUSED)
              {
                // the offset before is accumulated FIRST, the sums and the pixel of this offset fetched behind it, then the row recurrence of
                // this offset (round 6: with the fetches in front, every C wave's reads queued at the LDS at the top of a stage ahead of the
                // column recurrence's, whose chain is the stage's length: 14.2 -> 14.0 ms at 60 MP, same box)
                if(active && jb + M > 0) accumulate(std::integral_constant<int, (M + NR - 1) % NR>());
                env.sched_fence();
                fetch_sums(dyi * ndx + jb + M);
                if(active) fetch(m_tag, dyi * ndx + jb + M, jb + M == 0, dy, dx0 + jb + M);
                if(!(var & 64)) row_chain(m_tag);
                if(active && jb + M + 1 == ndx) accumulate(m_tag);
              }
              else if(active)
              {
                // accumulate first, fetch behind it: the first LDS reads of a stage then belong to the two recurrences -- the stage's latency
                // chains -- and not to the seven C waves (round 6: 21.47 -> 21.23 ms at 100 MP, same box, three runs each)
                if(jb + M > 0) accumulate(std::integral_constant<int, (M + NR - 1) % NR>());
                env.sched_fence();
                fetch(m_tag, dyi * ndx + jb + M, jb + M == 0, dy, dx0 + jb + M);
                if(jb + M + 1 == ndx) accumulate(m_tag);
              }
              env.sync();
            }
          }
        };
        run(std::integral_constant<int, 0>());
        run(std::integral_constant<int, 1>());
        run(std::integral_constant<int, 2>());
        run(std::integral_constant<int, 3>());
        run(std::integral_constant<int, 4>());
        run(std::integral_constant<int, 5>());
        run(std::integral_constant<int, 6>());
        run(std::integral_constant<int, 7>());
        run(std::integral_constant<int, 8>());
        run(std::integral_constant<int, 9>());
        if constexpr(NR > 10)
        {
          run(std::integral_constant<int, 10>());
          run(std::integral_constant<int, 11>());
          run(std::integral_constant<int, 12>());
          run(std::integral_constant<int, 13>());
        }
      }
    }
    // ---- normalise, blend (:490-521)
    if(!active) return;
    const int row = top + r;
    if(row < a.out_row0 || row >= a.out_row1) return;
#pragma unroll
    for(int i = 0; i < NPXL; i++)
    {
      if(cb + i >= cw) continue;
      const long o = (long)row * W + left + cb + i;
      F4 res;
      if(a.skip_blend)
      {
        res.x = accx[i] / accw[i];
        res.y = accy[i] / accw[i];
        res.z = accz[i] / accw[i];
        res.w = accw[i] / accw[i];
      }
      else
      {
        const F4 ip = in[o];
        res.x = (ip.x * (1.0f - a.luma)) + (accx[i] / accw[i] * a.luma);
        res.y = (ip.y * (1.0f - a.chroma)) + (accy[i] / accw[i] * a.chroma);
        res.z = (ip.z * (1.0f - a.chroma)) + (accz[i] / accw[i] * a.chroma);
        res.w = (ip.w * 0.0f) + (accw[i] / accw[i] * 1.0f);
      }
      out[o] = res;
      env.store_cell(a, o, res.x);
    }
  }
}

} // namespace nlm3
