// amaze_stream_body.h -- AMaZE demosaic with every plane of a tile ON CHIP: the body of the gfx950 kernel
// amaze_stream (demosaic_amaze.hip), written over an environment so that the same source also runs on the host in
// the CPU suite (tests/native/amaze_host.cpp: one fiber per thread, every LDS access checked for races and for ring
// slots read after they were overwritten) against the oracle before a GPU is involved.
//
// Reference: amaze_demosaic_RT(), src/iop/demosaic/amaze.cc:181-1419 -- per 160 x 160 tile a fixed sequence of stencil
// stages over tile-sized planes (1.5 MB per tile; the first device kernel, amaze_tiles, keeps them in a slab of global
// memory and moves 25 x the algorithmic bytes).  Here a tile is walked top to bottom in steps of R rows.  Every stage
// runs once per step on ITS rows of the step -- stage X works on tile rows [s R - L_X, s R - L_X + R), L_X being the
// number of rows X trails the load by: what X reads below a row must have been produced -- and its planes live in LDS
// as rings of just the rows between their producer and their last consumer (all but 640 bytes of a CU's 160 KB, ring depths rounded up to powers of two).  A phase = the stages of a
// step that do not depend on each other, then a workgroup barrier.
//
// What makes the result the reference's, bit for bit (oracle/src/demosaic_amaze.c restates it):
//   * the arithmetic of every stage is the first kernel's, operation for operation;
//   * the three Gauss-Seidel sweeps keep their order.  The two votes (amaze.cc:894-905, :1109-1126) read row r-1 voted and
//     row r+1 unvoted: one row per sub-phase.  The choice between the two colour-difference estimates (:585-705) reads the
//     UPDATED neighbour two sites back, along the row for hcd and down the column for vcd.  Down the column is the
//     direction of the walk (two sites per lane and step, in order).  Along the row it would be a chain of 76 dependent
//     steps -- but a site's result is one of only TWO values whatever its neighbour was (the bounded raw estimate or the
//     bounded alternative), so every site first computes both candidates, then which it would pick for either candidate of
//     its neighbour (two bits), and then finds its own pick by walking back to the nearest site whose pick does not depend
//     on its neighbour, counting the sites that invert: no serial sweep;
//   * the Nyquist refinement (:763-956) runs over the bounding box of the flagged sites, and only if that box has an
//     extent.  Neither is a dependency of the whole tile: a site of the second flag plane can only be set with four
//     flagged neighbours out of eight, which lie in two rows and two columns at least -- so where the reference skips the
//     refinement, or a site lies outside its box, the unconditional vote yields 0 and everything behind it does nothing
//     (tools/amaze_alias_probe.py bit 64 shows the same on the oracle);
//   * of the reference's plane sharings (amaze.cc:300-327) ONE reaches the kept 128 x 128 pixels of a full tile: the second
//     flag plane lives in cddiffsq's bytes and is cleared / written for tile rows 4..155 only, so the area weights of rows
//     150 and 151 read, as flags of rows 156 and 157, the BYTES of the squared colour differences of tile row 19, columns
//     80..119 (profiles/r03_amaze_alias_probe.txt, bit 16).  Those 80 floats are kept aside when row 19 passes.
// Tiles the frame cuts (the last tile row / column) are shorter / narrower planes with mirrored strips; the two kinds of them
// that a top-to-bottom walk cannot reproduce (stream_tile_ok()) stay with the first kernel's body.
#pragma once
#include <stdint.h>
#include <math.h>

#ifndef AMZ_FN
#define AMZ_FN static inline
#define AMZ_MEMBER static inline
#define AMZ_HD static inline // ... called from the launch code as well
#endif

namespace amz
{

constexpr int TS = 160, TSH = 80;
constexpr int R = 4;    // rows per step
constexpr int NT = 640; // threads: one per site of a full-width stage
constexpr int STREAM_THREADS = NT;
constexpr float EPS = 1e-5f, EPSSQ = 1e-10f, ARTHRESH = 0.75f;

// rows a stage trails the load by
constexpr int L_LOAD = 0, L_S1 = 2, L_S2 = 4, L_S3V = 6, L_S4 = 9, L_S5 = 11, L_S6 = 13, L_INT = 19, L_VOTE = 20, L_S8 = 22,
              L_S13 = 25, L_S14 = 26;
constexpr int L_S9 = 17, L_RB = 19, L_RBI = 20; // the diagonal branch: gradients, R/B estimates, their vote and R+B
constexpr int L_DQ = 5; // the squared gradients: formed late in a step for S5 of the NEXT step (lag 11 - 2 rows - R)
constexpr int STEPS = (TS - 17 + L_S14) / R + 1; // the last kept row is 143

// ---- the planes: WIDTH words per row (floats, or bytes for the flag planes), DEPTH rows
//      DEPTH = R + (last consumer's lag + the rows it reads above its own) - (producer's lag)
template <int OFF_, int W_, int D_> struct plane
{
  static constexpr int OFF = OFF_, W = W_, D = D_, END = OFF_ + W_ * D_;
  // (rows are below 256: with that known, the remainder by a depth that is no power of two needs no 32-bit multiply-high)
  AMZ_MEMBER int idx(const int r, const int c) { return OFF + (int)(((unsigned)r & 255u) % (unsigned)D) * W + c; }
  // the same in steps, for the planes whose depth is no power of two: the slot of a row once, those of the rows around it by
  // an addition and a wrap (-D < k < D)
  AMZ_MEMBER int slot(const int r) { return (int)(((unsigned)r & 255u) % (unsigned)D); }
  AMZ_MEMBER int step(const int sl, const int k)
  {
    int t = sl + k;
    t += t < 0 ? D : 0;
    return t >= D ? t - D : t;
  }
  AMZ_MEMBER int at(const int sl, const int c) { return OFF + sl * W + c; }
};
// (a power of two wherever LDS allows: the slot of a row is then one AND; 163 200 of the CU's 163 840 bytes)
// float planes, full width
typedef plane<0, TS, 32> P_CFA;               // load; the area weights read 7 rows up at lag 19, the output at 26: R + 26
typedef plane<P_CFA::END, TS, 8> P_D0;        // S1 -> S2 (2 rows up): R + 4
typedef plane<P_D0::END, TS, 8> P_D1;
typedef plane<P_D1::END, TS, 8> P_DQ;         // at lag 5, behind S5 in its step -> S5 of the next step (2 up)
typedef plane<P_DQ::END, TS, 16> P_VCD;       // S2 -> S4 (3 up at lag 9): R + 8
typedef plane<P_VCD::END, TS, R> P_HCD;       // S2 -> the horizontal half of S4 in the same step
typedef plane<P_HCD::END, TS, 8> P_VCDALT;    // S2 -> S3 down the columns (2 up at lag 6)
typedef plane<P_VCDALT::END, TS, R> P_HCDALT;
typedef plane<P_HCDALT::END, TS, R> P_HR0;    // the two candidates of a site of the row chains
typedef plane<P_HR0::END, TS, R> P_HR1;
typedef plane<P_HR1::END, TS, 16> P_DGV;      // dgintv: S2 -> S4 (2 up): R + 7
typedef plane<P_DGV::END, TS, R> P_DGH;
// float planes, one word per R/B site (or per pair of columns)
typedef plane<P_DGH::END, TSH, R> P_HWT;      // S2 -> the horizontal half of S4
typedef plane<P_HWT::END, TSH, 16> P_VWT;     // S2 -> S4: R + 5
typedef plane<P_VWT::END, TSH, 16> P_HVAR;    // the horizontal variances of S4, formed at lag 4 for lag 9
typedef plane<P_HVAR::END, TSH, 16> P_HVAR1;
typedef plane<P_HVAR1::END, TSH, 8> P_CDD;    // cddiffsq at lag 9 -> S5 (2 up)
typedef plane<P_CDD::END, TSH, 32> P_HVWT;    // S4 -> the output (1 up at lag 26): R + 18
typedef plane<P_HVWT::END, TSH, 32> P_VCDH;   // vcd / hcd at the R/B sites, lag 9 / 4 -> the curvature refinement at 22
typedef plane<P_VCDH::END, TSH, 32> P_HCDH;
typedef plane<P_HCDH::END, TSH, 16> P_GREEN;  // green at the R/B sites, lag 20 -> 26
typedef plane<P_GREEN::END, TSH, 16> P_DGO;   // G - R at R sites, G - B at B sites: lag 20 -> S13 (3 up at 25): R + 8
typedef plane<P_DGO::END, TSH, 8> P_DGP;      // the other difference at those sites: S13 -> output (1 up)
typedef plane<P_DGP::END, TSH, 8> P_GH;       // dgrb2
typedef plane<P_GH::END, TSH, 8> P_GV;
typedef plane<P_GV::END, TSH, 8> P_DELP;      // S9 -> R/B estimates (2 up)
typedef plane<P_DELP::END, TSH, 8> P_DELM;
typedef plane<P_DELM::END, TSH, 8> P_DSQP;
typedef plane<P_DSQP::END, TSH, 8> P_DSQM;
static_assert(70 % (R + 1) == 0 && 70 % (R + 3) == 0, "the offset that keeps a stepped slot's first row non-negative");
typedef plane<P_DSQM::END, TSH, R + 1> P_RBP;
typedef plane<P_RBP::END, TSH, R + 1> P_RBM;
typedef plane<P_RBM::END, TSH, R + 3> P_PMWT;   // lag 19 -> vote at 20 (1 up) -> S11 at 22
typedef plane<P_PMWT::END, TSH, 8> P_RBINT;     // lag 20 -> S11 (2 up)
constexpr int CDD19 = P_RBINT::END; // the 80 floats behind the second flag plane's rows 156..159
constexpr int FLOATS_END = CDD19 + TSH;
// byte planes (offsets in bytes)
typedef plane<FLOATS_END * 4, TS, R> P_HB;      // the two picks of a site of the row chains
typedef plane<P_HB::END, TSH, 8> P_NY;          // S5 -> S6 (2 up)
typedef plane<P_NY::END, TSH, 16> P_NY2;        // S6 -> area weights (6 up at lag 19) -> refinement at 22
constexpr int LDS_BYTES = (P_NY2::END + 15) & ~15;
static_assert(LDS_BYTES <= 160 * 1024, "LDS of a gfx950 CU");

struct args
{
  int width, height;
  uint32_t filters;
  int ex, ey;
  float clip_pt;
  // a row band of the frame: `in` holds the mosaic from frame row in_row0 on, `out` the frame rows
  // [out_row0, out_row1) and nothing else is written.  The whole frame: 0, 0, height
  int in_row0, out_row0, out_row1;
  int variant; // 0; a measuring run switches the first (1) / the second (2) vote off with it -- the output is then wrong
};

AMZ_FN float sqr(const float x) { return x * x; }
AMZ_FN float fmin2(const float a, const float b) { return b < a ? b : a; } // std::min
AMZ_FN float fmax2(const float a, const float b) { return a < b ? b : a; } // std::max
AMZ_FN float lim(const float a, const float b, const float c) { return fmax2(b, fmin2(a, c)); }
AMZ_FN float ulim(const float a, const float b, const float c) { return (b < c) ? lim(a, b, c) : lim(a, c, b); }
AMZ_FN float intp(const float a, const float b, const float c) { return a * (b - c) + c; }
AMZ_FN uint32_t f2u(const float f)
{
  union { float f; uint32_t u; } x;
  x.f = f;
  return x.u;
}
AMZ_FN float u2f(const uint32_t u)
{
  union { float f; uint32_t u; } x;
  x.u = u;
  return x.f;
}
// xmul2f / xdiv2f / xdivf, amaze.cc:77-121: exponent arithmetic unless the value is +-0
AMZ_FN float expo(const float d, const int n)
{
  const uint32_t u = f2u(d);
  return (u & 0x7FFFFFFFu) ? u2f(u + ((uint32_t)n << 23)) : d;
}
AMZ_FN float xmul2f(const float d) { return expo(d, 1); }
AMZ_FN float xdiv2f(const float d) { return expo(d, -1); }
AMZ_FN float xdivf(const float d, const int n) { return expo(d, -n); }
// clampnan(), amaze.cc:61-75: only infinities are clamped (the NaN branch is shadowed)
AMZ_FN float clampnan(const float x, const float m, const float M)
{
  const bool finite = (f2u(x) & 0x7F800000u) != 0x7F800000u;
  return finite ? x : (x < m ? m : (x > M ? M : x));
}
AMZ_FN int fct(const int r, const int c, const uint32_t filters) { return filters >> ((((r << 1) & 14) + (c & 1)) << 1) & 3; }

// (float)((double)a * 2.0 / (double)b), amaze.cc:1150-1190.  A binary64 quotient of two binary32 numbers rounded to binary32 IS
// the correctly rounded binary32 quotient (53 >= 2 * 24 + 2: the second rounding never meets a tie the first one made), and
// doubling is exact -- so unless 2 a leaves binary32's range, or the quotient its normal range (where the two roundings act
// on different grids), one binary32 division gives the same bits
AMZ_FN float div2_via_double(const float a, const float b)
{
  const float a2 = a + a, q = a2 / b;
  const uint32_t e = f2u(q) & 0x7F800000u, ea = f2u(a2) & 0x7F800000u;
  // (a quotient of +-0 -- not NaN, so b is neither 0 nor NaN -- comes from a = +-0 and is +-0 either way)
  if((e != 0u || a2 == 0.f) && e != 0x7F800000u && ea != 0x7F800000u) return q;
  return (float)((double)a * 2.0 / (double)b);
}

// the colour-difference variance a site of a chain compares (amaze.cc:590-600)
AMZ_FN float cdvar3(const float a, const float b, const float c) { return 3.f * (sqr(a) + sqr(b) + sqr(c)) - sqr(a + b + c); }
// ... and what becomes of the estimate h it picks (amaze.cc:611-705): bounded where green would overshoot or clip.  The
// reference has one branch for a green site and its mirror image for a red / blue one -- every quantity of the second is the
// first's with the sign of h flipped, and negation commutes with every rounding (-(a) + -(b) = -(a + b), (-a) x = -(a x),
// 1 + (-q) = 1 - q) -- so the red / blue case is the green one on -h with the result negated, bit for bit: neighbouring
// lanes are sites of both kinds, and one body runs where two did
AMZ_FN float chain_bound(const float h_in, const float before, const float here, const float after, const bool gsite, const float clip_pt)
{
  const uint32_t flip = gsite ? 0u : 0x80000000u;
  float h = u2f(f2u(h_in) ^ flip);
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
  return u2f(f2u(h) ^ flip);
}

// The tiles this kernel takes: all but those whose mirrored right strip (16 columns from ccmax) or bottom strip (16 rows from
// rrmax) runs past the 160 x 160 plane -- the right one wraps into the next row, the bottom one into the flag bytes behind the
// plane in the reference's buffer, and flags nobody raised break the argument that the Nyquist refinement needs no tile-wide
// state (see the header).  Those tiles, and the one below, keep the first kernel's body.
AMZ_HD bool stream_tile_ok(const int width, const int height, const int top, const int left)
{
  const int bottom = top + TS < height + 16 ? top + TS : height + 16, right = left + TS < width + 16 ? left + TS : width + 16;
  const int rrmax = bottom > height ? height - top : bottom - top, ccmax = right > width ? width - left : right - left;
  if((rrmax < bottom - top && rrmax > TS - 16) || (ccmax < right - left && ccmax > TS - 16)) return false;
  // a narrower tile of odd width: the split of the two colour differences (amaze.cc:1239-1244) stops one R/B site short of what
  // the last kept column's chrominance reads, and that word of the G - B plane still holds the alternative colour difference
  // of a tile row further down (the planes share memory): not something a top-to-bottom walk has at hand
  return !(right - left < TS && ((right - left) & 1));
}

// One tile.  All NT threads of the workgroup call this with the same arguments.
template <typename Env> AMZ_FN void tile(Env &env, const float *in, float *out, const args &a, const int top, const int left)
{
  const int tid = env.tid();
  const int width = a.width;
  const uint32_t filters = a.filters;
  const float clip_pt = a.clip_pt, clip_pt8 = 0.8f * a.clip_pt;
  // the tile in the frame, amaze.cc:339-350: rr1 x cc1 photosites of which rows [rrmin, rrmax) and columns [ccmin, ccmax) lie in
  // the frame, the rest (16 at a frame edge) is mirrored
  const int height = a.height;
  const int bottom = top + TS < height + 16 ? top + TS : height + 16, right = left + TS < width + 16 ? left + TS : width + 16;
  const int rr1 = bottom - top, cc1 = right - left;
  const int rrmin = top < 0 ? 16 : 0, ccmin = left < 0 ? 16 : 0;
  const int rrmax = bottom > height ? height - top : rr1, ccmax = right > width ? width - left : cc1;
  const int steps = rr1 > 32 ? (rr1 - 17 + L_S14) / R + 1 : 0; // the last kept row is rr1 - 17

#define AMZ_UNROLL _Pragma("unroll")
#define LD(P, r, c) env.ldf(P::idx((r), (c)), (r))
#define ST(P, r, c, v) env.stf(P::idx((r), (c)), (r), (v))
#define LDB(P, r, c) env.ldb(P::idx((r), (c)), (r))
#define STB(P, r, c, v) env.stb(P::idx((r), (c)), (r), (v))
#define IN_(x, lo, hi) ((x) >= (lo) && (x) < (hi))
// every thread over the sites of the R rows of a stage with lag L: all columns ...
// (eight of the ten waves lie in ONE tile row -- 160 columns are two and a half waves -- and with the row known to be the same
// in every lane the ring slots of the rows a site touches, times the pitch, are scalar arithmetic instead of 30 % of the
// kernel's vector instructions: the body is compiled twice, for a row in a scalar register and for a row per lane)
#define FOR_FULL(L)                                                                                                  \
  AMZ_UNROLL                                                                                                         \
  for(int _u = 0; _u < 2; _u++)                                                                                      \
    if((_u == 0) == full_one_row)                                                                                    \
      for(int rr = s * R - (L) + (_u == 0 ? env.uniform(full_row) : full_row), cc = full_col, _once = 1;             \
          _once && rr >= 0 && rr < TS; _once = 0)
// ... the R/B sites (q: the column parity of the R/B sites of the row; h: the site's word in a half-width plane)
#define FOR_RB(L)                                                                                         \
  for(int _k = tid; _k < R * TSH; _k += NT)                                                               \
    for(int rr = s * R - (L) + _k / TSH, h = _k % TSH, q = rr >= 0 ? (fct(rr, 2, filters) & 1) : 0, cc = q + 2 * h, _once = 1; \
        _once && rr >= 0 && rr < TS; _once = 0)

// ... all columns, two sites each on the upper half of the threads (for a stage that shares its phase with an R/B stage)
#define FOR_FULL_UPPER(L)                                     \
  for(int _k = tid - NT / 2; _k >= 0 && _k < R * TS; _k += NT / 2) \
    for(int rr = s * R - (L) + _k / TS, cc = _k % TS, _once = 1; _once && rr >= 0 && rr < TS; _once = 0)
// ... the same on the second half of the threads (for a stage that shares its phase with another R/B stage)
#define FOR_RB2(L)                                                                                        \
  for(int _k = tid - NT / 2; _k >= 0 && _k < R * TSH; _k += NT)                                           \
    for(int rr = s * R - (L) + _k / TSH, h = _k % TSH, q = rr >= 0 ? (fct(rr, 2, filters) & 1) : 0, cc = q + 2 * h, _once = 1; \
        _once && rr >= 0 && rr < TS; _once = 0)

  static_assert(NT == R * TS, "one photosite of the step's rows per thread");
  const int full_row = tid / TS, full_col = tid % TS;
  const bool full_one_row = env.uniform((int)(((tid & ~63) / TS) == ((tid | 63) / TS))) != 0;
  // tile rows from the mosaic, amaze.cc:352-460: the nine fills of the reference as one function of the photosite -- they do
  // not overlap in the tiles this kernel takes (stream_tile_ok()).  A strip mirrors about the frame edge, a corner about
  // row / column 32 of the FRAME on the side of a top / left edge (the reference's own rule); what no fill reaches stays 0
  auto mosaic = [&](const int rr, const int cc) -> float {
    const bool r0 = rr < rrmin, r2 = rr >= rrmax, c0 = cc < ccmin, c2 = cc >= ccmax;
    if((r2 && rr >= rrmax + 16) || (c2 && cc >= ccmax + 16)) return 0.f;
    const int row = r0 ? ((c0 || c2) ? 32 - rr : 32 - rr + top) : (r2 ? height - (rr - rrmax) - 2 : rr + top);
    const int col = c0 ? ((r0 || r2) ? 32 - cc : 32 - cc + left) : (c2 ? width - (cc - ccmax) - 2 : cc + left);
    return in[(size_t)(row - a.in_row0) * width + col];
  };
  float pre = mosaic(tid / TS, tid % TS);
  // the tile starts from zeros, like the reference's buffer in the oracle
  for(int k = tid; k < LDS_BYTES / 4; k += NT) env.zero(k);
  env.sync();

  for(int s = 0; s < steps; s++)
  {
    // ---- phase 1: the step's tile rows
    {
      // (one photosite per thread and step, fetched a step ahead: the round trip to memory runs under the step before)
      const int rr = s * R + tid / TS, cc = tid % TS;
      if(rr < TS) ST(P_CFA, rr, cc, pre);
      if(rr + R < TS) pre = mosaic(rr + R, cc);
    }
    env.sync();
    env.stamp(1);

    // ---- phase 2: S1 directional gradients, :463-473
    FOR_FULL(L_S1)
    {
      float v0 = 0.f, v1 = 0.f;
      if(IN_(rr, 2, rr1 - 2) && IN_(cc, 2, cc1 - 2))
      {
        const float c0 = LD(P_CFA, rr, cc);
        const float delh = fabsf(LD(P_CFA, rr, cc + 1) - LD(P_CFA, rr, cc - 1));
        const float delv = fabsf(LD(P_CFA, rr + 1, cc) - LD(P_CFA, rr - 1, cc));
        v0 = EPS + fabsf(LD(P_CFA, rr + 2, cc) - c0) + fabsf(c0 - LD(P_CFA, rr - 2, cc)) + delv;
        v1 = EPS + fabsf(LD(P_CFA, rr, cc + 2) - c0) + fabsf(c0 - LD(P_CFA, rr, cc - 2)) + delh;
      }
      ST(P_D0, rr, cc, v0);
      ST(P_D1, rr, cc, v1);
    }
    env.sync();
    env.stamp(2);

    // ---- phase 3: S2 colour differences by adaptive ratios and by Hamilton-Adams, :478-582
    FOR_FULL(L_S2)
    {
      float v = 0.f, hd = 0.f, va = 0.f, ha = 0.f, gv = 0.f, gh = 0.f;
      if(IN_(rr, 4, rr1 - 4) && IN_(cc, 4, cc1 - 4))
      {
        const bool gsite = fct(rr, cc, filters) & 1;
        const float c = LD(P_CFA, rr, cc);
        const float cu1 = LD(P_CFA, rr - 1, cc), cu2 = LD(P_CFA, rr - 2, cc), cd1 = LD(P_CFA, rr + 1, cc), cd2 = LD(P_CFA, rr + 2, cc);
        const float cl1 = LD(P_CFA, rr, cc - 1), cl2 = LD(P_CFA, rr, cc - 2), cr1 = LD(P_CFA, rr, cc + 1), cr2 = LD(P_CFA, rr, cc + 2);
        const float d0c = LD(P_D0, rr, cc), d0u2 = LD(P_D0, rr - 2, cc), d0d2 = LD(P_D0, rr + 2, cc);
        const float d0u1 = LD(P_D0, rr - 1, cc), d0d1 = LD(P_D0, rr + 1, cc);
        const float d1c = LD(P_D1, rr, cc), d1l2 = LD(P_D1, rr, cc - 2), d1r2 = LD(P_D1, rr, cc + 2);
        const float d1l1 = LD(P_D1, rr, cc - 1), d1r1 = LD(P_D1, rr, cc + 1);
        const float cru = cu1 * (d0u2 + d0c) / (d0u2 * (EPS + c) + d0c * (EPS + cu2));
        const float crd = cd1 * (d0d2 + d0c) / (d0d2 * (EPS + c) + d0c * (EPS + cd2));
        const float crl = cl1 * (d1l2 + d1c) / (d1l2 * (EPS + c) + d1c * (EPS + cl2));
        const float crr = cr1 * (d1r2 + d1c) / (d1r2 * (EPS + c) + d1c * (EPS + cr2));
        const float guha = cu1 + xdiv2f(c - cu2);
        const float gdha = cd1 + xdiv2f(c - cd2);
        const float glha = cl1 + xdiv2f(c - cl2);
        const float grha = cr1 + xdiv2f(c - cr2);
        float guar = fabsf(1.f - cru) < ARTHRESH ? c * cru : guha;
        float gdar = fabsf(1.f - crd) < ARTHRESH ? c * crd : gdha;
        float glar = fabsf(1.f - crl) < ARTHRESH ? c * crl : glha;
        float grar = fabsf(1.f - crr) < ARTHRESH ? c * crr : grha;
        const float hwt = d1l1 / (d1l1 + d1r1);
        const float vwt = d0u1 / (d0d1 + d0u1);
        const float Gintvha = vwt * gdha + (1.f - vwt) * guha;
        const float Ginthha = hwt * grha + (1.f - hwt) * glha;
        // (site - estimate at a green site, estimate - site at a red / blue one: x - y = -(y - x) exactly, one body for both)
        v = c - (vwt * gdar + (1.f - vwt) * guar);
        hd = c - (hwt * grar + (1.f - hwt) * glar);
        va = c - Gintvha;
        ha = c - Ginthha;
        if(!gsite)
        {
          v = -v;
          hd = -hd;
          va = -va;
          ha = -ha;
          // S4 forms the same two quotients at the R/B sites
          ST(P_HWT, rr, cc >> 1, hwt);
          ST(P_VWT, rr, cc >> 1, vwt);
        }
        if(c > clip_pt8 || Gintvha > clip_pt8 || Ginthha > clip_pt8)
        {
          guar = guha;
          gdar = gdha;
          glar = glha;
          grar = grha;
          v = va;
          hd = ha;
        }
        gv = fmin2(sqr(guha - gdha), sqr(guar - gdar));
        gh = fmin2(sqr(glha - grha), sqr(glar - grar));
      }
      ST(P_VCD, rr, cc, v);
      ST(P_HCD, rr, cc, hd);
      ST(P_VCDALT, rr, cc, va);
      ST(P_HCDALT, rr, cc, ha);
      ST(P_DGV, rr, cc, gv);
      ST(P_DGH, rr, cc, gh);
    }
    env.sync();
    env.stamp(3);

    // ---- phase 4: S3 (:585-705) down the columns -- a lane of the lower half of the threads per column and row parity, its two rows of
    //      the step in order -- and, on the upper half, the two candidates of every site of the row chains (two sites each)
    for(int _k = tid; _k < 2 * TS; _k += NT)
    {
      const int cc = _k % TS, par = _k / TS;
      if(!IN_(cc, 4, cc1 - 4)) continue;
      for(int j = 0; j < R / 2; j++)
      {
        const int rr = s * R - L_S3V + par + 2 * j;
        if(!IN_(rr, 4, rr1 - 4)) continue;
        const bool gsite = fct(rr, cc, filters) & 1;
        const float prev = LD(P_VCD, rr - 2, cc), c0 = LD(P_VCD, rr, cc), c1 = LD(P_VCD, rr + 2, cc);
        const float a0 = LD(P_VCDALT, rr - 2, cc), a1 = LD(P_VCDALT, rr, cc), a2 = LD(P_VCDALT, rr + 2, cc);
        const float hpick = cdvar3(a0, a1, a2) < cdvar3(prev, c0, c1) ? a1 : c0;
        ST(P_VCD, rr, cc, chain_bound(hpick, LD(P_CFA, rr - 1, cc), LD(P_CFA, rr, cc), LD(P_CFA, rr + 1, cc), gsite, clip_pt));
      }
    }
    FOR_FULL_UPPER(L_S2)
    {
      if(IN_(rr, 4, rr1 - 4) && IN_(cc, 4, cc1 - 4))
      {
        const bool gsite = fct(rr, cc, filters) & 1;
        const float before = LD(P_CFA, rr, cc - 1), here = LD(P_CFA, rr, cc), after = LD(P_CFA, rr, cc + 1);
        ST(P_HR0, rr, cc, chain_bound(LD(P_HCD, rr, cc), before, here, after, gsite, clip_pt));
        ST(P_HR1, rr, cc, chain_bound(LD(P_HCDALT, rr, cc), before, here, after, gsite, clip_pt));
      }
    }
    env.sync();
    env.stamp(4);

    // ---- phase 5: S4 the H/V weight from colour-difference variances, :707-760: what a site reads down its column (what it reads along
    //      its row was formed five rows ago, below), with it the squared difference of the two estimates at the R/B sites, :703.
    //      Upper half: which candidate a site of the row chains picks, for either candidate of its neighbour two columns back
    FOR_RB(L_S4)
    {
      const float v0 = LD(P_VCD, rr, cc), h0 = LD(P_HCDH, rr, h);
      const float cd = (IN_(rr, 4, rr1 - 4) && IN_(cc, 4, cc1 - 4)) ? sqr(v0 - h0) : 0.f;
      ST(P_CDD, rr, h, cd);
      ST(P_VCDH, rr, h, v0);
      if(rr == 19 && cc >= TSH) env.stf(CDD19 + cc - TSH, -1, cd);
      float w = 0.f;
      if(IN_(rr, 6, rr1 - 6) && IN_(cc, 6, cc1 - 6))
      {
        const float v1 = LD(P_VCD, rr - 1, cc), v2 = LD(P_VCD, rr - 2, cc), v3 = LD(P_VCD, rr - 3, cc);
        const float w1 = LD(P_VCD, rr + 1, cc), w2 = LD(P_VCD, rr + 2, cc), w3 = LD(P_VCD, rr + 3, cc);
        const float uave = v0 + v1 + v2 + v3;
        const float dave = v0 + w1 + w2 + w3;
        float vu = sqr(v0 - uave) + sqr(v1 - uave) + sqr(v2 - uave) + sqr(v3 - uave);
        float vd = sqr(v0 - dave) + sqr(w1 - dave) + sqr(w2 - dave) + sqr(w3 - dave);
        const float vwt = LD(P_VWT, rr, h);
        const float vcdvar = EPSSQ + vwt * vd + (1.f - vwt) * vu;
        const float hcdvar = LD(P_HVAR, rr, h);
        const float g0 = LD(P_DGV, rr, cc);
        vu = (g0) + (LD(P_DGV, rr - 1, cc)) + (LD(P_DGV, rr - 2, cc));
        vd = (g0) + (LD(P_DGV, rr + 1, cc)) + (LD(P_DGV, rr + 2, cc));
        const float vcdvar1 = EPSSQ + vwt * vd + (1.f - vwt) * vu;
        const float hcdvar1 = LD(P_HVAR1, rr, h);
        const float varwt = hcdvar / (vcdvar + hcdvar);
        const float diffwt = hcdvar1 / (vcdvar1 + hcdvar1);
        // the reference forms (0.5 - varwt) * (0.5 - diffwt) in binary64 (0.5 is a double literal there) and asks for > 0:
        // either factor is 0 or at least 2^-25 in size, the product cannot underflow -- both below 0.5 or both above
        if(((varwt < 0.5f && diffwt < 0.5f) || (varwt > 0.5f && diffwt > 0.5f)) && fabsf(0.5f - diffwt) < fabsf(0.5f - varwt))
          w = varwt;
        else
          w = diffwt;
      }
      ST(P_HVWT, rr, h, w);
    }
    FOR_FULL_UPPER(L_S2)
    {
      if(IN_(rr, 4, rr1 - 4) && IN_(cc, 4, cc1 - 4))
      {
        const float c0 = LD(P_HCD, rr, cc), c1 = LD(P_HCD, rr, cc + 2);
        const float altvar = cdvar3(LD(P_HCDALT, rr, cc - 2), LD(P_HCDALT, rr, cc), LD(P_HCDALT, rr, cc + 2));
        float p0, p1;
        if(cc - 2 >= 4)
        {
          p0 = LD(P_HR0, rr, cc - 2);
          p1 = LD(P_HR1, rr, cc - 2);
        }
        else
          p0 = p1 = LD(P_HCD, rr, cc - 2); // in front of the first site of the chain: never written
        const unsigned bits = (altvar < cdvar3(p0, c0, c1) ? 1u : 0u) | (altvar < cdvar3(p1, c0, c1) ? 2u : 0u);
        STB(P_HB, rr, cc, (unsigned char)bits);
      }
    }
    env.sync();
    env.stamp(5);

    // ---- phase 6: S5 Nyquist texture test, :763-820.  Upper half: the pick of a site of the row chains: back to the nearest site that
    //      picks the same whatever came before it (0 or 3), inverted once per site on the way whose pick is the opposite of its
    //      neighbour's (1; 2 copies it)
    FOR_RB(L_S5)
    {
      unsigned char flag = 0;
      if(IN_(rr, 6, rr1 - 6) && IN_(cc, 6, cc1 - 6))
      {
        const float gg0 = 0.5f * 0.07384411893421103f, gg1 = 0.5f * 0.06207511968171489f, gg2 = 0.5f * 0.0521818194747806f;
        const float gg3 = 0.5f * 0.03687419286733595f, gg4 = 0.5f * 0.03099732204057846f, gg5 = 0.5f * 0.018413194161458882f;
        const float go0 = 0.14659727707323927f, go1 = 0.103592713382435f, go2 = 0.0732036125103057f, go3 = 0.0365543548389495f;
#define CD(dr, dc) LD(P_CDD, rr + (dr), (cc + (dc)) >> 1)
#define DQ(dr, dc) LD(P_DQ, rr + (dr), cc + (dc))
        const float test
            = (go0 * CD(0, 0) + go1 * (CD(-1, -1) + CD(-1, 1) + CD(1, -1) + CD(1, 1))
               + go2 * (CD(-2, 0) + CD(0, -2) + CD(0, 2) + CD(2, 0)) + go3 * (CD(-2, -2) + CD(-2, 2) + CD(2, -2) + CD(2, 2)))
              - (gg0 * DQ(0, 0) + gg1 * (DQ(-1, 0) + DQ(0, 1) + DQ(0, -1) + DQ(1, 0))
                 + gg2 * (DQ(-1, -1) + DQ(-1, 1) + DQ(1, -1) + DQ(1, 1))
                 + gg3 * (DQ(-2, 0) + DQ(0, -2) + DQ(0, 2) + DQ(2, 0))
                 + gg4 * (DQ(-2, -1) + DQ(-2, 1) + DQ(-1, -2) + DQ(-1, 2) + DQ(1, -2) + DQ(1, 2) + DQ(2, -1) + DQ(2, 1))
                 + gg5 * (DQ(-2, -2) + DQ(-2, 2) + DQ(2, -2) + DQ(2, 2)));
#undef CD
#undef DQ
        flag = test > 0.f ? 1 : 0;
      }
      STB(P_NY, rr, h, flag);
    }
    FOR_FULL_UPPER(L_S2)
    {
      if(IN_(rr, 4, rr1 - 4) && IN_(cc, 4, cc1 - 4))
      {
        unsigned inv = 0, bits;
        for(int j = cc;; j -= 2)
        {
          bits = LDB(P_HB, rr, j);
          if(bits == 0u || bits == 3u) break;
          inv ^= (bits == 1u) ? 1u : 0u;
        }
        const unsigned pick = (bits & 1u) ^ inv;
        ST(P_HCD, rr, cc, pick ? LD(P_HR1, rr, cc) : LD(P_HR0, rr, cc));
      }
    }
    env.sync();
    env.stamp(6);

    // ---- phase 7: S6 majority vote on the flags, :832-845 (rows 156..159 of the second flag plane are bytes of the squared colour
    //      differences of row 19, see the header); S9 diagonal gradients and squared differences, :958-983.  Upper half: the
    //      half of S4 that reads along the row, on the rows whose chains phase 6 just resolved
    FOR_RB(L_S6)
    {
      unsigned char f2 = 0;
      if(IN_(rr, 8, rr1 - 8) && IN_(cc, 8, cc1 - 8))
      {
#define NY(dr, dc) LDB(P_NY, rr + (dr), (cc + (dc)) >> 1)
        const unsigned n = NY(-2, 0) + NY(-1, -1) + NY(-1, 1) + NY(0, -2) + NY(0, 2) + NY(1, -1) + NY(1, 1) + NY(2, 0);
        f2 = n > 4 ? 1 : (n < 4 ? 0 : NY(0, 0));
#undef NY
      }
      else if(rr >= TS - 4)
        f2 = (unsigned char)(f2u(env.ldf(CDD19 + (rr - (TS - 4)) * 20 + (h >> 2), -1)) >> (8 * (h & 3)));
      STB(P_NY2, rr, h, f2);
    }
    for(int _k = tid; _k < R * TSH; _k += NT)
    {
      const int rr = s * R - L_S9 + _k / TSH, hh = _k % TSH;
      if(rr < 0 || rr >= TS) continue;
      float vp = 0.f, vm = 0.f, sp = 0.f, sm = 0.f;
      const int c2 = 2 * hh; // the even column of the pair
      if(IN_(rr, 6, rr1 - 6) && IN_(c2, 6, cc1 - 6))
      {
        const bool odd = fct(rr, 2, filters) & 1;
        const int ga = odd ? c2 + 1 : c2, sb = odd ? c2 : c2 + 1; // gradients at the green site of the pair, squares at the other
        vp = fabsf(LD(P_CFA, rr - 1, ga + 1) - LD(P_CFA, rr + 1, ga - 1));
        vm = fabsf(LD(P_CFA, rr + 1, ga + 1) - LD(P_CFA, rr - 1, ga - 1));
        const float b = LD(P_CFA, rr, sb);
        sp = (sqr(b - LD(P_CFA, rr + 1, sb - 1)) + sqr(b - LD(P_CFA, rr - 1, sb + 1)));
        sm = (sqr(b - LD(P_CFA, rr - 1, sb - 1)) + sqr(b - LD(P_CFA, rr + 1, sb + 1)));
      }
      ST(P_DELP, rr, hh, vp);
      ST(P_DELM, rr, hh, vm);
      ST(P_DSQP, rr, hh, sp);
      ST(P_DSQM, rr, hh, sm);
    }
    for(int _k = tid - NT / 2; _k >= 0 && _k < R * TSH; _k += NT)
    {
      const int rr = s * R - L_S2 + _k / TSH, h = _k % TSH;
      if(rr < 0 || rr >= TS) continue;
      const int cc = (fct(rr, 2, filters) & 1) + 2 * h;
      const float h0 = LD(P_HCD, rr, cc);
      ST(P_HCDH, rr, h, h0);
      if(IN_(rr, 6, rr1 - 6) && IN_(cc, 6, cc1 - 6))
      {
        const float l1 = LD(P_HCD, rr, cc - 1), l2 = LD(P_HCD, rr, cc - 2), l3 = LD(P_HCD, rr, cc - 3);
        const float r1 = LD(P_HCD, rr, cc + 1), r2 = LD(P_HCD, rr, cc + 2), r3 = LD(P_HCD, rr, cc + 3);
        const float lave = h0 + l1 + l2 + l3;
        const float rave = h0 + r1 + r2 + r3;
        float hl = sqr(h0 - lave) + sqr(l1 - lave) + sqr(l2 - lave) + sqr(l3 - lave);
        float hr = sqr(h0 - rave) + sqr(r1 - rave) + sqr(r2 - rave) + sqr(r3 - rave);
        const float hwt = LD(P_HWT, rr, h);
        ST(P_HVAR, rr, h, EPSSQ + hwt * hr + (1.f - hwt) * hl);
        const float e0 = LD(P_DGH, rr, cc);
        hl = (e0) + (LD(P_DGH, rr, cc - 1)) + (LD(P_DGH, rr, cc - 2));
        hr = (e0) + (LD(P_DGH, rr, cc + 1)) + (LD(P_DGH, rr, cc + 2));
        ST(P_HVAR1, rr, h, EPSSQ + hwt * hr + (1.f - hwt) * hl);
      }
    }
    env.sync();
    env.stamp(7);

    // ---- phase 8: area interpolation of the weight in flagged regions, :850-890 (the lower half of the threads, a site each: few sites
    //      are flagged); then, on all of them, the diagonal R/B estimates, :986-1107
    FOR_RB(L_INT)
    {
      if(IN_(rr, 8, rr1 - 8) && IN_(cc, 8, cc1 - 8) && LDB(P_NY2, rr, h))
      {
        float sumcfa = 0.f, sumh = 0.f, sumv = 0.f, sumsqh = 0.f, sumsqv = 0.f, areawt = 0.f;
        for(int p = -6; p < 7; p += 2)
          for(int qq = -6; qq < 7; qq += 2)
            if(LDB(P_NY2, rr + p, (cc + qq) >> 1))
            {
              const float c = LD(P_CFA, rr + p, cc + qq);
              const float l = LD(P_CFA, rr + p, cc + qq - 1), r = LD(P_CFA, rr + p, cc + qq + 1);
              const float u = LD(P_CFA, rr + p - 1, cc + qq), d = LD(P_CFA, rr + p + 1, cc + qq);
              sumcfa += c;
              sumh += (l + r);
              sumv += (u + d);
              sumsqh += sqr(c - l) + sqr(c - r);
              sumsqv += sqr(c - u) + sqr(c - d);
              areawt += 1;
            }
        sumh = sumcfa - xdiv2f(sumh);
        sumv = sumcfa - xdiv2f(sumv);
        areawt = xdiv2f(areawt);
        const float hcdvar = EPSSQ + fabsf(areawt * sumsqh - sumh * sumh);
        const float vcdvar = EPSSQ + fabsf(areawt * sumsqv - sumv * sumv);
        ST(P_HVWT, rr, h, hcdvar / (vcdvar + hcdvar));
      }
    }
    // the diagonal estimates, two lanes a site: lane 2 k takes site k's nw - se diagonal ("m"), lane 2 k + 1 its ne - sw one ("p") --
    // the reference's two halves (:986-1107) are one expression on different neighbours and planes -- and they trade the
    // two variances for the weight by a lane swap.  a: the neighbour towards se / ne, b: towards nw / sw
    {
      const int _k = tid >> 1, half = tid & 1;
      const int rr = s * R - L_RB + _k / TSH, h = _k % TSH;
      const int q = rr >= 0 ? (fct(rr, 2, filters) & 1) : 0, cc = q + 2 * h;
      if(rr >= 0 && rr < TS)
      {
        float pw = 0.f, v = 0.f;
        if(IN_(rr, 8, rr1 - 8) && IN_(cc, 8, cc1 - 8))
        {
          const float ge0 = 0.13719494435797422f, ge1 = 0.05640252782101291f;
          const int dra = half ? -1 : 1; // a: one row down (se) or up (ne), one column right; b: the opposite site
          static_assert(P_DELP::W == P_DELM::W && P_DELP::D == P_DELM::D && P_DSQP::W == P_DSQM::W && P_DSQP::D == P_DSQM::D,
                        "one index for the planes of both diagonals");
          const int pdel = half ? P_DELP::OFF - P_DELM::OFF : 0, pdsq = half ? P_DSQP::OFF - P_DSQM::OFF : 0;
          const float c = LD(P_CFA, rr, cc);
          const float ca1 = LD(P_CFA, rr + dra, cc + 1), ca2 = LD(P_CFA, rr + 2 * dra, cc + 2);
          const float cb1 = LD(P_CFA, rr - dra, cc - 1), cb2 = LD(P_CFA, rr - 2 * dra, cc - 2);
          const float cra = xmul2f(ca1) / (EPS + c + (ca2));
          const float crb = xmul2f(cb1) / (EPS + c + (cb2));
          const float rba = fabsf(1.f - cra) < ARTHRESH ? c * cra : (ca1) + xdiv2f(c - ca2);
          const float rbb = fabsf(1.f - crb) < ARTHRESH ? c * crb : (cb1) + xdiv2f(c - cb2);
#define HD(dr, dc) env.ldf(P_DELM::idx(rr + (dr), (cc + (dc)) >> 1) + pdel, rr + (dr))
#define HS(dr, dc) env.ldf(P_DSQM::idx(rr + (dr), (cc + (dc)) >> 1) + pdsq, rr + (dr))
          const float d0 = HD(0, 0);
          const float wta = EPS + d0 + HD(dra, 1) + HD(2 * dra, 2);
          const float wtb = EPS + d0 + HD(-dra, -1) + HD(-2 * dra, -2);
          v = (wta * rbb + wtb * rba) / (wta + wtb);
          const float rbvar
              = EPSSQ + (ge0 * (HS(-1, 0) + HS(0, -1) + HS(0, 1) + HS(1, 0))
                         + ge1 * (HS(-2, -1) + HS(-2, 1) + HS(-1, -2) + HS(-1, 2) + HS(1, -2) + HS(1, 2) + HS(2, -1) + HS(2, 1)));
#undef HD
#undef HS
          const float rbvar_other = env.swap1(rbvar);
          // (the m lane's: rbvarm / (rbvarp + rbvarm); the p lane's quotient is not used)
          pw = rbvar / (rbvar_other + rbvar);
          if(v < c)
          {
            if(xmul2f(v) < c)
              v = ulim(v, cb1, ca1);
            else
            {
              const float wt = xmul2f(c - v) / (EPS + v + c);
              v = wt * v + (1.f - wt) * ulim(v, cb1, ca1);
            }
          }
          if(v > clip_pt) v = ulim(v, cb1, ca1);
        }
        {
          // (three planes whose depth is no power of two: the slot of the step's first row on the scalar unit, this thread's row
          // from there; + 70, a multiple of both depths, keeps the row of the first steps non-negative)
          const int jr = rr - (s * R - L_RB);
          if(!half) env.stf(P_PMWT::at(P_PMWT::step(P_PMWT::slot(s * R - L_RB + 70), jr), h), rr, pw);
          const int sr = P_RBP::step(P_RBP::slot(s * R - L_RB + 70), jr);
          static_assert(P_RBP::W == P_RBM::W && P_RBP::D == P_RBM::D, "one index for both estimates");
          env.stf(P_RBM::at(sr, h) + (half ? P_RBP::OFF - P_RBM::OFF : 0), rr, v);
        }
      }
    }
    env.sync();
    env.stamp(8);

    // ---- phase 9: the two votes, in place, row r sees row r-1 voted (:894-905, :1109-1126): ONE wave each walks its R rows in order, two
    //      sites per lane, in registers.  The other waves meanwhile form the squared gradients S5 reads in the next step, :463-473
    if(tid < 64 && !(a.variant & 1))
    {
      // lane l of the wave owns the sites 2 l and 2 l + 1 of every row.  Everything that does not depend on a vote -- the
      // unvoted values of the R rows and of the row below them, the row above as the step before left it -- is fetched at once,
      // by every lane and for every row (from a neighbouring row or site where one is not there: those values are not used),
      // so that the fetches are one straight stretch of code; then the rows are walked in registers, the one voted value per
      // row that belongs to the neighbouring lane (lane - 1 or lane + 1, by the row's CFA phase) arriving by a lane shuffle
      // instead of a round trip through LDS
      const int lane = tid & 63, ha = lane < TSH / 2 ? 2 * lane : TSH - 2, hb = ha + 1;
      const bool own = lane < TSH / 2;
      const int r0 = s * R - L_VOTE;
      // the rows fetched: r0 - 1 .. r0 + R, moved into the plane where the tile begins / ends (rows that are voted on are not moved)
      const int rb = r0 < 1 ? 1 : (r0 > TS - R - 2 ? TS - R - 2 : r0), sw = P_HVWT::slot(rb);
      float raw[R][2], below[R][3];
      float pa = env.ldf(P_HVWT::at(P_HVWT::step(sw, -1), ha), -1), pb = env.ldf(P_HVWT::at(P_HVWT::step(sw, -1), hb), -1);
      AMZ_UNROLL
      for(int j = 0; j < R; j++)
      {
        const int q = fct(rb + j, 2, filters) & 1, sj = P_HVWT::step(sw, j), sd = P_HVWT::step(sw, j + 1);
        // the three sites of the rows above / below that this lane's two sites see: ha + q - 1, ha + q, ha + q + 1
        const int n0 = ha + q - 1 < 0 ? 0 : ha + q - 1, n1 = ha + q, n2 = ha + q + 1 > TSH - 1 ? TSH - 1 : ha + q + 1;
        raw[j][0] = env.ldf(P_HVWT::at(sj, ha), -1);
        raw[j][1] = env.ldf(P_HVWT::at(sj, hb), -1);
        below[j][0] = env.ldf(P_HVWT::at(sd, n0), -1);
        below[j][1] = env.ldf(P_HVWT::at(sd, n1), -1);
        below[j][2] = env.ldf(P_HVWT::at(sd, n2), -1);
      }
      AMZ_UNROLL
      for(int j = 0; j < R; j++)
      {
        const int rr = r0 + j;
        const int q = (rr >= 0 ? fct(rr, 2, filters) : 0) & 1, ca = q + 2 * ha, cb = ca + 2;
        // the voted row above at ha + q - 1, ha + q, ha + q + 1: with q = 1 own, own, lane + 1's first; with q = 0 lane - 1's
        // second, own, own
        const float from_next = env.shfl_down1(pa), from_prev = env.shfl_up1(pb);
        const float a0 = q ? pa : from_prev, a1 = q ? pb : pa, a2 = q ? from_next : pb;
        // (env.add(): x + y with the operands in this order -- where both are NaNs the sum is the first one's, on x86 and on the GPU
        // alike, and the exponent arithmetic of xdivf() carries its sign into numbers)
        const float alta = xdivf(env.add(env.add(env.add(a0, a1), below[j][0]), below[j][1]), 2);
        const float altb = xdivf(env.add(env.add(env.add(a1, a2), below[j][1]), below[j][2]), 2);
        const bool row_in = own && IN_(rr, 8, rr1 - 8); // (then rb == r0: the fetched row j is row rr)
        const bool ina = row_in && IN_(ca, 8, cc1 - 8), inb = row_in && IN_(cb, 8, cc1 - 8);
        float wa = raw[j][0], wb = raw[j][1];
        const bool ta = ina && fabsf(0.5f - wa) < fabsf(0.5f - alta), tb = inb && fabsf(0.5f - wb) < fabsf(0.5f - altb);
        wa = ta ? alta : wa;
        wb = tb ? altb : wb;
        if(ta) env.stf(P_HVWT::at(P_HVWT::step(sw, j), ha), rr, wa);
        if(tb) env.stf(P_HVWT::at(P_HVWT::step(sw, j), hb), rr, wb);
        pa = wa;
        pb = wb;
      }
    }
    else if(tid >= 128 && tid < 192 && !(a.variant & 2))
    {
      // (the same walk over the diagonal weight)
      const int lane = tid & 63, ha = lane < TSH / 2 ? 2 * lane : TSH - 2, hb = ha + 1;
      const bool own = lane < TSH / 2;
      const int r0 = s * R - L_RBI;
      // the rows fetched: r0 - 1 .. r0 + R, moved into the plane where the tile begins / ends (rows that are voted on are not moved)
      const int rb = r0 < 1 ? 1 : (r0 > TS - R - 2 ? TS - R - 2 : r0), sw = P_PMWT::slot(rb);
      float raw[R][2], below[R][3];
      float pa = env.ldf(P_PMWT::at(P_PMWT::step(sw, -1), ha), -1), pb = env.ldf(P_PMWT::at(P_PMWT::step(sw, -1), hb), -1);
      AMZ_UNROLL
      for(int j = 0; j < R; j++)
      {
        const int q = fct(rb + j, 2, filters) & 1, sj = P_PMWT::step(sw, j), sd = P_PMWT::step(sw, j + 1);
        // the three sites of the rows above / below that this lane's two sites see: ha + q - 1, ha + q, ha + q + 1
        const int n0 = ha + q - 1 < 0 ? 0 : ha + q - 1, n1 = ha + q, n2 = ha + q + 1 > TSH - 1 ? TSH - 1 : ha + q + 1;
        raw[j][0] = env.ldf(P_PMWT::at(sj, ha), -1);
        raw[j][1] = env.ldf(P_PMWT::at(sj, hb), -1);
        below[j][0] = env.ldf(P_PMWT::at(sd, n0), -1);
        below[j][1] = env.ldf(P_PMWT::at(sd, n1), -1);
        below[j][2] = env.ldf(P_PMWT::at(sd, n2), -1);
      }
      AMZ_UNROLL
      for(int j = 0; j < R; j++)
      {
        const int rr = r0 + j;
        const int q = (rr >= 0 ? fct(rr, 2, filters) : 0) & 1, ca = q + 2 * ha, cb = ca + 2;
        // the voted row above at ha + q - 1, ha + q, ha + q + 1: with q = 1 own, own, lane + 1's first; with q = 0 lane - 1's
        // second, own, own
        const float from_next = env.shfl_down1(pa), from_prev = env.shfl_up1(pb);
        const float a0 = q ? pa : from_prev, a1 = q ? pb : pa, a2 = q ? from_next : pb;
        // (env.add(): x + y with the operands in this order -- where both are NaNs the sum is the first one's, on x86 and on the GPU
        // alike, and the exponent arithmetic of xdivf() carries its sign into numbers)
        const float alta = xdivf(env.add(env.add(env.add(a0, a1), below[j][0]), below[j][1]), 2);
        const float altb = xdivf(env.add(env.add(env.add(a1, a2), below[j][1]), below[j][2]), 2);
        const bool row_in = own && IN_(rr, 10, rr1 - 10); // (then rb == r0: the fetched row j is row rr)
        const bool ina = row_in && IN_(ca, 10, cc1 - 10), inb = row_in && IN_(cb, 10, cc1 - 10);
        float wa = raw[j][0], wb = raw[j][1];
        const bool ta = ina && fabsf(0.5f - wa) < fabsf(0.5f - alta), tb = inb && fabsf(0.5f - wb) < fabsf(0.5f - altb);
        wa = ta ? alta : wa;
        wb = tb ? altb : wb;
        if(ta) env.stf(P_PMWT::at(P_PMWT::step(sw, j), ha), rr, wa);
        if(tb) env.stf(P_PMWT::at(P_PMWT::step(sw, j), hb), rr, wb);
        pa = wa;
        pb = wb;
      }
    }
    // (the waves that do not vote share the gradients' 640 sites)
    for(int w = tid >= 192 ? tid - 128 : (tid >= 64 && tid < 128 ? tid - 64 : -1), _k = w; w >= 0 && _k < R * TS; _k += NT - 128)
      for(int rr = s * R - L_DQ + _k / TS, cc = _k % TS, _once = 1; _once && rr >= 0 && rr < TS; _once = 0)
    {
      float v = 0.f;
      if(IN_(rr, 2, rr1 - 2) && IN_(cc, 2, cc1 - 2))
      {
        const float delh = fabsf(LD(P_CFA, rr, cc + 1) - LD(P_CFA, rr, cc - 1));
        const float delv = fabsf(LD(P_CFA, rr + 1, cc) - LD(P_CFA, rr - 1, cc));
        v = sqr(delh) + sqr(delv);
      }
      ST(P_DQ, rr, cc, v);
    }
    env.sync();
    env.stamp(9);

    // ---- phase 10: green at the R/B sites and its curvature, :907-917
    FOR_RB(L_VOTE)
    {
      const float c = LD(P_CFA, rr, cc);
      float g = c, dg = 0.f, ch = 0.f, cv = 0.f;
      if(IN_(rr, 8, rr1 - 8) && IN_(cc, 8, cc1 - 8))
      {
        dg = intp(LD(P_HVWT, rr, h), LD(P_VCDH, rr, h), LD(P_HCDH, rr, h));
        g = c + dg;
        if(LDB(P_NY2, rr, h))
        {
          ch = sqr(g - xdiv2f(LD(P_CFA, rr, cc - 1) + LD(P_CFA, rr, cc + 1)));
          cv = sqr(g - xdiv2f(LD(P_CFA, rr - 1, cc) + LD(P_CFA, rr + 1, cc)));
        }
      }
      ST(P_GREEN, rr, h, g);
      ST(P_DGO, rr, h, dg);
      ST(P_GH, rr, h, ch);
      ST(P_GV, rr, h, cv);
    }
    // ... and on the other half of the threads R + B of the site from the two diagonal estimates and their voted weight, :1123
    FOR_RB2(L_RBI)
    {
      float rb = 0.f;
      const int jr = rr - (s * R - L_RBI);
      if(IN_(rr, 10, rr1 - 10) && IN_(cc, 10, cc1 - 10))
      {
        const float w = env.ldf(P_PMWT::at(P_PMWT::step(P_PMWT::slot(s * R - L_RBI + 70), jr), h), rr);
        const int sr = P_RBM::step(P_RBM::slot(s * R - L_RBI + 70), jr);
        rb = xdiv2f(env.add(env.add(LD(P_CFA, rr, cc), env.ldf(P_RBM::at(sr, h), rr) * (1.f - w)), env.ldf(P_RBP::at(sr, h), rr) * w));
      }
      ST(P_RBINT, rr, h, rb);
    }
    env.sync();
    env.stamp(10);

    // ---- phase 11: S8 refine flagged regions with the curvature of green, :923-956; then S11 where the diagonal estimate discriminates
    //      better, green from R + B, :1129-1236 (it overrides S8 at a site that takes both)
    //      Two lanes a site: the heavy part of S11 is the same expression along the column and along the row (:1150-1222), so
    //      lane 2 k walks site k's column, lane 2 k + 1 its row -- one instruction stream on different operands -- and the two
    //      trade their estimates by a lane swap; S8's two curvature sums likewise.  All ten waves work, none idles
    {
      const int _k = tid >> 1, half = tid & 1;
      const int rr = s * R - L_S8 + _k / TSH, h = _k % TSH;
      const int q = rr >= 0 ? (fct(rr, 2, filters) & 1) : 0, cc = q + 2 * h;
      if(rr >= 0 && rr < TS)
      {
        if(IN_(rr, 8, rr1 - 8) && IN_(cc, 8, cc1 - 8) && LDB(P_NY2, rr, h))
        {
          const float q0 = 0.169917f, q1 = 0.108947f, q2 = 0.069855f, q3 = 0.0287182f;
          // (the two planes have one shape: the lane's is a constant number of words behind the other)
          static_assert(P_GV::W == P_GH::W && P_GV::D == P_GH::D, "one index for both curvature planes");
          const int pl = half ? P_GV::OFF - P_GH::OFF : 0;
#define LDQ(r, c) env.ldf(P_GH::idx((r), (c)) + pl, (r))
          const float mine
              = (EPSSQ + (q0 * LDQ(rr, h) + q1 * (LDQ(rr - 1, (cc - 1) >> 1) + LDQ(rr - 1, (cc + 1) >> 1) + LDQ(rr + 1, (cc - 1) >> 1) + LDQ(rr + 1, (cc + 1) >> 1))
                          + q2 * (LDQ(rr - 2, h) + LDQ(rr, h - 1) + LDQ(rr, h + 1) + LDQ(rr + 2, h))
                          + q3 * (LDQ(rr - 2, h - 1) + LDQ(rr - 2, h + 1) + LDQ(rr + 2, h - 1) + LDQ(rr + 2, h + 1))));
#undef LDQ
          const float other = env.swap1(mine);
          const float gvarh = half ? other : mine, gvarv = half ? mine : other;
          const float dg = (LD(P_HCDH, rr, h) * gvarv + LD(P_VCDH, rr, h) * gvarh) / (gvarv + gvarh);
          if(!half)
          {
            ST(P_DGO, rr, h, dg);
            ST(P_GREEN, rr, h, LD(P_CFA, rr, cc) + dg);
          }
        }
        if(IN_(rr, 12, rr1 - 12) && IN_(cc, 12, cc1 - 12))
        {
          const float pm = LD(P_PMWT, rr, h), hv = LD(P_HVWT, rr, h);
          if(!(fabsf(0.5f - pm) < fabsf(0.5f - hv)))
          {
            // a: the neighbour above / to the left, b: below / to the right
            const int dr = half ? 0 : 1, dc = half ? 1 : 0;
            const float c = LD(P_CFA, rr, cc);
            const float ca = LD(P_CFA, rr - dr, cc - dc), cb = LD(P_CFA, rr + dr, cc + dc);
            const float rb = LD(P_RBINT, rr, h);
            const float rba = LD(P_RBINT, rr - 2 * dr, h - dc), rbb = LD(P_RBINT, rr + 2 * dr, h + dc);
            // the reference divides in binary64 (double literals) and rounds the quotient to binary32
            const float cra = div2_via_double(ca, EPS + rb + rba);
            const float crb = div2_via_double(cb, EPS + rb + rbb);
            const float ga = fabsf(1.f - cra) < ARTHRESH ? rb * cra : ca + xdiv2f(rb - rba);
            const float gb = fabsf(1.f - crb) < ARTHRESH ? rb * crb : cb + xdiv2f(rb - rbb);
            // the directional gradients of S1 at the two green neighbours, formed again from the mosaic
            const float ca2 = LD(P_CFA, rr - 2 * dr, cc - 2 * dc), ca3 = LD(P_CFA, rr - 3 * dr, cc - 3 * dc);
            const float cb2 = LD(P_CFA, rr + 2 * dr, cc + 2 * dc), cb3 = LD(P_CFA, rr + 3 * dr, cc + 3 * dc);
            const float da = EPS + fabsf(cb - ca) + fabsf(ca - ca3) + fabsf(c - ca2);
            const float db = EPS + fabsf(cb3 - cb) + fabsf(cb - ca) + fabsf(cb2 - c);
            // (the reference's two denominators are d0d + d0u and d1l + d1r: b + a and a + b -- binary32 addition commutes)
            float Gint = (da * gb + db * ga) / (db + da);
            if(Gint < rb)
            {
              if(2 * Gint < rb)
                Gint = ulim(Gint, ca, cb);
              else
              {
                const float wt = div2_via_double(rb - Gint, EPS + Gint + rb);
                Gint = wt * Gint + (1.f - wt) * ulim(Gint, ca, cb);
              }
            }
            if(Gint > clip_pt) Gint = ulim(Gint, ca, cb);
            const float other = env.swap1(Gint);
            const float Gintv = half ? other : Gint, Ginth = half ? Gint : other;
            const float g = Ginth * (1.f - hv) + Gintv * hv;
            if(!half)
            {
              ST(P_GREEN, rr, h, g);
              ST(P_DGO, rr, h, g - c);
            }
          }
        }
      }
    }
    env.sync();
    env.stamp(11);

    // ---- phase 12: S13 chrominance at the opposite R/B sites from the four diagonal neighbours, :1246-1276 (the split of :1239-1244 is in
    //      the indexing: a site keeps its own difference, this stage gives it the other one)
    FOR_RB(L_S13)
    {
      float v = 0.f;
      if(IN_(rr, 14, rr1 - 14) && IN_(cc, 14, cc1 - 14))
      {
#define D(dr, dc) LD(P_DGO, rr + (dr), (cc + (dc)) >> 1)
        const float nw1 = D(-1, -1), nw3 = D(-3, -3), se1 = D(1, 1), se3 = D(3, 3);
        const float ne1 = D(-1, 1), ne3 = D(-3, 3), sw1 = D(1, -1), sw3 = D(3, -3);
        const float wtnw = 1.f / (EPS + fabsf(nw1 - se1) + fabsf(nw1 - nw3) + fabsf(se1 - nw3));
        const float wtne = 1.f / (EPS + fabsf(ne1 - sw1) + fabsf(ne1 - ne3) + fabsf(sw1 - ne3));
        const float wtsw = 1.f / (EPS + fabsf(sw1 - ne1) + fabsf(sw1 - se3) + fabsf(ne1 - sw3)); // se3: as the reference has it
        const float wtse = 1.f / (EPS + fabsf(se1 - nw1) + fabsf(se1 - sw3) + fabsf(nw1 - se3)); // sw3: likewise
        v = (wtnw * (1.325f * nw1 - 0.175f * nw3 - 0.075f * D(-1, -3) - 0.075f * D(-3, -1))
             + wtne * (1.325f * ne1 - 0.175f * ne3 - 0.075f * D(-1, 3) - 0.075f * D(1, 1))
             + wtsw * (1.325f * sw1 - 0.175f * sw3 - 0.075f * D(1, -3) - 0.075f * D(-1, -1))
             + wtse * (1.325f * se1 - 0.175f * se3 - 0.075f * D(1, 3) - 0.075f * D(3, 1)))
            / (wtnw + wtne + wtsw + wtse);
#undef D
      }
      ST(P_DGP, rr, h, v);
    }
    env.sync();
    env.stamp(12);

    // ---- phase 13: S14 output, :1278-1411 (alpha is left as it is)
    FOR_FULL(L_S14)
    {
      if(IN_(rr, 16, rr1 - 16) && IN_(cc, 16, cc1 - 16) && rr + top < height && cc + left < width && IN_(rr + top, a.out_row0, a.out_row1))
      {
        float *const o = out + 4 * ((size_t)(rr + top - a.out_row0) * width + (cc + left));
        const int col = fct(rr, cc, filters);
        if(col & 1)
        {
          // a green site: the vertical neighbours are of one R/B colour, the horizontal ones of the other
          const bool vert_red = fct(rr - 1, cc, filters) == 0;
          const int hl = (cc - 1) >> 1, hr = (cc + 1) >> 1, hv = cc >> 1;
          const float wu = LD(P_HVWT, rr - 1, hv), wr = LD(P_HVWT, rr, hr), wl = LD(P_HVWT, rr, hl), wd = LD(P_HVWT, rr + 1, hv);
          const float temp = 1.f / (wu + 2.f - wr - wl + wd);
          // dgrb0 (G - R) and dgrb1 (G - B) at the four neighbours
          const float u0 = vert_red ? LD(P_DGO, rr - 1, hv) : LD(P_DGP, rr - 1, hv), u1 = vert_red ? LD(P_DGP, rr - 1, hv) : LD(P_DGO, rr - 1, hv);
          const float d0 = vert_red ? LD(P_DGO, rr + 1, hv) : LD(P_DGP, rr + 1, hv), d1 = vert_red ? LD(P_DGP, rr + 1, hv) : LD(P_DGO, rr + 1, hv);
          const float r0 = vert_red ? LD(P_DGP, rr, hr) : LD(P_DGO, rr, hr), r1 = vert_red ? LD(P_DGO, rr, hr) : LD(P_DGP, rr, hr);
          const float l0 = vert_red ? LD(P_DGP, rr, hl) : LD(P_DGO, rr, hl), l1 = vert_red ? LD(P_DGO, rr, hl) : LD(P_DGP, rr, hl);
          const float g = LD(P_CFA, rr, cc);
          env.store_rgb(o, clampnan(g - ((wu)*u0 + (1.f - wr) * r0 + (1.f - wl) * l0 + (wd)*d0) * temp, 0.0f, 1.0f), clampnan(g, 0.0f, 1.0f),
                        clampnan(g - ((wu)*u1 + (1.f - wr) * r1 + (1.f - wl) * l1 + (wd)*d1) * temp, 0.0f, 1.0f));
        }
        else
        {
          const int hh = cc >> 1;
          const float g = LD(P_GREEN, rr, hh), own = LD(P_DGO, rr, hh), opp = LD(P_DGP, rr, hh);
          env.store_rgb(o, clampnan(g - (col == 0 ? own : opp), 0.0f, 1.0f), clampnan(g, 0.0f, 1.0f),
                        clampnan(g - (col == 0 ? opp : own), 0.0f, 1.0f));
        }
      }
    }
    env.sync();
    env.stamp(13);
  }
#undef LD
#undef AMZ_UNROLL
#undef ST
#undef LDB
#undef STB
#undef IN_
#undef FOR_FULL
#undef FOR_FULL_UPPER
#undef FOR_RB
#undef FOR_RB2
}

} // namespace amz
