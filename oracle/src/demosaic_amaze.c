/* oracle/src/demosaic_amaze.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restatement of the AMaZE demosaic (Aliasing Minimization and Zipper Elimination, E. Martinec; the
 * RawTherapee port the reference carries): amaze_demosaic_RT(), src/iop/demosaic/amaze.cc:181-1419.
 *
 * The frame is processed in 160 x 160 tiles that overlap by 32 (16 on every side; the frame itself is
 * extended by 16 mirrored photosites).  Inside a tile the algorithm is a fixed sequence of stencil
 * stages over tile-sized planes.  The reference packs those planes into one allocation and lets
 * planes whose lifetimes do not overlap share memory; the restatement gives every plane its own
 * zero-initialised storage per tile (tests/test_oracle_vs_ref.py shows the two agree bit for bit).
 *
 * Three stages update a plane IN PLACE while reading neighbours that the same sweep has already
 * updated (Gauss-Seidel order); their row-major order is part of the result and is kept:
 *   S3  the choice between the two colour-difference estimates and the saturation bound
 *       (amaze.cc:585-705): hcd depends on hcd two columns to the left, vcd on vcd two rows up;
 *   S7  the diagonal-neighbour vote on the H/V weight (:894-917): row r reads row r-1 updated;
 *   S10 the same vote on the plus/minus diagonal weight (:1109-1126).
 * Every other stage reads planes written by earlier stages only.
 *
 * Half-resolution planes are indexed by (tile index) >> 1 as in the reference: they hold one value
 * per R/B photosite (or per photosite pair).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

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

static const float EPS = 1e-5f, EPSSQ = 1e-10f, ARTHRESH = 0.75f;
static const float GAUSSODD[4] = { 0.14659727707323927f, 0.103592713382435f, 0.0732036125103057f, 0.0365543548389495f };
static const float NYQTHRESH = 0.5f;
static const float GAUSSEVEN[2] = { 0.13719494435797422f, 0.05640252782101291f };
static const float GQUINC[4] = { 0.169917f, 0.108947f, 0.069855f, 0.0287182f };

static inline float sqr(const float x) { return x * x; }
static inline float fmin2(const float a, const float b) { return b < a ? b : a; } /* std::min(a, b) */
static inline float fmax2(const float a, const float b) { return a < b ? b : a; } /* std::max(a, b) */
static inline float lim(const float a, const float b, const float c) { return fmax2(b, fmin2(a, c)); }
static inline float ulim(const float a, const float b, const float c) { return (b < c) ? lim(a, b, c) : lim(a, c, b); }
static inline float intp(const float a, const float b, const float c) { return a * (b - c) + c; }

/* exponent tricks, amaze.cc:77-121: multiply / divide by a power of two unless the value is +-0 */
static inline float expo(float d, const int n)
{
  union { float f; uint32_t u; } x;
  x.f = d;
  if(x.u & 0x7FFFFFFF) x.u += (uint32_t)n << 23;
  return x.f;
}
static inline float xmul2f(const float d) { return expo(d, 1); }
static inline float xdiv2f(const float d) { return expo(d, -1); }
static inline float xdivf(const float d, const int n) { return expo(d, -n); }

/* clampnan(), amaze.cc:61-75 (NaN passes through: the isnan branch is shadowed by !isfinite) */
static inline float clampnan(const float x, const float m, const float M)
{
  if(!isfinite(x)) return isless(x, m) ? m : (isgreater(x, M) ? M : x);
  return x;
}

/* The reference carves its planes out of ONE allocation (amaze.cc:274-327) and lets planes with
 * disjoint lifetimes share memory.  A few stencils read positions their logical plane never wrote in
 * this tile and therefore see what the sharing partner left there (e.g. the G-B plane beyond the
 * columns the split at :1239 moves, which still holds the alternative vertical colour difference);
 * two border fills run past their plane into the next one.  The restatement therefore lays the planes
 * out exactly as the reference does -- offsets in floats, 32 floats (128 bytes) of padding between
 * planes -- and zeroes the whole buffer per tile. */
#define PAD 32
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
  O_DGRB2 = O_HVWT + TS * TSH + PAD, /* {h, v} pairs; shares with dgintv */
  O_DGINTH = O_DGRB2 + TS * TS + PAD,
  O_DSQ1M = O_DGINTH + TS * TS + PAD,
  O_DSQ1P = O_DSQ1M + TS * TSH + PAD,
  O_CFA = O_DSQ1P + TS * TSH + PAD,
  O_NYQUIST = O_CFA + TS * TS + PAD,              /* TS * TSH bytes */
  O_NYQUTEST = O_NYQUIST + TS * TSH / 4 + PAD,
  O_END = O_NYQUTEST + TS * TSH + PAD + 16 * TS /* slack for the bottom strip that runs past cfa */
};

typedef struct
{
  float buf[O_END];
  float priv[O_END]; /* probe only (oracle_amaze_unshare): private storage of planes taken out of their sharing */
} tile_t;

/* test hook: start every tile from this value instead of 0, to expose reads of never-written words */
static float g_amaze_poison = 0.0f;
void oracle_amaze_poison(const float v) { g_amaze_poison = v; }
/* test hook: 1 = keep the buffer from tile to tile and walk the tiles in row-major order on one thread,
 * i.e. exactly what the reference does with OMP_NUM_THREADS=1 (only the flag plane is cleared per tile) */
static int g_amaze_persistent = 0;
void oracle_amaze_persistent(const int on) { g_amaze_persistent = on; }
/* probe (tools/amaze_alias_probe.py): which of the reference's plane sharings (amaze.cc:300-327) the RESULT depends on.
 * Bit set = that plane gets zero-initialised storage of its own instead of its partner's memory:
 *   1 dgrb0 / dgrb1 (with vcdalt)   2 delp / delm / rbint (with cddiffsq)   4 pmwt (with delhvsqsum)
 *   8 rbm / rbp (with vcd)         16 the second Nyquist flag plane (with cddiffsq's bytes)   32 dgrb2 (with dgintv)
 *  64 (not a sharing) the Nyquist refinement runs over the whole tile interior instead of the bounding box of the flagged
 *     sites, and also where the reference skips it because that box has no extent (amaze.cc:800-828): a frame that does
 *     NOT change under this bit shows that the box is an optimisation, not a dependency of the whole tile
 * 0 = the reference's layout.  A frame that changes under a bit shows where a stencil reads what the partner left. */
static unsigned g_amaze_unshare = 0;

void oracle_amaze_unshare(const unsigned bits) { g_amaze_unshare = bits; }

static void amaze_tile(tile_t *t, const float *in, float *out, const int width, const int height, const int top,
                       const int left, const uint32_t filters, const int ex, const int ey, const float clip_pt)
{
  if(!g_amaze_persistent)
  {
    memset(t->buf, 0, sizeof(t->buf));
    if(g_amaze_poison != 0.0f)
      for(int k = 0; k < O_END; k++) t->buf[k] = g_amaze_poison;
  }
  if(g_amaze_unshare) memset(t->priv, 0, sizeof(t->priv));
  else
    memset((unsigned char *)(t->buf + O_NYQUIST) + 3 * TSH, 0, sizeof(unsigned char) * (TS - 6) * TSH); /* amaze.cc:337 */
  const float clip_pt8 = 0.8f * clip_pt;
  const int bottom = top + TS < height + 16 ? top + TS : height + 16;
  const int right = left + TS < width + 16 ? left + TS : width + 16;
  const int rr1 = bottom - top, cc1 = right - left;
  const int rrmin = top < 0 ? 16 : 0, ccmin = left < 0 ? 16 : 0;
  const int rrmax = bottom > height ? height - top : rr1, ccmax = right > width ? width - left : cc1;
  float *const B = t->buf;
  float *const cfa = B + O_CFA, *const green = B + O_GREEN, *const delhvsqsum = B + O_DELHVSQSUM;
  float *const dirwts0 = B + O_DIRWTS0, *const dirwts1 = B + O_DIRWTS1, *const vcd = B + O_VCD, *const hcd = B + O_HCD;
  float *const vcdalt = B + O_VCDALT, *const hcdalt = B + O_HCDALT, *const cddiffsq = B + O_CDDIFFSQ;
  float *const hvwt = B + O_HVWT, *const dgintv = B + O_DGRB2, *const dginth = B + O_DGINTH;
  float *const Q = t->priv;
  const unsigned un = g_amaze_unshare;
  float *const dgrb2 = (un & 32 ? Q : B) + O_DGRB2; /* [2 * k] = h, [2 * k + 1] = v */
  float *const dsq1m = B + O_DSQ1M, *const dsq1p = B + O_DSQ1P, *const nyqutest = B + O_NYQUTEST;
  unsigned char *const nyquist = (unsigned char *)(B + O_NYQUIST);
  /* shared storage, amaze.cc:300-327 */
  float *const dgrb0 = un & 1 ? Q + O_VCDALT : vcdalt, *const dgrb1 = dgrb0 + TS * TSH;
  float *const delp = un & 2 ? Q + O_CDDIFFSQ : cddiffsq, *const delm = delp + TS * TSH + PAD, *const rbint = delm;
  float *const pmwt = un & 4 ? Q + O_DELHVSQSUM : delhvsqsum;
  float *const rbm = un & 8 ? Q + O_VCD : vcd, *const rbp = rbm + TS * TSH + PAD;
  unsigned char *const nyquist2 = un & 16 ? (unsigned char *)(Q + O_NYQUIST) : (unsigned char *)cddiffsq;
#define FCT(r, c) oracle_fc((r), (c), filters)

  /* S0 tile load with 16 mirrored photosites beyond every frame edge, amaze.cc:352-460.
   * The fills are replayed in the reference's order with its raw tile indices, because two of them
   * run past their target when a tile is cut by its own size rather than by the frame:
   *   - the right strip always writes 16 columns from ccmax; with ccmax > 144 it wraps into columns
   *     0.. of the NEXT tile row (already filled) -- kept;
   *   - the bottom strip always writes 16 rows from rrmax; with rrmax > 144 it runs past the cfa plane
   *     into the Nyquist flag bytes that follow it in the reference's buffer (128 bytes of padding in
   *     between, amaze.cc:274-327), after those flags were cleared -- the flags then start from the
   *     bytes of those floats (the planes are laid out as in the reference, so this just happens). */
#define PUT(idx, v)            \
  do                           \
  {                            \
    const int _i = (idx);      \
    const float _v = (v);      \
    cfa[_i] = _v;   /* may run into the flag bytes behind the plane, as in the reference */ \
    green[_i] = _v; /* may run into delhvsqsum, overwritten by S1 where it is used */       \
  } while(0)
  if(rrmin > 0)
    for(int rr = 0; rr < 16; rr++)
      for(int cc = ccmin, row = 32 - rr + top; cc < ccmax; cc++) PUT(rr * TS + cc, in[(size_t)row * width + (cc + left)]);
  for(int rr = rrmin; rr < rrmax; rr++)
    for(int cc = ccmin; cc < ccmax; cc++) PUT(rr * TS + cc, in[(size_t)(rr + top) * width + (cc + left)]);
  if(rrmax < rr1)
    for(int rr = 0; rr < 16; rr++)
      for(int cc = ccmin; cc < ccmax; cc++) PUT((rrmax + rr) * TS + cc, in[(size_t)(height - rr - 2) * width + (left + cc)]);
  if(ccmin > 0)
    for(int rr = rrmin; rr < rrmax; rr++)
      for(int cc = 0, row = rr + top; cc < 16; cc++) PUT(rr * TS + cc, in[(size_t)row * width + (32 - cc + left)]);
  if(ccmax < cc1)
    for(int rr = rrmin; rr < rrmax; rr++)
      for(int cc = 0; cc < 16; cc++) PUT(rr * TS + ccmax + cc, in[(size_t)(top + rr) * width + (width - cc - 2)]);
  if(rrmin > 0 && ccmin > 0)
    for(int rr = 0; rr < 16; rr++)
      for(int cc = 0; cc < 16; cc++) PUT(rr * TS + cc, in[(size_t)(32 - rr) * width + (32 - cc)]);
  if(rrmax < rr1 && ccmax < cc1)
    for(int rr = 0; rr < 16; rr++)
      for(int cc = 0; cc < 16; cc++) PUT((rrmax + rr) * TS + ccmax + cc, in[(size_t)(height - rr - 2) * width + (width - cc - 2)]);
  if(rrmin > 0 && ccmax < cc1)
    for(int rr = 0; rr < 16; rr++)
      for(int cc = 0; cc < 16; cc++) PUT(rr * TS + ccmax + cc, in[(size_t)(32 - rr) * width + (width - cc - 2)]);
  if(rrmax < rr1 && ccmin > 0)
    for(int rr = 0; rr < 16; rr++)
      for(int cc = 0; cc < 16; cc++) PUT((rrmax + rr) * TS + cc, in[(size_t)(height - rr - 2) * width + (32 - cc)]);
#undef PUT

  /* S1 gradients, :463-473 */
  for(int rr = 2; rr < rr1 - 2; rr++)
    for(int cc = 2, i = rr * TS + cc; cc < cc1 - 2; cc++, i++)
    {
      const float delh = fabsf(cfa[i + 1] - cfa[i - 1]);
      const float delv = fabsf(cfa[i + V1] - cfa[i - V1]);
      dirwts0[i] = EPS + fabsf(cfa[i + V2] - cfa[i]) + fabsf(cfa[i] - cfa[i - V2]) + delv;
      dirwts1[i] = EPS + fabsf(cfa[i + 2] - cfa[i]) + fabsf(cfa[i] - cfa[i - 2]) + delh;
      delhvsqsum[i] = sqr(delh) + sqr(delv);
    }

  /* S2 colour differences by adaptive ratios and by Hamilton-Adams, :478-582 */
  const float *d0 = dirwts0, *d1 = dirwts1;
  for(int rr = 4; rr < rr1 - 4; rr++)
    for(int cc = 4, i = rr * TS + cc; cc < cc1 - 4; cc++, i++)
    {
      const int sign = (FCT(rr, cc) & 1) ? 1 : 0; /* fcswitch: green site */
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
      if(sign)
      {
        vcd[i] = cfa[i] - (vwt * gdar + (1.f - vwt) * guar);
        hcd[i] = cfa[i] - (hwt * grar + (1.f - hwt) * glar);
        vcdalt[i] = cfa[i] - Gintvha;
        hcdalt[i] = cfa[i] - Ginthha;
      }
      else
      {
        vcd[i] = (vwt * gdar + (1.f - vwt) * guar) - cfa[i];
        hcd[i] = (hwt * grar + (1.f - hwt) * glar) - cfa[i];
        vcdalt[i] = Gintvha - cfa[i];
        hcdalt[i] = Ginthha - cfa[i];
      }
      if(cfa[i] > clip_pt8 || Gintvha > clip_pt8 || Ginthha > clip_pt8)
      {
        guar = guha;
        gdar = gdha;
        glar = glha;
        grar = grha;
        vcd[i] = vcdalt[i];
        hcd[i] = hcdalt[i];
      }
      dgintv[i] = fmin2(sqr(guha - gdha), sqr(guar - gdar));
      dginth[i] = fmin2(sqr(glha - grha), sqr(glar - grar));
    }

  /* S3 choose the smoother estimate, bound it in saturated regions -- in place, row-major, :585-705 */
  for(int rr = 4; rr < rr1 - 4; rr++)
    for(int cc = 4, i = rr * TS + cc; cc < cc1 - 4; cc++, i++)
    {
      const int gsite = FCT(rr, cc) & 1;
      const float hcdvar = 3.f * (sqr(hcd[i - 2]) + sqr(hcd[i]) + sqr(hcd[i + 2])) - sqr(hcd[i - 2] + hcd[i] + hcd[i + 2]);
      const float hcdaltvar = 3.f * (sqr(hcdalt[i - 2]) + sqr(hcdalt[i]) + sqr(hcdalt[i + 2]))
                              - sqr(hcdalt[i - 2] + hcdalt[i] + hcdalt[i + 2]);
      const float vcdvar = 3.f * (sqr(vcd[i - V2]) + sqr(vcd[i]) + sqr(vcd[i + V2])) - sqr(vcd[i - V2] + vcd[i] + vcd[i + V2]);
      const float vcdaltvar = 3.f * (sqr(vcdalt[i - V2]) + sqr(vcdalt[i]) + sqr(vcdalt[i + V2]))
                              - sqr(vcdalt[i - V2] + vcdalt[i] + vcdalt[i + V2]);
      if(hcdaltvar < hcdvar) hcd[i] = hcdalt[i];
      if(vcdaltvar < vcdvar) vcd[i] = vcdalt[i];
      float Ginth, Gintv;
      if(gsite)
      {
        Ginth = -hcd[i] + cfa[i];
        Gintv = -vcd[i] + cfa[i];
        if(hcd[i] > 0)
        {
          if(3.f * hcd[i] > (Ginth + cfa[i]))
            hcd[i] = -ulim(Ginth, cfa[i - 1], cfa[i + 1]) + cfa[i];
          else
          {
            const float hwt = 1.f - 3.f * hcd[i] / (EPS + Ginth + cfa[i]);
            hcd[i] = hwt * hcd[i] + (1.f - hwt) * (-ulim(Ginth, cfa[i - 1], cfa[i + 1]) + cfa[i]);
          }
        }
        if(vcd[i] > 0)
        {
          if(3.f * vcd[i] > (Gintv + cfa[i]))
            vcd[i] = -ulim(Gintv, cfa[i - V1], cfa[i + V1]) + cfa[i];
          else
          {
            const float vwt = 1.f - 3.f * vcd[i] / (EPS + Gintv + cfa[i]);
            vcd[i] = vwt * vcd[i] + (1.f - vwt) * (-ulim(Gintv, cfa[i - V1], cfa[i + V1]) + cfa[i]);
          }
        }
        if(Ginth > clip_pt) hcd[i] = -ulim(Ginth, cfa[i - 1], cfa[i + 1]) + cfa[i];
        if(Gintv > clip_pt) vcd[i] = -ulim(Gintv, cfa[i - V1], cfa[i + V1]) + cfa[i];
      }
      else
      {
        Ginth = hcd[i] + cfa[i];
        Gintv = vcd[i] + cfa[i];
        if(hcd[i] < 0)
        {
          if(3.f * hcd[i] < -(Ginth + cfa[i]))
            hcd[i] = ulim(Ginth, cfa[i - 1], cfa[i + 1]) - cfa[i];
          else
          {
            const float hwt = 1.f + 3.f * hcd[i] / (EPS + Ginth + cfa[i]);
            hcd[i] = hwt * hcd[i] + (1.f - hwt) * (ulim(Ginth, cfa[i - 1], cfa[i + 1]) - cfa[i]);
          }
        }
        if(vcd[i] < 0)
        {
          if(3.f * vcd[i] < -(Gintv + cfa[i]))
            vcd[i] = ulim(Gintv, cfa[i - V1], cfa[i + V1]) - cfa[i];
          else
          {
            const float vwt = 1.f + 3.f * vcd[i] / (EPS + Gintv + cfa[i]);
            vcd[i] = vwt * vcd[i] + (1.f - vwt) * (ulim(Gintv, cfa[i - V1], cfa[i + V1]) - cfa[i]);
          }
        }
        if(Ginth > clip_pt) hcd[i] = ulim(Ginth, cfa[i - 1], cfa[i + 1]) - cfa[i];
        if(Gintv > clip_pt) vcd[i] = ulim(Gintv, cfa[i - V1], cfa[i + V1]) - cfa[i];
        cddiffsq[i] = sqr(vcd[i] - hcd[i]);
      }
    }

  /* S4 H/V weight at R/B sites from colour-difference variances, :707-760 */
  for(int rr = 6; rr < rr1 - 6; rr++)
    for(int cc = 6 + (FCT(rr, 2) & 1), i = rr * TS + cc; cc < cc1 - 6; cc += 2, i += 2)
    {
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
      /* the product is formed in binary64 (0.5 is a double literal there) */
      if((0.5 - (double)varwt) * (0.5 - (double)diffwt) > 0 && fabsf(0.5f - diffwt) < fabsf(0.5f - varwt))
        hvwt[i >> 1] = varwt;
      else
        hvwt[i >> 1] = diffwt;
    }

  /* S5 Nyquist texture test, :763-820 */
  const float gg[6] = { NYQTHRESH * 0.07384411893421103f, NYQTHRESH * 0.06207511968171489f, NYQTHRESH * 0.0521818194747806f,
                        NYQTHRESH * 0.03687419286733595f, NYQTHRESH * 0.03099732204057846f, NYQTHRESH * 0.018413194161458882f };
  const float *cd = cddiffsq, *dq = delhvsqsum;
  int nystartrow = 0, nyendrow = 0, nystartcol = TS + 1, nyendcol = 0;
  for(int rr = 6; rr < rr1 - 6; rr++)
    for(int cc = 6 + (FCT(rr, 2) & 1), i = rr * TS + cc; cc < cc1 - 6; cc += 2, i += 2)
    {
      const float test
          = (GAUSSODD[0] * cd[i] + GAUSSODD[1] * (cd[i - M1] + cd[i + P1] + cd[i - P1] + cd[i + M1])
             + GAUSSODD[2] * (cd[i - V2] + cd[i - 2] + cd[i + 2] + cd[i + V2])
             + GAUSSODD[3] * (cd[i - M2] + cd[i + P2] + cd[i - P2] + cd[i + M2]))
            - (gg[0] * dq[i] + gg[1] * (dq[i - V1] + dq[i + 1] + dq[i - 1] + dq[i + V1])
               + gg[2] * (dq[i - M1] + dq[i + P1] + dq[i - P1] + dq[i + M1])
               + gg[3] * (dq[i - V2] + dq[i - 2] + dq[i + 2] + dq[i + V2])
               + gg[4] * (dq[i - V2 - 1] + dq[i - V2 + 1] + dq[i - TS - 2] + dq[i - TS + 2] + dq[i + TS - 2] + dq[i + TS + 2]
                          + dq[i + V2 - 1] + dq[i + V2 + 1])
               + gg[5] * (dq[i - M2] + dq[i + P2] + dq[i - P2] + dq[i + M2]));
      nyqutest[i >> 1] = test;
      if(test > 0.f)
      {
        nyquist[i >> 1] = 1;
        nystartrow = nystartrow ? nystartrow : rr;
        nyendrow = rr;
        nystartcol = nystartcol > cc ? cc : nystartcol;
        nyendcol = nyendcol < cc ? cc : nyendcol;
      }
    }
  int do_nyquist = nystartrow != nyendrow && nystartcol != nyendcol;
  if(g_amaze_unshare & 64)
  {
    /* probe: every site of the interior is voted on, flagged or not */
    do_nyquist = 1;
    nystartrow = 0;
    nyendrow = rr1;
    nystartcol = 0;
    nyendcol = cc1;
  }
  if(do_nyquist)
  {
    nyendrow++;
    nyendcol++;
    nystartcol -= (nystartcol & 1);
    nystartrow = nystartrow > 8 ? nystartrow : 8;
    nyendrow = nyendrow < rr1 - 8 ? nyendrow : rr1 - 8;
    nystartcol = nystartcol > 8 ? nystartcol : 8;
    nyendcol = nyendcol < cc1 - 8 ? nyendcol : cc1 - 8;
    /* S6 majority vote on the flags, then area interpolation of the weight in flagged regions, :832-890 */
    const unsigned char *ny = nyquist;
    memset(&nyquist2[4 * TSH], 0, sizeof(char) * (TS - 8) * TSH); /* amaze.cc:830; the bytes are shared with cddiffsq */
    for(int rr = nystartrow; rr < nyendrow; rr++)
      for(int i = rr * TS + nystartcol + (FCT(rr, 2) & 1); i < rr * TS + nyendcol; i += 2)
      {
        const unsigned n = ny[(i - V2) >> 1] + ny[(i - M1) >> 1] + ny[(i + P1) >> 1] + ny[(i - 2) >> 1] + ny[(i + 2) >> 1]
                           + ny[(i - P1) >> 1] + ny[(i + M1) >> 1] + ny[(i + V2) >> 1];
        nyquist2[i >> 1] = n > 4 ? 1 : (n < 4 ? 0 : ny[i >> 1]);
      }
    for(int rr = nystartrow; rr < nyendrow; rr++)
      for(int i = rr * TS + nystartcol + (FCT(rr, 2) & 1); i < rr * TS + nyendcol; i += 2)
      {
        if(!nyquist2[i >> 1]) continue;
        float sumcfa = 0.f, sumh = 0.f, sumv = 0.f, sumsqh = 0.f, sumsqv = 0.f, areawt = 0.f;
        for(int a = -6; a < 7; a += 2)
        {
          int i1 = i + (a * TS) - 6;
          for(int b = -6; b < 7; b += 2, i1 += 2)
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

  /* S7 green at R/B sites -- the weight vote is in place, row r sees row r-1 already voted, :894-917 */
  for(int rr = 8; rr < rr1 - 8; rr++)
    for(int i = rr * TS + 8 + (FCT(rr, 2) & 1); i < rr * TS + cc1 - 8; i += 2)
    {
      const float alt = xdivf(hvwt[(i - M1) >> 1] + hvwt[(i + P1) >> 1] + hvwt[(i - P1) >> 1] + hvwt[(i + M1) >> 1], 2);
      hvwt[i >> 1] = fabsf(0.5f - hvwt[i >> 1]) < fabsf(0.5f - alt) ? alt : hvwt[i >> 1];
      dgrb0[i >> 1] = intp(hvwt[i >> 1], vcd[i], hcd[i]);
      green[i] = cfa[i] + dgrb0[i >> 1];
      const int flagged = do_nyquist && nyquist2[i >> 1];
      dgrb2[2 * (i >> 1)] = flagged ? sqr(green[i] - xdiv2f(green[i - 1] + green[i + 1])) : 0.f;
      dgrb2[2 * (i >> 1) + 1] = flagged ? sqr(green[i] - xdiv2f(green[i - V1] + green[i + V1])) : 0.f;
    }

  /* S8 refine flagged regions with the curvature of green, :923-956 */
  if(do_nyquist)
    for(int rr = nystartrow; rr < nyendrow; rr++)
      for(int i = rr * TS + nystartcol + (FCT(rr, 2) & 1); i < rr * TS + nyendcol; i += 2)
      {
        if(!nyquist2[i >> 1]) continue;
#define GH(k) dgrb2[2 * (k)]
#define GV(k) dgrb2[2 * (k) + 1]
        const float gvarh
            = EPSSQ + (GQUINC[0] * GH(i >> 1) + GQUINC[1] * (GH((i - M1) >> 1) + GH((i + P1) >> 1) + GH((i - P1) >> 1) + GH((i + M1) >> 1))
                       + GQUINC[2] * (GH((i - V2) >> 1) + GH((i - 2) >> 1) + GH((i + 2) >> 1) + GH((i + V2) >> 1))
                       + GQUINC[3] * (GH((i - M2) >> 1) + GH((i + P2) >> 1) + GH((i - P2) >> 1) + GH((i + M2) >> 1)));
        const float gvarv
            = EPSSQ + (GQUINC[0] * GV(i >> 1) + GQUINC[1] * (GV((i - M1) >> 1) + GV((i + P1) >> 1) + GV((i - P1) >> 1) + GV((i + M1) >> 1))
                       + GQUINC[2] * (GV((i - V2) >> 1) + GV((i - 2) >> 1) + GV((i + 2) >> 1) + GV((i + V2) >> 1))
                       + GQUINC[3] * (GV((i - M2) >> 1) + GV((i + P2) >> 1) + GV((i - P2) >> 1) + GV((i + M2) >> 1)));
        dgrb0[i >> 1] = (hcd[i] * gvarv + vcd[i] * gvarh) / (gvarv + gvarh);
        green[i] = cfa[i] + dgrb0[i >> 1];
      }
#undef GH
#undef GV

  /* S9 diagonal gradients and squared diagonal differences, :958-983; diagonal R/B estimates, :986-1107 */
  for(int rr = 6; rr < rr1 - 6; rr++)
  {
    const int odd = FCT(rr, 2) & 1;
    for(int cc = 6, i = rr * TS + cc; cc < cc1 - 6; cc += 2, i += 2)
    {
      const int a = odd ? i + 1 : i; /* photosite the gradients are taken at */
      const int b = odd ? i : i + 1; /* photosite the squared differences are taken at */
      delp[i >> 1] = fabsf(cfa[a + P1] - cfa[a - P1]);
      delm[i >> 1] = fabsf(cfa[a + M1] - cfa[a - M1]);
      dsq1p[i >> 1] = (sqr(cfa[b] - cfa[b - P1]) + sqr(cfa[b] - cfa[b + P1]));
      dsq1m[i >> 1] = (sqr(cfa[b] - cfa[b - M1]) + sqr(cfa[b] - cfa[b + M1]));
    }
  }
  const float *sm = dsq1m, *sp = dsq1p;
  for(int rr = 8; rr < rr1 - 8; rr++)
    for(int cc = 8 + (FCT(rr, 2) & 1), i = rr * TS + cc, h = i >> 1; cc < cc1 - 8; cc += 2, i += 2, h++)
    {
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
          = EPSSQ + (GAUSSEVEN[0] * (sm[(i - V1) >> 1] + sm[(i - 1) >> 1] + sm[(i + 1) >> 1] + sm[(i + V1) >> 1])
                     + GAUSSEVEN[1] * (sm[(i - V2 - 1) >> 1] + sm[(i - V2 + 1) >> 1] + sm[(i - 2 - V1) >> 1] + sm[(i + 2 - V1) >> 1]
                                       + sm[(i - 2 + V1) >> 1] + sm[(i + 2 + V1) >> 1] + sm[(i + V2 - 1) >> 1] + sm[(i + V2 + 1) >> 1]));
      pmwt[h]
          = rbvarm / ((EPSSQ + (GAUSSEVEN[0] * (sp[(i - V1) >> 1] + sp[(i - 1) >> 1] + sp[(i + 1) >> 1] + sp[(i + V1) >> 1])
                                + GAUSSEVEN[1] * (sp[(i - V2 - 1) >> 1] + sp[(i - V2 + 1) >> 1] + sp[(i - 2 - V1) >> 1] + sp[(i + 2 - V1) >> 1]
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
      rbp[h] = vp;
      rbm[h] = vm;
    }

  /* S10 vote on the diagonal weight (in place, row r sees row r-1 voted) and R+B, :1109-1126 */
  for(int rr = 10; rr < rr1 - 10; rr++)
    for(int cc = 10 + (FCT(rr, 2) & 1), i = rr * TS + cc, h = i >> 1; cc < cc1 - 10; cc += 2, i += 2, h++)
    {
      const float alt = xdivf(pmwt[(i - M1) >> 1] + pmwt[(i + P1) >> 1] + pmwt[(i - P1) >> 1] + pmwt[(i + M1) >> 1], 2);
      if(fabsf(0.5f - pmwt[h]) < fabsf(0.5f - alt)) pmwt[h] = alt;
      rbint[h] = xdiv2f(cfa[i] + rbm[h] * (1.f - pmwt[h]) + rbp[h] * pmwt[h]);
    }

  /* S11 where the diagonal estimate discriminates better, redo green from R+B, :1129-1236 */
  for(int rr = 12; rr < rr1 - 12; rr++)
    for(int cc = 12 + (FCT(rr, 2) & 1), i = rr * TS + cc, h = i >> 1; cc < cc1 - 12; cc += 2, i += 2, h++)
    {
      if(fabsf(0.5f - pmwt[i >> 1]) < fabsf(0.5f - hvwt[i >> 1])) continue;
      /* the ratios are formed in binary64 (2.0 is a double literal there) and rounded once */
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
      green[i] = Ginth * (1.f - hvwt[h]) + Gintv * hvwt[h];
      dgrb0[i >> 1] = green[i] - cfa[i];
    }

  /* S12 split G-B from G-R: the B coset moves to the second plane, :1239-1244 */
  for(int rr = 13 - ey; rr < rr1 - 12; rr += 2)
    for(int h = (rr * TS + 13 - ex) >> 1; h < (rr * TS + cc1 - 12) >> 1; h++)
    {
      dgrb1[h] = dgrb0[h];
      dgrb0[h] = 0;
    }

  /* S13 chrominance at the opposite R/B sites from the four diagonal neighbours, :1246-1276 */
  for(int rr = 14; rr < rr1 - 14; rr++)
    for(int cc = 14 + (FCT(rr, 2) & 1), i = rr * TS + cc; cc < cc1 - 14; cc += 2, i += 2)
    {
      float *D = (1 - FCT(rr, cc) / 2) ? dgrb1 : dgrb0;
      const float wtnw = 1.f / (EPS + fabsf(D[(i - M1) >> 1] - D[(i + M1) >> 1]) + fabsf(D[(i - M1) >> 1] - D[(i - M3) >> 1])
                                + fabsf(D[(i + M1) >> 1] - D[(i - M3) >> 1]));
      const float wtne = 1.f / (EPS + fabsf(D[(i + P1) >> 1] - D[(i - P1) >> 1]) + fabsf(D[(i + P1) >> 1] - D[(i + P3) >> 1])
                                + fabsf(D[(i - P1) >> 1] - D[(i + P3) >> 1]));
      const float wtsw = 1.f / (EPS + fabsf(D[(i - P1) >> 1] - D[(i + P1) >> 1]) + fabsf(D[(i - P1) >> 1] - D[(i + M3) >> 1])
                                + fabsf(D[(i + P1) >> 1] - D[(i - P3) >> 1]));
      const float wtse = 1.f / (EPS + fabsf(D[(i + M1) >> 1] - D[(i - M1) >> 1]) + fabsf(D[(i + M1) >> 1] - D[(i - P3) >> 1])
                                + fabsf(D[(i - M1) >> 1] - D[(i + M3) >> 1]));
      D[i >> 1] = (wtnw * (1.325f * D[(i - M1) >> 1] - 0.175f * D[(i - M3) >> 1] - 0.075f * D[(i - M1 - 2) >> 1] - 0.075f * D[(i - M1 - V2) >> 1])
                   + wtne * (1.325f * D[(i + P1) >> 1] - 0.175f * D[(i + P3) >> 1] - 0.075f * D[(i + P1 + 2) >> 1] - 0.075f * D[(i + P1 + V2) >> 1])
                   + wtsw * (1.325f * D[(i - P1) >> 1] - 0.175f * D[(i - P3) >> 1] - 0.075f * D[(i - P1 - 2) >> 1] - 0.075f * D[(i - P1 - V2) >> 1])
                   + wtse * (1.325f * D[(i + M1) >> 1] - 0.175f * D[(i + M3) >> 1] - 0.075f * D[(i + M1 + 2) >> 1] - 0.075f * D[(i + M1 + V2) >> 1]))
                  / (wtnw + wtne + wtsw + wtse);
    }

  /* S14 output: R and B at green sites from the four neighbours, copied at R/B sites; then green, :1278-1411 */
  for(int rr = 16; rr < rr1 - 16; rr++)
  {
    const int row = rr + top;
    const int gfirst = (FCT(rr, 2) & 1) == 1; /* the photosite at an even tile column is green */
    for(int cc = 16; cc < cc1 - 16; cc++)
    {
      const int col = cc + left, i = rr * TS + cc;
      if(!(col < width && row < height)) continue;
      float *o = out + 4 * ((size_t)row * width + col);
      const int at_green = gfirst ? !(cc & 1) : (cc & 1);
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
#undef FCT
}

int oracle_demosaic_amaze(float *out, const float *in, const dt_hip_roi_t *roi_out, const dt_hip_roi_t *roi_in,
                          const uint32_t filters, const float clip_pt)
{
  (void)roi_out;
  const int width = roi_in->width, height = roi_in->height;
  int ex, ey;
  if(oracle_fc(0, 0, filters) == 1)
  {
    if(oracle_fc(0, 1, filters) == 0) { ey = 0; ex = 1; }
    else { ey = 1; ex = 0; }
  }
  else
  {
    if(oracle_fc(0, 0, filters) == 0) { ey = 0; ex = 0; }
    else { ey = 1; ex = 1; }
  }
  const int nty = (height + 16 + (TS - 32) - 1) / (TS - 32), ntx = (width + 16 + (TS - 32) - 1) / (TS - 32);
  if(g_amaze_persistent)
  {
    tile_t *t = (tile_t *)calloc(1, sizeof(tile_t));
    for(int ty = 0; ty < nty; ty++)
      for(int tx = 0; tx < ntx; tx++)
      {
        const int top = -16 + ty * (TS - 32), left = -16 + tx * (TS - 32);
        if(top < height && left < width) amaze_tile(t, in, out, width, height, top, left, filters, ex, ey, clip_pt);
      }
    free(t);
    return 0;
  }
#pragma omp parallel
  {
    tile_t *t = (tile_t *)malloc(sizeof(tile_t));
#pragma omp for collapse(2) schedule(dynamic)
    for(int ty = 0; ty < nty; ty++)
      for(int tx = 0; tx < ntx; tx++)
      {
        const int top = -16 + ty * (TS - 32), left = -16 + tx * (TS - 32);
        if(top < height && left < width) amaze_tile(t, in, out, width, height, top, left, filters, ex, ey, clip_pt);
      }
    free(t);
  }
  return 0;
}

/* Where the reference's AMaZE output depends on what the previous tile of the same OpenMP thread
 * left in the shared buffer (it only clears the flag plane per tile): a few stencils near the end of
 * a tile's valid region read words the current tile never wrote.  Observed and bounded by comparing
 * the per-tile-zeroed restatement with the persistent one (oracle_amaze_persistent, which equals the
 * reference on one thread bit for bit): the R or B channel of the LAST output row of a tile row
 * (frame rows 127, 255, ...), of the last frame row and of the last frame column, depending on the
 * CFA phase and on the parity of the frame size.  The restatement -- and the device -- zero the buffer
 * per tile, which makes every pixel a function of the frame alone. */
void oracle_amaze_stale_mask(uint8_t *mask, const int width, const int height)
{
  memset(mask, 0, (size_t)width * height);
  for(int j = 0; j < height; j++)
  {
    const int whole_row = (j % (TS - 32) == TS - 33) || j == height - 1;
    for(int i = 0; i < width; i++) mask[(size_t)j * width + i] = (whole_row || i == width - 1) ? 1 : 0;
  }
}
