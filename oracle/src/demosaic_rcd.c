/* oracle/src/demosaic_rcd.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restatement of Ansel's ratio-corrected demosaic for Bayer sensors:
 *   rcd_demosaic()    src/iop/demosaic/rcd.c:274-564
 *   rcd_ppg_border()  src/iop/demosaic/rcd.c:92-272
 * reached from process() in src/iop/demosaic.c:1041-1253 with the roi-shifted filter word
 * (dt_dev_get_roi_filters, src/develop/imageop.c:139-142).
 *
 * The reference works on 112 x 112 tiles (RCD_TILESIZE) of which the inner 94 x 94
 * (RCD_TILEVALID, border RCD_BORDER = 9, RCD_MARGIN = 6 at the image edge) are written out, and
 * the result near a tile border is NOT independent of where the tile sits (VH_Dir is zero on the
 * 4-px tile rim, rcd.c:303-306).  The tile grid is therefore part of the algorithm and is
 * reproduced exactly: same tile origins, same per-tile index arithmetic, including the quirks
 *   - the gradient sums of steps 3.1, 4.2 and 4.3 call fabs(), not fabsf() (rcd.c:414-417,
 *     478-481, 508-527): each |difference| is a double, so `eps + fabs() + fabs() ...` is summed
 *     in binary64 and rounded to binary32 once, at the assignment;
 *   - P/Q_CDiff_Hpf are computed at odd columns only (rcd.c:444-451) and addressed by indx/2
 *     (rcd.c:455-461), so on rows whose red/blue sites sit on odd columns the three samples are
 *     (r-1,c) (r,c) (r+1,c+2) rather than the diagonal;
 *   - PQ_Dir shares its buffer with the low-pass filter `lpf` (rcd.c:318), so PQ_Dir read one row
 *     or column outside the region step 4.1 wrote returns the lpf value there (rcd.c:470-471).
 *
 * Reads of per-thread scratch the current tile never wrote.  The reference reuses its scratch
 * buffers from tile to tile without clearing them (only VH_Dir and, on partial tiles, rgb are
 * memset), and a few output pixels depend on such stale words:
 *   (a) rgb[c] at red/blue sites 3 px outside the region step 4.2 computes, read by step 4.3
 *       for the outermost output row/column of an edge tile (image rows 6 and H-7, cols 6 and W-7).
 *       No tile ever writes those words, so the reference returns whatever the allocation held;
 *   (b) P/Q_CDiff_Hpf at column tileCols-3 when tileCols is even: a partial last tile column then
 *       reads what the previous (full) tile of the same OpenMP thread left there, which reaches
 *       image columns W-9..W-7.  This one depends on thread scheduling: the reference is not
 *       reproducible run to run on those three columns.
 * This restatement defines every such word as 0.0f -- what the reference computes whenever its
 * scratch is fresh.  With the scratch allocator of oracle/_ref zero-filling, the restatement is
 * bit-identical to the reference's own code on every pixel of single-tile and multi-tile images
 * except case (b) (tests/test_oracle_vs_ref.py); oracle_rcd_stale_mask() marks (a) and (b).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stddef.h>
#include "oracle.h"
#include "ppg_core.h"

#define TS 112 /* RCD_TILESIZE, rcd.c:53-55 */
#define RCD_BORDER 9
#define RCD_MARGIN 6
#define TV (TS - 2 * RCD_BORDER) /* RCD_TILEVALID */
#define W1 TS
#define W2 (2 * TS)
#define W3 (3 * TS)
#define W4 (4 * TS)
#define EPS 1e-5f
#define EPSSQ 1e-10f

static inline float sqf(const float x) { return x * x; }
static inline float intp(const float a, const float b, const float c) { return a * (b - c) + c; } /* demosaic.c:250-257 */
static inline float pos(const float a) { return fmaxf(0.0f, a); }
static inline int imin(const int a, const int b) { return a < b ? a : b; }

/* FTZ/DAZ: rcd_demosaic() runs with MXCSR flush-to-zero and denormals-are-zero set
 * (dt_fp_init(DT_FP_MODE_FAST), rcd.c:300; src/system/fp_mode.h:45-58). */
#if defined(__x86_64__)
#include <xmmintrin.h>
static unsigned rcd_fast_fp_enter(void)
{
  const unsigned old = _mm_getcsr();
  _mm_setcsr(old | _MM_FLUSH_ZERO_ON | 0x0040u);
  return old;
}
static void rcd_fast_fp_leave(const unsigned old) { _mm_setcsr(old); }
#else
static unsigned rcd_fast_fp_enter(void) { return 0; }
static void rcd_fast_fp_leave(const unsigned old) { (void)old; }
#endif

/* ---- border: rcd_ppg_border(), rcd.c:92-272: the per-pixel PPG of ppg_core.h with clamped
 * samples, kept for the outer RCD_MARGIN ring only (the tiles overwrite the rest) ---------- */

/* ---- one RCD tile ------------------------------------------------------------------- */

typedef struct
{
  float cfa[TS * TS];
  float vh[TS * TS];      /* VH_Dir */
  float g[TS * TS / 2];   /* rgb[1] at red/blue sites (indx / 2); = cfa where step 3.1 did not run */
  float x[TS * TS / 2];   /* lpf, then PQ_Dir (same buffer, rcd.c:318) */
  float co[TS * TS / 2];  /* rgb[2 - FC] at red/blue sites: the step 4.2 result, 0 elsewhere */
} rcd_tile_t;

static inline float hpf(const float *c, const int i, const int s)
{
  return sqf((c[i - 3 * s] - c[i - s] - c[i + s] + c[i + 3 * s]) - 3.0f * (c[i - 2 * s] + c[i + 2 * s]) + 6.0f * c[i]);
}

/* P/Q_CDiff_Hpf word `idx2` (rcd.c:444-451): the value computed at (row, odd col) with
 * idx2 == (row * TS + col) / 2, or 0 where this tile's step 4.0 does not write it */
static inline float pq_hpf(const float *cfa, const int idx2, const int tileRows, const int tileCols, const int s)
{
  const int row = idx2 / (TS / 2);
  const int col = 2 * (idx2 - row * (TS / 2)) + 1;
  if(row < 3 || row >= tileRows - 3 || col < 3 || col >= tileCols - 3) return 0.0f;
  return hpf(cfa, row * TS + col, s);
}

static void rcd_tile(rcd_tile_t *t, float *out, const float *in, const int width, const int height,
                     const uint32_t filters, const float scaler, const float revscaler, const int tile_vertical,
                     const int tile_horizontal, const int num_vertical, const int num_horizontal)
{
  const int rowStart = tile_vertical * TV, rowEnd = imin(rowStart + TS, height);
  const int colStart = tile_horizontal * TV, colEnd = imin(colStart + TS, width);
  const int tileRows = imin(rowEnd - rowStart, TS), tileCols = imin(colEnd - colStart, TS);
  float *const cfa = t->cfa, *const vh = t->vh, *const g = t->g, *const x = t->x, *const co = t->co;

  memset(vh, 0, sizeof(t->vh));
  memset(co, 0, sizeof(t->co));
  memset(x, 0, sizeof(t->x));
  memset(cfa, 0, sizeof(t->cfa));
  memset(g, 0, sizeof(t->g));

  /* step 0, rcd.c:345-354: safe_in() = fmaxf(0, a) * scale; rgb[1] starts as cfa everywhere */
  for(int row = rowStart; row < rowEnd; row++)
    for(int col = colStart, indx = (row - rowStart) * TS; col < colEnd; col++, indx++)
      cfa[indx] = pos(in[(size_t)row * width + col]) * revscaler;
  for(int row = 0; row < tileRows; row++)
    for(int col = (oracle_fc(row, 0, filters) & 1); col < tileCols; col += 2) g[(row * TS + col) / 2] = cfa[row * TS + col];

  /* step 1, rcd.c:356-390: VH_Dir = V_Stat / (V_Stat + H_Stat) */
  for(int row = 4; row < tileRows - 4; row++)
    for(int col = 4, indx = row * TS + col; col < tileCols - 4; col++, indx++)
    {
      const float V_Stat = fmaxf(EPSSQ, hpf(cfa, indx - W1, W1) + hpf(cfa, indx, W1) + hpf(cfa, indx + W1, W1));
      const float H_Stat = fmaxf(EPSSQ, hpf(cfa, indx - 1, 1) + hpf(cfa, indx, 1) + hpf(cfa, indx + 1, 1));
      vh[indx] = V_Stat / (V_Stat + H_Stat);
    }

  /* step 2.1, rcd.c:394-402: low-pass filter at red/blue sites */
  for(int row = 2; row < tileRows - 2; row++)
    for(int col = 2 + (oracle_fc(row, 0, filters) & 1), indx = row * TS + col; col < tileCols - 2; col += 2, indx += 2)
      x[indx / 2] = cfa[indx] + 0.5f * (cfa[indx - W1] + cfa[indx + W1] + cfa[indx - 1] + cfa[indx + 1])
                    + 0.25f * (cfa[indx - W1 - 1] + cfa[indx - W1 + 1] + cfa[indx + W1 - 1] + cfa[indx + W1 + 1]);

  /* step 3.1, rcd.c:406-440: green at red/blue sites */
  for(int row = 4; row < tileRows - 4; row++)
    for(int col = 4 + (oracle_fc(row, 0, filters) & 1), indx = row * TS + col; col < tileCols - 4; col += 2, indx += 2)
    {
      const int lpindx = indx / 2;
      const float cfai = cfa[indx];
      const float N_Grad = EPS + fabs(cfa[indx - W1] - cfa[indx + W1]) + fabs(cfai - cfa[indx - W2]) + fabs(cfa[indx - W1] - cfa[indx - W3]) + fabs(cfa[indx - W2] - cfa[indx - W4]);
      const float S_Grad = EPS + fabs(cfa[indx - W1] - cfa[indx + W1]) + fabs(cfai - cfa[indx + W2]) + fabs(cfa[indx + W1] - cfa[indx + W3]) + fabs(cfa[indx + W2] - cfa[indx + W4]);
      const float W_Grad = EPS + fabs(cfa[indx - 1] - cfa[indx + 1]) + fabs(cfai - cfa[indx - 2]) + fabs(cfa[indx - 1] - cfa[indx - 3]) + fabs(cfa[indx - 2] - cfa[indx - 4]);
      const float E_Grad = EPS + fabs(cfa[indx - 1] - cfa[indx + 1]) + fabs(cfai - cfa[indx + 2]) + fabs(cfa[indx + 1] - cfa[indx + 3]) + fabs(cfa[indx + 2] - cfa[indx + 4]);
      const float lpfi = x[lpindx];
      const float N_Est = cfa[indx - W1] * (lpfi + lpfi) / (EPS + lpfi + x[lpindx - W1]);
      const float S_Est = cfa[indx + W1] * (lpfi + lpfi) / (EPS + lpfi + x[lpindx + W1]);
      const float W_Est = cfa[indx - 1] * (lpfi + lpfi) / (EPS + lpfi + x[lpindx - 1]);
      const float E_Est = cfa[indx + 1] * (lpfi + lpfi) / (EPS + lpfi + x[lpindx + 1]);
      const float V_Est = (S_Grad * N_Est + N_Grad * S_Est) / (N_Grad + S_Grad);
      const float H_Est = (W_Grad * E_Est + E_Grad * W_Est) / (E_Grad + W_Grad);
      const float VH_Central_Value = vh[indx];
      const float VH_Neighbourhood_Value = 0.25f * (vh[indx - W1 - 1] + vh[indx - W1 + 1] + vh[indx + W1 - 1] + vh[indx + W1 + 1]);
      const float VH_Disc = (fabs(0.5f - VH_Central_Value) < fabs(0.5f - VH_Neighbourhood_Value)) ? VH_Neighbourhood_Value : VH_Central_Value;
      g[lpindx] = intp(VH_Disc, H_Est, V_Est);
    }

  /* steps 4.0 + 4.1, rcd.c:444-463: PQ_Dir written over lpf */
  for(int row = 4; row < tileRows - 4; row++)
    for(int col = 4 + (oracle_fc(row, 0, filters) & 1), indx = row * TS + col; col < tileCols - 4; col += 2, indx += 2)
    {
      const int indx2 = indx / 2, indx3 = (indx - W1 - 1) / 2, indx4 = (indx + W1 - 1) / 2;
      const float P_Stat = fmaxf(EPSSQ, pq_hpf(cfa, indx3, tileRows, tileCols, W1 + 1) + pq_hpf(cfa, indx2, tileRows, tileCols, W1 + 1) + pq_hpf(cfa, indx4 + 1, tileRows, tileCols, W1 + 1));
      const float Q_Stat = fmaxf(EPSSQ, pq_hpf(cfa, indx3 + 1, tileRows, tileCols, W1 - 1) + pq_hpf(cfa, indx2, tileRows, tileCols, W1 - 1) + pq_hpf(cfa, indx4, tileRows, tileCols, W1 - 1));
      x[indx2] = P_Stat / (P_Stat + Q_Stat);
    }

  /* step 4.2, rcd.c:466-496: red at blue sites and blue at red sites.  rgb[c] at the diagonal
   * neighbours and rgb[1] two sites away are native samples / step-3.1 greens. */
  for(int row = 4; row < tileRows - 4; row++)
    for(int col = 4 + (oracle_fc(row, 0, filters) & 1), indx = row * TS + col; col < tileCols - 4; col += 2, indx += 2)
    {
      const int pqindx = indx / 2, pqindx2 = (indx - W1 - 1) / 2, pqindx3 = (indx + W1 - 1) / 2;
      const float PQ_Central_Value = x[pqindx];
      const float PQ_Neighbourhood_Value = 0.25f * (x[pqindx2] + x[pqindx2 + 1] + x[pqindx3] + x[pqindx3 + 1]);
      const float PQ_Disc = (fabs(0.5f - PQ_Central_Value) < fabs(0.5f - PQ_Neighbourhood_Value)) ? PQ_Neighbourhood_Value : PQ_Central_Value;
#define G1(o) g[(indx + (o)) / 2]
      const float NW_Grad = EPS + fabs(cfa[indx - W1 - 1] - cfa[indx + W1 + 1]) + fabs(cfa[indx - W1 - 1] - cfa[indx - W3 - 3]) + fabs(G1(0) - G1(-W2 - 2));
      const float NE_Grad = EPS + fabs(cfa[indx - W1 + 1] - cfa[indx + W1 - 1]) + fabs(cfa[indx - W1 + 1] - cfa[indx - W3 + 3]) + fabs(G1(0) - G1(-W2 + 2));
      const float SW_Grad = EPS + fabs(cfa[indx - W1 + 1] - cfa[indx + W1 - 1]) + fabs(cfa[indx + W1 - 1] - cfa[indx + W3 - 3]) + fabs(G1(0) - G1(W2 - 2));
      const float SE_Grad = EPS + fabs(cfa[indx - W1 - 1] - cfa[indx + W1 + 1]) + fabs(cfa[indx + W1 + 1] - cfa[indx + W3 + 3]) + fabs(G1(0) - G1(W2 + 2));
      const float NW_Est = cfa[indx - W1 - 1] - G1(-W1 - 1);
      const float NE_Est = cfa[indx - W1 + 1] - G1(-W1 + 1);
      const float SW_Est = cfa[indx + W1 - 1] - G1(W1 - 1);
      const float SE_Est = cfa[indx + W1 + 1] - G1(W1 + 1);
      const float P_Est = (NW_Grad * SE_Est + SE_Grad * NW_Est) / (NW_Grad + SE_Grad);
      const float Q_Est = (NE_Grad * SW_Est + SW_Grad * NE_Est) / (NE_Grad + SW_Grad);
      co[pqindx] = G1(0) + intp(PQ_Disc, Q_Est, P_Est);
#undef G1
    }

  /* step 4.3 + output, rcd.c:499-555.  Only pixels that are written out are evaluated. */
  const int first_vertical = rowStart + ((tile_vertical == 0) ? RCD_MARGIN : RCD_BORDER);
  const int last_vertical = rowEnd - ((tile_vertical == num_vertical - 1) ? RCD_MARGIN : RCD_BORDER);
  const int first_horizontal = colStart + ((tile_horizontal == 0) ? RCD_MARGIN : RCD_BORDER);
  const int last_horizontal = colEnd - ((tile_horizontal == num_horizontal - 1) ? RCD_MARGIN : RCD_BORDER);
  for(int irow = first_vertical; irow < last_vertical; irow++)
    for(int icol = first_horizontal; icol < last_horizontal; icol++)
    {
      const int row = irow - rowStart, col = icol - colStart, indx = row * TS + col;
      const int f = oracle_fc(row, col, filters);
      float rgb[3];
      if(f & 1)
      {
        /* a green site; step 4.3 ran here iff 4 <= row < tileRows - 4 and 4 <= col < tileCols - 4,
         * which holds for every output pixel (margin >= 6) */
        const float VH_Central_Value = vh[indx];
        const float VH_Neighbourhood_Value = 0.25f * (vh[indx - W1 - 1] + vh[indx - W1 + 1] + vh[indx + W1 - 1] + vh[indx + W1 + 1]);
        const float VH_Disc = (fabs(0.5f - VH_Central_Value) < fabs(0.5f - VH_Neighbourhood_Value)) ? VH_Neighbourhood_Value : VH_Central_Value;
        const float rgb1 = cfa[indx];
        const float N1 = EPS + fabs(rgb1 - cfa[indx - W2]);
        const float S1 = EPS + fabs(rgb1 - cfa[indx + W2]);
        const float W1_ = EPS + fabs(rgb1 - cfa[indx - 2]);
        const float E1 = EPS + fabs(rgb1 - cfa[indx + 2]);
        const float rgb1mw1 = g[(indx - W1) / 2], rgb1pw1 = g[(indx + W1) / 2];
        const float rgb1m1 = g[(indx - 1) / 2], rgb1p1 = g[(indx + 1) / 2];
        /* colour of the horizontal neighbours is native there; the other colour comes from step 4.2 */
        const int ch = oracle_fc(row, col + 1, filters); /* 0 or 2 */
        rgb[1] = rgb1;
        for(int c = 0; c <= 2; c += 2)
        {
          const int hnative = (c == ch);
          const float cN1 = hnative ? co[(indx - W1) / 2] : cfa[indx - W1];
          const float cS1 = hnative ? co[(indx + W1) / 2] : cfa[indx + W1];
          const float cN3 = hnative ? co[(indx - W3) / 2] : cfa[indx - W3];
          const float cS3 = hnative ? co[(indx + W3) / 2] : cfa[indx + W3];
          const float cW1 = hnative ? cfa[indx - 1] : co[(indx - 1) / 2];
          const float cE1 = hnative ? cfa[indx + 1] : co[(indx + 1) / 2];
          const float cW3 = hnative ? cfa[indx - 3] : co[(indx - 3) / 2];
          const float cE3 = hnative ? cfa[indx + 3] : co[(indx + 3) / 2];
          const float SNabs = fabs(cN1 - cS1);
          const float EWabs = fabs(cW1 - cE1);
          const float N_Grad = N1 + SNabs + fabs(cN1 - cN3);
          const float S_Grad = S1 + SNabs + fabs(cS1 - cS3);
          const float W_Grad = W1_ + EWabs + fabs(cW1 - cW3);
          const float E_Grad = E1 + EWabs + fabs(cE1 - cE3);
          const float N_Est = cN1 - rgb1mw1;
          const float S_Est = cS1 - rgb1pw1;
          const float W_Est = cW1 - rgb1m1;
          const float E_Est = cE1 - rgb1p1;
          const float V_Est = (N_Grad * S_Est + S_Grad * N_Est) / (N_Grad + S_Grad);
          const float H_Est = (E_Grad * W_Est + W_Grad * E_Est) / (E_Grad + W_Grad);
          rgb[c] = rgb1 + intp(VH_Disc, H_Est, V_Est);
        }
      }
      else
      {
        rgb[f] = cfa[indx];
        rgb[1] = g[indx / 2];
        rgb[2 - f] = co[indx / 2];
      }
      float *o = out + ((size_t)irow * width + icol) * 4;
      o[0] = scaler * fmaxf(0.0f, rgb[0]);
      o[1] = scaler * fmaxf(0.0f, rgb[1]);
      o[2] = scaler * fmaxf(0.0f, rgb[2]);
      o[3] = 0.0f;
    }
}

static int rcd_run(float *out, const float *in, const int width, const int height, const uint32_t filters,
                   const float processed_maximum[4])
{
  if(width < 16 || height < 16) return 0; /* rcd.c:280-284: logs and leaves the output untouched */
  /* border first (rcd.c:286), then the tiles overwrite everything from row/col RCD_MARGIN on */
  const ppg_ctx_t k = { in, width, height, width, height, 0, 0, filters, 1, NULL };
  for(int j = 0; j < height; j++)
    for(int i = 0; i < width; i++)
    {
      if(!ppg_ring_lt(&k, j, i, RCD_MARGIN))
      {
        i = width - RCD_MARGIN - 1;
        continue;
      }
      float *o = out + ((size_t)j * width + i) * 4;
      ppg_pixel(&k, j, i, o);
      if(!ppg_ring_lt(&k, j, i, 3)) o[3] = 0.0f; /* rcd.c:190: alpha written by pass 2 only */
    }
  const float scaler = fmaxf(processed_maximum[0], fmaxf(processed_maximum[1], processed_maximum[2]));
  const float revscaler = 1.0f / scaler;
  const int num_vertical = 1 + (height - 2 * RCD_BORDER - 1) / TV;
  const int num_horizontal = 1 + (width - 2 * RCD_BORDER - 1) / TV;
  rcd_tile_t *t = (rcd_tile_t *)malloc(sizeof(rcd_tile_t));
  if(!t) return 1;
  const unsigned old = rcd_fast_fp_enter();
  for(int tv = 0; tv < num_vertical; tv++)
    for(int th = 0; th < num_horizontal; th++)
      rcd_tile(t, out, in, width, height, filters, scaler, revscaler, tv, th, num_vertical, num_horizontal);
  rcd_fast_fp_leave(old);
  free(t);
  return 0;
}

/* Marks (1) the output pixels whose value in the reference depends on scratch words the tile
 * never wrote -- see the header comment.  mask is width*height bytes. */
void oracle_rcd_stale_mask(uint8_t *mask, const int width, const int height, const uint32_t filters)
{
  memset(mask, 0, (size_t)width * height);
  if(width < 16 || height < 16) return;
  const int num_horizontal = 1 + (width - 2 * RCD_BORDER - 1) / TV;
  const int lastColStart = (num_horizontal - 1) * TV;
  const int lastCols = imin(width - lastColStart, TS);
  for(int j = RCD_MARGIN; j < height - RCD_MARGIN; j++)
    for(int i = RCD_MARGIN; i < width - RCD_MARGIN; i++)
    {
      const int green = oracle_fc(j, i, filters) & 1;
      int m = 0;
      if(green && (j == RCD_MARGIN || j == height - RCD_MARGIN - 1 || i == RCD_MARGIN || i == width - RCD_MARGIN - 1)) m = 1; /* (a) */
      if((lastCols % 2 == 0) && lastCols < TS && num_horizontal > 1 && i >= width - 9 && i <= width - 7) m = 1; /* (b) */
      mask[(size_t)j * width + i] = (uint8_t)m;
    }
}

uint32_t oracle_shift_dcraw_filters(uint32_t filters, uint32_t x, uint32_t y)
{
  /* dt_rawspeed_crop_dcraw_filters -> rawspeed ColorFilterArray::shiftDcrawFilter
   * (src/imageio/imageio_rawspeed.cc:146-151; rawspeed is not vendored).  Pinned by the identity
   * FC(r + y, c + x, f) == FC(r, c, shifted), tests/test_filters.py. */
  if(!filters || filters == 9u) return filters;
  uint32_t out = 0;
  for(int r = 0; r < 8; r++)
    for(int c = 0; c < 2; c++)
      out |= (uint32_t)oracle_fc(r + y, c + x, filters) << ((((r << 1) & 14) + (c & 1)) << 1);
  return out;
}

int oracle_demosaic_ppg(float *out, const float *in, const dt_hip_roi_t *roi_out, const dt_hip_roi_t *roi_in, uint32_t filters,
                        float median_thrs);
int oracle_demosaic_amaze(float *out, const float *in, const dt_hip_roi_t *roi_out, const dt_hip_roi_t *roi_in,
                          uint32_t filters, float clip_pt);

static int demosaic_methods(const dt_hip_piece_t *piece, const dt_hip_demosaic_data_t *d, const float *in, void *out)
{
  const uint32_t filters = oracle_shift_dcraw_filters(piece->filters, piece->roi_in.x, piece->roi_in.y);
  if(d->demosaicing_method == DT_HIP_DEMOSAIC_RCD)
    return rcd_run((float *)out, (const float *)in, piece->roi_in.width, piece->roi_in.height, filters,
                   piece->processed_maximum);
  if(d->demosaicing_method == DT_HIP_DEMOSAIC_PPG)
  {
    dt_hip_roi_t roo = piece->roi_out;
    roo.x = roo.y = 0;
    return oracle_demosaic_ppg((float *)out, (const float *)in, &roo, &piece->roi_in, filters, d->median_thrs);
  }
  if(d->demosaicing_method == DT_HIP_DEMOSAIC_AMAZE)
  {
    /* amaze.cc:191-192 */
    const float clip_pt = fminf(piece->processed_maximum[0], fminf(piece->processed_maximum[1], piece->processed_maximum[2]));
    dt_hip_roi_t roo = piece->roi_out;
    roo.x = roo.y = 0;
    return oracle_demosaic_amaze((float *)out, (const float *)in, &roo, &piece->roi_in, filters, clip_pt);
  }
  return 1;
}

/* green_equilibration_lavg(), src/iop/demosaic/basic.c:248-293: a green site on the lattice anchored at (oj, oi) is
 * scaled by mean(diagonal greens) / mean(axial greens at distance 2) where both neighbourhoods are flat (mean absolute
 * difference of the six pairs below thr) and the site is below 0.95.  Sums run in the reference's operand order. */
static float mean_abs_pairs(const float g[4])
{
  /* pairs (0,1) (0,2) (0,3) (1,2) (2,3) (1,3), accumulated left to right */
  static const int pa[6] = { 0, 0, 0, 1, 2, 1 }, pb[6] = { 1, 2, 3, 2, 3, 3 };
  float acc = fabsf(g[pa[0]] - g[pb[0]]);
  for(int k = 1; k < 6; k++) acc = acc + fabsf(g[pa[k]] - g[pb[k]]);
  return acc / 6.0f;
}

/* green_equilibration_favg(), src/iop/demosaic/basic.c:296-329: the greens of the even rows are scaled by
 * sum(greens of the odd rows) / sum(greens of the even rows), both sums in binary64 over the 2x2 cells of the frame.
 * The reference adds inside an OpenMP reduction: the low bits of its sums follow the thread count.  This is the sum in
 * index order (the reference on one thread); tests allow 1 ulp of binary32 per pixel against either. */
static void green_eq_favg(float *out, const float *in, const int width, const int height, const uint32_t filters, const int x,
                          const int y)
{
  const int oi = (oracle_fc(y, x, filters) & 1) != 1 ? 1 : 0;
  const int g2_offset = oi ? -1 : 1;
  double sum1 = 0.0, sum2 = 0.0;
  memcpy(out, in, sizeof(float) * (size_t)width * height);
  for(int j = 0; j < height - 1; j += 2)
    for(int i = oi; i < width - 1 - g2_offset; i += 2)
    {
      sum1 += in[(size_t)j * width + i];
      sum2 += in[(size_t)(j + 1) * width + i + g2_offset];
    }
  if(!(sum1 > 0.0 && sum2 > 0.0)) return;
  const double gr_ratio = sum2 / sum1;
  for(int j = 0; j < height - 1; j += 2)
    for(int i = oi; i < width - 1 - g2_offset; i += 2) out[(size_t)j * width + i] = in[(size_t)j * width + i] * gr_ratio;
}

static void green_eq_lavg(float *out, const float *in, const int width, const int height, const uint32_t filters, const int x,
                          const int y, const float thr)
{
  /* the lattice origin: the first green of the second kind at or after (2, 2), basic.c:253-256 */
  int oj = 2, oi = 2;
  if(oracle_fc(oj + y, oi + x, filters) != 1) oj++;
  if(oracle_fc(oj + y, oi + x, filters) != 1) oi++;
  if(oracle_fc(oj + y, oi + x, filters) != 1) oj--;
  memcpy(out, in, sizeof(float) * (size_t)width * height);
#pragma omp parallel for
  for(int j = oj; j < height - 2; j += 2)
    for(int i = oi; i < width - 2; i += 2)
    {
      const float *c = in + (size_t)j * width + i;
      const float diag[4] = { c[-width - 1], c[-width + 1], c[width - 1], c[width + 1] };
      const float axial[4] = { c[-2 * width], c[2 * width], c[-2], c[2] };
      const float m_diag = (((diag[0] + diag[1]) + diag[2]) + diag[3]) / 4.0f;
      const float m_axial = (((axial[0] + axial[1]) + axial[2]) + axial[3]) / 4.0f;
      if(!(m_axial > 0.0f) || !(m_diag > 0.0f) || !(m_diag / m_axial < 2.0f)) continue;
      if((c[0] < 0.95f) && (mean_abs_pairs(diag) < thr) && (mean_abs_pairs(axial) < thr))
        out[(size_t)j * width + i] = c[0] * m_diag / m_axial;
    }
}

/* color_smoothing(), src/iop/demosaic/basic.c:191-243: `passes` x (red, blue): the channel minus green goes through a
 * 3x3 median, found with the reference's 19 compare-exchanges in its order (which decides where a NaN ends up) */
static const unsigned char k_median9[19][2] = { { 1, 2 }, { 4, 5 }, { 7, 8 }, { 0, 1 }, { 3, 4 }, { 6, 7 }, { 1, 2 },
                                                { 4, 5 }, { 7, 8 }, { 0, 3 }, { 5, 8 }, { 4, 7 }, { 3, 6 }, { 1, 4 },
                                                { 2, 5 }, { 4, 7 }, { 4, 2 }, { 6, 4 }, { 4, 2 } };

void oracle_color_smoothing(float *out, int width, int height, int passes);
static void color_smoothing(float *out, const int width, const int height, const int passes)
{
  oracle_color_smoothing(out, width, height, passes);
}
void oracle_color_smoothing(float *out, const int width, const int height, const int passes)
{
  const size_t npix = (size_t)width * height;
  for(int pass = 0; pass < passes; pass++)
    for(int ch = 0; ch < 3; ch += 2)
    {
      /* the channel is parked in alpha for the whole frame first, :198-204 */
      for(size_t k = 0; k < npix; k++) out[4 * k + 3] = out[4 * k + ch];
#pragma omp parallel for
      for(int j = 1; j < height - 1; j++)
        for(int i = 1; i < width - 1; i++)
        {
          float *px = out + 4 * ((size_t)j * width + i);
          float v[9];
          for(int dj = -1, n = 0; dj <= 1; dj++)
            for(int di = -1; di <= 1; di++, n++)
            {
              const float *q = px + 4 * ((ptrdiff_t)dj * width + di);
              v[n] = q[3] - q[1];
            }
          for(int e = 0; e < 19; e++)
          {
            const int lo = k_median9[e][0], hi = k_median9[e][1];
            if(v[lo] > v[hi])
            {
              const float t = v[lo];
              v[lo] = v[hi];
              v[hi] = t;
            }
          }
          px[ch] = fmaxf(v[4] + px[1], 0.0f);
        }
    }
}

/* process(), src/iop/demosaic.c:1041-1253, Bayer branch: [green equilibration, local average] -> demosaic ->
 * [colour smoothing]. */
int oracle_demosaic_vng4(float *out, const float *in, int width, int height, int rx, int ry, uint32_t filters);
int oracle_dual_demosaic(float *rgb, const float *raw, int width, int height, int rx, int ry, uint32_t filters, float dual_threshold,
                         const float wb[4]);
int oracle_demosaic(const dt_hip_piece_t *piece, const dt_hip_demosaic_data_t *d, const void *in_, void *out)
{
  if(d->green_eq > 3 || d->color_smoothing > 5) return 1;
  if(d->median_thrs != 0.0f && d->demosaicing_method != DT_HIP_DEMOSAIC_PPG) return 1;
  if(!piece->filters || piece->filters == 9u) return 1;
  const float *in = (const float *)in_;
  float *geq = NULL;
  float *aux = NULL;
  if(d->green_eq)
  {
    /* demosaic.c:1137-1163 */
    const size_t bytes = sizeof(float) * (size_t)piece->roi_in.width * piece->roi_in.height;
    geq = (float *)malloc(bytes);
    if(d->green_eq == 3) aux = (float *)malloc(bytes);
    if(!geq || (d->green_eq == 3 && !aux)) return 1;
    if(d->green_eq >= 2)
      green_eq_favg(aux ? aux : geq, in, piece->roi_in.width, piece->roi_in.height, piece->filters, piece->roi_in.x, piece->roi_in.y);
    if(d->green_eq & 1)
      green_eq_lavg(geq, aux ? aux : in, piece->roi_in.width, piece->roi_in.height, piece->filters, piece->roi_in.x,
                    piece->roi_in.y, d->green_eq_threshold);
    in = geq;
  }
  int rc;
  if(d->demosaicing_method == DT_HIP_DEMOSAIC_PASSTHROUGH_MONOCHROME || d->demosaicing_method == DT_HIP_DEMOSAIC_PASSTHROUGH_COLOR)
  {
    /* passthrough_monochrome() / passthrough_color(), src/iop/demosaic/passthrough.c:21-67, called in front of the Bayer
     * branch (demosaic.c:1111-1118): the mosaic as the module received it, whatever green_eq says; the colour of a photosite
     * from the unshifted filters at the OUTPUT position (process() zeroes roo.x / roo.y: the roi offset plays no part);
     * channel 3 is not written */
    const float *const px = (const float *)in_;
    float *const o = (float *)out;
    const int w = piece->roi_out.width, h = piece->roi_out.height;
    for(int row = 0; row < h; row++)
      for(int col = 0; col < w; col++)
      {
        const float v = px[(size_t)row * piece->roi_in.width + col];
        float *const q = o + 4 * ((size_t)row * w + col);
        if(d->demosaicing_method == DT_HIP_DEMOSAIC_PASSTHROUGH_MONOCHROME)
          q[0] = q[1] = q[2] = v;
        else
        {
          q[0] = q[1] = q[2] = 0.0f;
          q[oracle_fc(row, col, piece->filters)] = v;
        }
      }
    rc = 0;
  }
  else if(d->demosaicing_method == DT_HIP_DEMOSAIC_VNG4)
    rc = oracle_demosaic_vng4((float *)out, in, piece->roi_in.width, piece->roi_in.height, piece->roi_in.x, piece->roi_in.y,
                              piece->filters);
  else if(d->demosaicing_method & DT_HIP_DEMOSAIC_DUAL)
  {
    /* demosaic.c:1243-1247: the high-frequency method, then the blend with VNG4 of the mosaic as the module received it */
    dt_hip_demosaic_data_t base = *d;
    base.demosaicing_method = d->demosaicing_method & ~DT_HIP_DEMOSAIC_DUAL;
    rc = demosaic_methods(piece, &base, in, out);
    if(rc == 0)
      rc = oracle_dual_demosaic((float *)out, (const float *)in_, piece->roi_in.width, piece->roi_in.height, piece->roi_in.x,
                                piece->roi_in.y, piece->filters, d->dual_thrs, d->wb_coeffs);
  }
  else
    rc = demosaic_methods(piece, d, in, out);
  free(geq);
  free(aux);
  if(rc == 0 && d->color_smoothing) color_smoothing((float *)out, piece->roi_out.width, piece->roi_out.height, (int)d->color_smoothing);
  return rc;
}
