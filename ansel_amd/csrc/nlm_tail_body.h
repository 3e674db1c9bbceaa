// nlm_tail_body.h -- the rows of a TALL non-local-means chunk behind its 64th (round 5).
//
// The reference's chunk height is a function of the frame (compute_slice_height(), src/pixel/nlmeans_core.c:264-295): 69
// rows at 24 MP (6000 x 4000), 68 at 42 MP, 67 at 150 MP.  The fused chunk kernel (nlm3_body.h FUSED, nlm_chunks_v4) holds
// at most 64 rows in one workgroup -- 69 rows x 8 lanes of its weights' role are nine waves (2 + 6 + 9 = 17 > 16) and three
// tables + the window 169 KB -- so until round 5 those frames took the second version at half the rate per pixel.  Now a
// tall chunk is cut where the running sums allow it: the recurrence down the rows (nlmeans_core.c:437-488) is one sum per
// table column and offset, so the fused kernel runs the chunk's first 64 rows unchanged ("head": same window, same tables,
// same chains) and its column-recurrence waves EXPORT, per offset, the 72 + 4 column sums they hold behind row 63
// (304 bytes); this body continues them through the 1 - 5 rows that are left ("tail"), every term, sum and weight formed
// by the same operations on the same operands in the same order as the bodies of nlm2_body.h / nlm3_body.h form them.
//
// One workgroup of 512 threads per interior chunk, four stages in flight, one barrier per offset:
//   (a) offset s      waves 2-7   a lane = (table column x, tail row j): the term D(row + P) - D(row - P - 1) of the column
//                                 recurrence (the lane's own two pixels stay in registers for the chunk)      -> T[s % 3]
//   (b) offset s - 1  wave 1   a lane = a table column (its first twelve lanes: two): the exported sum + its <= 5 terms, in place
//   (c) offset s - 2  wave 0   eight lanes = a tail row: the sliding row sum (:405-415), the chain passed from lane to lane
//                                 by a DPP shift as in the fused chunk kernel                                  -> D[s % 2]
//   (d) offset s - 3  waves 2-7   a lane = a pixel: weight 2^-(distortion x sharpness), four accumulations (:416-436)
// then the normalisation and the blend (:490-521).  A workgroup needs ~35 KB of LDS (the window is 19 + <= 5 rows): four of
// them share a CU and hide each other's row chain.  Compiled for the host by tests/native/nlm2_host.cpp like the others.
#pragma once

#include "nlm2_body.h"

#define NLT_THREADS 512
#define NLT_WP 92         // window pitch in pixels (chunk width + 2 x reach <= 92, as nlm2_body.h's tight layout)
#define NLT_TP 80         // table pitch in floats: slots 0 .. chunk width + 4
#define NLT_ROWS 5        // most tail rows: chunks are at most 69 rows high, the head takes 64
#define NLT_SEED_PITCH 80 // floats per offset in the export: slot x of the table at word x
#define NLT_HEAD_ROWS 64

namespace nlmt
{
using nlm2::f2;
using nlm2::imin;

// LDS floats of one workgroup
inline size_t lds_floats(const int tail_rows, const int reach, const int npatch)
{
  const int wh = tail_rows + 2 * reach;
  return (size_t)wh * NLT_WP * 3 + 3 * NLT_ROWS * NLT_TP + 2 * NLT_ROWS * NLT_TP + 3 * ((npatch + 3) & ~3);
}

template <int P = 2> inline bool fits(const int chk_w, const int chk_h, const int radius, const int reach, const int npatch)
{
  return radius == P && chk_h > NLT_HEAD_ROWS && chk_h - NLT_HEAD_ROWS <= NLT_ROWS && chk_w + 2 * reach <= NLT_WP && chk_w + 2 * P + 1 <= NLT_TP
         && reach >= P + 1 && (chk_w + 2 * P) * NLT_ROWS <= NLT_THREADS - 128 && npatch <= 4096;
}

// Env: tid(), bid(), lds(), sync(), lane_shr1(), cvt_i32_sat(), int_as_float(), max_num().  Args: nlm_args of nlmeans.hip.
// seeds: this chunk's exported column sums, [npatch][NLT_SEED_PITCH].
// BORDER: a chunk of the outermost ring (the head ran nlm3::body<..., BORDER, FUSED, TALL> on it).  As there, all clipping is
// "this squared difference is not there": it is +0 where the row or column of one of its two pixels lies outside the frame
// (nlm3_body.h, BORDER: why that reproduces init_column_sums() and the three branches of nlmeans_core.c:437-488), a pixel whose
// shifted pixel lies outside gets weight +0 for that offset (:398-404), and the window holds zeros there.  The chunk may be
// narrower / lower than the grid's.
// P, CENTER (round 6): patch radius 1 beside 2; the weight with the centre pixel's term (nlm3_body.h CENTER) in stage (d).
template <bool BORDER = false, int P = 2, bool CENTER = false, class Env, class Args, class F4, class I2>
NLM2_FN void body(const Env &env, const F4 *__restrict__ in, F4 *__restrict__ out, const Args &a, const I2 *__restrict__ patches,
                  const float *__restrict__ seeds)
{
  constexpr int S = 2 * P + 1, WP = NLT_WP, TP = NLT_TP;
  const int tid = env.tid();
  const int W = a.W, H = a.H;
  const int cy_launch = env.bid() / a.nchx, cx = env.bid() - cy_launch * a.nchx;
  const int cy = cy_launch + a.cy0;
  const int top = cy * a.chk_h, left = cx * a.chk_w;
  const int bot = imin(top + a.chk_h, H), right = imin(left + a.chk_w, W);
  const int ch = bot - top, cw = right - left;
  const int reach = a.reach;
  // interior (the test of nlm3::body()): the chunk is whole and no patch of any offset reaches past the frame
  const bool interior = top >= reach && bot + reach <= H && left >= reach && right + reach <= W && ch == a.chk_h && cw == a.chk_w;
  if(BORDER ? (interior || ch <= NLT_HEAD_ROWS || cw < 1) : !interior) return;
  const int R0 = NLT_HEAD_ROWS, TR = ch - R0; // the tail: chunk rows R0 .. ch - 1
  const int n = a.npatch;
  const int ncol = cw + 2 * P; // table slots 1 .. ncol (slot x = frame column left - P - 1 + x); slot 0 is never summed

  float *const lds = env.lds();
  const int wh = TR + 2 * reach; // window row wy = frame row top + R0 - reach + wy, window column wx = frame column left - reach + wx
  f2 *const XY = (f2 *)lds;                  // [wh][WP]
  float *const Z = lds + 2 * wh * WP;        // [wh][WP]
  float *const Tb = Z + wh * WP;             // [3][NLT_ROWS][TP]: terms, then column sums
  float *const Db = Tb + 3 * NLT_ROWS * TP;  // [2][NLT_ROWS][TP]: distortions
  int *const dsv = (int *)(Db + 2 * NLT_ROWS * TP); // window shift of every offset
  int *const pdy = dsv + ((n + 3) & ~3), *const pdx = pdy + ((n + 3) & ~3); // BORDER: its row and column shift
  const int r0 = top + R0 - reach, c0 = left - reach;
  const float n0 = a.norm[0], n1 = a.norm[1], n2 = a.norm[2];

  // (four fetches of a thread in flight, then their stores: nlm3_body.h)
  for(int i0 = tid; i0 < wh * WP; i0 += 4 * NLT_THREADS)
  {
    F4 v[4];
#pragma unroll
    for(int u = 0; u < 4; u++)
    {
      const int i = i0 + u * NLT_THREADS;
      const int wy = i / WP, wx = i - wy * WP;
      const int c = c0 + wx;
      v[u].x = v[u].y = v[u].z = v[u].w = 0.0f;
      // the pitch may run past the frame's right edge; the rows of an interior chunk's window are inside the frame
      if(i < wh * WP && c < W && (!BORDER || (c >= 0 && r0 + wy >= 0 && r0 + wy < H))) v[u] = in[(long)(r0 + wy) * W + c];
    }
#pragma unroll
    for(int u = 0; u < 4; u++)
    {
      const int i = i0 + u * NLT_THREADS;
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
  for(int i = tid; i < n; i += NLT_THREADS)
  {
    dsv[i] = patches[i].x * WP + patches[i].y;
    if(BORDER)
    {
      pdy[i] = patches[i].x;
      pdx[i] = patches[i].y;
    }
  }
  env.sync();

  const int w = tid >> 6, lane = tid & 63;
  const int u = tid - 128;
  // ---- (a): the lane's term, and its two own pixels
  const bool a_on = w >= 2 && u < ncol * TR;
  const int aj = a_on ? u / ncol : 0, ax = a_on ? 1 + u - aj * ncol : 1;
  const int awc = reach - P - 1 + ax;
  const int a_enter = (aj + P + reach) * WP + awc, a_leave = (aj - P - 1 + reach) * WP + awc; // rows R0 + aj + P and R0 + aj - P - 1
  const f2 oe_xy = XY[a_enter], ol_xy = XY[a_leave];
  const float oe_z = Z[a_enter], ol_z = Z[a_leave];
  // BORDER: the frame rows / column of the lane's own pixels
  const int a_re = top + R0 + aj + P, a_rl = top + R0 + aj - P - 1, a_c = left - P - 1 + ax;
  const bool own_e = !BORDER || ((unsigned)a_re < (unsigned)H), own_l = !BORDER || ((unsigned)a_rl < (unsigned)H),
             own_c = !BORDER || ((unsigned)a_c < (unsigned)W);
  // ---- (b): the lane's table column
  // wave 1: slot 1 + lane, and its first twelve lanes slot 65 + lane as well (wave 0 carries the row chain: nothing else)
  const bool b_on = w == 1 && 1 + lane <= ncol, b2_on = w == 1 && 65 + lane <= ncol;
  const int bx = 1 + lane, bx2 = 65 + lane;
  // ---- (d): the lane's pixel
  const bool d_on = w >= 2 && u < cw * TR;
  const int dj = d_on ? u / cw : 0, dc = d_on ? u - dj * cw : 0;
  const int d_win = (dj + reach) * WP + reach + dc;
  float accx = 0.0f, accy = 0.0f, accz = 0.0f, accw = 0.0f;
  const float sharp_m23 = a.sharpness * -8388608.0f;
  // CENTER: the lane's own pixel, the divisor 1 + center_weight and its refined reciprocal (nlm3_body.h)
  [[maybe_unused]] const f2 own_xy = CENTER ? XY[d_win] : f2{ 0.0f, 0.0f };
  [[maybe_unused]] const float own_z = CENTER ? Z[d_win] : 0.0f;
  [[maybe_unused]] const float cden = 1.0f + a.center_weight;
  [[maybe_unused]] const float crcp = Env::rcp_refined(cden);
  float seed = b_on ? seeds[bx] : 0.0f, seed2 = b2_on ? seeds[bx2] : 0.0f; // offset 0's, for stage 1

  for(int s = 0; s < n + 3; s++)
  {
    if(a_on && s < n)
    {
      // nlmeans_core.c:437-488 as nlm3_body.h's A1 forms it: (own - shifted)^2 per channel of the entering and of the leaving
      // row, term = ((nx2 - px2) n0 + (ny2 - py2) n1) + (nz2 - pz2) n2
      const int dS = dsv[s];
      const f2 se = XY[a_enter + dS], sl = XY[a_leave + dS];
      const float sez = Z[a_enter + dS], slz = Z[a_leave + dS];
      const float ex = oe_xy.x - se.x, ey = oe_xy.y - se.y, ez = oe_z - sez;
      const float lx = ol_xy.x - sl.x, ly = ol_xy.y - sl.y, lz = ol_z - slz;
      float nx2 = ex * ex, ny2 = ey * ey, nz2 = ez * ez;
      float px2 = lx * lx, py2 = ly * ly, pz2 = lz * lz;
      if(BORDER)
      {
        const int dy = pdy[s], dx = pdx[s];
        const bool col_ok = own_c && (unsigned)(a_c + dx) < (unsigned)W;
        if(!(col_ok && own_e && (unsigned)(a_re + dy) < (unsigned)H)) nx2 = ny2 = nz2 = 0.0f;
        if(!(col_ok && own_l && (unsigned)(a_rl + dy) < (unsigned)H)) px2 = py2 = pz2 = 0.0f;
      }
      Tb[((s % 3) * NLT_ROWS + aj) * TP + ax] = ((nx2 - px2) * n0 + (ny2 - py2) * n1) + (nz2 - pz2) * n2;
    }
    if(b_on && s >= 1 && s <= n)
    {
      // the column recurrence continued from the exported sum (nlm3_body.h column_chain: v = v + term)
      float *const col = Tb + (((s - 1) % 3) * NLT_ROWS) * TP + bx;
      float v = seed;
      if(s < n) seed = seeds[(size_t)s * NLT_SEED_PITCH + bx]; // the next offset's, a stage ahead
#pragma unroll
      for(int j = 0; j < NLT_ROWS; j++)
      {
        if(j < TR)
        {
          v = v + col[j * TP];
          col[j * TP] = v;
        }
      }
      if(b2_on)
      {
        float *const col2 = Tb + (((s - 1) % 3) * NLT_ROWS) * TP + bx2;
        float v2 = seed2;
        if(s < n) seed2 = seeds[(size_t)s * NLT_SEED_PITCH + bx2];
#pragma unroll
        for(int j = 0; j < NLT_ROWS; j++)
        {
          if(j < TR)
          {
            v2 = v2 + col2[j * TP];
            col2[j * TP] = v2;
          }
        }
      }
    }
    if(w == 0 && s >= 2 && s <= n + 1)
    {
      // the sliding row sum (:405-415) as nlm3_body.h's fused variant forms it (row_chain()): a tail row is eight lanes, a lane
      // nine adjacent columns; it reads the 9 + 5 column sums its columns' patches span, forms its nine (entering - leaving)
      // terms, and the chain of the row runs through the eight lanes in eight phases -- the carry moves to the next lane by
      // a DPP row shift, the row's first lane starts from the sum of the first four slots (slot 0 is never summed); lanes
      // that are done recompute.  One wave, ~110 instructions per offset (a first version walked a row in ONE lane: ~400
      // instructions of a wave that issues one every ~4.7 cycles: 1.64 ms of a 24 MP frame's 7.97).  All 64 lanes run it
      // (the shift is a wave operation); lanes 40 - 63 read table rows that do not exist (the words do) and store nothing
      constexpr int NPX = 9;
      const int cj = lane >> 3, j8 = lane & 7, cb = NPX * j8;
      const float *const T = Tb + (((s - 2) % 3) * NLT_ROWS + cj) * TP + cb;
      float cs[NPX + S];
#pragma unroll
      for(int i = 0; i < NPX + S; i++) cs[i] = T[i];
      const bool row_head = j8 == 0;
      float e[NPX];
      e[0] = cs[S] - (row_head ? 0.0f : cs[0]);
#pragma unroll
      for(int i = 1; i < NPX; i++) e[i] = cs[i + S] - cs[i];
      float first = 0.0f;
#pragma unroll
      for(int kk = 1; kk < S; kk++) first += cs[kk];
      float carry = first;
      float dist[NPX];
#pragma unroll
      for(int ph = 0; ph < 8; ph++)
      {
        float d = carry;
#pragma unroll
        for(int i = 0; i < NPX; i++)
        {
          d = d + e[i];
          dist[i] = d;
        }
        if(ph + 1 < 8)
        {
          const float from_left = env.lane_shr1(d);
          carry = row_head ? first : from_left;
        }
      }
      if(cj < TR)
      {
        float *const d = Db + (((s - 2) & 1) * NLT_ROWS + cj) * TP + cb;
#pragma unroll
        for(int i = 0; i < NPX; i++) d[i] = dist[i];
      }
    }
    if(d_on && s >= 3)
    {
      const int p = s - 3;
      const int wo = d_win + dsv[p];
      const float dist = Db[((p & 1) * NLT_ROWS + dj) * TP + dc];
      const f2 q = XY[wo];
      const float qz = Z[wo];
      float wgt;
      if constexpr(CENTER)
      {
        // nlmeans_core.c:416-424, as nlm3_body.h's accumulate() forms it
        const float dx_ = own_xy.x - q.x, dy_ = own_xy.y - q.y, dz_ = own_z - qz;
        const float num = dist + (dx_ * dx_ * a.cpn + dy_ * dy_ * a.cpn + dz_ * dz_ * a.cpn);
        const float v = Env::min_num(0.0f, Env::div_uniform(num, cden, crcp) * sharp_m23 + 16777216.0f); // (nlm3_body.h: why this is max(0, q s - 2) x -2^23)
        const int k0 = (int)(0x3f800000u + (unsigned)Env::cvt_i32_sat(v));
        wgt = Env::int_as_float(k0 >= 0x800000 ? k0 : 0);
      }
      else
        wgt = nlm2::mexp2_scaled<Env>(dist, sharp_m23);
      if(BORDER && !((unsigned)(top + R0 + dj + pdy[p]) < (unsigned)H && (unsigned)(left + dc + pdx[p]) < (unsigned)W)) wgt = 0.0f;
      accx = accx + q.x * wgt;
      accy = accy + q.y * wgt;
      accz = accz + qz * wgt;
      accw = accw + 1.0f * wgt;
    }
    env.sync();
  }

  // ---- normalise, blend (:490-521)
  if(!d_on) return;
  const int row = top + R0 + dj;
  if(row < a.out_row0 || row >= a.out_row1) return;
  const long o = (long)row * W + left + dc;
  F4 res;
  if(a.skip_blend)
  {
    res.x = accx / accw;
    res.y = accy / accw;
    res.z = accz / accw;
    res.w = accw / accw;
  }
  else
  {
    const F4 ip = in[o];
    res.x = (ip.x * (1.0f - a.luma)) + (accx / accw * a.luma);
    res.y = (ip.y * (1.0f - a.chroma)) + (accy / accw * a.chroma);
    res.z = (ip.z * (1.0f - a.chroma)) + (accz / accw * a.chroma);
    res.w = (ip.w * 0.0f) + (accw / accw * 1.0f);
  }
  out[o] = res;
  env.store_cell(a, o, res.x);
}

} // namespace nlmt
