// tests/native/amaze_host.cpp -- TEST INFRASTRUCTURE: the body of the gfx950 kernel amaze_stream
// (ansel_amd/csrc/amaze_stream_body.h) compiled for the host.  A workgroup is NT fibers (ucontext) that a scheduler
// runs one after the other from barrier to barrier, its LDS a heap block with a shadow: every access is checked for
//   * a race -- a word read in the phase another thread writes it in, or written in the phase another thread reads or
//     writes it in (the sequential schedule would hide those); inside one wave the unit is the stretch between two
//     wave_sync(), across waves the stretch between two workgroup barriers --, and
//   * a stale ring slot -- a plane's row read after a later row has overwritten its slot, or before it was produced.
// So the CPU suite compares the kernel with the oracle bit for bit, and proves its schedule, before a GPU is involved.
// -ffp-contract=off like the device build.
//
//   g++ -O2 -std=c++17 -ffp-contract=off -fPIC -shared -I ansel_amd/csrc tests/native/amaze_host.cpp -o tests/native/libamaze_host.so
#include <ucontext.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <string>
#include <vector>

#include "amaze_stream_body.h"

namespace
{

struct Shared
{
  std::vector<uint8_t> lds;
  std::vector<int16_t> tag;            // the row a word holds (-1: the zero the tile starts from)
  std::vector<int32_t> wphase, rphase; // the round of the last write / read (a round: every runnable fiber up to its next yield)
  std::vector<int32_t> wblock, rblock; // ... and the workgroup phase (between two workgroup barriers) it belongs to
  std::vector<int16_t> wtid, rtid;     // ... and who (-2: readers of several waves, -3: several readers of one wave)
  int phase = 1, block = 1;
  std::vector<char> at_block;
  std::vector<float> xchg; // lane shuffles
  long errors = 0;
  std::string first;
  ucontext_t main;
  std::vector<ucontext_t> ctx;
  std::vector<char> done;
  std::vector<std::vector<char>> stacks;
  // job
  const float *in;
  float *out;
  amz::args a;
  std::vector<std::pair<int, int>> tiles; // top, left
};

Shared *G = nullptr;

struct HostEnv
{
  int tid_;
  Shared *sh;
  int tid() const { return tid_; }
  void fail(const char *what, const int addr, const int row) const
  {
    if(sh->errors++ == 0)
    {
      char b[256];
      snprintf(b, sizeof(b), "%s: byte %d (row %d, holds row %d, written in phase %d by %d, read in phase %d by %d), thread %d, phase %d",
               what, addr, row, (int)sh->tag[addr], sh->wphase[addr], (int)sh->wtid[addr], sh->rphase[addr], (int)sh->rtid[addr], tid_, sh->phase);
      sh->first = b;
    }
  }
  // two accesses conflict when nothing orders them: threads of different waves inside one workgroup phase; threads of one
  // wave inside one round (a wave's LDS accesses execute in program order, and wave_sync() ends a round)
  bool unordered(const int other, const int oround, const int oblock) const
  {
    if(other == tid_ || oblock != sh->block) return false;
    if(other < 0) return other == -2 || oround == sh->phase;
    return (other >> 6) != (tid_ >> 6) || oround == sh->phase;
  }
  void on_read(const int addr, const int row) const
  {
    if(unordered(sh->wtid[addr], sh->wphase[addr], sh->wblock[addr])) fail("read of a word another thread writes in this phase", addr, row);
    if(row >= 0 && sh->tag[addr] >= 0 && sh->tag[addr] != row) fail("stale ring slot", addr, row);
    if(sh->rblock[addr] == sh->block && sh->rtid[addr] != tid_)
    {
      // keep the strongest record: readers of another wave conflict with any later write of the phase
      const int o = sh->rtid[addr];
      if(o == -2 || (o >= 0 && (o >> 6) != (tid_ >> 6)))
        sh->rtid[addr] = -2;
      else if(sh->rphase[addr] == sh->phase)
        sh->rtid[addr] = -3;
      else
      {
        sh->rphase[addr] = sh->phase;
        sh->rtid[addr] = (int16_t)tid_;
      }
    }
    else
    {
      sh->rphase[addr] = sh->phase;
      sh->rblock[addr] = sh->block;
      sh->rtid[addr] = (int16_t)tid_;
    }
  }
  void on_write(const int addr, const int row) const
  {
    if(unordered(sh->wtid[addr], sh->wphase[addr], sh->wblock[addr])) fail("word written by two threads in one phase", addr, row);
    if(unordered(sh->rtid[addr], sh->rphase[addr], sh->rblock[addr])) fail("write of a word another thread reads in this phase", addr, row);
    sh->wphase[addr] = sh->phase;
    sh->wblock[addr] = sh->block;
    sh->wtid[addr] = (int16_t)tid_;
    sh->tag[addr] = (int16_t)row;
  }
  float ldf(const int idx, const int row) const
  {
    on_read(idx * 4, row);
    float v;
    memcpy(&v, &sh->lds[(size_t)idx * 4], 4);
    return v;
  }
  void stf(const int idx, const int row, const float v) const
  {
    on_write(idx * 4, row);
    memcpy(&sh->lds[(size_t)idx * 4], &v, 4);
  }
  unsigned char ldb(const int bidx, const int row) const
  {
    on_read(bidx, row);
    return sh->lds[bidx];
  }
  void stb(const int bidx, const int row, const unsigned char v) const
  {
    on_write(bidx, row);
    sh->lds[bidx] = v;
  }
  void zero(const int word) const
  {
    for(int b = 0; b < 4; b++)
    {
      on_write(word * 4 + b, -1);
      sh->lds[(size_t)word * 4 + b] = 0;
    }
  }
  void sync() const
  {
    sh->at_block[tid_] = 1;
    swapcontext(&sh->ctx[tid_], &sh->main);
  }
  // orders the LDS accesses of ONE wave (on the device they execute in program order; here the fibers of the wave meet)
  void wave_sync() const { swapcontext(&sh->ctx[tid_], &sh->main); }
  // the value of lane - 1 / lane + 1 of the wave (every lane of the wave calls it; the first / last lane gets its own)
  float shfl_up1(const float v) const { return shuffle(v, -1); }
  float shfl_down1(const float v) const { return shuffle(v, 1); }
  int uniform(const int v) const { return v; }
  // the value of lane ^ 1 (both lanes of the pair call it)
  float swap1(const float v) const
  {
    sh->xchg[tid_] = v;
    swapcontext(&sh->ctx[tid_], &sh->main);
    const float r = sh->xchg[tid_ ^ 1];
    swapcontext(&sh->ctx[tid_], &sh->main);
    return r;
  }
  float shuffle(const float v, const int d) const
  {
    sh->xchg[tid_] = v;
    swapcontext(&sh->ctx[tid_], &sh->main);
    const int src = (tid_ & 63) + d;
    const float r = (src < 0 || src > 63) ? v : sh->xchg[(tid_ & ~63) + src];
    swapcontext(&sh->ctx[tid_], &sh->main);
    return r;
  }
  float add(const float x, const float y) const { return x + y; }
  void stamp(int) const {}
  void store_rgb(float *const o, const float r, const float g, const float b) const
  {
    o[0] = r;
    o[1] = g;
    o[2] = b;
  }
};

void fiber(const int t)
{
  Shared *const sh = G;
  HostEnv env{ t, sh };
  for(const auto &tl : sh->tiles) amz::tile(env, sh->in, sh->out, sh->a, tl.first, tl.second);
  sh->done[t] = 1;
  swapcontext(&sh->ctx[t], &sh->main);
}

} // namespace

extern "C" int amaze_host_lds_bytes(void) { return amz::LDS_BYTES; }

// Runs the kernel body over every tile of the frame that amz::stream_tile_ok() admits (the pixels of the other tiles are
// left as they are in `out`).  Returns the number of schedule errors (0 = none), the first one as text in err.
static int run_band(const float *in, float *out, int width, int height, uint32_t filters, float clip_pt, int in_row0, int out_row0,
                    int out_row1, int ty0, int ty1, int *stream_tiles, int *all_tiles, char *err, int errlen);

extern "C" int amaze_host_run(const float *in, float *out, int width, int height, uint32_t filters, float clip_pt,
                              int *stream_tiles, int *all_tiles, char *err, int errlen)
{
  return run_band(in, out, width, height, filters, clip_pt, 0, 0, height, 0, 1 << 30, stream_tiles, all_tiles, err, errlen);
}

// The same on a row band: `in` holds the mosaic from frame row in_row0 on, `out` the frame rows [out_row0, out_row1), the
// tile rows [ty0, ty1) of the frame's grid are walked (amaze_demosaic_launch() with a band does exactly this)
extern "C" int amaze_host_run_band(const float *in, float *out, int width, int height, uint32_t filters, float clip_pt, int in_row0,
                                   int out_row0, int out_row1, int ty0, int ty1, int *stream_tiles, int *all_tiles, char *err, int errlen)
{
  return run_band(in, out, width, height, filters, clip_pt, in_row0, out_row0, out_row1, ty0, ty1, stream_tiles, all_tiles, err, errlen);
}

static int run_band(const float *in, float *out, int width, int height, uint32_t filters, float clip_pt, int in_row0, int out_row0,
                    int out_row1, int ty0, int ty1, int *stream_tiles, int *all_tiles, char *err, int errlen)
{
  Shared sh;
  G = &sh;
  sh.lds.assign(amz::LDS_BYTES, 0xA5); // the body zeroes it itself
  sh.tag.assign(amz::LDS_BYTES, -1);
  sh.wphase.assign(amz::LDS_BYTES, 0);
  sh.rphase.assign(amz::LDS_BYTES, 0);
  sh.wblock.assign(amz::LDS_BYTES, 0);
  sh.rblock.assign(amz::LDS_BYTES, 0);
  sh.at_block.assign(amz::NT, 0);
  sh.xchg.assign(amz::NT, 0.f);
  sh.wtid.assign(amz::LDS_BYTES, -1);
  sh.rtid.assign(amz::LDS_BYTES, -1);
  sh.in = in;
  sh.out = out;
  sh.a.width = width;
  sh.a.height = height;
  sh.a.filters = filters;
  sh.a.ex = sh.a.ey = 0;
  sh.a.clip_pt = clip_pt;
  sh.a.variant = 0;
  sh.a.in_row0 = in_row0;
  sh.a.out_row0 = out_row0;
  sh.a.out_row1 = out_row1;
  const int ntx = (width + 16 + (amz::TS - 32) - 1) / (amz::TS - 32), nty = (height + 16 + (amz::TS - 32) - 1) / (amz::TS - 32);
  int all = 0;
  for(int ty = ty0 > 0 ? ty0 : 0; ty < nty && ty < ty1; ty++)
    for(int tx = 0; tx < ntx; tx++)
    {
      const int top = -16 + ty * (amz::TS - 32), left = -16 + tx * (amz::TS - 32);
      if(!(top < height && left < width)) continue;
      all++;
      if(amz::stream_tile_ok(width, height, top, left)) sh.tiles.push_back({ top, left });
    }
  if(stream_tiles) *stream_tiles = (int)sh.tiles.size();
  if(all_tiles) *all_tiles = all;
  const int NT = amz::NT;
  sh.ctx.resize(NT);
  sh.done.assign(NT, 0);
  sh.stacks.resize(NT);
  for(int t = 0; t < NT; t++)
  {
    sh.stacks[t].resize(256 * 1024);
    getcontext(&sh.ctx[t]);
    sh.ctx[t].uc_stack.ss_sp = sh.stacks[t].data();
    sh.ctx[t].uc_stack.ss_size = sh.stacks[t].size();
    sh.ctx[t].uc_link = &sh.main;
    makecontext(&sh.ctx[t], (void (*)())fiber, 1, t);
  }
  // a round: every fiber that is not waiting at a workgroup barrier runs up to its next yield; when all of them wait
  // there, the barrier opens
  for(;;)
  {
    int live = 0, ran = 0;
    for(int t = 0; t < NT; t++)
      if(!sh.done[t])
      {
        live++;
        if(sh.at_block[t]) continue;
        swapcontext(&sh.main, &sh.ctx[t]);
        ran++;
      }
    if(!live) break;
    sh.phase++;
    if(!ran)
    {
      for(int t = 0; t < NT; t++) sh.at_block[t] = 0;
      sh.block++;
    }
  }
  if(err && errlen > 0)
  {
    strncpy(err, sh.first.c_str(), errlen - 1);
    err[errlen - 1] = 0;
  }
  G = nullptr;
  return (int)(sh.errors > 0x7fffffff ? 0x7fffffff : sh.errors);
}
