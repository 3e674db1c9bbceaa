// hip_common.h -- internals shared by the translation units of libansel_hip.
// Not part of the C-ABI (that is include/ansel_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "ansel_hip.h"

// The A/B switches of the kernels' development (superseded kernel versions, roles switched off, clock reads) exist in the
// MEASURING build only: `python -m ansel_amd.build --measuring` compiles every translation unit with
// -DANSEL_HIP_MEASURING into ansel_amd/libansel_hip_measuring.so, which tools/ load through ANSEL_HIP_LIB.  The product
// library carries one kernel per job and reads no environment variable on a launch path: measuring_env() is a constant
// there, and the superseded kernels are not compiled.
#ifdef ANSEL_HIP_MEASURING
#include <stdlib.h>
static inline const char *measuring_env(const char *name) { return getenv(name); }
#else
static inline constexpr const char *measuring_env(const char *) { return nullptr; }
#endif

namespace ansel
{

// Test hooks (testhooks.hip, dt_hip_test_dispatch(); not part of include/ansel_hip.h): send a launch to one of the FALLBACK
// kernels the product carries anyway -- the second non-local-means kernel that takes the chunk grids the third does not,
// AMaZE's first kernel that takes the tiles the on-chip one does not -- on a frame where the primary kernel applies, so that
// the tests can compare the two and the oracle on the same frame.  A word in memory set through the C entry, no environment.
enum dispatch_key_t
{
  DISPATCH_NLM_V2 = 0,      // non-local means: the second version wherever the third / fused one would run
  DISPATCH_NLM_FUSED,       // ... the fused variant wherever it fits (also on the third version's grids)
  DISPATCH_AMAZE_UNFUSED,   // AMaZE: one launch per kind of tile instead of the mixed queue
  DISPATCH_AMAZE_SLAB,      // AMaZE: every tile on the first kernel
  DISPATCH_AMAZE_BLOCKS,    // AMaZE: at most this many workgroups (a workgroup then walks many tiles)
  DISPATCH_BILAT_BLUR_SPLIT, // bilateral grid: the blur's x-pass in a launch of its own (what grids beyond 2^22 cells take)
  DISPATCH_KEYS
};
int dispatch_override(dispatch_key_t key); // 0: not set

void set_last_error(const char *fmt, ...);

// per-device state owned by runtime.cpp
hipStream_t stream_of(int devid); // also makes the device current for the calling thread
// a small host table -> device memory on the device's stream, through a pinned staging ring: no wait for the stream (runtime.cpp)
int upload_small(int devid, void *dst_dev, const void *src_host, size_t bytes);
bool make_current(int devid);
int hip_device_of(int devid); // the HIP ordinal behind a runtime device id (-1: no such device)
bool valid_device(int devid);

// Tagged launch bracket: the HIP peer of dt_opencl_events_get_slot() + the event the
// OpenCL enqueue fills (src/common/opencl.c:3048-3240).  When profiling is enabled for the
// device, a start/stop hipEvent pair is recorded on the device's stream around the launch;
// dt_hip_events_profiling() later harvests them per tag.
struct launch_scope
{
  int devid;
  const char *tag;
  hipEvent_t start, stop;
  bool active;
  launch_scope(int devid, const char *tag);
  ~launch_scope();
};

static inline int check_launch(const char *what)
{
  const hipError_t e = hipGetLastError();
  if(e != hipSuccess)
  {
    set_last_error("%s: %s", what, hipGetErrorString(e));
    return DT_HIP_DEFAULT_ERROR;
  }
  return DT_HIP_SUCCESS;
}

#define ANSEL_HIP_CHECK(expr)                                                                   \
  do                                                                                            \
  {                                                                                             \
    const hipError_t _e = (expr);                                                               \
    if(_e != hipSuccess)                                                                        \
    {                                                                                           \
      ansel::set_last_error("%s:%d %s: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));  \
      return DT_HIP_DEFAULT_ERROR;                                                              \
    }                                                                                           \
  } while(0)

// Streaming (non-temporal) 16-byte store: module outputs are written once and read by the next
// kernel from HBM/L2, never re-read by the writer -- the device analogue of the reference's
// dt_store_simd_nontemporal() (src/system/simd.h:160-181).
typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store(float4 *p, const float4 v)
{
  const v4f_t t = { v.x, v.y, v.z, v.w };
  __builtin_nontemporal_store(t, reinterpret_cast<v4f_t *>(p));
}

// A large by-value kernel argument, read in place from the kernarg segment at its byte offset.  As a
// by-value argument every field is loaded in the kernel's entry block; a struct of a few hundred dwords then
// exceeds the ~100 scalar registers and comes back one v_readlane at a time (rgb_chain: 162 SGPRs spilled,
// 4.95 -> 4.35 ms once the fields are loaded where they are used).  The argument stays in the signature so
// that the host marshals it; the kernel must not name it.
template <typename T> __device__ __forceinline__ const T &kernarg_at(const int offset)
{
  typedef const T __attribute__((address_space(4))) *p4;
  return *(const T *)(p4)((const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr() + offset);
}
// where a by-value argument of type T sits in the kernarg segment when the arguments in front of it have the types
// Before...: every argument at its natural alignment, in declaration order (the code-object metadata's .offset; the
// tests read it back: tests/test_kernel_resources.py).  The kernels that call kernarg_at() spell their offset with
// this, and static_assert the number their comment quotes.
template <typename T, typename... Before> constexpr int kernarg_offset_after()
{
  size_t off = 0;
  ((off = (off + alignof(Before) - 1) / alignof(Before) * alignof(Before) + sizeof(Before)), ...);
  return (int)((off + alignof(T) - 1) / alignof(T) * alignof(T));
}

// XCD-aware 2-D launch for stencil kernels that walk the frame row by row.  The eight XCDs of an MI355X have
// private L2s and workgroups are handed to them round-robin by linear workgroup id (x fastest); with gridDim.x not
// a multiple of 8 the workgroups of a column block land on a different XCD row after row, and the rows a stencil
// re-reads come from HBM / the Infinity Cache every time (dn_decompose: 5.2 planes fetched per plane of input).
// With gridDim.x padded to a multiple of 8, column block bx always runs on XCD bx % 8 and that XCD sweeps the rows
// in order: vertical taps hit its L2.  The padding workgroups (blockIdx.x >= the real count) must exit at once.
// The padding would leave the XCDs that only get padding columns idle (38 column blocks: six XCDs sweep five, two
// sweep four), so the assignment rotates by one XCD every 64 rows of the walk: xcd_col() is the logical column
// block of this workgroup, >= gx for padding.
static inline unsigned xcd_pad(const int gx) { return (unsigned)((gx + 7) / 8) * 8; }
__device__ __forceinline__ int xcd_col() { return (int)((blockIdx.x + blockIdx.y / 64) % gridDim.x); }

// Journal of the highlight-clip pass (pointwise.hip, pipe_fused.hip): number of photosites above the
// threshold and the first 25 of them {index in the output buffer, unclipped value}.  In band mode
// the leading count is summed over all bands before highlights_resolve_launch() decides the bypass.
#define HL_MIN_CLIPPED 25
struct hl_journal
{
  unsigned long long count;
  unsigned long long index[HL_MIN_CLIPPED];
  float value[HL_MIN_CLIPPED];
};
int highlights_resolve_launch(int devid, float *out, const hl_journal *journal);

// A row band of a frame for the RCD launch (multi-GPU row bands)
struct rcd_band_t
{
  int tv0, tv1;          // tile rows of the frame's own 94-row tile grid
  int in_row0, in_rows;  // frame rows held by the input buffer
  int out_row0, out_rows; // frame rows held by the output buffer
};

// A row band in front of a stencil module (pipe.cpp): the input buffer holds frame rows
// from buf_row0 on (own rows + the halo fetched from the neighbours), the band owns frame rows [row0, row1)
struct band_view_t
{
  int frame_h;
  int buf_row0;
  int row0, row1;
};
// stencil modules on a row band
int nlmeans_halo_rows(const dt_hip_piece_t *piece, const dt_hip_nlmeans_data_t *d);
int nlmeans_process_band(int devid, const dt_hip_piece_t *piece, const dt_hip_nlmeans_data_t *d, const band_view_t *band,
                         dt_hip_mem_t dev_in, dt_hip_mem_t dev_out);
int diffuse_halo_rows(const dt_hip_piece_t *piece, const dt_hip_diffuse_data_t *d);
int diffuse_process_rows(int devid, const dt_hip_piece_t *piece, const dt_hip_diffuse_data_t *d, int first_row,
                         dt_hip_mem_t dev_in, dt_hip_mem_t dev_out);
// ... followed by "rgb_to_lab" as the tail of its last kernel; DT_HIP_INVALID_ARG when that kernel has no tail
int diffuse_process_post_lab(int devid, const dt_hip_piece_t *piece, const dt_hip_diffuse_data_t *d, dt_hip_mem_t dev_in,
                             dt_hip_mem_t dev_out, const dt_hip_lab_data_t *lab);
// denoiseprofile: -1 when the frame is too small for the module to do anything but copy
int denoiseprofile_halo_rows(const dt_hip_piece_t *piece, const dt_hip_denoiseprofile_data_t *d);
struct dn_band_job_t;
// non-local means mode: one call (job == nullptr on return, dev_out = the band's own rows).  Wavelets mode:
// preconditions the buffer (buf_rows rows: own rows + 2 halo rows) and returns a job; denoiseprofile_band_step() then
// runs one decomposition per call and says what the band needs next -- 1: the neighbours' *halo_rows rows of
// *halo_buf ([min(h, row0)][rows][min(h, H - row1)], RGBA float), 2: an all-reduce (SUM) of *sums (the frame-wide table
// of partial sums of detail^2: own rows filled, the rest zero, so the sum is exact in any order), 0: nothing, call
// denoiseprofile_band_finish() (thresholds, synthesis, inverse transform into dev_out = own rows; frees the job)
int denoiseprofile_band_begin(int devid, const dt_hip_piece_t *piece, const dt_hip_denoiseprofile_data_t *d,
                              const band_view_t *band, int buf_rows, dt_hip_mem_t dev_in, dt_hip_mem_t dev_out,
                              dn_band_job_t **job);
int denoiseprofile_band_step(dn_band_job_t *job, dt_hip_mem_t *halo_buf, int *halo_rows, double **sums, size_t *sum_count);
int denoiseprofile_band_finish(dn_band_job_t *job, dt_hip_mem_t dev_out);
void denoiseprofile_band_abort(dn_band_job_t *job);
// guided_filter.hip: guided_filter() of src/pixel/guided_filter.c:369 -- `mask` (width x height floats) filtered in place,
// guided by width x height float4 pixels
int guided_filter_launch(int devid, const float4 *guide, float *mask, int width, int height, int w, float sqrt_eps,
                         float guide_weight, float minv, float maxv);
// detailmask.hip: the refinement a blend's details threshold applies to its form mask (`form` NULL: the constant `fill`)
int detail_refine_launch(int devid, const float *rawdetail, const float *form, float fill, float level, int width, int height,
                         float *refined);
// local contrast (bilateral grid) on row bands: the grid is one accumulation over the frame in pixel order, so the bands
// take turns (bilat.hip).  begin: the zeroed grid of the frame; splat: the band's rows on top of what the grid holds;
// finish: blur of the complete grid (this band's copy) and the slice of the band's rows
int bilat_band_supported(const dt_hip_piece_t *piece, const dt_hip_bilat_data_t *d);
int blend_refines_with_detail_mask(const dt_hip_blend_data_t *d); // blend.hip
int bilat_band_begin(int devid, const dt_hip_piece_t *piece, const dt_hip_bilat_data_t *d, dt_hip_mem_t *grid, size_t *bytes);
int bilat_band_splat(int devid, const dt_hip_piece_t *piece, const dt_hip_bilat_data_t *d, dt_hip_mem_t grid,
                     dt_hip_mem_t in_rows, int row0, int rows);
int bilat_band_finish(int devid, const dt_hip_piece_t *piece, const dt_hip_bilat_data_t *d, dt_hip_mem_t grid,
                      dt_hip_mem_t in_rows, dt_hip_mem_t out_rows, int row0, int rows);

// Grid for a grid-stride streaming kernel: enough workgroups to fill 256 CUs x 8 and no more
// (cdna_hip_programming.md Guideline 11).
static inline unsigned stream_grid(size_t work_items, unsigned block)
{
  size_t g = (work_items + block - 1) / block;
  if(g > 2048u * 4u) g = 2048u * 4u;
  if(g < 1) g = 1;
  return (unsigned)g;
}

// Grid for a one-pixel-per-thread kernel (256-thread workgroups).  Used where a kernel carries a few
// hundred uniform parameters: inside a grid-stride loop those stay live across iterations and spill
// out of the 100-odd SGPRs; used once per wave they do not (rgb_chain: 154 -> 32 VGPRs, 6.7 -> 4.9 ms).
static inline unsigned pixel_grid(size_t pixels)
{
  const size_t g = (pixels + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 0x7fffffffu ? 0x7fffffffu : g));
}

} // namespace ansel
