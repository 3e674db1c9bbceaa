// pipe_fused.h -- launchers of the fused stage groups (pipe_fused.hip), used by the executor (pipe.cpp)
#pragma once
#include "hip_common.h"

namespace ansel
{

// ---- fused CFA group: rawprepare [-> temperature] [-> highlights(clip)] --------------------
struct raw_group_t
{
  dt_hip_piece_t rawprepare_piece;
  dt_hip_rawprepare_data_t rawprepare;
  bool has_temperature;
  dt_hip_piece_t temperature_piece;
  dt_hip_temperature_data_t temperature;
  bool has_highlights;
  dt_hip_piece_t highlights_piece;
  dt_hip_highlights_data_t highlights;
};
int raw_group_launch(int devid, const raw_group_t &g, dt_hip_mem_t dev_in, dt_hip_mem_t dev_out,
                     dt_hip_mem_t deferred_journal = nullptr);
bool raw_group_supported(const raw_group_t &g);

// ---- fused RGBA group: any order of exposure, colorin, channelmixerrgb, filmicrgb, colorout,
//      optionally opened / closed by the Lab glue of the pipe and closed by the float -> u16 conversion ------------------------------------------
enum rgb_op_t { RGB_OP_EXPOSURE = 0, RGB_OP_COLORIN, RGB_OP_CHANNELMIXER, RGB_OP_FILMIC, RGB_OP_COLOROUT, RGB_OP_END };
struct rgb_group_t
{
  int width, height;
  int n_ops;
  int ops[8];
  dt_hip_exposure_data_t exposure;
  dt_hip_conversion_t colorin, colorout;
  dt_hip_channelmixerrgb_data_t channelmixer;
  dt_hip_filmicrgb_data_t filmic;
  int to_u16; // 0: float4 out, 1: RGBA u16, 2: RGB u16 rows (export_u16 + export_rows)
  // the pipe's colourspace glue next to the run: "lab_to_rgb" in front of it, "rgb_to_lab" behind it (never with to_u16)
  int pre_lab, post_lab;
  dt_hip_lab_data_t lab_pre, lab_post;
};
struct bilat_slice_args;
// pre_slice (px_bilat.h): local contrast's bilateral-grid slice as the run's first stage (bilat.hip bilat_process_chain())
int rgb_group_launch(int devid, const rgb_group_t &g, dt_hip_mem_t dev_in, dt_hip_mem_t dev_out, const bilat_slice_args *pre_slice = nullptr);
// the kernel arguments of a run (rgb_chain_kernel.h): cm_kind = the chromatic adaptation its kernel is instantiated for
// (CM_NONE without color calibration), fm = the filmic mode (FM_NONE without filmic)
struct chain_args;
int rgb_group_fill_args(const rgb_group_t &g, chain_args &a, int &cm_kind, int &fm);
// denoiseprofile.hip: denoise (profiled), wavelets, with the run `chain` (no filmic, float output) applied in its last
// kernel; DT_HIP_INVALID_ARG when the combination has no fused kernel (the caller then runs the two one after the other)
int denoiseprofile_process_chain(int devid, const dt_hip_piece_t *piece, const dt_hip_denoiseprofile_data_t *d,
                                 dt_hip_mem_t dev_in, dt_hip_mem_t dev_out, const rgb_group_t *chain);

// bilat.hip: local contrast in bilateral-grid mode with the run `chain` behind it -- the slice is the run's first stage;
// DT_HIP_INVALID_ARG when the module is in another mode
// cells: the frame's lightness cells, written by the module in front (nlmeans.hip nlmeans_process_cells()), or NULL
int bilat_process_chain(int devid, const dt_hip_piece_t *piece, const dt_hip_bilat_data_t *d, dt_hip_mem_t dev_in, dt_hip_mem_t dev_out,
                        const rgb_group_t *chain, dt_hip_mem_t cells = nullptr);
int bilat_process_cells(int devid, const dt_hip_piece_t *piece, const dt_hip_bilat_data_t *d, dt_hip_mem_t dev_in, dt_hip_mem_t dev_out,
                        dt_hip_mem_t cells);
int bilat_cell_params(const dt_hip_piece_t *piece, const dt_hip_bilat_data_t *d, float *sigma_r, int *size_z);
int nlmeans_process_cells(int devid, const dt_hip_piece_t *piece, const dt_hip_nlmeans_data_t *d, dt_hip_mem_t dev_in, dt_hip_mem_t dev_out,
                          dt_hip_mem_t cells, float sigma_r, int size_z);

} // namespace ansel
