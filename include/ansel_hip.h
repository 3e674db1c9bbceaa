/*
 * ansel_hip.h -- C-ABI of libansel_hip: the MI355X-native peer of Ansel's OpenCL layer
 * and the device entry points of the hot iop modules of the export pixelpipe.
 *
 * Everything here is `extern "C"`, plain pointers and sizes; no torch/HIP types leak
 * through.  Two groups of symbols:
 *
 *   1. dt_hip_*            device runtime.  Stands where the reference has
 *                          src/common/opencl.{c,h} (dt_opencl_*, opencl.h:333-651);
 *                          same return conventions (0 = success, negative = error,
 *                          allocators return NULL, dt_hip_finish TRUE on success).
 *   2. dt_hip_iop_<op>_*   one process entry per hot module: what that module's
 *                          process_cl() (src/iop/iop_api.h:277-278) calls instead of
 *                          dt_opencl_set_kernel_arg()/dt_opencl_enqueue_kernel_2d().
 *                          Arguments are (devid, piece view, module data, dev_in, dev_out):
 *                          the same information process_cl() reads from `piece`.
 *
 * Device buffers are linear, tightly packed, row-major (pitch = bpp * width), exactly the
 * host layout of a pixelpipe cacheline (src/develop/pixelpipe_hb.c:985): 1 x u16 or 1 x f32
 * per photosite before demosaic, float4 RGBA after.
 *
 * INTEGRATION.md shows the process_cl() stub a maintainer adds for each entry.
 */
#ifndef ANSEL_HIP_H
#define ANSEL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes: mirror src/common/opencl.h:61-62 ------------------------------- */
#define DT_HIP_SUCCESS 0
#define DT_HIP_DEFAULT_ERROR (-999)
#define DT_HIP_SYSMEM_ALLOCATION (-998)
#define DT_HIP_INVALID_ARG (-997)
#define DT_HIP_WRITER_FAILED (-996) /* a batch's writer callback returned non-zero (dt_hip_batch_set_writer) */
#define DT_HIP_MAX_ERRORS 5 /* src/common/opencl.h:50 */

/* opaque device allocation, stands for cl_mem */
typedef void *dt_hip_mem_t;

/* ---- shared plain views of host structs ----------------------------------------- */

/* == dt_iop_roi_t, src/pixel/format.h:44-48 (same field order and types) */
typedef struct dt_hip_roi_t
{
  int x, y, width, height;
  double scale;
} dt_hip_roi_t;

/* The read-only slice of dt_dev_pixelpipe_iop_t (src/develop/pixelpipe_hb.h:101-166) and of
 * its dt_iop_buffer_dsc_t dsc_in (src/pixel/format.h:82-123) that the hot modules read. */
typedef struct dt_hip_piece_t
{
  dt_hip_roi_t roi_in, roi_out;
  uint32_t filters;           /* piece->dsc_in.filters: dcraw Bayer word, 0 = not mosaiced */
  uint32_t channels;          /* piece->dsc_in.channels: 1 or 4 */
  uint32_t datatype;          /* piece->dsc_in.datatype: DT_HIP_TYPE_* */
  uint32_t _pad;
  float processed_maximum[4]; /* piece->dsc_in.processed_maximum */
} dt_hip_piece_t;

#define DT_HIP_TYPE_FLOAT 1  /* TYPE_FLOAT  (src/pixel/format.h:52) */
#define DT_HIP_TYPE_UINT16 2 /* TYPE_UINT16 */

/* == dt_develop_tiling_t, src/develop/tiling.h:39-59 */
typedef struct dt_hip_tiling_t
{
  float factor, factor_cl;
  float maxbuf, maxbuf_cl;
  unsigned overhead;
  unsigned overlap;
  unsigned xalign, yalign;
} dt_hip_tiling_t;
/* The tiling callbacks below (dt_hip_iop_<op>_tiling) fill the struct like the module's tiling_callback()
 * (iop_api.h:119-120): factor / maxbuf / overlap / alignment as the reference states them -- the host's own
 * tiling and ROI planning keep working -- and factor_cl / maxbuf_cl with what THIS implementation holds on the
 * device, which is what dt_hip_image_fits_device() is asked about.  Modules without a callback of their own take
 * default_tiling_callback(), src/develop/tiling.c:1423-1463: */
void dt_hip_default_tiling(const dt_hip_piece_t *piece, int before_demosaic, dt_hip_tiling_t *tiling);
/* default_process_tiling_cl() for a module that does not move pixels (roi_in == roi_out),
 * _default_process_tiling_cl_ptp(), src/develop/tiling.c:842-1067 -- what the host falls back to when
 * dt_hip_image_fits_device() says no (pixelpipe_gpu.c:514): the frame stays in HOST memory, tiles of it go through
 * the device with `overlap` pixels of context on every side and only their interior comes back.
 * dt_hip_plan_tiles_ptp() is the tile plan of :868-979 as a pure function (no device needed);
 * dt_hip_default_process_tiling_ptp() runs module `op` (the names of dt_hip_pipe_add_node) over it:
 * host_in / host_out are full frames of in_bpp / out_bpp bytes per pixel, `tiling` is the module's callback
 * result, available_bytes = 0 asks the device (a smaller value forces tiling).  As in the reference, a tile is an
 * ordinary module run on a cropped frame: results equal the untiled run for every pixel whose dependencies stay
 * within `overlap` -- all of them for pointwise modules. */
typedef struct dt_hip_tile_plan_t
{
  int32_t width, height;     /* largest tile, overlap included */
  int32_t tile_wd, tile_ht;  /* step between tile origins = the part of a tile that is kept */
  int32_t tiles_x, tiles_y;
  int32_t overlap;           /* the module's overlap rounded up to its alignment */
} dt_hip_tile_plan_t;
int dt_hip_plan_tiles_ptp(int roi_width, int roi_height, int in_bpp, int out_bpp, const dt_hip_tiling_t *tiling,
                          unsigned filters, size_t available_bytes, size_t memalloc_bytes, int max_width, int max_height,
                          dt_hip_tile_plan_t *plan);
int dt_hip_default_process_tiling_ptp(int devid, const char *op, const dt_hip_piece_t *piece, const void *data,
                                      size_t data_size, const dt_hip_tiling_t *tiling, const void *host_in, void *host_out,
                                      int in_bpp, int out_bpp, size_t available_bytes);
/* ... and for roi_in != roi_out (_default_process_tiling_cl_roi(), src/develop/tiling.c:1076-1390): on the export path
 * that is finalscale, whose modify_roi_in() (src/iop/finalscale.c:76-107) relates the regions.  dt_hip_plan_tiles_roi()
 * is the tile grid of :1100-1220, dt_hip_tile_rois_finalscale() the three regions of one tile (:1228-1300) -- both pure
 * functions -- and dt_hip_default_process_tiling_roi() the loop of :1222-1370.  Like the reference, every tile is
 * resampled as an image of its own (finalscale ignores the region origins, finalscale.c:124-129). */
typedef struct dt_hip_tile_plan_roi_t
{
  int32_t width, height;     /* largest buffer of a tile (input or output) */
  int32_t tile_wd, tile_ht;  /* the good part of a tile, in output pixels */
  int32_t tiles_x, tiles_y;
  int32_t overlap_in, overlap_out, delta, xyalign;
} dt_hip_tile_plan_roi_t;
#define DT_HIP_TILE_EMPTY 2 /* dt_hip_tile_rois_finalscale(): the tile has no output pixels (the grid step is rounded up) */
int dt_hip_plan_tiles_roi(const dt_hip_roi_t *roi_in, const dt_hip_roi_t *roi_out, int in_bpp, int out_bpp,
                          const dt_hip_tiling_t *tiling, unsigned filters, size_t available_bytes, size_t memalloc_bytes,
                          int max_width, int max_height, dt_hip_tile_plan_roi_t *plan);
int dt_hip_tile_rois_finalscale(const dt_hip_tile_plan_roi_t *plan, const dt_hip_roi_t *roi_in, const dt_hip_roi_t *roi_out,
                                int tx, int ty, dt_hip_roi_t *iroi_full, dt_hip_roi_t *oroi_full, dt_hip_roi_t *oroi_good);
int dt_hip_default_process_tiling_roi(int devid, const char *op, const dt_hip_piece_t *piece, const void *data, size_t data_size,
                                      const dt_hip_tiling_t *tiling, const void *host_in, void *host_out, int in_bpp,
                                      int out_bpp, size_t available_bytes);

/* ---- 1. device runtime (peer of src/common/opencl.h) ------------------------------ */

/* lifecycle: dt_opencl_init/cleanup/is_inited (opencl.h:333-336,460) */
int dt_hip_init(void);
void dt_hip_cleanup(void);
int dt_hip_is_inited(void);
int dt_hip_get_num_devices(void);                            /* opencl.h:364 */
const char *dt_hip_get_device_name(int devid);
size_t dt_hip_get_device_available(int devid);               /* free HBM bytes; opencl.h:405 */
size_t dt_hip_get_device_memalloc(int devid);                /* largest single allocation */
/* device exclusivity: dt_opencl_reserve_device_for_pipe / reserve_device_by_id / try_reserve_device_by_id /
 * release_device (opencl.h:351-419, opencl.c:1642-1755).  There is one lock per device; reserving a device and
 * locking it are the same act.  _for_pipe returns the device id or -1 when every device is busy; _by_id blocks;
 * try_ returns 0 when the device was reserved (pthread convention) and never waits. */
int dt_hip_reserve_device_for_pipe(int pipetype);
void dt_hip_reserve_device_by_id(int devid);
int dt_hip_try_reserve_device_by_id(int devid);
void dt_hip_release_device(int devid);
/* dt_opencl_get_device_max_image_size / _max_global_mem (opencl.h:399-402): TRUE when written */
int dt_hip_get_device_max_image_size(int devid, int *width, int *height);
size_t dt_hip_get_device_max_global_mem(int devid);
/* dt_opencl_report_pipe_error (opencl.h:415): 1 = this run failed, retry; 2 = fifth failure, the device path is
 * off for the session and dt_hip_is_enabled() turns 0.  dt_opencl_is_enabled / update_settings (:464-470) */
int dt_hip_report_pipe_error(void);
int dt_hip_is_enabled(void);
int dt_hip_update_settings(void);
/* per-device knobs the host queries (opencl.h:584-591, 648-650); constants on this part */
void dt_hip_check_tuning(int devid);
int dt_hip_avoid_atomics(int devid);
int dt_hip_micro_nap(int devid);
int dt_hip_use_pinned_memory(int devid);
int dt_hip_dev_roundup_width(int size, int devid);
int dt_hip_dev_roundup_height(int size, int devid);
/* dt_opencl_image_fits_device (opencl.h:571): would factor * w*h*bpp + overhead fit? */
int dt_hip_image_fits_device(int devid, size_t width, size_t height, unsigned bpp, float factor, size_t overhead);
/* dt_opencl_image_fits_device_reason (opencl.h:577): 0 fits, 1 one buffer exceeds the largest allocation, 2 the
 * total exceeds free memory; *needed / *limit (may be NULL) report the pair that decided */
int dt_hip_image_fits_device_reason(int devid, size_t width, size_t height, unsigned bpp, float factor, size_t overhead,
                                    size_t *needed, size_t *limit);

/* stream each device's work is enqueued on.  By default the runtime creates one
 * non-blocking stream per device; a host that already owns a stream (e.g. the one a
 * torch allocator orders its frees on) can adopt it.  `stream` is a hipStream_t. */
void *dt_hip_get_stream(int devid);
int dt_hip_set_stream(int devid, void *stream);

/* memory: dt_opencl_alloc_device[_buffer], release, get_mem_object_size (opencl.h:508-561) */
dt_hip_mem_t dt_hip_alloc_device(int devid, int width, int height, int bpp);
dt_hip_mem_t dt_hip_alloc_device_buffer(int devid, size_t size);
void dt_hip_release_mem_object(dt_hip_mem_t mem);
size_t dt_hip_get_mem_object_size(dt_hip_mem_t mem);
void dt_hip_memory_statistics(int devid, size_t *current, size_t *peak); /* opencl.h:648 */
/* dt_opencl_get_image_width / _height / _element_size / get_mem_context_id (opencl.h:554-560): the geometry
 * dt_hip_alloc_device() was called with (0 for plain buffers); the device an object lives on (-1: not ours) */
int dt_hip_get_image_width(dt_hip_mem_t mem);
int dt_hip_get_image_height(dt_hip_mem_t mem);
int dt_hip_get_image_element_size(dt_hip_mem_t mem);
int dt_hip_get_mem_context_id(dt_hip_mem_t mem);
/* dt_opencl_alloc_device_use_host_pointer (opencl.h:521): a device view of page-locked host memory (zero copy);
 * NULL unless `host` came from dt_hip_alloc_host_pinned().  dt_opencl_map_buffer / map_image / unmap_mem_object
 * (:545-550, used by pixelpipe_cache.c for pinned cache lines): such an object maps to its host memory (after
 * draining the stream when `blocking`); device-only memory has no host mapping and returns NULL. */
dt_hip_mem_t dt_hip_alloc_device_use_host_pointer(int devid, int width, int height, int bpp, void *host, int flags);
void *dt_hip_map_buffer(int devid, dt_hip_mem_t buffer, int blocking, int flags, size_t offset, size_t size);
void *dt_hip_map_image(int devid, dt_hip_mem_t buffer, int blocking, int flags, size_t width, size_t height, int bpp);
int dt_hip_unmap_mem_object(int devid, dt_hip_mem_t mem, void *mapped_ptr);
/* page-locked host memory for the two ends of an export (the sensor buffer going up, the exported frame
 * coming down): the peer of the reference's pinned transfer buffers (dt_opencl_use_pinned_memory(),
 * CL_MEM_ALLOC_HOST_PTR in dt_opencl_alloc_device_use_host_pointer(), opencl.h:541-561, 650-651).
 * Copies from / to such memory run at PCIe rate and can be non-blocking. */
void *dt_hip_alloc_host_pinned(size_t size);
void dt_hip_free_host_pinned(void *host);
int dt_hip_is_pinned_memory(const void *host);

/* copies: dt_opencl_write_host_to_device / read_host_from_device (+_rowpitch,
 * _non_blocking) and enqueue_copy_* (opencl.h:473-537).  rowpitch in bytes. */
int dt_hip_write_host_to_device(int devid, const void *host, dt_hip_mem_t device, int width, int height, int bpp);
int dt_hip_write_host_to_device_rowpitch(int devid, const void *host, dt_hip_mem_t device, int width, int height, int bpp, size_t rowpitch, int blocking);
int dt_hip_read_host_from_device(int devid, void *host, dt_hip_mem_t device, int width, int height, int bpp);
int dt_hip_read_host_from_device_rowpitch(int devid, void *host, dt_hip_mem_t device, int width, int height, int bpp, size_t rowpitch, int blocking);
/* dt_opencl_read_host_from_device_raw / write_host_to_device_raw (opencl.h:489-506): the window {origin[2],
 * region[2]} (pixels) of a 2-D image made by dt_hip_alloc_device() against host memory of `rowpitch` bytes per row
 * (0: packed); dt_opencl_enqueue_copy_image (:516): a window between two such images.  What the host tilers move. */
int dt_hip_read_host_from_device_raw(int devid, void *host, dt_hip_mem_t device, const size_t *origin, const size_t *region,
                                     int rowpitch, int blocking);
int dt_hip_write_host_to_device_raw(int devid, const void *host, dt_hip_mem_t device, const size_t *origin, const size_t *region,
                                    int rowpitch, int blocking);
int dt_hip_enqueue_copy_image(int devid, dt_hip_mem_t src, dt_hip_mem_t dst, const size_t *orig_src, const size_t *orig_dst,
                              const size_t *region);
/* dt_opencl_copy_host_to_device[_rowpitch|_constant] (opencl.h:508-514): allocate + upload, NULL on failure;
 * dt_opencl_copy_device_to_host (:473); dt_opencl_read_buffer_from_device / write_buffer_to_device (:533-537) */
dt_hip_mem_t dt_hip_copy_host_to_device(int devid, void *host, int width, int height, int bpp);
dt_hip_mem_t dt_hip_copy_host_to_device_rowpitch(int devid, void *host, int width, int height, int bpp, int rowpitch);
dt_hip_mem_t dt_hip_copy_host_to_device_constant(int devid, size_t size, void *host);
int dt_hip_copy_device_to_host(int devid, void *host, dt_hip_mem_t device, int width, int height, int bpp);
int dt_hip_read_buffer_from_device(int devid, void *host, dt_hip_mem_t device, size_t offset, size_t size, int blocking);
int dt_hip_write_buffer_to_device(int devid, const void *host, dt_hip_mem_t device, size_t offset, size_t size, int blocking);
int dt_hip_enqueue_copy_buffer_to_buffer(int devid, dt_hip_mem_t src, dt_hip_mem_t dst, size_t srcoffset, size_t dstoffset, size_t size);
/* copy a w x h window of `bpp`-byte pixels between two linear images (used by tiling and
 * by the row-band halo exchange) */
int dt_hip_enqueue_copy_region(int devid, dt_hip_mem_t src, int src_width, int src_x, int src_y,
                               dt_hip_mem_t dst, int dst_width, int dst_x, int dst_y, int width, int height, int bpp);

/* sync + profiling: dt_opencl_finish (TRUE on success), events_* (opencl.h:343-346,594-608) */
int dt_hip_finish(int devid);
/* dt_opencl_enqueue_barrier (opencl.h:346): the stream is in order, so this only validates devid */
int dt_hip_enqueue_barrier(int devid);
/* dt_opencl_events_wait_for / events_flush (opencl.h:601-605) */
void dt_hip_events_wait_for(int devid);
int dt_hip_events_flush(int devid, int reset);
void dt_hip_events_enable(int devid, int enable);
void dt_hip_events_reset(int devid);
/* wait for all tagged launches, aggregate per tag; returns number of distinct tags.
 * Fills up to `max` entries: tags[i] -> static string, ms[i] = summed duration, counts[i]. */
int dt_hip_events_profiling(int devid, const char **tags, float *ms, int *counts, int max);
/* last error text of this thread (never NULL) */
const char *dt_hip_last_error(void);

/* ---- 2. module entry points --------------------------------------------------------- */
/* All return DT_HIP_SUCCESS or a negative error; a process_cl() stub returns (err == 0).
 * All enqueue on dt_hip_get_stream(devid) and do not synchronise. */

/* rawprepare: src/iop/rawprepare.c:467-520 (raw mosaic u16 or f32 -> normalised f32) */
typedef struct dt_hip_rawprepare_data_t
{
  int32_t x, y, width, height; /* sensor crop, rawprepare.c:94-98 */
  float sub[4];
  float div[4];
} dt_hip_rawprepare_data_t;
int dt_hip_iop_rawprepare_process(int devid, const dt_hip_piece_t *piece, const dt_hip_rawprepare_data_t *d,
                                  dt_hip_mem_t dev_in, dt_hip_mem_t dev_out);

/* temperature (white balance): src/iop/temperature.c:487-600, Bayer and 4-channel paths */
typedef struct dt_hip_temperature_data_t
{
  float coeffs[4];
} dt_hip_temperature_data_t;
int dt_hip_iop_temperature_process(int devid, const dt_hip_piece_t *piece, const dt_hip_temperature_data_t *d,
                                   dt_hip_mem_t dev_in, dt_hip_mem_t dev_out);

/* highlights, clip mode only: src/iop/highlights.c:680-789 + highlights/clip.c:62-84,
 * including the "fewer than 25 clipped photosites -> copy input" early bypass
 * (highlights.c:266-300, highlights/common.h:218). */
typedef struct dt_hip_highlights_data_t
{
  int mode;   /* DT_HIP_HIGHLIGHTS_CLIP only; other modes return DT_HIP_INVALID_ARG */
  float clip;
} dt_hip_highlights_data_t;
#define DT_HIP_HIGHLIGHTS_CLIP 0
int dt_hip_iop_highlights_process(int devid, const dt_hip_piece_t *piece, const dt_hip_highlights_data_t *d,
                                  dt_hip_mem_t dev_in, dt_hip_mem_t dev_out);
/* Row-band form (a frame split over several devices): the bypass above depends on the clipped
 * count of the WHOLE frame.  _deferred clips this band and journals into `journal` (device memory,
 * DT_HIP_HIGHLIGHTS_JOURNAL_BYTES, zeroed by the call; its leading uint64 is the band's clipped
 * count); the caller sums that uint64 over all bands, then _resolve restores the band's journalled
 * photosites if the sum is below 25.  Mosaic (1-channel) input only. */
#define DT_HIP_HIGHLIGHTS_JOURNAL_BYTES 320
int dt_hip_iop_highlights_process_deferred(int devid, const dt_hip_piece_t *piece,
                                           const dt_hip_highlights_data_t *d, dt_hip_mem_t dev_in,
                                           dt_hip_mem_t dev_out, dt_hip_mem_t journal);
int dt_hip_iop_highlights_resolve(int devid, dt_hip_mem_t dev_out, dt_hip_mem_t journal);

/* demosaic: src/iop/demosaic.c:1041-1253 dispatching to
 * rcd_demosaic (src/iop/demosaic/rcd.c:274-564) or demosaic_ppg (src/iop/demosaic/ppg.c:20-217).
 * piece->filters is the UNSHIFTED dsc_in.filters; the roi_in shift
 * (dt_dev_get_roi_filters, src/develop/imageop.c:139) is applied inside. */
#define DT_HIP_DEMOSAIC_PPG 0   /* DT_IOP_DEMOSAIC_PPG   (demosaic.c enum) */
#define DT_HIP_DEMOSAIC_AMAZE 1 /* DT_IOP_DEMOSAIC_AMAZE */
#define DT_HIP_DEMOSAIC_VNG4 2  /* DT_IOP_DEMOSAIC_VNG4  (vng.c:34-221: the interpolation the dual methods blend with) */
#define DT_HIP_DEMOSAIC_RCD 5   /* DT_IOP_DEMOSAIC_RCD   */
#define DT_HIP_DEMOSAIC_PASSTHROUGH_MONOCHROME 3 /* DT_IOP_DEMOSAIC_PASSTHROUGH_MONOCHROME: the photosite in R, G and B
                                                    (passthrough_monochrome(), src/iop/demosaic/passthrough.c:21-41) */
#define DT_HIP_DEMOSAIC_PASSTHROUGH_COLOR 4      /* DT_IOP_DEMOSAIC_PASSTHROUGH_COLOR: the photosite in its own channel, 0 in the
                                                    other two (passthrough_color(), passthrough.c:44-67).  Both take the mosaic as
                                                    it came in -- no green equilibration, demosaic.c:1111-1118 -- and leave alpha */
#define DT_HIP_DEMOSAIC_DUAL 2048 /* DEMOSAIC_DUAL (demosaic.c:110) or-ed onto RCD / AMaZE: DT_IOP_DEMOSAIC_RCD_VNG, _AMAZE_VNG --
                                     dual_demosaic(), src/iop/demosaic/dual.c:40-110: VNG4 of the mosaic (as it came in, not
                                     green-equilibrated) with two passes of colour smoothing, blended with the high-frequency
                                     interpolation by the blurred sigmoid of ITS raw detail mask around dual_thrs */
/* Around the interpolation, process() (demosaic.c:1137-1250) runs the module's optional steps:
 *   green_eq         0 DT_IOP_GREEN_EQ_NO, 1 _LOCAL: green_equilibration_lavg() (demosaic/basic.c:248-293) on the mosaic,
 *                    with green_eq_threshold = 0.0001f * img->exif_iso (demosaic.c:1049).  2 _FULL:
 *                    green_equilibration_favg() (:296-329), 3 _BOTH: _favg then _lavg.  _favg scales by the ratio of two
 *                    binary64 sums over the frame which the reference adds in an OpenMP reduction (their low bits follow
 *                    the thread count); the device adds them as double-double pairs in a fixed order: a pixel is within
 *                    one ulp of binary32 of any run of the reference (in practice equal).  Not available on row bands.
 *   median_thrs      > 0: pre_median() (basic.c:136-186) of the green sites before PPG (ppg.c:58-67); PPG only
 *   color_smoothing  0..5 passes of color_smoothing() (basic.c:191-243) on the output */
typedef struct dt_hip_demosaic_data_t
{
  uint32_t green_eq;
  uint32_t color_smoothing;
  uint32_t demosaicing_method;
  float median_thrs;
  float green_eq_threshold;
  float dual_thrs;    /* dt_iop_demosaic_data_t.dual_thrs; read with DT_HIP_DEMOSAIC_DUAL only (<= 0: no blend, dual.c:52) */
  float wb_coeffs[4]; /* piece->dsc_in.temperature.coeffs: the detail mask divides by them (dual.c:84); DUAL only */
} dt_hip_demosaic_data_t;
int dt_hip_iop_demosaic_process(int devid, const dt_hip_piece_t *piece, const dt_hip_demosaic_data_t *d,
                                dt_hip_mem_t dev_in, dt_hip_mem_t dev_out);
void dt_hip_iop_demosaic_tiling(const dt_hip_piece_t *piece, const dt_hip_demosaic_data_t *d, dt_hip_tiling_t *tiling);

/* exposure: src/iop/exposure.c:501-545, out = (in - black) * scale */
typedef struct dt_hip_exposure_data_t
{
  float black;
  float scale;
} dt_hip_exposure_data_t;
int dt_hip_iop_exposure_process(int devid, const dt_hip_piece_t *piece, const dt_hip_exposure_data_t *d,
                                dt_hip_mem_t dev_in, dt_hip_mem_t dev_out);

/* colorin / colorout matrix path: dt_colorspaces_apply_conversion -> _apply_matrix
 * (src/colorprofiles/conversion.c:593-682) with its optional source curves, clipping detour
 * and target curves (_apply_target_curves, conversion.c:546-582).  The fields are the ones
 * the accessors at conversion.c:756-834 hand to a device kernel.  LUTs are device buffers of
 * DT_HIP_LUT_SAMPLES floats; NULL or lut[0] < 0 means "channel is linear". */
#define DT_HIP_LUT_SAMPLES 0x10000
typedef struct dt_hip_conversion_t
{
  float matrix[3][4];       /* rows of conversion->matrix, 4th column ignored */
  float clip_matrix[3][4];
  int has_clipping;
  int nonlinear_source;
  int nonlinear_target;
  int blue_mapping;         /* colorin legacy hook, colorin.c:690-709 */
  float coeffs_source[3][3];
  float coeffs_target[3][3];
  dt_hip_mem_t lut_source[3];
  dt_hip_mem_t lut_target[3];
  /* lut[c][0] of each curve, copied host side when the curve is uploaded so that the
   * "linear channel" sentinel needs no device read */
  float lut_source_first[3];
  float lut_target_first[3];
} dt_hip_conversion_t;
int dt_hip_iop_colorin_process(int devid, const dt_hip_piece_t *piece, const dt_hip_conversion_t *d,
                               dt_hip_mem_t dev_in, dt_hip_mem_t dev_out);
int dt_hip_iop_colorout_process(int devid, const dt_hip_piece_t *piece, const dt_hip_conversion_t *d,
                                dt_hip_mem_t dev_in, dt_hip_mem_t dev_out);

/* color calibration: loop_switch, src/iop/channelmixerrgb.c:766-960.  Arguments are
 * loop_switch()'s own (filled by process() at channelmixerrgb.c:1960-2079). */
#define DT_HIP_ADAPTATION_LINEAR_BRADFORD 0 /* src/pixel/chromatic_adaptation.h:30-38 */
#define DT_HIP_ADAPTATION_CAT16 1
#define DT_HIP_ADAPTATION_FULL_BRADFORD 2
#define DT_HIP_ADAPTATION_XYZ 3
#define DT_HIP_ADAPTATION_RGB 4
typedef struct dt_hip_channelmixerrgb_data_t
{
  float XYZ_to_RGB[3][4];
  float RGB_to_XYZ[3][4];
  float MIX[3][4];
  float illuminant[4];
  float saturation[4];
  float lightness[4];
  float grey[4];
  float p;
  float gamut;
  int clip;
  int apply_grey;
  int adaptation; /* DT_HIP_ADAPTATION_* */
  int version;    /* CHANNELMIXERRGB_V_1..3 = 0..2 */
} dt_hip_channelmixerrgb_data_t;
int dt_hip_iop_channelmixerrgb_process(int devid, const dt_hip_piece_t *piece,
                                       const dt_hip_channelmixerrgb_data_t *d, dt_hip_mem_t dev_in,
                                       dt_hip_mem_t dev_out);

/* filmic RGB tone mapping: filmic_agx / filmic_v5 / filmic_chroma_v4 / filmic_split_v4,
 * src/iop/filmicrgb.c:2153-2587, and the 2019-2020 colour sciences filmic_split_v1 / _v2_v3, filmic_chroma_v1 /
 * _v2_v3 (:1534-1737; sigma_toe / sigma_shoulder of commit_params :4101 are derived from the spline's latitude; the
 * per-channel variants pass the input's alpha through where the reference leaves it unwritten); highlight
 * reconstruction bypassed (hl_deprecated, :2733).
 * spline == dt_iop_filmic_rgb_spline_t (:216-223).  work_*, export_* are the 3x3 parts of
 * the work / export dt_iop_order_iccprofile_info_t matrix_in, matrix_out. */
typedef struct dt_hip_filmic_spline_t
{
  float M1[4], M2[4], M3[4], M4[4], M5[4];
  float latitude_min, latitude_max;
  float y[5];
  float x[5];
  int type[2];
} dt_hip_filmic_spline_t;
typedef struct dt_hip_filmicrgb_data_t
{
  float white_source, grey_source, black_source;
  float dynamic_range;
  float saturation;
  float output_power;
  float agx_beta_hue;
  int preserve_color; /* dt_iop_filmicrgb_methods_type_t */
  int version;        /* dt_iop_filmicrgb_colorscience_type_t, 0..9 */
  int use_output_profile;
  dt_hip_filmic_spline_t spline;
  float work_matrix_in[3][4], work_matrix_out[3][4];
  float export_matrix_in[3][4], export_matrix_out[3][4];
} dt_hip_filmicrgb_data_t;
int dt_hip_iop_filmicrgb_process(int devid, const dt_hip_piece_t *piece, const dt_hip_filmicrgb_data_t *d,
                                 dt_hip_mem_t dev_in, dt_hip_mem_t dev_out);

/* final float -> integer conversion of the export driver:
 * src/imageio/imageio_core.c:706-737 (RGBA f32 -> RGBA u16 / u8) */
int dt_hip_export_convert_u16(int devid, int width, int height, dt_hip_mem_t dev_in, dt_hip_mem_t dev_out);
int dt_hip_export_convert_u8(int devid, int width, int height, dt_hip_mem_t dev_in, dt_hip_mem_t dev_out);
/* The row loop of the format writers (src/imageio/format/tiff.c:293-360; png.c / jpeg.c likewise): `layers`
 * (3, or 1 for tiff's grayscale "shortfile" mode) of the 4 samples of every pixel, packed -- the bytes of the
 * scanlines, so the download carries 3/4 of the frame.  bpp = bits per sample of dev_in: 32, 16 or 8.
 * In a pipe: node "export_rows" (data dt_hip_export_rows_t) after "export_u16" (fused into the RGBA chain) or after
 * "export_u8" (the 8-bit writers: jpeg.c, png.c at 8 bits). */
typedef struct dt_hip_export_rows_t
{
  int32_t bpp, layers;
} dt_hip_export_rows_t;
int dt_hip_export_pack_rows(int devid, int width, int height, int bpp, int layers, dt_hip_mem_t dev_in,
                            dt_hip_mem_t dev_out);

/* diffuse or sharpen: process(), src/iop/diffuse.c:1155-1258 -> wavelets_process() (:978-1106),
 * decompose_2D_Bspline() (src/pixel/bspline.h:351-377), heat_PDE_diffusion() (diffuse.c:760-968).
 * The struct is dt_iop_diffuse_params_t (diffuse.c:76-105; commit_params memcpy's it, :133-138)
 * followed by pipe->iscale, which process() reads through dt_dev_get_module_scale()
 * (src/develop/imageop.c:134-137).  threshold > 0: build_mask() / inpaint_mask() (diffuse.c:1106-1152) -- pixels
 * above the threshold are re-seeded with Box-Muller noise from a generator keyed on the pixel's position
 * (src/iop/noise_generator.h:36-93: splitmix32, xoshiro128+, logf / cosf / sinf) and only they are diffused. */
typedef struct dt_hip_diffuse_data_t
{
  int iterations;
  float sharpness;
  int radius;
  float regularization;
  float variance_threshold;
  float anisotropy_first, anisotropy_second, anisotropy_third, anisotropy_fourth;
  float threshold;
  float first, second, third, fourth;
  int radius_center;
  float iscale; /* dt_dev_pixelpipe_t::iscale, 1 for a full-resolution export */
} dt_hip_diffuse_data_t;
int dt_hip_iop_diffuse_process(int devid, const dt_hip_piece_t *piece, const dt_hip_diffuse_data_t *d,
                               dt_hip_mem_t dev_in, dt_hip_mem_t dev_out);
/* tiling_callback(), diffuse.c:585-610 */
void dt_hip_iop_diffuse_tiling(const dt_hip_piece_t *piece, const dt_hip_diffuse_data_t *d, dt_hip_tiling_t *tiling);

/* denoise (profiled), wavelets mode: process_wavelets(), src/iop/denoiseprofile.c:1289-1447, with
 * eaw_dn_decompose() / eaw_synthesize() (src/pixel/eaw.c:242-327, :157-175), the variance-stabilising
 * transforms precondition*() / backtransform*() (denoiseprofile.c:852-1095) and the BayesShrink
 * threshold variance_stabilizing_xform() (:1223-1287).
 * Fields are those of dt_iop_denoiseprofile_data_t (denoiseprofile.c:352-371) the path reads;
 * force[][] is the per-band curve sampling commit_params() leaves there (:3088-3100);
 * wb_coeffs = piece->dsc_in.temperature.coeffs (read by compute_wb_factors(), :1097-1128).
 * mode: DT_HIP_DENOISEPROFILE_WAVELETS or DT_HIP_DENOISEPROFILE_NLMEANS.
 *
 * The sum of squared detail coefficients per band is an OpenMP reduction in the reference, i.e. its
 * value depends on the host's thread count; the device (and the oracle) define it as the binary64
 * sum in the fixed order documented in DESIGN.md, rounded once to binary32. */
#define DT_HIP_DENOISEPROFILE_BANDS 7
#define DT_HIP_DENOISEPROFILE_NLMEANS 0
#define DT_HIP_DENOISEPROFILE_WAVELETS 1
/* MODE_NLMEANS_AUTO / MODE_WAVELETS_AUTO (denoiseprofile.c:124-125): the "auto" sliders are resolved into the same
 * fields at commit time, process() dispatches them like the manual modes (:2617-2621).  MODE_VARIANCE (2, a
 * noise-profiling aid) is refused. */
#define DT_HIP_DENOISEPROFILE_NLMEANS_AUTO 3
#define DT_HIP_DENOISEPROFILE_WAVELETS_AUTO 4
#define DT_HIP_DENOISEPROFILE_IS_NLMEANS(mode) ((mode) == DT_HIP_DENOISEPROFILE_NLMEANS || (mode) == DT_HIP_DENOISEPROFILE_NLMEANS_AUTO)
#define DT_HIP_DENOISEPROFILE_IS_WAVELETS(mode) ((mode) == DT_HIP_DENOISEPROFILE_WAVELETS || (mode) == DT_HIP_DENOISEPROFILE_WAVELETS_AUTO)
#define DT_HIP_DENOISEPROFILE_RGB 0
#define DT_HIP_DENOISEPROFILE_Y0U0V0 1
typedef struct dt_hip_denoiseprofile_data_t
{
  float radius, nbhood, strength, shadows, bias, scattering, central_pixel_weight, overshooting;
  float a[3], b[3];
  int mode;
  float force[6][DT_HIP_DENOISEPROFILE_BANDS]; /* rows: all, R, G, B, Y0, U0V0 */
  int wb_adaptive_anscombe;
  int fix_anscombe_and_nlmeans_norm;
  int use_new_vst;
  int wavelet_color_mode;
  float wb_coeffs[4];
} dt_hip_denoiseprofile_data_t;
int dt_hip_iop_denoiseprofile_process(int devid, const dt_hip_piece_t *piece,
                                      const dt_hip_denoiseprofile_data_t *d, dt_hip_mem_t dev_in,
                                      dt_hip_mem_t dev_out);
/* tiling_callback(), denoiseprofile.c:796-848 */
void dt_hip_iop_denoiseprofile_tiling(const dt_hip_piece_t *piece, const dt_hip_denoiseprofile_data_t *d,
                                      dt_hip_tiling_t *tiling);

/* denoise (non-local means): process() -> process_cpu(), src/iop/nlmeans.c:416-465, over
 * nlmeans_denoise(), src/pixel/nlmeans_core.c:315-532 (Lab input; patch radius P = ceil(radius * scale),
 * search radius K = ceil(7 * scale), centre weight < 0 branch, luma/chroma blend).
 * dt_hip_nlmeans_data_t == dt_iop_nlmeans_params_t (nlmeans.c:81-88).
 * The same core runs denoise (profiled) in DT_HIP_DENOISEPROFILE_NLMEANS mode
 * (process_nlmeans_cpu(), src/iop/denoiseprofile.c:1599-1648). */
typedef struct dt_hip_nlmeans_data_t
{
  float radius, strength, luma, chroma;
} dt_hip_nlmeans_data_t;
int dt_hip_iop_nlmeans_process(int devid, const dt_hip_piece_t *piece, const dt_hip_nlmeans_data_t *d,
                               dt_hip_mem_t dev_in, dt_hip_mem_t dev_out);
/* tiling_callback(), nlmeans.c:400-414 */
void dt_hip_iop_nlmeans_tiling(const dt_hip_piece_t *piece, const dt_hip_nlmeans_data_t *d, dt_hip_tiling_t *tiling);

/* RGB <-> Lab glue the pixelpipe runs around Lab modules (src/develop/pixelpipe_cpu.c:59-75 ->
 * dt_ioppr_transform_image_colorspace(), src/colorprofiles/iop_profile.c:540-596) for a linear matrix
 * work profile: _transform_rgb_to_lab_matrix() / _transform_lab_to_rgb_matrix() (:377-463).
 * matrix = the profile's RGB -> XYZ(D50) (to Lab) resp. XYZ(D50) -> RGB (from Lab) 3x3, rows padded
 * to 4.  Alpha is carried over (the pipe converts in place).  dev_in == dev_out is allowed.
 * A work profile with tone curves (profile_info->nonlinearlut, e.g. sRGB or Adobe RGB chosen as the work profile):
 * nonlinearlut != 0 and lut / unbounded_coeffs / lut_first = the profile's lut_in / unbounded_coeffs_in (to Lab, curves
 * ahead of the matrix, :389-393) resp. lut_out / unbounded_coeffs_out (from Lab, curves behind it, :455-462), as
 * _apply_tonecurves() (:332-372) reads them; device buffers of DT_HIP_LUT_SAMPLES floats, NULL or lut_first[c] < 0 =
 * the channel is linear.  Such a conversion runs in its own launch (a linear one is fused into its neighbours). */
typedef struct dt_hip_lab_data_t
{
  float matrix[3][4];
  int nonlinearlut;
  float unbounded_coeffs[3][3];
  dt_hip_mem_t lut[3];
  float lut_first[3]; /* lut[c][0], copied host side like dt_hip_conversion_t's */
} dt_hip_lab_data_t;
int dt_hip_transform_rgb_to_lab(int devid, const dt_hip_piece_t *piece, const dt_hip_lab_data_t *d,
                                dt_hip_mem_t dev_in, dt_hip_mem_t dev_out);
int dt_hip_transform_lab_to_rgb(int devid, const dt_hip_piece_t *piece, const dt_hip_lab_data_t *d,
                                dt_hip_mem_t dev_in, dt_hip_mem_t dev_out);

/* local contrast, both modes: process(), src/iop/bilat.c:330-361; bilateral grid -> dt_bilateral_init /
 * _splat / _blur / _slice, src/pixel/bilateral.c:157-393 (Lab input, only L changes).
 * dt_hip_bilat_data_t == dt_iop_bilat_params_t (bilat.c:78-86) + pipe->iscale.
 * DT_HIP_BILAT_LOCAL_LAPLACIAN (the module's default mode): local_laplacian_internal(),
 * src/pixel/locallaplacian.c:354-563, regular mode (an export has no preview boundary);
 * sigma_r / sigma_s then carry the highlights / shadows sliders, midtone the curve width.
 *
 * dt_bilateral_splat() accumulates the grid in binary32 per OpenMP slice and then merges the
 * slices, so its rounding depends on the host's thread count; the device accumulates every grid cell
 * in pixel row-major order, which is exactly what the reference computes on one thread. */
#define DT_HIP_BILAT_BILATERAL 0
#define DT_HIP_BILAT_LOCAL_LAPLACIAN 1
typedef struct dt_hip_bilat_data_t
{
  int mode;
  float sigma_r, sigma_s, detail, midtone;
  float iscale;
} dt_hip_bilat_data_t;
int dt_hip_iop_bilat_process(int devid, const dt_hip_piece_t *piece, const dt_hip_bilat_data_t *d,
                             dt_hip_mem_t dev_in, dt_hip_mem_t dev_out);
/* tiling_callback(), bilat.c:252-297 */
void dt_hip_iop_bilat_tiling(const dt_hip_piece_t *piece, const dt_hip_bilat_data_t *d, dt_hip_tiling_t *tiling);

/* finalscale: process(), src/iop/finalscale.c:117-131 -> dt_iop_clip_and_zoom_roi()
 * (src/develop/imageop_math.c:146-152) -> dt_interpolation_resample_roi(), src/pixel/interpolation.c:898-1062:
 * separable resampling from roi_in (width, height, scale) to roi_out with the export interpolator
 * (plans of _prepare_resampling_plan(), :711-895; kernels bilinear / bicubic / Mitchell, :175-296).
 * roi x/y are ignored like the module does.  scale_out == scale_in copies. */
#define DT_HIP_INTERPOLATION_BILINEAR 0
#define DT_HIP_INTERPOLATION_BICUBIC 1
#define DT_HIP_INTERPOLATION_MITCHELL 2 /* DT_INTERPOLATION_DEFAULT */
typedef struct dt_hip_finalscale_data_t
{
  int interpolation; /* the "plugins/lighttable/export/pixel_interpolator" preference */
} dt_hip_finalscale_data_t;
int dt_hip_iop_finalscale_process(int devid, const dt_hip_piece_t *piece, const dt_hip_finalscale_data_t *d,
                                  dt_hip_mem_t dev_in, dt_hip_mem_t dev_out);
/* initialscale: process(), src/iop/initialscale.c:120-127 -- dt_iop_clip_and_zoom_roi() with the regions as they are:
 * roi_in the whole input buffer at scale 1 (modify_roi_in(), :72-83), roi_out a region of the scaled image at its offset;
 * at scale 1 a crop.  Same data (the interpolator) as finalscale. */
int dt_hip_iop_initialscale_process(int devid, const dt_hip_piece_t *piece, const dt_hip_finalscale_data_t *d,
                                  dt_hip_mem_t dev_in, dt_hip_mem_t dev_out);

/* basebuffer: process(), src/iop/basebuffer.c:118-160 -- the first node of every pipe copies the region
 * roi_out of the full sensor buffer (host memory of the mipmap cache, iwidth x iheight pixels of bpp bytes,
 * unpadded rows) into the pipe's first cacheline.  On the device that is the frame's upload: one 2-D
 * host-to-device copy on the device's stream (PCIe; excluded from every HBM roofline figure).  Rows and
 * columns of roi_out beyond the sensor are left untouched, as the reference leaves them. */
int dt_hip_iop_basebuffer_process(int devid, const dt_hip_piece_t *piece, int iwidth, int iheight, int bpp,
                                  const void *host_full, dt_hip_mem_t dev_out);

/* The wire end in front of the pipe: sensor data as the file holds it -- a bit stream of `bits`-bit
 * photosites, every row starting on a byte -- unpacked into the u16 mosaic rawprepare takes, so that a batch export
 * uploads 1.25 / 1.5 / 1.75 bytes per photosite instead of 2.  In Ansel this is rawspeed's job (its decompressors for
 * uncompressed packed data; the submodule src/external/rawspeed is not part of the reference tree, so PARITY IS UNPINNED
 * by the reference: the two layouts below are restated from the formats themselves -- TIFF / DNG BitsPerSample < 16 with
 * FillOrder 1, and the little-endian packing of the 10 / 12 / 14-bit vendor containers -- and pinned against numpy's
 * unpackbits in tests/test_raw_unpack.py).
 *   DT_HIP_RAW_PACK_MSB: photosite x of a row occupies bits [x * bits, (x + 1) * bits) of the row's bytes read most
 *                        significant bit first (12 bits: b0 = p0[11:4], b1 = p0[3:0] p1[11:8], b2 = p1[7:0])
 *   DT_HIP_RAW_PACK_LSB: the row's bytes form a little-endian integer, photosite x its bits [x * bits, (x + 1) * bits)
 *                        (12 bits: p0 = b0 | (b1 & 15) << 8, p1 = b1 >> 4 | b2 << 4)
 * bits: 8, 10, 12, 14 or 16 (16: MSB = big-endian words, LSB = a copy).  row_bytes >= ceil(width * bits / 8). */
#define DT_HIP_RAW_PACK_MSB 0
#define DT_HIP_RAW_PACK_LSB 1
int dt_hip_raw_unpack(int devid, dt_hip_mem_t dev_packed, int width, int height, size_t row_bytes, int bits, int order,
                      dt_hip_mem_t dev_out_u16);

/* ---- 2b. the blend stage ---------------------------------------------------------------- */
/* dt_develop_blend_process(), src/develop/blend.c:657-965: what the pixelpipe runs after the
 * process() of every blending-capable module (src/develop/pixelpipe_cpu.c:137-228) -- build the
 * opacity mask, then blend the module's input into its output, IN PLACE in the output buffer,
 * leaving the per-pixel opacity in the alpha channel.  The struct carries the fields of
 * dt_develop_blend_params_t (src/develop/blend.h:199-244) the supported paths read, under the
 * reference's names, plus the work profile's RGB -> XYZ(D50) matrix that
 * dt_develop_blendif_init_masking_profile() (blend.c:322-353) turns into the masking profile.
 * Built: all four blend colourspaces -- "RGB (scene)" (src/develop/blends/blendif_rgb_jzczhz.c), "RGB (display)"
 * (blendif_rgb_hsl.c), "Lab" (blendif_lab.c), "raw" (blendif_raw.c: one-channel buffers before demosaic); the RGB
 * ones with a LINEAR work profile (nonlinearlut == 0) for the gray channel of the mask.  Mask modes uniform and
 * parametric -- gray, R, G, B and Jz, Cz, hz resp. H, S, L; L, a, b, C, h -- on input and output, all combine / invert
 * variants; the post operations mask blur (the recursive gaussian of src/pixel/gaussian.c, blend.c:869-881) and mask
 * tone curve (contrast / brightness, blend.c:626-655); every operator of the four files (16, 30, 27, 17) and the
 * reverse flag.  Drawn and raster masks -- and the details threshold that refines them -- arrive the way the reference's
 * own device blend takes them (blend.c:1278-1325: the host rasterises the forms, combines them with the raster mask,
 * refines with the detail mask, and uploads ONE float plane): `form_mask`, a plane of roi_out's size.  It replaces the
 * constant form mask of a parametric-only blend in make_mask() (blendif_*.c) and is followed by the same post operations;
 * a raster mask alone (no drawn form, no active parametric channel) is form * opacity (blend.c:740-745).
 * Feathering (blend.c:603-623, :825-852): the guided filter of src/pixel/guided_filter.c over the mask, guided by the
 * module's input or output (feathering_guide), before or after the blur as _develop_mask_get_post_operations() orders
 * them (blend.c:427-469) -- its 512-pixel tile grid and Kahan box means reproduced, bit-identical to the CPU path.
 * The details threshold (blend.c:361-425) runs on the device when the raw detail mask is at hand (`detail_mask`).
 * Refused with DT_HIP_INVALID_ARG (never approximated): GUI mask display, a drawn / raster mask mode without the plane,
 * a details threshold with neither the raw detail mask nor a plane refined by the host. */
#define DT_HIP_BLEND_CS_RAW 1 /* dt_develop_blend_colorspace_t, blend.h:51-58 */
#define DT_HIP_BLEND_CS_LAB 2
#define DT_HIP_BLEND_CS_RGB_DISPLAY 3
#define DT_HIP_BLEND_CS_RGB_SCENE 4
#define DT_HIP_MASK_ENABLED 1u /* dt_develop_mask_mode_t, blend.h:110-118 */
#define DT_HIP_MASK_SHAPE 2u
#define DT_HIP_MASK_PARAMETRIC 4u
#define DT_HIP_MASK_RASTER 8u
#define DT_HIP_COMBINE_INV 1u /* dt_develop_mask_combine_mode_t, blend.h:120-131 */
#define DT_HIP_COMBINE_INCL 2u
#define DT_HIP_MASK_GUIDE_IN_BEFORE_BLUR 0x01u /* dt_develop_mask_feathering_guide_t, blend.h:133-139 */
#define DT_HIP_MASK_GUIDE_OUT_BEFORE_BLUR 0x02u
#define DT_HIP_MASK_GUIDE_IN_AFTER_BLUR 0x05u
#define DT_HIP_MASK_GUIDE_OUT_AFTER_BLUR 0x06u
#define DT_HIP_BLEND_REVERSE 0x80000000u /* dt_develop_blend_mode_t flag, blend.h:106 */
#define DT_HIP_BLENDIF_SIZE 16
typedef struct dt_hip_blend_data_t
{
  uint32_t mask_mode;     /* DT_HIP_MASK_* bits */
  int32_t blend_cst;      /* DT_HIP_BLEND_CS_* */
  uint32_t blend_mode;    /* DEVELOP_BLEND_* value (blend.h:61-107), DT_HIP_BLEND_REVERSE or-ed in */
  float blend_parameter;  /* exposure fulcrum of the scene-referred operators, EV */
  float opacity;          /* 0 .. 100 */
  uint32_t mask_combine;  /* DT_HIP_COMBINE_* bits */
  uint32_t blendif;       /* bit i: channel i active; bit 16 + i: channel i inverted (blend.h:141-197) */
  float feathering_radius, blur_radius, details; /* details: see detail_mask */
  float contrast, brightness;                    /* mask tone curve */
  float blendif_parameters[4 * DT_HIP_BLENDIF_SIZE];
  float blendif_boost_factors[DT_HIP_BLENDIF_SIZE];
  float matrix_in[3][4]; /* dt_iop_order_iccprofile_info_t.matrix_in of the work profile */
  dt_hip_mem_t form_mask; /* NULL, or the host-rendered form mask: roi_out.width x roi_out.height floats on the device
                             (dt_hip_copy_host_to_device(devid, mask, width, height, 4)); required when mask_mode has
                             DT_HIP_MASK_SHAPE / _RASTER or details != 0 */
  uint32_t feathering_guide; /* DT_HIP_MASK_GUIDE_*; read when feathering_radius > 0.1 */
  dt_hip_mem_t detail_mask;  /* NULL, or the raw detail mask of the hidden "detailmask" stage (dt_hip_iop_detailmask_process):
                                roi_out.width x roi_out.height floats on the device.  With details != 0 the blend then
                                refines the form mask itself, _refine_with_detail_mask() (blend.c:361-425, :789): the
                                sigmoid of dt_masks_calc_detail_mask() around the threshold, its 9 x 9 blur, times the
                                form mask (or the neutral fill of a parametric-only blend).  The mask must have the
                                geometry of roi_out (no distorting module between the two stages: the warp of
                                dt_dev_distort_detail_mask() is not built).  NULL with details != 0: form_mask is taken as
                                already refined by the host */
} dt_hip_blend_data_t;
/* dev_in: the module's input, roi_in; dev_out: the module's output, roi_out, blended in place */
int dt_hip_develop_blend_process(int devid, const dt_hip_piece_t *piece, const dt_hip_blend_data_t *d,
                                 dt_hip_mem_t dev_in, dt_hip_mem_t dev_out);

/* The hidden "detailmask" stage behind demosaic (src/iop/detailmask.c:111-170): copies its input and leaves the raw
 * detail mask of dt_masks_calc_rawdetail_mask() (src/develop/masks/detail.c:283-316) -- the Scharr gradient magnitude of
 * sqrt(mean of the white-balance-normalised, clipped-at-0 RGB), borders extended by one -- in `mask`
 * (roi_out.width x roi_out.height floats on the device, the caller's buffer).  wb: piece->dsc_in.temperature.coeffs when
 * the white balance is enabled, else 1, 1, 1.  Frames below 3 x 3 are refused. */
typedef struct dt_hip_detailmask_data_t
{
  float wb[4];
  dt_hip_mem_t mask;
} dt_hip_detailmask_data_t;
int dt_hip_iop_detailmask_process(int devid, const dt_hip_piece_t *piece, const dt_hip_detailmask_data_t *d,
                                  dt_hip_mem_t dev_in, dt_hip_mem_t dev_out);

/* ---- 3. export-pipe executor ----------------------------------------------------------- */
/* The device-resident part of dt_dev_pixelpipe_process_rec() (src/develop/pixelpipe_hb.c:881-1282)
 * for an export: an ordered list of module nodes, each with its piece view and committed data,
 * run on one device from a raw buffer in HBM to the exported buffer in HBM.  Intermediate module
 * outputs are taken from and returned to the runtime's pool (they are cachelines nobody re-reads
 * in an export).  With fusion enabled (default) maximal runs of pointwise modules execute as one
 * kernel each -- rawprepare/temperature/highlights before demosaic; exposure/colorin/
 * channelmixerrgb/filmicrgb/colorout/export_u16 after -- with results bit-identical to the
 * module-by-module chain.  `op` is the module's op name ("rawprepare", "temperature",
 * "highlights", "demosaic", "denoiseprofile", "exposure", "colorin", "channelmixerrgb", "diffuse",
 * "nlmeans", "filmicrgb", "colorout"), the colourspace glue "rgb_to_lab" / "lab_to_rgb", or
 * "export_u16" (data NULL) for the final float -> u16 of src/imageio/imageio_core.c:729. */
typedef struct dt_hip_pipe_t dt_hip_pipe_t;
dt_hip_pipe_t *dt_hip_pipe_new(int devid);
void dt_hip_pipe_free(dt_hip_pipe_t *pipe);
int dt_hip_pipe_add_node(dt_hip_pipe_t *pipe, const char *op, const dt_hip_piece_t *piece, const void *data,
                         size_t data_size);
void dt_hip_pipe_set_fusion(dt_hip_pipe_t *pipe, int enabled);
/* number of kernel groups the current node list is executed as (after fusion planning) */
int dt_hip_pipe_num_groups(dt_hip_pipe_t *pipe);
int dt_hip_pipe_process(dt_hip_pipe_t *pipe, dt_hip_mem_t dev_in, dt_hip_mem_t dev_out);

/* ---- 3a. a stream of frames through one pipe (batch export) ---------------------------------- */
/* The export loop of a batch (src/cli + dt_imageio_export_with_flags(), src/imageio/imageio_core.c) pushes one
 * frame after the other through the same pipe.  Serially a frame costs upload + kernels + download; here the three
 * legs of consecutive frames overlap: `depth` slots, each with its own device input and output buffer, the upload
 * on one copy stream, the kernels on the device's compute stream, the download on a second copy stream, ordered by
 * events.  Host buffers must be pinned (dt_hip_alloc_host_pinned) and stay untouched until the slot is waited for.
 *   slot = dt_hip_batch_submit(b, host_in, host_out)   enqueue a frame; blocks only when every slot is in flight
 *   dt_hip_batch_wait(b, slot)                          the frame submitted into `slot` is in host_out
 *   dt_hip_batch_drain(b)                               all of them are */
typedef struct dt_hip_batch_t dt_hip_batch_t;
dt_hip_batch_t *dt_hip_batch_new(dt_hip_pipe_t *pipe, int depth, size_t in_bytes, size_t out_bytes);
void dt_hip_batch_free(dt_hip_batch_t *batch);
int dt_hip_batch_submit(dt_hip_batch_t *batch, const void *host_in, void *host_out);
int dt_hip_batch_wait(dt_hip_batch_t *batch, int slot);
int dt_hip_batch_drain(dt_hip_batch_t *batch);
/* The fourth leg -- the file encoder.  dt_imageio_export_with_flags() calls format->write_image() on the exported buffer
 * after the pipe has run (src/imageio/imageio_core.c:965), frame after frame on the exporting thread; with a writer set
 * the batch calls it for frame n on a host thread of its own as soon as the frame's download has landed, while frames
 * n + 1 ... are uploaded, processed and downloaded.  One writer thread: frames are written in submission order, `seq`
 * counts them from 0 since dt_hip_batch_new().  A frame's slot is done (dt_hip_batch_wait) when its writer has returned --
 * host_out is the writer's until then -- and a writer that returns non-zero makes that wait (and a drain, and the submit
 * that needs the slot) return DT_HIP_WRITER_FAILED; the frames behind it are still written.  Set or cleared (NULL)
 * between frames: the call drains the batch first. */
typedef int (*dt_hip_batch_writer_t)(void *user, long seq, void *host_out, size_t bytes);
int dt_hip_batch_set_writer(dt_hip_batch_t *batch, dt_hip_batch_writer_t writer, void *user);

/* ---- 3b. one frame over several devices: row bands ---------------------------------------- */
/* Replaces default_process_tiling_cl() / _default_process_tiling_cl_ptp() (src/develop/tiling.c:842,
 * :1394) for a frame that is split because there are several GPUs, not because it does not fit.
 * The reference re-runs each module on overlapping tiles and accepts seam differences for modules
 * whose result depends on the tile origin (RCD); here a band runs the frame's OWN RCD tile rows, so
 * the assembled bands are bit-identical to the unsplit frame.
 *
 * A band owns frame rows [row0, row0 + rows) of every module output (all modules of the export
 * pipe run at scale 1 with identical geometry; rawprepare's crop is the only offset).  Bands are
 * cut on RCD tile rows: row0 = 94 * tv0 + 9 (0 for the first band).  The demosaic of a band reads
 * `halo` = 9 mosaic rows of each neighbour; those rows are exchanged once, after the CFA stages.  With the
 * AMaZE demosaic the cuts are its own tile rows (row0 = 128 * tv0, amaze.cc:181-350) and the halo 16 rows:
 *
 *   dt_hip_pipe_band_begin()   CFA stages on the band's own rows, into a buffer laid out as
 *                              [halo_top rows][rows][halo_bottom rows] of `row_bytes` each
 *   -- caller: all-reduce `clipped_count` (a device uint64; NULL if the pipe has no highlights) --
 *   dt_hip_pipe_band_resolve() highlights bypass decision on the summed count (own rows final)
 *   -- caller: fill the halo rows from the neighbours' last / first own rows (send/recv) --
 *   dt_hip_pipe_band_finish()  demosaic on the frame's tile rows, RGBA stages on the band's own
 *                              rows -> dev_out_band (rows x width)
 *
 * dev_in_band holds the band's input rows only: input rows [crop_y + row0, crop_y + row0 + rows),
 * full input width.
 *
 * Stencil modules after the demosaic (denoiseprofile, diffuse, nlmeans): dt_hip_pipe_band_finish() is
 * resumable.  It returns DT_HIP_BAND_EXCHANGE (> 0) when the band needs something from the others, with the
 * request in `state`; the caller serves it and calls again with the same arguments, until DT_HIP_SUCCESS
 * (or an error, which ends the walk and frees the state):
 *
 *   state->halo_rows = h > 0   halo_buf holds RGBA rows [min(h, row0)][rows][min(h, H - row0 - rows)] of
 *                              `row_bytes`; fill the first and last part with the neighbours' last / first
 *                              own rows of the same buffer.  h is the same on every band: in front of a
 *                              module it is dt_hip_band_halo_rows(); the profiled wavelets then stop once
 *                              more before each later scale k with h = 2 * 2^k rows of their own coarse
 *                              plane.  A neighbour must own at least h rows.
 *   state->sum_buf != NULL     sum_count doubles to all-reduce (SUM) over the bands: the frame-wide table of
 *                              partial sums of the profiled wavelets (eaw.c:253-255), each band having filled
 *                              the entries of its own rows and zeroed the rest, so the reduced table -- and
 *                              the thresholds taken from it in a fixed order -- do not depend on how the
 *                              collective associates.
 *
 * What a band computes is bit-identical to the rows of the unsplit frame: the wavelets compute their own rows
 * of every scale from own + fetched rows of the scale before; diffuse runs on
 * [halo][rows][halo] as on a frame of its own (h covers every stencil of every iteration / scale, so only halo
 * rows see the artificial border); non-local means runs the chunk rows of the FRAME's grid that intersect the
 * band (nlmeans_core.c:264-313: the grid is a function of the frame size).  Blend nodes with uniform, parametric or
 * host-rendered form masks (form_mask = the FRAME's plane, whole on every band's device) are pointwise and run on the
 * band; local contrast's bilateral grid is relayed from band to band (relay_buf below).  Refused in band mode: the local
 * laplacian, finalscale / initialscale, blends with a mask blur or mask feathering. */
#define DT_HIP_BAND_EXCHANGE 1
typedef struct dt_hip_band_t
{
  int32_t row0, rows;
  int32_t halo_top, halo_bottom; /* mosaic rows needed from the bands above / below */
  int32_t tile_row0, tile_row1;  /* RCD (or AMaZE) tile rows [tile_row0, tile_row1) of the frame */
} dt_hip_band_t;
typedef struct dt_hip_band_state_t
{
  dt_hip_mem_t halo_buf; /* NULL when the pipe has no demosaic node */
  size_t row_bytes;
  dt_hip_mem_t clipped_count;
  void *priv;
  /* requests of a dt_hip_pipe_band_finish() that returned DT_HIP_BAND_EXCHANGE */
  int32_t halo_rows;
  int32_t sum_planes; /* sum_buf = sum_planes tables of [frame rows][segments][4] doubles; a band's own rows are the
                         only non-zero entries of its table, so SUM over the bands is an all-gather of row segments */
  double *sum_buf;
  size_t sum_count;
  /* a RELAY stop (local contrast, bilateral grid: one accumulation over the frame in pixel order): relay_buf is this
   * band's copy of the frame's grid (relay_bytes).  The driver, for the bands in order 0 .. n-1: copies band k-1's
   * relay_buf into band k's (k > 0), calls dt_hip_pipe_band_relay() for band k (its rows are accumulated on top);
   * then copies band n-1's relay_buf -- the complete grid -- into every other band's, and resumes the walk. */
  dt_hip_mem_t relay_buf;
  size_t relay_bytes;
} dt_hip_band_state_t;
/* pure function (no device needed): cut a height-row frame into n_bands bands for the given
 * demosaic method (DT_HIP_DEMOSAIC_RCD, or -1 for a pipe without demosaic: 2-row aligned cuts) */
int dt_hip_plan_bands(int width, int height, int demosaic_method, int n_bands, dt_hip_band_t *bands);
/* pure function: rows of a neighbour a band needs in front of module `op` (0 for a pointwise module, -1 when
 * the module has no band mode or passes a frame of this size through).  piece = the module's FRAME geometry */
int dt_hip_band_halo_rows(const char *op, const dt_hip_piece_t *piece, const void *data, size_t data_size);
int dt_hip_pipe_band_begin(dt_hip_pipe_t *pipe, const dt_hip_band_t *band, dt_hip_mem_t dev_in_band,
                           dt_hip_band_state_t *state);
int dt_hip_pipe_band_resolve(dt_hip_pipe_t *pipe, const dt_hip_band_t *band, dt_hip_band_state_t *state);
int dt_hip_pipe_band_finish(dt_hip_pipe_t *pipe, const dt_hip_band_t *band, dt_hip_band_state_t *state,
                            dt_hip_mem_t dev_out_band);
/* the band's turn in a relay stop (see dt_hip_band_state_t.relay_buf) */
int dt_hip_pipe_band_relay(dt_hip_pipe_t *pipe, const dt_hip_band_t *band, dt_hip_band_state_t *state);
/* give up a band whose walk has not ended (an error on another rank, a cancelled export): frees the state */
void dt_hip_pipe_band_abort(dt_hip_pipe_t *pipe, dt_hip_band_state_t *state);
/* The whole walk from ONE process, for a host that is one C process like the reference (src/develop/pixelpipe_hb.c:1470
 * runs every pipe of the application; default_process_tiling_cl(), tiling.c:1394, is its only way to split a frame):
 * band k of the frame runs on pipes[k] -- n pipes loaded with the same node list, each on its own device (or all on
 * one: the single-GPU test) -- from dev_in[k] (the band's rows of the input, resident on that device) into
 * dev_out[k].  One host thread per band inside the call; halo rows, the wavelets' partial sums and the bilateral grid
 * travel as peer copies between the devices (xGMI; peer copies, not RCCL -- one process owns every device), ordered by
 * hipEvents between the bands' streams: a band waits for the two bands it reads from, no stream is drained inside the
 * walk; the 8-byte clipped count travels through the host.  Returns when every band's rows are written (streams
 * drained).  The assembled rows are bit-identical to dt_hip_pipe_process() on the whole frame. */
int dt_hip_pipe_process_bands(dt_hip_pipe_t *const *pipes, int n_bands, const dt_hip_band_t *bands,
                              const dt_hip_mem_t *dev_in, const dt_hip_mem_t *dev_out);

/* What the last dt_hip_pipe_process_bands() call did between the devices: bands, distinct devices, exchange stops per
 * band, device-to-device copies and their bytes (zero when every band ran on one device), the host time the band threads
 * spent waiting for each other to get as far enqueueing (never for a device), and ordered device pairs WITHOUT peer
 * access (their copies are staged through the host by the runtime: correct, not xGMI speed). */
typedef struct dt_hip_band_stats_t
{
  int32_t bands, devices, exchange_stops, pairs_without_peer_access;
  uint64_t peer_copies, peer_bytes, host_wait_ns;
} dt_hip_band_stats_t;
void dt_hip_pipe_bands_stats(dt_hip_band_stats_t *out);
/* Before the first multi-device walk: can every device of `devids` reach every other (hipDeviceCanAccessPeer), and does a
 * device-to-device copy ordered by a cross-device event arrive intact?  0, or an error naming the pair
 * (dt_hip_last_error()).  With one device in the list it checks the same calls on that device. */
int dt_hip_peer_selftest(const int *devids, int n);

/* layout self-check for language bindings: sizeof() of the struct named `name` as compiled */
size_t dt_hip_abi_sizeof(const char *name);
/* dt_rawspeed_crop_dcraw_filters(): shift a dcraw Bayer filter word by a crop origin */
uint32_t dt_hip_crop_dcraw_filters(uint32_t filters, uint32_t crop_x, uint32_t crop_y);

#ifdef __cplusplus
}
#endif
#endif /* ANSEL_HIP_H */
