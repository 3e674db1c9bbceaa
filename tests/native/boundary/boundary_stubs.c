/* boundary_stubs.c -- TEST INFRASTRUCTURE (tests/test_boundary_compile.py).
 *
 * Linked with the reference's OWN src/develop/pixelpipe_gpu.c and src/develop/tiling.c (compiled from where they lie,
 * unmodified, against include/ansel_opencl_peer.h) and with libansel_hip.so:
 *   * the host-side services those two files call -- cache lines, logging, the CPU fallback's entry -- in the simplest
 *     form that works for one module run (a cache line is a host buffer with at most one device payload);
 *   * modules written the way INTEGRATION.md section 2 tells a maintainer to write them (integration_stubs.h, over the
 *     reference's own dt_iop_<op>_data_t): `exposure` (pointwise) and `diffuse` (a stencil module: tiles overlap), and the
 *     blend stage dt_develop_blend_process_cl() behind a module that supports blending;
 *   * the drivers the test calls: the reference's pixelpipe_process_on_GPU() on exposure, plain and blended (device path,
 *     output synced to the host cache line), and the reference's default_process_tiling_cl() on either module (host-tiled
 *     path, the budget squeezed so that the frame takes many tiles; every process_cl() call's region is logged).
 */
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "common/opencl.h" /* = include/ansel_opencl_peer.h + boundary_host.h */
#include "develop/pixelpipe_process.h"
#include "develop/pixelpipe_cpu.h"
#include "develop/pixelpipe_gpu.h"
#include "develop/tiling.h"

#include "integration_stubs.h" /* the process_cl() / blend bodies of INTEGRATION.md section 2 over the reference's own data types */

/* ---- logging ------------------------------------------------------------------------------------------------------ */
static unsigned int g_debug = 0;
static char g_last_message[512];
void dt_print(dt_debug_thread_t thread, const char *msg, ...)
{
  if(!(g_debug & thread)) return;
  va_list ap;
  va_start(ap, msg);
  vfprintf(stderr, msg, ap);
  va_end(ap);
}
void dt_vprint(dt_debug_thread_t thread, const char *msg, ...)
{
  (void)thread;
  (void)msg;
}
unsigned int dt_get_debug_flags(void) { return g_debug; }
void dt_pipeline_message(const char *format, ...)
{
  va_list ap;
  va_start(ap, format);
  vsnprintf(g_last_message, sizeof(g_last_message), format, ap);
  va_end(ap);
}
const char *boundary_last_message(void) { return g_last_message; }

/* ---- host memory (src/caches/pixelpipe_cache_alloc.h, src/system/sys_resources.h) ---------------------------------- */
void *dt_pixelpipe_cache_alloc_align_cache_impl(size_t size, int id, const char *name)
{
  (void)id;
  (void)name;
  return aligned_alloc(64, (size + 63) & ~(size_t)63);
}
void dt_pixelpipe_cache_free_align_cache(void **mem, const char *message)
{
  (void)message;
  if(mem && *mem)
  {
    free(*mem);
    *mem = NULL;
  }
}
size_t dt_pixelpipe_cache_get_largest_free_run(void) { return (size_t)8 << 30; }
size_t dt_get_available_mem() { return (size_t)8 << 30; }
void dt_dev_pixelpipe_cache_get_usage(size_t *current, size_t *max)
{
  if(current) *current = 0;
  if(max) *max = (size_t)8 << 30;
}
int dt_dev_pixel_pipe_cache_remove_lru(void) { return 1; }

/* ---- cache lines (src/caches/pixelpipe_cache.h) ---------------------------------------------------------------------- */
void *dt_pixel_cache_entry_get_data(struct dt_pixel_cache_entry_t *entry) { return entry ? entry->data : NULL; }
void *dt_pixel_cache_alloc(struct dt_pixel_cache_entry_t *entry)
{
  if(entry && !entry->data) entry->data = aligned_alloc(64, (entry->size + 63) & ~(size_t)63);
  return entry ? entry->data : NULL;
}
void dt_dev_pixelpipe_cache_wrlock_entry(gboolean lock, struct dt_pixel_cache_entry_t *entry)
{
  (void)lock;
  (void)entry;
}
void dt_dev_pixelpipe_cache_rdlock_entry(gboolean lock, struct dt_pixel_cache_entry_t *entry)
{
  (void)lock;
  (void)entry;
}
gboolean dt_dev_pixelpipe_cache_gpu_device_buffer(const dt_dev_pixelpipe_t *pipe, const dt_pixel_cache_entry_t *cache_entry)
{
  (void)cache_entry;
  return pipe && pipe->devid >= 0;
}
int dt_dev_pixelpipe_cache_sync_cl_buffer(int devid, void *host_ptr, void *cl_mem_buffer, const dt_iop_roi_t *roi, int cl_mode,
                                          size_t bpp, struct dt_iop_module_t *module, const char *message)
{
  (void)module;
  (void)message;
  if(!host_ptr || !cl_mem_buffer) return 1;
  const int err = (cl_mode & CL_MAP_WRITE)
                      ? dt_opencl_write_host_to_device(devid, host_ptr, cl_mem_buffer, roi->width, roi->height, (int)bpp)
                      : dt_opencl_read_host_from_device(devid, host_ptr, cl_mem_buffer, roi->width, roi->height, (int)bpp);
  return err != CL_SUCCESS;
}
float *dt_dev_pixelpipe_cache_restore_cl_buffer(struct dt_dev_pixelpipe_t *pipe, float *input, void *cl_mem_input,
                                                const dt_iop_roi_t *roi_in, struct dt_iop_module_t *module, size_t in_bpp,
                                                struct dt_pixel_cache_entry_t *input_entry, const char *message)
{
  if(!input) input = dt_pixel_cache_alloc(input_entry);
  if(input && cl_mem_input
     && dt_dev_pixelpipe_cache_sync_cl_buffer(pipe->devid, input, cl_mem_input, roi_in, CL_MAP_READ, in_bpp, module, message))
    return NULL;
  return input;
}
void dt_dev_pixelpipe_cache_release_cl_buffer(void **cl_mem_buffer, struct dt_pixel_cache_entry_t *entry, void *host_ptr,
                                              gboolean cache_device)
{
  (void)host_ptr;
  if(!cl_mem_buffer || !*cl_mem_buffer) return;
  if(cache_device && entry && !entry->cl_mem)
  {
    entry->cl_mem = *cl_mem_buffer; /* the line keeps its device payload for the next module */
    entry->cl_width = dt_opencl_get_image_width(*cl_mem_buffer);
    entry->cl_height = dt_opencl_get_image_height(*cl_mem_buffer);
    entry->cl_bpp = dt_opencl_get_image_element_size(*cl_mem_buffer);
  }
  else if(!entry || entry->cl_mem != *cl_mem_buffer)
    dt_opencl_release_mem_object(*cl_mem_buffer);
  *cl_mem_buffer = NULL;
}
void *dt_dev_pixelpipe_cache_borrow_cl_payload(struct dt_pixel_cache_entry_t *entry, int devid, int width, int height, int bpp)
{
  (void)devid;
  if(entry && entry->cl_mem && entry->cl_width == width && entry->cl_height == height && entry->cl_bpp == bpp) return entry->cl_mem;
  return NULL;
}
void dt_dev_pixelpipe_cache_return_cl_payload(struct dt_pixel_cache_entry_t *entry, void *mem)
{
  (void)entry;
  (void)mem;
}
void *dt_dev_pixelpipe_cache_alloc_cl_device_buffer(int devid, const dt_iop_roi_t *roi, size_t bpp,
                                                    const struct dt_iop_module_t *module, const char *message, void *keep)
{
  (void)module;
  (void)message;
  (void)keep;
  return dt_opencl_alloc_device(devid, roi->width, roi->height, (int)bpp);
}
void *dt_dev_pixelpipe_cache_get_cl_buffer(int devid, void *host_ptr, const dt_iop_roi_t *roi, size_t bpp,
                                           struct dt_iop_module_t *module, const char *message,
                                           struct dt_pixel_cache_entry_t *entry, gboolean *out_reused, void *keep)
{
  (void)host_ptr;
  (void)entry;
  if(out_reused) *out_reused = FALSE;
  return dt_dev_pixelpipe_cache_alloc_cl_device_buffer(devid, roi, bpp, module, message, keep);
}
int dt_dev_pixelpipe_cache_prepare_cl_input(struct dt_dev_pixelpipe_t *pipe, struct dt_iop_module_t *module, float *input,
                                            void **cl_mem_input, const dt_iop_roi_t *roi_in, size_t in_bpp,
                                            struct dt_pixel_cache_entry_t *input_entry,
                                            struct dt_pixel_cache_entry_t **locked_input_entry, void *keep)
{
  (void)input_entry;
  if(locked_input_entry) *locked_input_entry = NULL;
  if(*cl_mem_input) return 0; /* borrowed from the line */
  if(!input) return 1;
  *cl_mem_input = dt_dev_pixelpipe_cache_alloc_cl_device_buffer(pipe->devid, roi_in, in_bpp, module, "input", keep);
  if(!*cl_mem_input) return 1;
  return dt_dev_pixelpipe_cache_sync_cl_buffer(pipe->devid, input, *cl_mem_input, roi_in, CL_MAP_WRITE, in_bpp, module, "input");
}
gboolean dt_dev_pixelpipe_cache_flush_host_pinned_image(void *host_ptr, struct dt_pixel_cache_entry_t *entry_hint, int devid)
{
  (void)host_ptr;
  (void)entry_hint;
  (void)devid;
  return FALSE;
}
void dt_dev_pixelpipe_cache_flush_clmem(const int devid) { (void)devid; }

/* ---- what the harness' module never reaches: blending, colourspace conversion, the CPU fallback ------------------------ */
static int g_unexpected = 0;
int boundary_unexpected_calls(void) { return g_unexpected; }
void dt_dev_pixelpipe_debug_dump_module_io(dt_dev_pixelpipe_t *pipe, dt_iop_module_t *module, const char *stage, gboolean is_cl,
                                           const dt_iop_buffer_dsc_t *in_dsc, const dt_iop_buffer_dsc_t *out_dsc,
                                           const dt_iop_roi_t *roi_in, const dt_iop_roi_t *roi_out, size_t in_bpp, size_t out_bpp,
                                           int cst_before, int cst_after)
{
  (void)pipe; (void)module; (void)stage; (void)is_cl; (void)in_dsc; (void)out_dsc; (void)roi_in; (void)roi_out; (void)in_bpp;
  (void)out_bpp; (void)cst_before; (void)cst_after;
}
dt_pixelpipe_blend_transform_t dt_dev_pixelpipe_transform_for_blend(const dt_iop_module_t *self, const dt_dev_pixelpipe_iop_t *piece,
                                                                    const dt_iop_buffer_dsc_t *output_dsc)
{
  (void)self; (void)piece; (void)output_dsc;
  return DT_DEV_PIXELPIPE_BLEND_TRANSFORM_NONE;
}
int dt_develop_blend_process(struct dt_iop_module_t *self, struct dt_dev_pixelpipe_t *pipe,
                             const struct dt_dev_pixelpipe_iop_t *piece, const void *const i, void *const o)
{
  (void)self; (void)pipe; (void)piece; (void)i; (void)o;
  g_unexpected++;
  return 1;
}
static int g_blend_cl_calls = 0;
int dt_develop_blend_process_cl(struct dt_iop_module_t *self, struct dt_dev_pixelpipe_t *pipe,
                                const struct dt_dev_pixelpipe_iop_t *piece, void *dev_in, void *dev_out)
{
  g_blend_cl_calls++;
  return stub_develop_blend_process_cl(self, pipe, piece, dev_in, dev_out); /* INTEGRATION.md section 2, the blend stage */
}
dt_iop_colorspace_type_t dt_develop_blend_colorspace(const struct dt_dev_pixelpipe_iop_t *const piece, dt_iop_colorspace_type_t cst)
{
  (void)piece;
  return cst;
}
/* the pipe's work profile: none for the modules that do not ask, the one the test installed for those that do */
static dt_iop_order_iccprofile_info_t g_work_profile;
static int g_have_work_profile = 0;
dt_iop_order_iccprofile_info_t *dt_ioppr_get_pipe_work_profile_info(const struct dt_dev_pixelpipe_t *pipe)
{
  (void)pipe;
  return g_have_work_profile ? &g_work_profile : NULL;
}

/* ---- the prepared conversion behind the reference's accessor API (colorprofiles/conversion.h:225-345): a test double that
 *      holds what dt_colorspaces_prepare_conversion() would have derived from two profiles */
struct dt_colorspaces_conversion_t
{
  dt_colormatrix_t matrix, clip_matrix;
  int has_clipping;
  const float *source_curve[3], *target_curve[3];
  float source_coeffs[9], target_coeffs[9];
};
uint64_t dt_colorspaces_conversion_identity(const dt_colorspaces_conversion_t *const c) { return (uint64_t)(uintptr_t)c; }
gboolean dt_colorspaces_conversion_is_matrix(const dt_colorspaces_conversion_t *const c) { return c != NULL; }
gboolean dt_colorspaces_conversion_has_clipping(const dt_colorspaces_conversion_t *const c) { return c && c->has_clipping; }
gboolean dt_colorspaces_conversion_matrix(const dt_colorspaces_conversion_t *const c, dt_colormatrix_t m)
{
  if(!c) return FALSE;
  memcpy(m, c->matrix, sizeof(dt_colormatrix_t));
  return TRUE;
}
gboolean dt_colorspaces_conversion_source_matrix(const dt_colorspaces_conversion_t *const c, dt_colormatrix_t m)
{
  return dt_colorspaces_conversion_matrix(c, m);
}
gboolean dt_colorspaces_conversion_clip_matrix(const dt_colorspaces_conversion_t *const c, dt_colormatrix_t m)
{
  if(!c || !c->has_clipping) return FALSE;
  memcpy(m, c->clip_matrix, sizeof(dt_colormatrix_t));
  return TRUE;
}
const float *dt_colorspaces_conversion_source_curve(const dt_colorspaces_conversion_t *const c, const int ch) { return c->source_curve[ch]; }
const float *dt_colorspaces_conversion_target_curve(const dt_colorspaces_conversion_t *const c, const int ch) { return c->target_curve[ch]; }
const float *dt_colorspaces_conversion_source_coeffs(const dt_colorspaces_conversion_t *const c) { return c->source_coeffs; }
const float *dt_colorspaces_conversion_target_coeffs(const dt_colorspaces_conversion_t *const c) { return c->target_coeffs; }
int dt_ioppr_get_iop_order(GList *iop_order_list, const char *op_name, const int multi_priority)
{
  (void)iop_order_list; (void)op_name; (void)multi_priority;
  return 0;
}
void dt_colorspaces_apply_profile(const char *const op_name, const char *const instance_name, const float *const image_in,
                                  float *const image_out, const int width, const int height, const int cst_from,
                                  const int cst_to, int *converted_cst, const dt_iop_order_iccprofile_info_t *const profile_info)
{
  (void)op_name; (void)instance_name; (void)image_in; (void)image_out; (void)width; (void)height; (void)cst_from; (void)cst_to;
  (void)converted_cst; (void)profile_info;
  g_unexpected++;
}
int dt_colorspaces_apply_profile_cl(const char *const op_name, const char *const instance_name, const int devid, void *dev_img_in,
                                    void *dev_img_out, const int width, const int height, const int cst_from, const int cst_to,
                                    int *converted_cst, const dt_iop_order_iccprofile_info_t *const profile_info)
{
  (void)op_name; (void)instance_name; (void)devid; (void)dev_img_in; (void)dev_img_out; (void)width; (void)height; (void)cst_from;
  (void)cst_to; (void)converted_cst; (void)profile_info;
  g_unexpected++;
  return 0;
}
/* src/develop/pixelpipe_cpu.c: there is no CPU fallback behind this library */
static int g_cpu_fallbacks = 0;
int boundary_cpu_fallbacks(void) { return g_cpu_fallbacks; }
int pixelpipe_process_on_CPU(dt_dev_pixelpipe_t *pipe, const dt_dev_pixelpipe_iop_t *piece, const dt_dev_pixelpipe_iop_t *previous_piece,
                             dt_develop_tiling_t *tiling, dt_pixelpipe_flow_t *pixelpipe_flow, gboolean *const cache_output,
                             dt_pixel_cache_entry_t *input_entry, dt_pixel_cache_entry_t *output_entry)
{
  (void)pipe; (void)piece; (void)previous_piece; (void)tiling; (void)pixelpipe_flow; (void)cache_output; (void)input_entry;
  (void)output_entry;
  g_cpu_fallbacks++;
  return 1;
}

/* ---- the modules: exposure and diffuse or sharpen, written as INTEGRATION.md section 2 prescribes (integration_stubs.h) -- */
static int g_process_cl_calls = 0;
static size_t g_tile_budget = 0; /* bytes one buffer of a tile may take (0: the device's real budget) */
static int g_module_flags = IOP_FLAGS_ALLOW_TILING;
#define TILE_LOG_MAX 4096
static int g_tile_log[TILE_LOG_MAX][4]; /* roi_in of every process_cl() call: x, y, width, height */
/* the test rebuilds the reference's tile plan from what the module was actually asked to process */
int boundary_tile_log(int i, int *xywh)
{
  if(i < 0 || i >= g_process_cl_calls || i >= TILE_LOG_MAX) return 0;
  memcpy(xywh, g_tile_log[i], sizeof(g_tile_log[i]));
  return 1;
}
static void log_call(const struct dt_dev_pixelpipe_iop_t *piece)
{
  if(g_process_cl_calls < TILE_LOG_MAX)
  {
    int *e = g_tile_log[g_process_cl_calls];
    e[0] = piece->roi_in.x; e[1] = piece->roi_in.y; e[2] = piece->roi_in.width; e[3] = piece->roi_in.height;
  }
  g_process_cl_calls++;
}
static void squeeze(const struct dt_dev_pixelpipe_t *pipe, struct dt_develop_tiling_t *tiling)
{
  if(g_tile_budget) /* the test squeezes the per-buffer budget: factor_cl = free memory / budget */
    tiling->factor_cl = (float)dt_opencl_get_device_available(pipe->devid) / (float)g_tile_budget;
}
static int module_flags(void) { return g_module_flags; }

static const char *exposure_name(void) { return "exposure"; }
static int exposure_process_cl(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe,
                               const struct dt_dev_pixelpipe_iop_t *piece, void *dev_in, void *dev_out)
{
  log_call(piece);
  return stub_exposure_process_cl(self, pipe, piece, dev_in, dev_out);
}
static void exposure_tiling_callback(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe,
                                     const struct dt_dev_pixelpipe_iop_t *piece, struct dt_develop_tiling_t *tiling)
{
  default_tiling_callback(self, pipe, piece, tiling); /* the reference's own default, tiling.c:1423-1463 */
  squeeze(pipe, tiling);
}
static const char *diffuse_name(void) { return "diffuse"; }
static int diffuse_process_cl(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe,
                              const struct dt_dev_pixelpipe_iop_t *piece, void *dev_in, void *dev_out)
{
  log_call(piece);
  return stub_diffuse_process_cl(self, pipe, piece, dev_in, dev_out);
}
static void diffuse_tiling_callback(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe,
                                    const struct dt_dev_pixelpipe_iop_t *piece, struct dt_develop_tiling_t *tiling)
{
  stub_diffuse_tiling_callback(self, pipe, piece, tiling); /* overlap = what the module's stencils reach */
  squeeze(pipe, tiling);
}
/* round 4: a mosaic-stage module, a stencil module with overlap 128, a colour module -- each the stub of INTEGRATION.md section 2 */
static const char *demosaic_name(void) { return "demosaic"; }
static int demosaic_process_cl(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe,
                               const struct dt_dev_pixelpipe_iop_t *piece, void *dev_in, void *dev_out)
{
  log_call(piece);
  return stub_demosaic_process_cl(self, pipe, piece, dev_in, dev_out);
}
static void demosaic_tiling_callback(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe,
                                     const struct dt_dev_pixelpipe_iop_t *piece, struct dt_develop_tiling_t *tiling)
{
  stub_demosaic_tiling_callback(self, pipe, piece, tiling); /* overlap 10, xalign = yalign = 2 (demosaic.c:1532-1600) */
  squeeze(pipe, tiling);
}
static const char *denoiseprofile_name(void) { return "denoiseprofile"; }
static int denoiseprofile_process_cl(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe,
                                     const struct dt_dev_pixelpipe_iop_t *piece, void *dev_in, void *dev_out)
{
  log_call(piece);
  return stub_denoiseprofile_process_cl(self, pipe, piece, dev_in, dev_out);
}
static void denoiseprofile_tiling_callback(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe,
                                           const struct dt_dev_pixelpipe_iop_t *piece, struct dt_develop_tiling_t *tiling)
{
  stub_denoiseprofile_tiling_callback(self, pipe, piece, tiling); /* overlap 2^(max scale): 128 (denoiseprofile.c:796-848) */
  squeeze(pipe, tiling);
}
static const char *colorin_name(void) { return "colorin"; }
static int colorin_process_cl(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe,
                              const struct dt_dev_pixelpipe_iop_t *piece, void *dev_in, void *dev_out)
{
  log_call(piece);
  return stub_colorin_process_cl(self, pipe, piece, dev_in, dev_out);
}

static int module_process_tiling_cl(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe,
                                    const struct dt_dev_pixelpipe_iop_t *piece, const void *const i, void *const o, const int bpp)
{
  return default_process_tiling_cl(self, pipe, piece, i, o, bpp);
}

enum { M_EXPOSURE = 0, M_DIFFUSE = 1, M_DEMOSAIC = 2, M_DENOISEPROFILE = 3, M_COLORIN = 4 };
static void make_module(dt_iop_module_t *m, dt_develop_t *dev, const int kind)
{
  static const char *const ops[] = { "exposure", "diffuse", "demosaic", "denoiseprofile", "colorin" };
  typedef const char *(*name_fn)(void);
  static const name_fn names[] = { exposure_name, diffuse_name, demosaic_name, denoiseprofile_name, colorin_name };
  memset(m, 0, sizeof(*m));
  memset(dev, 0, sizeof(*dev));
  strcpy(m->op, ops[kind]);
  m->dev = dev;
  m->name = names[kind];
  m->flags = module_flags;
  m->tiling_callback = kind == M_DIFFUSE ? diffuse_tiling_callback
                       : (kind == M_DEMOSAIC ? demosaic_tiling_callback
                                             : (kind == M_DENOISEPROFILE ? denoiseprofile_tiling_callback : exposure_tiling_callback));
  m->process_cl = kind == M_DIFFUSE ? diffuse_process_cl
                  : (kind == M_DEMOSAIC ? demosaic_process_cl
                                        : (kind == M_DENOISEPROFILE ? denoiseprofile_process_cl
                                                                    : (kind == M_COLORIN ? colorin_process_cl : exposure_process_cl)));
  m->process_tiling_cl = module_process_tiling_cl;
  g_module_flags = IOP_FLAGS_ALLOW_TILING;
  static volatile int keep = 0; /* every stub of integration_stubs.h stays referenced: compiled, type-checked, linked */
  (void)integration_stub_table(keep);
}
static void make_piece(dt_dev_pixelpipe_iop_t *piece, dt_iop_module_t *m, void *d, int w, int h)
{
  memset(piece, 0, sizeof(*piece));
  piece->module = m;
  piece->data = d;
  piece->enabled = TRUE;
  piece->iwidth = w;
  piece->iheight = h;
  const dt_iop_roi_t roi = { 0, 0, w, h, 1.0 };
  piece->buf_in = piece->buf_out = piece->roi_in = piece->roi_out = roi;
  piece->process_cl_ready = 1;
  piece->process_tiling_ready = 1;
  piece->dsc_in.channels = piece->dsc_out.channels = 4;
  piece->dsc_in.datatype = piece->dsc_out.datatype = TYPE_FLOAT;
  piece->dsc_in.bpp = piece->dsc_out.bpp = 16;
  piece->dsc_in.cst = piece->dsc_out.cst = IOP_CS_RGB;
  for(int c = 0; c < 4; c++) piece->dsc_in.processed_maximum[c] = piece->dsc_out.processed_maximum[c] = 1.0f;
}
static dt_iop_exposure_data_t exposure_data(const float black, const float scale)
{
  dt_iop_exposure_data_t d; /* what commit_params() leaves in piece->data (exposure.c:381-420) */
  memset(&d, 0, sizeof(d));
  d.black = black;
  d.scale = scale;
  return d;
}

/* the reference's pixelpipe_process_on_GPU() on the module: device path, output synced back into the host line.
 * Returns its return value; *flow = the dt_pixelpipe_flow_t bits it set, *calls = process_cl() invocations.
 * blend != NULL: the module supports blending and carries these blend parameters (the fields of dt_hip_blend_data_t,
 * moved into the reference's dt_develop_blend_params_t by name) -- dt_develop_blend_process_cl() runs behind process_cl() */
static int run_exposure_gpu(const float *in, float *out, int w, int h, float black, float scale, const dt_hip_blend_data_t *blend,
                            int *flow, int *calls)
{
  if(dt_hip_init() != DT_HIP_SUCCESS) return -1;
  dt_iop_module_t m;
  dt_develop_t dev;
  dt_dev_pixelpipe_iop_t piece;
  dt_iop_exposure_data_t d = exposure_data(black, scale);
  make_module(&m, &dev, 0);
  make_piece(&piece, &m, &d, w, h);
  dt_develop_blend_params_t bp;
  memset(&bp, 0, sizeof(bp));
  if(blend)
  {
    g_module_flags |= IOP_FLAGS_SUPPORTS_BLENDING;
    bp.mask_mode = blend->mask_mode;
    bp.blend_cst = blend->blend_cst;
    bp.blend_mode = blend->blend_mode;
    bp.blend_parameter = blend->blend_parameter;
    bp.opacity = blend->opacity;
    bp.mask_combine = blend->mask_combine;
    bp.blendif = blend->blendif;
    bp.feathering_radius = blend->feathering_radius;
    bp.feathering_guide = blend->feathering_guide;
    bp.blur_radius = blend->blur_radius;
    bp.contrast = blend->contrast;
    bp.brightness = blend->brightness;
    bp.details = blend->details;
    memcpy(bp.blendif_parameters, blend->blendif_parameters, sizeof(bp.blendif_parameters));
    memcpy(bp.blendif_boost_factors, blend->blendif_boost_factors, sizeof(bp.blendif_boost_factors));
    memcpy(g_stub_work_matrix_in, blend->matrix_in, sizeof(g_stub_work_matrix_in));
    piece.blendop_data = &bp;
  }
  dev.image_storage.dsc = piece.dsc_in;
  dt_dev_pixelpipe_t pipe;
  memset(&pipe, 0, sizeof(pipe));
  pipe.dev = &dev;
  pipe.type = DT_DEV_PIXELPIPE_EXPORT;
  pipe.devid = dt_opencl_reserve_device_for_pipe(pipe.type);
  pipe.opencl_enabled = TRUE;
  if(pipe.devid < 0) return -2;
  const size_t bytes = (size_t)w * h * 16;
  dt_pixel_cache_entry_t ein = { 1, (void *)in, bytes, NULL, 0, 0, 0 }, eout = { 2, out, bytes, NULL, 0, 0, 0 };
  dt_develop_tiling_t tiling;
  memset(&tiling, 0, sizeof(tiling));
  m.tiling_callback(&m, &pipe, &piece, &tiling);
  dt_pixelpipe_flow_t fl = PIXELPIPE_FLOW_NONE;
  gboolean cache_output = TRUE; /* the test reads the result from the host line */
  g_process_cl_calls = 0;
  g_blend_cl_calls = 0;
  g_tile_budget = 0;
  const int rc = pixelpipe_process_on_GPU(&pipe, &piece, NULL, &tiling, &fl, &cache_output, &ein, &eout);
  dt_opencl_finish(pipe.devid);
  if(eout.cl_mem) dt_opencl_release_mem_object(eout.cl_mem); /* the payload the line kept */
  if(ein.cl_mem) dt_opencl_release_mem_object(ein.cl_mem);
  dt_opencl_release_device(pipe.devid);
  if(flow) *flow = (int)fl;
  if(calls) *calls = g_process_cl_calls;
  return rc;
}
int boundary_run_exposure_gpu(const float *in, float *out, int w, int h, float black, float scale, int *flow, int *calls)
{
  return run_exposure_gpu(in, out, w, h, black, scale, NULL, flow, calls);
}
int boundary_run_exposure_blended_gpu(const float *in, float *out, int w, int h, float black, float scale,
                                      const dt_hip_blend_data_t *blend, int *flow, int *blend_calls)
{
  int calls = 0;
  const int rc = run_exposure_gpu(in, out, w, h, black, scale, blend, flow, &calls);
  if(blend_calls) *blend_calls = g_blend_cl_calls;
  return rc;
}

/* the reference's default_process_tiling_cl() on a module with `budget` bytes per tile buffer: TRUE on success.
 * diffuse == NULL: exposure (black, scale); else diffuse or sharpen with these parameters (a stencil module: the
 * reference's plan cuts overlapping tiles) */
static int run_tiled(const float *in, float *out, int w, int h, float black, float scale, const dt_hip_diffuse_data_t *diffuse,
                     size_t budget, int *calls)
{
  if(dt_hip_init() != DT_HIP_SUCCESS) return -1;
  dt_iop_module_t m;
  dt_develop_t dev;
  dt_dev_pixelpipe_iop_t piece;
  dt_iop_exposure_data_t d = exposure_data(black, scale);
  dt_iop_diffuse_data_t dd;
  memset(&dd, 0, sizeof(dd));
  if(diffuse)
  {
    dd.iterations = diffuse->iterations; dd.sharpness = diffuse->sharpness; dd.radius = diffuse->radius;
    dd.regularization = diffuse->regularization; dd.variance_threshold = diffuse->variance_threshold;
    dd.anisotropy_first = diffuse->anisotropy_first; dd.anisotropy_second = diffuse->anisotropy_second;
    dd.anisotropy_third = diffuse->anisotropy_third; dd.anisotropy_fourth = diffuse->anisotropy_fourth;
    dd.threshold = diffuse->threshold; dd.first = diffuse->first; dd.second = diffuse->second; dd.third = diffuse->third;
    dd.fourth = diffuse->fourth; dd.radius_center = diffuse->radius_center;
  }
  make_module(&m, &dev, diffuse != NULL ? M_DIFFUSE : M_EXPOSURE);
  make_piece(&piece, &m, diffuse ? (void *)&dd : (void *)&d, w, h);
  dt_dev_pixelpipe_t pipe;
  memset(&pipe, 0, sizeof(pipe));
  pipe.dev = &dev;
  pipe.type = DT_DEV_PIXELPIPE_EXPORT;
  pipe.devid = dt_opencl_reserve_device_for_pipe(pipe.type);
  pipe.opencl_enabled = TRUE;
  if(pipe.devid < 0) return -2;
  g_process_cl_calls = 0;
  g_tile_budget = budget;
  const int ok = m.process_tiling_cl(&m, &pipe, &piece, in, out, 16);
  dt_opencl_finish(pipe.devid);
  g_tile_budget = 0;
  dt_opencl_release_device(pipe.devid);
  if(calls) *calls = g_process_cl_calls;
  return ok;
}
int boundary_run_exposure_tiled(const float *in, float *out, int w, int h, float black, float scale, size_t budget, int *calls)
{
  return run_tiled(in, out, w, h, black, scale, NULL, budget, calls);
}
int boundary_run_diffuse_tiled(const float *in, float *out, int w, int h, const dt_hip_diffuse_data_t *diffuse, size_t budget,
                               int *calls)
{
  return run_tiled(in, out, w, h, 0.0f, 1.0f, diffuse, budget, calls);
}


/* ---- round 4: the reference's default_process_tiling_cl() on a MOSAIC-stage module (one channel in, four out, tiles aligned to the
 *      2 x 2 cell, overlap 10) and on denoise (profiled) wavelets (overlap 128); its pixelpipe_process_on_GPU() on colorin ---- */
static int run_module_tiled(const int kind, void *data, const float *in, float *out, int w, int h, int in_channels, uint32_t filters,
                            const float *wb, size_t budget, int *calls)
{
  if(dt_hip_init() != DT_HIP_SUCCESS) return -1;
  dt_iop_module_t m;
  dt_develop_t dev;
  dt_dev_pixelpipe_iop_t piece;
  make_module(&m, &dev, kind);
  make_piece(&piece, &m, data, w, h);
  if(in_channels == 1)
  {
    piece.dsc_in.channels = 1;
    piece.dsc_in.bpp = 4;
    piece.dsc_in.cst = IOP_CS_RAW;
    piece.dsc_in.filters = filters;
  }
  for(int c = 0; c < 4; c++)
  {
    piece.dsc_in.temperature.coeffs[c] = piece.dsc_out.temperature.coeffs[c] = wb ? wb[c] : 1.0f;
    if(wb) piece.dsc_in.processed_maximum[c] = piece.dsc_out.processed_maximum[c] = wb[c];
  }
  piece.dsc_in.temperature.enabled = wb != NULL;
  dt_dev_pixelpipe_t pipe;
  memset(&pipe, 0, sizeof(pipe));
  pipe.dev = &dev;
  pipe.type = DT_DEV_PIXELPIPE_EXPORT;
  pipe.devid = dt_opencl_reserve_device_for_pipe(pipe.type);
  pipe.opencl_enabled = TRUE;
  if(pipe.devid < 0) return -2;
  g_process_cl_calls = 0;
  g_tile_budget = budget;
  const int ok = m.process_tiling_cl(&m, &pipe, &piece, in, out, (int)piece.dsc_in.bpp);
  dt_opencl_finish(pipe.devid);
  g_tile_budget = 0;
  dt_opencl_release_device(pipe.devid);
  if(calls) *calls = g_process_cl_calls;
  return ok;
}
int boundary_run_demosaic_tiled(const float *mosaic, float *out, int w, int h, unsigned filters, int method, size_t budget, int *calls)
{
  dt_iop_demosaic_data_t d; /* what commit_params() leaves in piece->data (demosaic.c:1630-1720) */
  memset(&d, 0, sizeof(d));
  d.demosaicing_method = (uint32_t)method;
  return run_module_tiled(M_DEMOSAIC, &d, mosaic, out, w, h, 1, filters, NULL, budget, calls);
}
int boundary_run_denoiseprofile_tiled(const float *in, float *out, int w, int h, const dt_hip_denoiseprofile_data_t *hd, size_t budget,
                                      int *calls)
{
  dt_iop_denoiseprofile_data_t d; /* the reference's struct, filled field by field from the test's parameters */
  memset(&d, 0, sizeof(d));
  d.radius = hd->radius; d.nbhood = hd->nbhood; d.strength = hd->strength; d.shadows = hd->shadows; d.bias = hd->bias;
  d.scattering = hd->scattering; d.central_pixel_weight = hd->central_pixel_weight; d.overshooting = hd->overshooting;
  memcpy(d.a, hd->a, sizeof(d.a));
  memcpy(d.b, hd->b, sizeof(d.b));
  d.mode = (dt_iop_denoiseprofile_mode_t)hd->mode;
  memcpy(d.force, hd->force, sizeof(d.force));
  d.wb_adaptive_anscombe = hd->wb_adaptive_anscombe;
  d.fix_anscombe_and_nlmeans_norm = hd->fix_anscombe_and_nlmeans_norm;
  d.use_new_vst = hd->use_new_vst;
  d.wavelet_color_mode = (dt_iop_denoiseprofile_wavelet_mode_t)hd->wavelet_color_mode;
  return run_module_tiled(M_DENOISEPROFILE, &d, in, out, w, h, 4, 0, hd->wb_coeffs, budget, calls);
}
/* colorin through pixelpipe_process_on_GPU(): `conv` carries HOST curves (lut_source[c] = 65536 floats or NULL) -- the stub
 * uploads them through the accessor API as a port would */
int boundary_run_colorin_gpu(const float *in, float *out, int w, int h, const dt_hip_conversion_t *conv, int *flow, int *calls)
{
  if(dt_hip_init() != DT_HIP_SUCCESS) return -1;
  dt_iop_module_t m;
  dt_develop_t dev;
  dt_dev_pixelpipe_iop_t piece;
  struct dt_colorspaces_conversion_t c;
  memset(&c, 0, sizeof(c));
  for(int r = 0; r < 3; r++)
    for(int k = 0; k < 4; k++)
    {
      c.matrix[r][k] = conv->matrix[r][k];
      c.clip_matrix[r][k] = conv->clip_matrix[r][k];
    }
  c.has_clipping = conv->has_clipping;
  for(int ch = 0; ch < 3; ch++)
  {
    c.source_curve[ch] = (const float *)conv->lut_source[ch];
    for(int k = 0; k < 3; k++) c.source_coeffs[3 * ch + k] = conv->coeffs_source[ch][k];
  }
  dt_iop_colorin_data_t d;
  memset(&d, 0, sizeof(d));
  d.conversion = &c;
  d.blue_mapping = conv->blue_mapping;
  make_module(&m, &dev, M_COLORIN);
  make_piece(&piece, &m, &d, w, h);
  dev.image_storage.dsc = piece.dsc_in;
  dt_dev_pixelpipe_t pipe;
  memset(&pipe, 0, sizeof(pipe));
  pipe.dev = &dev;
  pipe.type = DT_DEV_PIXELPIPE_EXPORT;
  pipe.devid = dt_opencl_reserve_device_for_pipe(pipe.type);
  pipe.opencl_enabled = TRUE;
  if(pipe.devid < 0) return -2;
  const size_t bytes = (size_t)w * h * 16;
  dt_pixel_cache_entry_t ein = { 1, (void *)in, bytes, NULL, 0, 0, 0 }, eout = { 2, out, bytes, NULL, 0, 0, 0 };
  dt_develop_tiling_t tiling;
  memset(&tiling, 0, sizeof(tiling));
  m.tiling_callback(&m, &pipe, &piece, &tiling);
  dt_pixelpipe_flow_t fl = PIXELPIPE_FLOW_NONE;
  gboolean cache_output = TRUE;
  g_process_cl_calls = 0;
  g_tile_budget = 0;
  const int rc = pixelpipe_process_on_GPU(&pipe, &piece, NULL, &tiling, &fl, &cache_output, &ein, &eout);
  dt_opencl_finish(pipe.devid);
  if(eout.cl_mem) dt_opencl_release_mem_object(eout.cl_mem);
  if(ein.cl_mem) dt_opencl_release_mem_object(ein.cl_mem);
  dt_opencl_release_device(pipe.devid);
  if(flow) *flow = (int)fl;
  if(calls) *calls = g_process_cl_calls;
  return rc;
}
