/* boundary_host.h -- TEST INFRASTRUCTURE (tests/test_boundary_compile.py).
 *
 * The reference's src/develop/pixelpipe_gpu.c and src/develop/tiling.c are compiled FROM WHERE THEY LIE, unmodified,
 * against include/ansel_opencl_peer.h (installed as common/opencl.h) and libansel_hip.so.  Those two files also
 * include the application's big headers (develop.h, imageop.h, pixelpipe_hb.h: GTK, the database, the GUI); this header
 * stands in for them with the slice the two files actually touch -- the members they read, with the reference's names
 * and types (src/develop/pixelpipe_hb.h:101-166, src/develop/imageop.h, src/caches/pixelpipe_cache.h) -- and the
 * prototypes of the host-side services they call, which boundary_stubs.c implements in the simplest way that works.
 * Nothing of the reference is copied: every real header that is free of GUI dependencies (pixel/format.h,
 * system/macros.h, develop/tiling.h, develop/pixelpipe_process.h, ...) is included as it is. */
#pragma once
#include <glib.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include "system/macros.h"
#include "pixel/format.h" /* dt_iop_roi_t, dt_iop_buffer_dsc_t, dt_iop_colorspace_type_t: the real contract types */

struct dt_iop_module_t;
struct dt_dev_pixelpipe_t;
struct dt_dev_pixelpipe_iop_t;
struct dt_develop_tiling_t;
struct dt_develop_t;

/* src/caches/pixelpipe_cache.h: one cache line */
#define DT_PIXELPIPE_CACHE_HASH_INVALID ((uint64_t)-1)
typedef struct dt_pixel_cache_entry_t
{
  uint64_t hash;
  void *data;
  size_t size;
  void *cl_mem; /* the vRAM payload bound to this line on the pipe's device, if any */
  int cl_width, cl_height, cl_bpp;
} dt_pixel_cache_entry_t;

/* src/develop/imageop.h: flags, colourspaces and the slice of dt_iop_module_t the two files read */
typedef enum dt_iop_flags_t
{
  IOP_FLAGS_NONE = 0,
  IOP_FLAGS_SUPPORTS_BLENDING = 1 << 1,
  IOP_FLAGS_ALLOW_TILING = 1 << 4,
  IOP_FLAGS_PREVIEW_NON_OPENCL = 1 << 8,
  IOP_FLAGS_TILING_FULL_ROI = 1 << 10,
  IOP_FLAGS_TAKE_NO_INPUT = 1 << 14
} dt_iop_flags_t;

typedef enum dt_dev_pixelpipe_display_mask_t
{
  DT_DEV_PIXELPIPE_DISPLAY_NONE = 0,
  DT_DEV_PIXELPIPE_DISPLAY_ANY = 0xff << 2
} dt_dev_pixelpipe_display_mask_t;

typedef enum dt_dev_pixelpipe_type_t
{
  DT_DEV_PIXELPIPE_NONE = 0,
  DT_DEV_PIXELPIPE_EXPORT = 1 << 0,
  DT_DEV_PIXELPIPE_FULL = 1 << 1,
  DT_DEV_PIXELPIPE_PREVIEW = 1 << 2,
  DT_DEV_PIXELPIPE_THUMBNAIL = 1 << 3
} dt_dev_pixelpipe_type_t;

typedef struct dt_iop_module_t
{
  char op[20];
  char multi_name[128];
  int iop_order;
  struct dt_develop_t *dev;
  dt_dev_pixelpipe_display_mask_t request_mask_display;
  const char *(*name)(void);
  int (*flags)(void);
  void (*tiling_callback)(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe,
                          const struct dt_dev_pixelpipe_iop_t *piece, struct dt_develop_tiling_t *tiling);
  void (*modify_roi_in)(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe, struct dt_dev_pixelpipe_iop_t *piece,
                        const dt_iop_roi_t *roi_out, dt_iop_roi_t *roi_in);
  int (*process)(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe, const struct dt_dev_pixelpipe_iop_t *piece,
                 const void *const i, void *const o);
  int (*process_cl)(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe,
                    const struct dt_dev_pixelpipe_iop_t *piece, void *dev_in, void *dev_out);
  int (*process_tiling)(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe,
                        const struct dt_dev_pixelpipe_iop_t *piece, const void *const i, void *const o, const int bpp);
  int (*process_tiling_cl)(struct dt_iop_module_t *self, const struct dt_dev_pixelpipe_t *pipe,
                           const struct dt_dev_pixelpipe_iop_t *piece, const void *const i, void *const o, const int bpp);
} dt_iop_module_t;

/* src/develop/develop.h: the image's storage contract is all the two files read of it */
typedef struct dt_develop_t
{
  struct
  {
    dt_iop_buffer_dsc_t dsc;
  } image_storage;
  gboolean gui_attached;
  struct dt_iop_module_t *gui_module;
  struct dt_dev_pixelpipe_t *pipe;
} dt_develop_t;

/* src/develop/iop_profile.h: opaque here */
typedef struct dt_iop_order_iccprofile_info_t dt_iop_order_iccprofile_info_t;

/* src/develop/pixelpipe_hb.h:101-166 and the pipe itself */
typedef struct dt_dev_pixelpipe_iop_t
{
  struct dt_iop_module_t *module;
  void *data;
  void *blendop_data;
  gboolean enabled;
  int iwidth, iheight;
  dt_iop_roi_t buf_in, buf_out;
  dt_iop_roi_t roi_in, roi_out;
  int process_cl_ready;
  int process_tiling_ready;
  dt_iop_buffer_dsc_t dsc_in, dsc_out, dsc_mask;
  gboolean bypass_cache;
  dt_pixel_cache_entry_t cache_entry;
  gboolean cache_output_on_ram;
} dt_dev_pixelpipe_iop_t;

typedef struct dt_dev_pixelpipe_t
{
  struct dt_develop_t *dev;
  dt_dev_pixelpipe_type_t type;
  int devid;
  int tiling;
  gboolean opencl_enabled;
  gboolean realtime;
  gboolean reentry, no_cache, bypass_cache;
  GList *iop_order_list;
} dt_dev_pixelpipe_t;

/* src/common/logging.h */
typedef enum dt_debug_thread_t
{
  DT_DEBUG_OPENCL = 1 << 7,
  DT_DEBUG_MEMORY = 1 << 9,
  DT_DEBUG_VERBOSE = 1 << 22,
  DT_DEBUG_TILING = 1 << 23
} dt_debug_thread_t;
void dt_print(dt_debug_thread_t thread, const char *msg, ...);
void dt_vprint(dt_debug_thread_t thread, const char *msg, ...);
unsigned int dt_get_debug_flags(void);
void dt_pipeline_message(const char *format, ...);
#ifndef _
#define _(s) (s) /* gettext */
#endif

/* src/caches/pixelpipe_cache.h: the cache services pixelpipe_gpu.c calls (boundary_stubs.c: one line = one host buffer
 * + at most one device payload) */
void *dt_pixel_cache_entry_get_data(struct dt_pixel_cache_entry_t *entry);
void *dt_pixel_cache_alloc(struct dt_pixel_cache_entry_t *entry);
void dt_dev_pixelpipe_cache_wrlock_entry(gboolean lock, struct dt_pixel_cache_entry_t *entry);
void dt_dev_pixelpipe_cache_rdlock_entry(gboolean lock, struct dt_pixel_cache_entry_t *entry);
int dt_dev_pixelpipe_cache_sync_cl_buffer(int devid, void *host_ptr, void *cl_mem_buffer, const dt_iop_roi_t *roi, int cl_mode,
                                          size_t bpp, struct dt_iop_module_t *module, const char *message);
float *dt_dev_pixelpipe_cache_restore_cl_buffer(struct dt_dev_pixelpipe_t *pipe, float *input, void *cl_mem_input,
                                                const dt_iop_roi_t *roi_in, struct dt_iop_module_t *module, size_t in_bpp,
                                                struct dt_pixel_cache_entry_t *input_entry, const char *message);
void dt_dev_pixelpipe_cache_release_cl_buffer(void **cl_mem_buffer, struct dt_pixel_cache_entry_t *entry, void *host_ptr,
                                              gboolean cache_device);
int dt_dev_pixelpipe_cache_prepare_cl_input(struct dt_dev_pixelpipe_t *pipe, struct dt_iop_module_t *module, float *input,
                                            void **cl_mem_input, const dt_iop_roi_t *roi_in, size_t in_bpp,
                                            struct dt_pixel_cache_entry_t *input_entry,
                                            struct dt_pixel_cache_entry_t **locked_input_entry, void *keep);
void *dt_dev_pixelpipe_cache_get_cl_buffer(int devid, void *host_ptr, const dt_iop_roi_t *roi, size_t bpp,
                                           struct dt_iop_module_t *module, const char *message,
                                           struct dt_pixel_cache_entry_t *entry, gboolean *out_reused, void *keep);
gboolean dt_dev_pixelpipe_cache_flush_host_pinned_image(void *host_ptr, struct dt_pixel_cache_entry_t *entry_hint, int devid);
void dt_dev_pixelpipe_cache_flush_clmem(const int devid);
void *dt_dev_pixelpipe_cache_borrow_cl_payload(struct dt_pixel_cache_entry_t *entry, int devid, int width, int height, int bpp);
void dt_dev_pixelpipe_cache_return_cl_payload(struct dt_pixel_cache_entry_t *entry, void *mem);
void *dt_dev_pixelpipe_cache_alloc_cl_device_buffer(int devid, const dt_iop_roi_t *roi, size_t bpp,
                                                    const struct dt_iop_module_t *module, const char *message, void *keep);
void dt_dev_pixelpipe_cache_get_usage(size_t *current, size_t *max);
int dt_dev_pixel_pipe_cache_remove_lru(void);
size_t dt_pixelpipe_cache_get_largest_free_run(void);

/* src/develop/blend.h, src/colorprofiles/iop_profile.h, src/develop/iop_order.h: not exercised by the harness' module
 * (no blending, one colourspace), declared so that the host files compile; the stubs fail loudly if reached */
int dt_develop_blend_process(struct dt_iop_module_t *self, struct dt_dev_pixelpipe_t *pipe,
                             const struct dt_dev_pixelpipe_iop_t *piece, const void *const i, void *const o);
int dt_develop_blend_process_cl(struct dt_iop_module_t *self, struct dt_dev_pixelpipe_t *pipe,
                                const struct dt_dev_pixelpipe_iop_t *piece, void *dev_in, void *dev_out);
dt_iop_colorspace_type_t dt_develop_blend_colorspace(const struct dt_dev_pixelpipe_iop_t *const piece, dt_iop_colorspace_type_t cst);
dt_iop_order_iccprofile_info_t *dt_ioppr_get_pipe_work_profile_info(const struct dt_dev_pixelpipe_t *pipe);
int dt_ioppr_get_iop_order(GList *iop_order_list, const char *op_name, const int multi_priority);
void dt_colorspaces_apply_profile(const char *const op_name, const char *const instance_name, const float *const image_in,
                                  float *const image_out, const int width, const int height, const int cst_from,
                                  const int cst_to, int *converted_cst, const dt_iop_order_iccprofile_info_t *const profile_info);
int dt_colorspaces_apply_profile_cl(const char *const op_name, const char *const instance_name, const int devid, void *dev_img_in,
                                    void *dev_img_out, const int width, const int height, const int cst_from, const int cst_to,
                                    int *converted_cst, const dt_iop_order_iccprofile_info_t *const profile_info);
