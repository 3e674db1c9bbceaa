/* oracle/ref_shim/ref_host.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Minimal stand-ins for the host-side types the reference's pixel code touches
 * (dt_iop_module_t, dt_dev_pixelpipe_t, dt_dev_pixelpipe_iop_t ...), so that the
 * reference's own process() bodies can be compiled *from where they lie* under
 * /root/reference/src without GTK/sqlite/lcms.  Only the fields the hot path reads
 * are present (reference: src/develop/pixelpipe_hb.h:101-166, src/iop/iop_api.h).
 * Everything arithmetic (system/simd.h, math/math.h, pixel/format.h, ...) is the
 * reference's real header, found through -I/root/reference/src.
 */
#ifndef ANSEL_REF_HOST_H
#define ANSEL_REF_HOST_H

#include <glib.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <float.h>
#include <assert.h>

#include "system/macros.h"
#include "system/mem_alloc.h"
#include "system/openmp.h"
#include "system/target_clones.h"
#include "system/simd.h"
#include "system/fp_mode.h"
#include "math/math.h"
#include "math/matrices.h"
#include "pixel/format.h"
#include "caches/pixelpipe_cache_alloc.h"

#ifdef __cplusplus
extern "C" {
#endif

#ifndef _
#define _(x) (x)
#endif
#ifndef N_
#define N_(x) (x)
#endif

/* module registration macro of src/common/module_api.h: the extractor may carry an invocation
 * along with the declaration that follows it */
#define DT_MODULE_INTROSPECTION(version, type)

#define dt_control_log(...) ((void)0)
#define dt_print(...) ((void)0)
#define dt_iop_fmt_log(...) ((void)0)

typedef struct dt_image_t
{
  int flags;
  float exif_iso;
  uint16_t raw_white_point;
} dt_image_t;

typedef struct dt_develop_t
{
  dt_image_t image_storage;
  int gui_attached;
} dt_develop_t;

typedef enum dt_dev_pixelpipe_type_t
{
  DT_DEV_PIXELPIPE_NONE = 0,
  DT_DEV_PIXELPIPE_EXPORT = 1 << 0,
  DT_DEV_PIXELPIPE_FULL = 1 << 1,
  DT_DEV_PIXELPIPE_PREVIEW = 1 << 2,
  DT_DEV_PIXELPIPE_THUMBNAIL = 1 << 3,
} dt_dev_pixelpipe_type_t;

#define DT_DEV_PIXELPIPE_DISPLAY_MASK 1

typedef struct dt_dev_pixelpipe_t
{
  int type;
  int mask_display;
  dt_develop_t *dev;
  float iscale;
} dt_dev_pixelpipe_t;

struct dt_iop_module_t;
typedef struct dt_dev_pixelpipe_iop_t
{
  struct dt_iop_module_t *module;
  void *data;
  dt_iop_roi_t roi_in, roi_out;
  dt_iop_roi_t buf_in, buf_out;
  dt_iop_buffer_dsc_t dsc_in, dsc_out;
  int process_cl_ready, process_tiling_ready;
  void *blendop_data;
} dt_dev_pixelpipe_iop_t;

typedef struct dt_iop_module_t
{
  dt_develop_t *dev;
  void *global_data;
} dt_iop_module_t;

/* src/develop/imageop.c:134-137 */
#define dt_dev_get_module_scale(pipe, roi_in) ((float)((pipe)->iscale / (roi_in)->scale))

/* src/common/imagebuf.h: plain float copy of a w x h x ch buffer (units that compile the real
 * common/imagebuf.c define REF_REAL_IMAGEBUF and get the reference's own) */
#ifndef REF_REAL_IMAGEBUF
static inline void dt_iop_image_copy_by_size(float *const out, const float *const in, const size_t w,
                                             const size_t h, const size_t ch)
{
  memcpy(out, in, sizeof(float) * w * h * ch);
}
#endif

static inline void dt_iop_alpha_copy(const void *i, void *o, int w, int h) { (void)i; (void)o; (void)w; (void)h; }

#ifdef __cplusplus
}
#endif
#endif
