/* oracle/ref_wrap/ref_blend_raw.c -- TEST INFRASTRUCTURE ONLY.
 * The "raw" blend colourspace (one channel, before demosaic): src/develop/blends/blendif_raw.c lifted
 * verbatim at build time.  Driven by ref_develop_blend() in ref_blend.c. */
#define REF_REAL_IMAGEBUF 1
#include "ref_piece.h"
#include "common/imagebuf.h"
#include "math/openmp_maths.h"

typedef char dt_dev_operation_t[20];
typedef int dt_dev_pixelpipe_display_mask_t;
#define DT_DEV_PIXELPIPE_DISPLAY_NONE 0
#define DT_DEV_PIXELPIPE_DISPLAY_ANY 0x3fc /* never requested here */
typedef struct dt_iop_module_so_t dt_iop_module_so_t;
typedef struct dt_iop_order_iccprofile_info_t dt_iop_order_iccprofile_info_t;
#include "gen/blend_h.inc"

/* the row-function type of blendif_raw.c:32-33 (a function typedef, which extract.py does not lift) */
typedef void(_blend_row_func)(const float *const restrict a, const float *const restrict b,
                              float *const restrict out, const float *const restrict mask, const size_t stride);
#include "gen/blendif_raw.inc"
