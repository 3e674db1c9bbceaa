/* oracle/ref_wrap/ref_blend_hsl.c -- TEST INFRASTRUCTURE ONLY.
 * The "RGB (display)" blend colourspace: src/develop/blends/blendif_rgb_hsl.c lifted verbatim at build time
 * (parametric mask on gray / R / G / B / H / S / L and all blend operators).  Driven by ref_develop_blend() in
 * ref_blend.c. */
#define REF_REAL_IMAGEBUF 1
#include "ref_piece.h"
#include "common/imagebuf.h"
#include "common/colorspaces_inline_conversions.h"
#include "math/openmp_maths.h"

typedef int dt_colorspaces_color_profile_type_t;
typedef int dt_colorspaces_color_mode_t;
typedef enum dt_iop_color_intent_t { DT_INTENT_PERCEPTUAL = 0 } dt_iop_color_intent_t;
#define DT_IOP_COLOR_ICC_LEN 512
#include "gen/iop_profile.inc"
#include "gen/iop_profile_info.inc"

typedef char dt_dev_operation_t[20];
typedef int dt_dev_pixelpipe_display_mask_t;
#define DT_DEV_PIXELPIPE_DISPLAY_NONE 0
#define DT_DEV_PIXELPIPE_DISPLAY_ANY 0x3fc /* never requested here */
typedef struct dt_iop_module_so_t dt_iop_module_so_t;
#include "gen/blend_h.inc"

void dt_develop_blendif_process_parameters(float *const restrict parameters, const dt_develop_blend_params_t *const params);
int dt_develop_blendif_init_masking_profile(const struct dt_dev_pixelpipe_t *pipe, const struct dt_dev_pixelpipe_iop_t *piece,
                                            dt_iop_order_iccprofile_info_t *blending_profile, dt_develop_blend_colorspace_t cst);

/* GUI channel display is never requested on an export */
static void _display_channel(const float *a, float *b, const float *mask, size_t stride, int channel, const float *boost,
                             const dt_iop_order_iccprofile_info_t *profile)
{
  (void)a; (void)b; (void)mask; (void)stride; (void)channel; (void)boost; (void)profile;
}
/* the row-function type of blendif_rgb_hsl.c:36-37 (a function typedef, which extract.py does not lift) */
typedef void(_blend_row_func)(const float *const restrict a, const float *const restrict b,
                              float *const restrict out, const float *const restrict mask, const size_t stride);
#include "gen/blendif_rgb_hsl.inc"
