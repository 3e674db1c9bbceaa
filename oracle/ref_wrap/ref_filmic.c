/* oracle/ref_wrap/ref_filmic.c -- TEST INFRASTRUCTURE ONLY.
 * The reference's filmic RGB tone mapping (src/iop/filmicrgb.c: filmic_agx, filmic_v5,
 * filmic_chroma_v4, filmic_split_v4, the 2019-2020 variants filmic_split_v1 / _v2_v3 and filmic_chroma_v1 / _v2_v3,
 * and everything they call, plus the spline solver
 * dt_iop_filmic_rgb_compute_spline), lifted verbatim at build time, behind C entry points that
 * take the C-ABI structs of include/ansel_hip.h. */
#include "ref_piece.h"
#include "common/colorspaces_inline_conversions.h"
#include "pixel/chromatic_adaptation.h"
#include "math/openmp_maths.h"
#include "math/gaussian_elimination.h"
#include "iop/noise_generator.h"

typedef int dt_colorspaces_color_profile_type_t;
typedef int dt_colorspaces_color_mode_t;
typedef enum dt_iop_color_intent_t { DT_INTENT_PERCEPTUAL = 0 } dt_iop_color_intent_t;
#define DT_IOP_COLOR_ICC_LEN 512
typedef void *GtkWidget;

#include "gen/iop_profile.inc"
#include "gen/iop_profile_info.inc"
#include "gen/filmicrgb.inc"

/* user parameters, == the fields of dt_iop_filmicrgb_params_t (filmicrgb.c:255-286) the spline
 * and the tone mapping depend on */
typedef struct ref_filmic_params_t
{
  float grey_point_source, black_point_source, white_point_source;
  float security_factor, grey_point_target, black_point_target, white_point_target;
  float output_power, latitude, contrast, saturation, balance;
  int preserve_color, version, auto_hardness, custom_grey;
  int shadows, highlights, spline_version;
} ref_filmic_params_t;

/* follows commit_params(), filmicrgb.c:4005-4110, for the fields the device path reads */
int ref_filmicrgb_commit(const ref_filmic_params_t *u, dt_hip_filmicrgb_data_t *d)
{
  ref_reset_fp_mode();
  dt_iop_filmicrgb_params_t p;
  memset(&p, 0, sizeof(p));
  p.grey_point_source = u->grey_point_source;
  p.black_point_source = u->black_point_source;
  p.white_point_source = u->white_point_source;
  p.reconstruct_threshold = 16.0f; /* FILMIC_RECONSTRUCT_DEPRECATED: reconstruction bypassed */
  p.reconstruct_feather = 3.0f;
  p.security_factor = u->security_factor;
  p.grey_point_target = u->grey_point_target;
  p.black_point_target = u->black_point_target;
  p.white_point_target = u->white_point_target;
  p.output_power = u->output_power;
  p.latitude = u->latitude;
  p.contrast = u->contrast;
  p.saturation = u->saturation;
  p.balance = u->balance;
  p.preserve_color = u->preserve_color;
  p.version = u->version;
  p.auto_hardness = u->auto_hardness;
  p.custom_grey = u->custom_grey;
  p.shadows = u->shadows;
  p.highlights = u->highlights;
  p.spline_version = u->spline_version;

  float grey_source = 0.1845f;
  if(p.custom_grey) grey_source = p.grey_point_source / 100.0f;
  const float white_source = p.white_point_source;
  const float black_source = p.black_point_source;
  const float dynamic_range = white_source - black_source;

  memset(d, 0, sizeof(*d));
  d->white_source = white_source;
  d->dynamic_range = dynamic_range;
  d->black_source = black_source;
  d->grey_source = grey_source;
  d->output_power = p.output_power;
  d->version = p.version;
  d->preserve_color = p.preserve_color;

  dt_iop_filmic_rgb_spline_t spline;
  memset(&spline, 0, sizeof(spline));
  dt_iop_filmic_rgb_compute_spline(&p, &spline);
  for(int k = 0; k < 4; k++)
  {
    d->spline.M1[k] = spline.M1[k];
    d->spline.M2[k] = spline.M2[k];
    d->spline.M3[k] = spline.M3[k];
    d->spline.M4[k] = spline.M4[k];
    d->spline.M5[k] = spline.M5[k];
  }
  d->spline.latitude_min = spline.latitude_min;
  d->spline.latitude_max = spline.latitude_max;
  for(int k = 0; k < 5; k++)
  {
    d->spline.x[k] = spline.x[k];
    d->spline.y[k] = spline.y[k];
  }
  d->spline.type[0] = spline.type[0];
  d->spline.type[1] = spline.type[1];

  if(p.version >= DT_FILMIC_COLORSCIENCE_V4)
    d->saturation = p.saturation / 100.0f;
  else
    d->saturation = (2.0f * p.saturation / 100.0f + 1.0f);
  const float agx_axis = CLAMPF(p.saturation / 100.0f, -1.f, 1.f);
  d->agx_beta_hue = 0.5f * (agx_axis + 1.f);
  return 0;
}

static void fill_profile(dt_iop_order_iccprofile_info_t *pr, const float in[3][4], const float out[3][4])
{
  memset(pr, 0, sizeof(*pr));
  for(int r = 0; r < 3; r++)
    for(int c = 0; c < 3; c++)
    {
      pr->matrix_in[r][c] = in[r][c];
      pr->matrix_out[r][c] = out[r][c];
    }
  pr->lutsize = 0x10000;
  pr->nonlinearlut = 0;
}

/* follows the tail of process(), filmicrgb.c:2842-2890 (highlight reconstruction bypassed) */
int ref_filmicrgb(const dt_hip_piece_t *v, const dt_hip_filmicrgb_data_t *d, const void *in, void *out)
{
  ref_reset_fp_mode();
  dt_iop_filmicrgb_data_t data;
  memset(&data, 0, sizeof(data));
  data.white_source = d->white_source;
  data.grey_source = d->grey_source;
  data.black_source = d->black_source;
  data.dynamic_range = d->dynamic_range;
  data.saturation = d->saturation;
  data.output_power = d->output_power;
  data.agx_beta_hue = d->agx_beta_hue;
  data.preserve_color = d->preserve_color;
  data.version = d->version;
  data.hl_deprecated = TRUE;
  /* commit_params(), filmicrgb.c:4101-4102 */
  data.sigma_toe = powf(d->spline.latitude_min / 3.0f, 2.0f);
  data.sigma_shoulder = powf((1.0f - d->spline.latitude_max) / 3.0f, 2.0f);
  for(int k = 0; k < 4; k++)
  {
    data.spline.M1[k] = d->spline.M1[k];
    data.spline.M2[k] = d->spline.M2[k];
    data.spline.M3[k] = d->spline.M3[k];
    data.spline.M4[k] = d->spline.M4[k];
    data.spline.M5[k] = d->spline.M5[k];
  }
  data.spline.latitude_min = d->spline.latitude_min;
  data.spline.latitude_max = d->spline.latitude_max;
  for(int k = 0; k < 5; k++)
  {
    data.spline.x[k] = d->spline.x[k];
    data.spline.y[k] = d->spline.y[k];
  }
  data.spline.type[0] = d->spline.type[0];
  data.spline.type[1] = d->spline.type[1];

  dt_iop_order_iccprofile_info_t work, export_;
  fill_profile(&work, d->work_matrix_in, d->work_matrix_out);
  fill_profile(&export_, d->export_matrix_in, d->export_matrix_out);
  const dt_iop_order_iccprofile_info_t *const export_profile = d->use_output_profile ? &export_ : NULL;

  const size_t width = v->roi_out.width, height = v->roi_out.height;
  const float white_display = powf(data.spline.y[4], data.output_power);
  const float black_display = powf(data.spline.y[0], data.output_power);
  const float *const fin = (const float *)in;
  float *const fout = (float *)out;

  if(_filmic_is_agx(data.version))
    filmic_agx(fin, fout, &work, export_profile, &data, data.spline, width, height, 4, black_display, white_display);
  else if(data.version == DT_FILMIC_COLORSCIENCE_V5)
    filmic_v5(fin, fout, &work, export_profile, &data, data.spline, width, height, 4, black_display, white_display);
  else if(data.version == DT_FILMIC_COLORSCIENCE_V4)
  {
    if(data.preserve_color == DT_FILMIC_METHOD_NONE)
      filmic_split_v4(fin, fout, &work, export_profile, &data, data.spline, data.preserve_color, width, height, 4,
                      data.version, black_display, white_display);
    else
      filmic_chroma_v4(fin, fout, &work, export_profile, &data, data.spline, data.preserve_color, width, height, 4,
                       data.version, black_display, white_display);
  }
  else
  {
    /* the per-channel variants leave the output's alpha unwritten (the pipeline ignores it): the input's here */
    memcpy(fout, fin, sizeof(float) * 4 * width * height);
    if(data.preserve_color == DT_FILMIC_METHOD_NONE)
    {
      if(data.version == DT_FILMIC_COLORSCIENCE_V1)
        filmic_split_v1(fin, fout, &work, &data, data.spline, width, height);
      else
        filmic_split_v2_v3(fin, fout, &work, &data, data.spline, width, height);
    }
    else if(data.version == DT_FILMIC_COLORSCIENCE_V1)
      filmic_chroma_v1(fin, fout, &work, &data, data.spline, data.preserve_color, width, height);
    else
      filmic_chroma_v2_v3(fin, fout, &work, &data, data.spline, data.preserve_color, width, height, 4, data.version);
  }
  return 0;
}
