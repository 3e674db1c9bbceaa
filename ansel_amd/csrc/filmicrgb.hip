// filmicrgb.hip -- filmic RGB tone mapping on gfx950, highlight reconstruction bypassed (the
// default of every new edit: hl_deprecated, src/iop/filmicrgb.c:2733, :4104).
//
//   filmic_agx()        filmicrgb.c:2495-2587   colour science v8 "AgX" (versions 5..9; default 7)
//   filmic_v5()         filmicrgb.c:2247-2300   v7
//   filmic_chroma_v4()  filmicrgb.c:2153-2198   v6, norm-preserving
//   filmic_split_v4()   filmicrgb.c:2201-2244   v6, per channel
// with log_tonemapping :1047, filmic_spline :1063-1160, get_pixel_norm_simd :976-1035, the Ych /
// gamut-mapping helpers :1740-2030, filmic_v4_prepare_matrices :2033-2064, the AgX bracket
// :2344-2459 and filmic_agx_compress_negatives :2461-2492.
//
// One pointwise kernel per colour science, 16 B in + 16 B out per pixel (32 B/px algorithmic).
// It is the one ALU-heavy pointwise stage of the pipe: per pixel 3 log2f + 3..9 powf (all in
// binary64 inside devmath.h, to return glibc's bits), ~10 3x3 products, ~25 divisions and 3 sqrt.
// The reference measured 1.03 s on CPU and 0.22 s on its OpenCL path for 24 MP (filmicrgb.c:2684).
//
// The per-call matrix preparation the reference does at the top of each of these functions runs
// on the host here (filmic_prepare below), in the same binary32 operation order.
#include "px_filmicrgb.h"

using namespace ansel;

namespace
{

template <int MODE, bool EXPORT>
__global__ __launch_bounds__(256) void filmic_kernel(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                      const size_t npixels, const fargs a_by_value)
{
  constexpr int at = kernarg_offset_after<fargs, const float4 *, float4 *, size_t>(); // after two pointers and a size_t (hip_common.h)
  static_assert(at == 24, "filmic_kernel: the by-value parameter block follows two pointers and a size_t");
  const fargs &a = kernarg_at<fargs>(at);
  (void)a_by_value;
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; // one pixel per thread (pixel_grid)
  if(k < npixels)
    nt_store(out + k, px_filmicrgb<MODE>(in[k], a, EXPORT));
}

// ---------------------------------------------------------------------------------------------
// host: per-call preparation (binary32, reference operation order)
// ---------------------------------------------------------------------------------------------
typedef float mat_t[4][4];

const mat_t XYZ_D50_to_D65_CAT16 = { { 9.89466254e-01f, -4.00304626e-02f, 4.40530317e-02f, 0.f },
                                     { -5.40518733e-03f, 1.00666069e+00f, -1.75551955e-03f, 0.f },
                                     { -4.03920992e-04f, 1.50768030e-02f, 1.30210211e+00f, 0.f } };
const mat_t XYZ_D65_to_D50_CAT16 = { { 1.01085433e+00f, 4.07086103e-02f, -3.41445825e-02f, 0.f },
                                     { 5.42814201e-03f, 9.93581926e-01f, 1.15592039e-03f, 0.f },
                                     { 2.50722468e-04f, -1.14918759e-02f, 7.67964947e-01f, 0.f } };
const mat_t XYZ_D65_to_LMS_2006_D65 = { { 0.257085f, 0.859943f, -0.031061f, 0.f },
                                        { -0.394427f, 1.175800f, 0.106423f, 0.f },
                                        { 0.064856f, -0.076250f, 0.559067f, 0.f } };
const mat_t LMS_2006_D65_to_XYZ_D65 = { { 1.80794659f, -1.29971660f, 0.34785879f, 0.f },
                                        { 0.61783960f, 0.39595453f, -0.04104687f, 0.f },
                                        { -0.12546960f, 0.20478038f, 1.74274183f, 0.f } };
const mat_t filmlightRGB_D65_to_LMS_D65 = { { 0.95f, 0.38f, 0.00f, 0.f }, { 0.05f, 0.62f, 0.03f, 0.f }, { 0.00f, 0.00f, 0.97f, 0.f } };
const mat_t LMS_D65_to_filmlightRGB_D65 = { { 1.0877193f, -0.66666667f, 0.02061856f, 0.f },
                                            { -0.0877193f, 1.66666667f, -0.05154639f, 0.f },
                                            { 0.f, 0.f, 1.03092784f, 0.f } };

// dt_colormatrix_mul(), src/math/matrices.h:167-179
void mat_mul(mat_t dst, const mat_t m1, const mat_t m2)
{
  mat_t t;
  for(int k = 0; k < 3; ++k)
    for(int i = 0; i < 4; i++)
    {
      float sum = 0.0f;
      for(int j = 0; j < 3; j++) sum += m1[k][j] * m2[j][i];
      t[k][i] = sum;
    }
  for(int i = 0; i < 4; i++) t[3][i] = 0.f;
  memcpy(dst, t, sizeof(mat_t));
}

// mat3SSEinv(), src/math/matrices.h:37-66
int mat_inv(mat_t dst, const mat_t src)
{
#define A(y, x) src[(y - 1)][(x - 1)]
#define B(y, x) dst[(y - 1)][(x - 1)]
  const float det = A(1, 1) * (A(3, 3) * A(2, 2) - A(3, 2) * A(2, 3)) - A(2, 1) * (A(3, 3) * A(1, 2) - A(3, 2) * A(1, 3))
                    + A(3, 1) * (A(2, 3) * A(1, 2) - A(2, 2) * A(1, 3));
  if(fabsf(det) < 1e-7f) return 1;
  const float invDet = 1.f / det;
  B(1, 1) = invDet * (A(3, 3) * A(2, 2) - A(3, 2) * A(2, 3));
  B(1, 2) = -invDet * (A(3, 3) * A(1, 2) - A(3, 2) * A(1, 3));
  B(1, 3) = invDet * (A(2, 3) * A(1, 2) - A(2, 2) * A(1, 3));
  B(2, 1) = -invDet * (A(3, 3) * A(2, 1) - A(3, 1) * A(2, 3));
  B(2, 2) = invDet * (A(3, 3) * A(1, 1) - A(3, 1) * A(1, 3));
  B(2, 3) = -invDet * (A(2, 3) * A(1, 1) - A(2, 1) * A(1, 3));
  B(3, 1) = invDet * (A(3, 2) * A(2, 1) - A(3, 1) * A(2, 2));
  B(3, 2) = -invDet * (A(3, 2) * A(1, 1) - A(3, 1) * A(1, 2));
  B(3, 3) = invDet * (A(2, 2) * A(1, 1) - A(2, 1) * A(1, 2));
#undef A
#undef B
  return 0;
}

// dot_product() (matrices.h:201-206) with scalar_product()'s accumulation (math.h:186-194)
void dotp(const float v[4], const mat_t M, float o[4])
{
  for(int i = 0; i < 3; i++)
  {
    float acc = 0.f;
    for(int c = 0; c < 3; c++) acc += v[c] * M[i][c];
    o[i] = acc;
  }
}

void agx_xyz_D50_to_Yrg(const float xyz_D50[4], float Yrg[4])
{
  float xyz_D65[4] = { 0.f }, lms[4] = { 0.f };
  dotp(xyz_D50, XYZ_D50_to_D65_CAT16, xyz_D65);
  dotp(xyz_D65, XYZ_D65_to_LMS_2006_D65, lms);
  const float Y = 0.68990272f * lms[0] + 0.34832189f * lms[1];
  const float a = lms[0] + lms[1] + lms[2];
  float n[4] = { 0.f }, rgb[4] = { 0.f };
  for(int c = 0; c < 4; c++) n[c] = (a == 0.f) ? 0.f : lms[c] / a;
  dotp(n, LMS_D65_to_filmlightRGB_D65, rgb);
  Yrg[0] = Y;
  Yrg[1] = rgb[0];
  Yrg[2] = rgb[1];
}

void agx_Yrg_to_xyz_D50(const float Yrg[4], float xyz_D50[4])
{
  const float Y = Yrg[0], r = Yrg[1], g = Yrg[2];
  const float b = 1.f - r - g;
  const float rgb[4] = { r, g, b, 0.f };
  float lms[4] = { 0.f }, LMS[4] = { 0.f }, xyz_D65[4] = { 0.f };
  dotp(rgb, filmlightRGB_D65_to_LMS_D65, lms);
  const float denom = (0.68990272f * lms[0] + 0.34832189f * lms[1]);
  const float a = (denom == 0.f) ? 0.f : Y / denom;
  for(int c = 0; c < 4; c++) LMS[c] = lms[c] * a;
  dotp(LMS, LMS_2006_D65_to_XYZ_D65, xyz_D65);
  dotp(xyz_D65, XYZ_D65_to_D50_CAT16, xyz_D50);
}

// _filmic_agx_build_displaced(), filmicrgb.c:2344-2388
bool agx_build_displaced(const mat_t work_in, const mat_t work_out, const float inset[3], const float rotation[3], mat_t M)
{
  float white_xyz[4] = { 0.f }, white_Yrg[4] = { 0.f };
  for(int r = 0; r < 3; r++)
    for(int c = 0; c < 3; c++) white_xyz[r] += work_in[r][c];
  agx_xyz_D50_to_Yrg(white_xyz, white_Yrg);
  mat_t P = { { 0.f } };
  for(int i = 0; i < 3; i++)
  {
    const float primary_xyz[4] = { work_in[0][i], work_in[1][i], work_in[2][i], 0.f };
    float primary_Yrg[4] = { 0.f };
    agx_xyz_D50_to_Yrg(primary_xyz, primary_Yrg);
    const float dr = primary_Yrg[1] - white_Yrg[1];
    const float dg = primary_Yrg[2] - white_Yrg[2];
    const float in_i = inset[i];
    const float scale = 1.f - (in_i >= 0.f ? (in_i <= 0.9f ? in_i : 0.9f) : 0.f);
    const float cos_a = cosf(rotation[i]);
    const float sin_a = sinf(rotation[i]);
    const float displaced_Yrg[4] = { primary_Yrg[0], white_Yrg[1] + scale * (cos_a * dr - sin_a * dg),
                                     white_Yrg[2] + scale * (sin_a * dr + cos_a * dg), 0.f };
    float displaced_xyz[4] = { 0.f };
    agx_Yrg_to_xyz_D50(displaced_Yrg, displaced_xyz);
    for(int r = 0; r < 3; r++) P[r][i] = displaced_xyz[r];
  }
  mat_t Pinv = { { 0.f } };
  if(mat_inv(Pinv, P)) return false;
  float s[4] = { 0.f };
  dotp(white_xyz, Pinv, s);
  for(int r = 0; r < 3; r++)
    for(int c = 0; c < 3; c++) P[r][c] *= s[c];
  mat_mul(M, work_out, P);
  return true;
}

void mat_identity(mat_t M)
{
  for(int r = 0; r < 4; r++)
    for(int c = 0; c < 4; c++) M[r][c] = (r == c && r < 3) ? 1.f : 0.f;
}

// filmic_agx_prepare_bracket(), filmicrgb.c:2390-2459
void agx_prepare_bracket(const mat_t work_in, const mat_t work_out, const int variant, mat_t inset, mat_t outset)
{
  // { inset[3], rotation[3], outset[3], outset_rotation[3] } per bleach variant, versions 5..9
  static const float K[5][12] = {
    { +0.5991055f, +0.6000000f, +0.3300009f, +0.0571015f, +0.1999891f, +0.0886110f, 0.761433f, 0.752267f, 0.465293f, -0.0034297f, +0.1952448f, -0.0480109f },
    { +0.6410825f, +0.6898110f, +0.3194529f, +0.0405734f, +0.1631286f, +0.0350584f, 0.784757f, 0.789387f, 0.445403f, -0.0057845f, +0.1593207f, -0.0592955f },
    { +0.6509540f, +0.7488775f, +0.3517703f, +0.0278602f, +0.1214671f, -0.0228829f, 0.793082f, 0.815169f, 0.460318f, -0.0053781f, +0.1187604f, -0.0794801f },
    { +0.6379749f, +0.7878689f, +0.3753822f, +0.0106096f, +0.0582598f, -0.0696729f, 0.790237f, 0.831376f, 0.465406f, -0.0080070f, +0.0571100f, -0.0912220f },
    { +0.5770235f, +0.8102094f, +0.4000390f, -0.0081060f, -0.0034008f, -0.1035236f, 0.766420f, 0.838020f, 0.465130f, -0.0122011f, -0.0021732f, -0.0971215f },
  };
  const float *k = K[(variant >= 5 && variant <= 9) ? variant - 5 : 0];
  mat_t rec = { { 0.f } };
  if(!agx_build_displaced(work_in, work_out, k + 0, k + 3, inset) || !agx_build_displaced(work_in, work_out, k + 6, k + 9, rec)
     || mat_inv(outset, rec))
  {
    mat_identity(inset);
    mat_identity(outset);
  }
}

void to_mat(mat_t m, const float a[3][4])
{
  memset(m, 0, sizeof(mat_t));
  for(int r = 0; r < 3; r++)
    for(int c = 0; c < 3; c++) m[r][c] = a[r][c];
}

void to_m3(m3 &o, const mat_t m)
{
  for(int r = 0; r < 3; r++)
    for(int c = 0; c < 3; c++) o.r[r][c] = m[r][c];
}

void filmic_prepare(const dt_hip_filmicrgb_data_t *d, fargs &a)
{
  memset(&a, 0, sizeof(a));
  mat_t work_in, work_out, exp_in, exp_out, tmp, m;
  to_mat(work_in, d->work_matrix_in);
  to_mat(work_out, d->work_matrix_out);
  to_mat(exp_in, d->export_matrix_in);
  to_mat(exp_out, d->export_matrix_out);
  // filmic_v4_prepare_matrices(), filmicrgb.c:2033-2064
  mat_mul(tmp, XYZ_D50_to_D65_CAT16, work_in);
  mat_mul(m, XYZ_D65_to_LMS_2006_D65, tmp);
  to_m3(a.input, m);
  mat_mul(tmp, XYZ_D65_to_D50_CAT16, LMS_2006_D65_to_XYZ_D65);
  mat_mul(m, work_out, tmp);
  to_m3(a.output, m);
  if(d->use_output_profile)
  {
    mat_mul(tmp, XYZ_D65_to_D50_CAT16, LMS_2006_D65_to_XYZ_D65);
    mat_mul(m, exp_out, tmp);
    to_m3(a.export_output, m);
    mat_mul(tmp, XYZ_D50_to_D65_CAT16, exp_in);
    mat_mul(m, XYZ_D65_to_LMS_2006_D65, tmp);
    to_m3(a.export_input, m);
  }
  if(d->version >= 5)
  {
    mat_t inset = { { 0.f } }, outset = { { 0.f } };
    agx_prepare_bracket(work_in, work_out, d->version, inset, outset);
    to_m3(a.inset, inset);
    to_m3(a.outset, outset);
  }
  for(int c = 0; c < 3; c++) a.luma[c] = work_in[1][c];
  // exp_tonemapping_v2(), filmicrgb.c:1054-1060
  a.norm_min = d->grey_source * exp2f(d->dynamic_range * 0.f + d->black_source);
  a.norm_max = d->grey_source * exp2f(d->dynamic_range * 1.f + d->black_source);
  a.display_white = powf(d->spline.y[4], d->output_power);
  a.display_black = powf(d->spline.y[0], d->output_power);
  a.grey_source = d->grey_source;
  a.black_source = d->black_source;
  a.dynamic_range = d->dynamic_range;
  a.output_power = d->output_power;
  a.saturation = d->saturation;
  a.beta_hue = d->agx_beta_hue;
  for(int k = 0; k < 3; k++)
  {
    a.M1[k] = d->spline.M1[k];
    a.M2[k] = d->spline.M2[k];
    a.M3[k] = d->spline.M3[k];
    a.M4[k] = d->spline.M4[k];
    a.M5[k] = d->spline.M5[k];
  }
  a.inv_M2[0] = 1.f / d->spline.M2[0];
  a.inv_M2[1] = 1.f / d->spline.M2[1];
  a.latitude_min = d->spline.latitude_min;
  a.latitude_max = d->spline.latitude_max;
  a.y0 = d->spline.y[0];
  a.y4 = d->spline.y[4];
  a.type0 = d->spline.type[0];
  a.type1 = d->spline.type[1];
  a.preserve_color = d->preserve_color;
  a.sigma_toe = powf(d->spline.latitude_min / 3.0f, 2.0f);
  a.sigma_shoulder = powf((1.0f - d->spline.latitude_max) / 3.0f, 2.0f);
  a.legacy_version = d->version;
}

template <int MODE>
void launch_m(const bool exp, const unsigned grid, hipStream_t s, const float4 *in, float4 *out, const size_t np, const fargs &a)
{
  if(exp)
    filmic_kernel<MODE, true><<<grid, 256, 0, s>>>(in, out, np, a);
  else
    filmic_kernel<MODE, false><<<grid, 256, 0, s>>>(in, out, np, a);
}

} // namespace

namespace ansel
{
int filmicrgb_fill_args(const dt_hip_filmicrgb_data_t *d, fargs &a)
{
  if(d->version < 0 || d->version > 9)
  {
    set_last_error("filmicrgb: no colour science %d (dt_iop_filmicrgb_colorscience_type_t is 0..9)", d->version);
    return DT_HIP_INVALID_ARG;
  }
  filmic_prepare(d, a);
  a.use_export = d->use_output_profile != 0;
  if(d->version >= 5)
    a.mode = MODE_AGX;
  else if(d->version == 4)
    a.mode = MODE_V5;
  else if(d->version < 3) // process(), filmicrgb.c:2862-2887
    a.mode = d->preserve_color == 0 ? MODE_SPLIT_LEGACY : (d->version == 0 ? MODE_CHROMA_V1 : MODE_CHROMA_V2_V3);
  else
    a.mode = d->preserve_color == 0 ? MODE_SPLIT_V4 : MODE_CHROMA_V4;
  return DT_HIP_SUCCESS;
}
} // namespace ansel

extern "C" int dt_hip_iop_filmicrgb_process(int devid, const dt_hip_piece_t *piece, const dt_hip_filmicrgb_data_t *d,
                                            dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  if(!valid_device(devid) || !piece || !d || !dev_in || !dev_out) return DT_HIP_INVALID_ARG;
  if(piece->channels != 4) return DT_HIP_INVALID_ARG;
  fargs a;
  const int err = filmicrgb_fill_args(d, a);
  if(err != DT_HIP_SUCCESS) return err;
  const size_t np = (size_t)piece->roi_out.width * piece->roi_out.height;
  if(np == 0) return DT_HIP_SUCCESS;
  const unsigned grid = pixel_grid(np);
  hipStream_t s = stream_of(devid);
  const float4 *in = (const float4 *)dev_in;
  float4 *out = (float4 *)dev_out;
  const bool exp = a.use_export != 0;
  launch_scope ls(devid, "filmicrgb");
  switch(a.mode)
  {
    case MODE_AGX: launch_m<MODE_AGX>(exp, grid, s, in, out, np, a); break;
    case MODE_V5: launch_m<MODE_V5>(exp, grid, s, in, out, np, a); break;
    case MODE_SPLIT_V4: launch_m<MODE_SPLIT_V4>(exp, grid, s, in, out, np, a); break;
    case MODE_SPLIT_LEGACY: launch_m<MODE_SPLIT_LEGACY>(false, grid, s, in, out, np, a); break;
    case MODE_CHROMA_V1: launch_m<MODE_CHROMA_V1>(false, grid, s, in, out, np, a); break;
    case MODE_CHROMA_V2_V3: launch_m<MODE_CHROMA_V2_V3>(false, grid, s, in, out, np, a); break;
    default: launch_m<MODE_CHROMA_V4>(exp, grid, s, in, out, np, a); break;
  }
  return check_launch("filmicrgb");
}
