/* oracle/src/basic.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Restatement of the streaming glue modules of the raw -> RGB export pipe. */
#include <math.h>
#include <string.h>
#include "oracle.h"

/* rawprepare process(), raw mosaic branches: src/iop/rawprepare.c:467-563.
 * crop offsets: compute_proper_crop(), rawprepare.c:206-210. */
int oracle_rawprepare(const dt_hip_piece_t *piece, const dt_hip_rawprepare_data_t *d, const void *ivoid, void *ovoid)
{
  const int width = piece->roi_out.width, height = piece->roi_out.height;
  const int input_width = piece->roi_in.width;
  const int cfa_x = piece->roi_out.x + d->x, cfa_y = piece->roi_out.y + d->y;
  const int csx = (int)roundf((double)d->x * piece->roi_in.scale);
  const int csy = (int)roundf((double)d->y * piece->roi_in.scale);
  if(!(piece->filters && piece->channels == 1)) return 1;
  float inv_div[4];
  for(int k = 0; k < 4; k++) inv_div[k] = 1.0f / d->div[k];
  float *const out = (float *)ovoid;
  #pragma omp parallel for schedule(static)
  for(int j = 0; j < height; j++)
  {
    const size_t pin = (size_t)input_width * (j + csy) + csx;
    const size_t pout = (size_t)j * width;
    const int row_phase = ((j + cfa_y) & 1) << 1;
    for(int i = 0; i < width; i++)
    {
      const int id = row_phase + ((cfa_x + i) & 1);
      const float v = (piece->datatype == DT_HIP_TYPE_UINT16) ? (float)((const uint16_t *)ivoid)[pin + i]
                                                               : ((const float *)ivoid)[pin + i];
      out[pout + i] = (v - d->sub[id]) * inv_div[id];
    }
  }
  return 0;
}

/* temperature process(): src/iop/temperature.c:487-600 (Bayer and 4-channel branches) */
int oracle_temperature(const dt_hip_piece_t *piece, const dt_hip_temperature_data_t *d, const void *ivoid, void *ovoid)
{
  const int width = piece->roi_out.width, height = piece->roi_out.height;
  const float *const in = (const float *)ivoid;
  float *const out = (float *)ovoid;
  if(piece->filters == 9u) return 1;
  if(piece->filters)
  {
    #pragma omp parallel for schedule(static)
    for(int j = 0; j < height; j++)
      for(int i = 0; i < width; i++)
      {
        const size_t p = (size_t)j * width + i;
        out[p] = in[p] * d->coeffs[oracle_fc(j + piece->roi_out.y, i + piece->roi_out.x, piece->filters)];
      }
    return 0;
  }
  if(piece->channels != 4) return 1;
  const size_t npixels = (size_t)width * height;
  #pragma omp parallel for schedule(static)
  for(size_t k = 0; k < npixels; k++)
  {
    out[4 * k + 0] = in[4 * k + 0] * d->coeffs[0];
    out[4 * k + 1] = in[4 * k + 1] * d->coeffs[1];
    out[4 * k + 2] = in[4 * k + 2] * d->coeffs[2];
    out[4 * k + 3] = in[4 * k + 3];
  }
  return 0;
}

/* highlights process(), clip mode: src/iop/highlights.c:680-789 with _hl_count_thresholds
 * (:232-255), _hl_count_clipped (:266-292), _hl_copy_input (:296-302) and process_clip
 * (src/iop/highlights/clip.c:62-84). */
int oracle_highlights(const dt_hip_piece_t *piece, const dt_hip_highlights_data_t *d, const void *ivoid, void *ovoid)
{
  if(d->mode != DT_HIP_HIGHLIGHTS_CLIP) return 1;
  const float *const in = (const float *)ivoid;
  float *const out = (float *)ovoid;
  const size_t npixels = (size_t)piece->roi_out.width * piece->roi_out.height;
  float pmax[4];
  for(int c = 0; c < 4; c++) pmax[c] = (piece->processed_maximum[c] > 0.f) ? piece->processed_maximum[c] : 1.0f;
  const float clip = d->clip * fminf(pmax[0], fminf(pmax[1], pmax[2]));
  const float thr[3] = { clip, clip, clip }; /* clip mode: factor 0 -> scalar clip on all channels */
  const size_t ch = piece->filters ? 1 : piece->channels;
  size_t clipped = 0;
  if(piece->filters)
  {
    const float raw_threshold = fminf(fminf(thr[0], thr[1]), thr[2]);
#pragma omp parallel for schedule(static) reduction(+ : clipped)
    for(size_t k = 0; k < npixels; k++) clipped += (in[k] > raw_threshold);
  }
  else
  {
    const size_t ncol = ch < 3 ? ch : 3;
    for(size_t k = 0; k < npixels; k++)
    {
      int over = 0;
      for(size_t c = 0; c < ncol; c++) over |= (in[k * ch + c] > thr[c]);
      clipped += (over != 0);
    }
  }
  if(clipped < 25) /* DT_HL_MIN_CLIPPED_PIXELS, src/iop/highlights/common.h:218 */
  {
    memcpy(out, in, sizeof(float) * npixels * ch);
    return 0;
  }
  #pragma omp parallel for schedule(static)
  for(size_t k = 0; k < npixels * ch; k++) out[k] = (clip < in[k]) ? clip : in[k]; /* MIN(clip, in[k]) */
  return 0;
}

/* exposure process(): src/iop/exposure.c:501-545 */
int oracle_exposure(const dt_hip_piece_t *piece, const dt_hip_exposure_data_t *d, const void *ivoid, void *ovoid)
{
  const float *const in = (const float *)ivoid;
  float *const out = (float *)ovoid;
  const size_t n = (size_t)piece->roi_out.width * piece->roi_out.height * piece->channels;
  #pragma omp parallel for schedule(static)
  for(size_t k = 0; k < n; k++) out[k] = (in[k] - d->black) * d->scale;
  return 0;
}

/* _export_final_buffer_to_uint16 / _clamp_float_to_uint8: src/imageio/imageio_core.c:706-737.
 * CLAMP is glib's, CLAMPF is src/math/math.h:91. */
int oracle_export_convert_u16(int width, int height, const float *in, uint16_t *out)
{
  const size_t n = (size_t)width * height * 4;
  #pragma omp parallel for schedule(static)
  for(size_t k = 0; k < n; k++)
  {
    const float x = roundf(in[k] * 65535.f);
    const float c = (x > 65535.f) ? 65535.f : ((x < 0.f) ? 0.f : x);
    /* a NaN survives CLAMP(); the reference's (uint16_t) cast of it is cvttss2si -> 0x80000000
     * -> low 16 bits 0 on x86.  Written out here so the oracle does not lean on UB. */
    out[k] = (c != c) ? 0 : (uint16_t)c;
  }
  return 0;
}

int oracle_export_convert_u8(int width, int height, const float *in, uint8_t *out)
{
  const size_t n = (size_t)width * height * 4;
  #pragma omp parallel for schedule(static)
  for(size_t k = 0; k < n; k++)
  {
    const float x = roundf(in[k] * 255.f);
    const float c = (x >= 0.f) ? ((x <= 255.f) ? x : 255.f) : 0.f; /* NaN -> 0.f */
    out[k] = (uint8_t)c;
  }
  return 0;
}

/* dt_hip_raw_unpack() (include/ansel_hip.h): a packed bit stream of photosites -> u16.  NOT a restatement of reference
 * code -- the unpacking is rawspeed's, which is not part of the reference tree: PARITY UNPINNED by the reference.  Written
 * here in the plainest form (one bit at a time) as the checker of the device kernel; tests/test_raw_unpack.py pins it
 * against numpy's unpackbits. */
int oracle_raw_unpack(const uint8_t *packed, int width, int height, size_t row_bytes, int bits, int order, uint16_t *out)
{
  if(bits < 1 || bits > 16 || (order != 0 && order != 1) || row_bytes < ((size_t)width * bits + 7) / 8) return 1;
  for(int y = 0; y < height; y++)
    for(int x = 0; x < width; x++)
    {
      unsigned v = 0;
      for(int b = 0; b < bits; b++)
      {
        const size_t bit = (size_t)x * bits + b; /* b-th bit of the photosite in stream order */
        const unsigned byte = packed[(size_t)y * row_bytes + bit / 8];
        if(order == 0) v = v << 1 | ((byte >> (7 - bit % 8)) & 1u);     /* most significant bit first */
        else v |= ((byte >> (bit % 8)) & 1u) << b;                       /* least significant bit first */
      }
      out[(size_t)y * width + x] = (uint16_t)v;
    }
  return 0;
}
