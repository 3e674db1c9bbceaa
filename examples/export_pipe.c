/* examples/export_pipe.c -- the C-ABI from C, the reference's own language: no Python, no torch.
 *
 * What a host like src/develop/pixelpipe_hb.c does with libansel_hip.so for one export: take a device, upload
 * the sensor buffer (basebuffer), hand the enabled nodes to the executor, run it, read the exported frame back.
 * The mosaic is synthetic (a 64-bit LCG, reproducible from tests/test_gpu_c_example.py); the program prints the
 * FNV-1a hash of the exported RGBA u16 frame and the time of one pass.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/export_pipe.c -Lansel_amd -lansel_hip -Wl,-rpath,'$ORIGIN/../ansel_amd' -o examples/export_pipe
 *   examples/export_pipe [width height [bands]]
 *
 * bands > 1: the same frame cut into row bands and walked by dt_hip_pipe_process_bands() -- one pipe per band, band k on
 * device k % (number of devices) -- which must print the same hash as the unsplit run.
 * bands < 0: a batch export of -bands frames (the same frame every time) through dt_hip_batch_*: upload, kernels, download
 * and the "encoder" -- a writer callback that hashes the frame, where format->write_image() would compress and write it
 * (src/imageio/imageio_core.c:965) -- of consecutive frames overlapped.  Prints written=N and the hash every frame had.
 */
#define _POSIX_C_SOURCE 199309L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ansel_hip.h"

#define CHECK(call)                                                                                  \
  do                                                                                                 \
  {                                                                                                  \
    const int rc_ = (call);                                                                          \
    if(rc_ != DT_HIP_SUCCESS)                                                                        \
    {                                                                                                \
      fprintf(stderr, "%s failed with %d: %s\n", #call, rc_, dt_hip_last_error());                   \
      return 1;                                                                                      \
    }                                                                                                \
  } while(0)

static dt_hip_piece_t piece_of(int w, int h, uint32_t filters, uint32_t channels, uint32_t datatype, const float *pmax)
{
  dt_hip_piece_t p;
  memset(&p, 0, sizeof(p));
  p.roi_in.width = p.roi_out.width = w;
  p.roi_in.height = p.roi_out.height = h;
  p.roi_in.scale = p.roi_out.scale = 1.0;
  p.filters = filters;
  p.channels = channels;
  p.datatype = datatype;
  for(int c = 0; c < 4; c++) p.processed_maximum[c] = pmax[c];
  return p;
}

/* the batch's writer: called on the batch's writer thread, frame after frame in submission order */
typedef struct
{
  long frames;
  uint64_t first_hash;
  int all_equal;
} encoder_t;

static int write_image(void *user, long seq, void *host_out, size_t bytes)
{
  encoder_t *e = (encoder_t *)user;
  uint64_t fnv = 0xcbf29ce484222325ull;
  const unsigned char *b = (const unsigned char *)host_out;
  for(size_t k = 0; k < bytes; k++) fnv = (fnv ^ b[k]) * 0x100000001b3ull;
  if(seq != e->frames) return 1; /* out of order */
  if(seq == 0) e->first_hash = fnv;
  else if(fnv != e->first_hash) e->all_equal = 0;
  e->frames++;
  return 0;
}

int main(int argc, char **argv)
{
  const int w = argc > 2 ? atoi(argv[1]) : 1504, h = argc > 2 ? atoi(argv[2]) : 1000;
  const int arg3 = argc > 3 ? atoi(argv[3]) : 1;
  const int nframes = arg3 < 0 ? -arg3 : 0, nbands = arg3 < 0 ? 1 : arg3;
  if(nbands < 1 || nbands > 16 || nframes > 64) return 1;
  const uint32_t RGGB = 0x94949494u;
  const float wb[4] = { 2.1f, 1.0f, 1.6f, 1.0f }, ones[4] = { 1.f, 1.f, 1.f, 1.f };

  if(dt_hip_init() != DT_HIP_SUCCESS)
  {
    fprintf(stderr, "dt_hip_init: %s\n", dt_hip_last_error());
    return 2; /* no GPU: there is no CPU fallback */
  }
  const int dev = dt_hip_reserve_device_for_pipe(0);
  if(dev < 0) return 2;

  /* the "sensor buffer": 14-bit values from a 64-bit LCG, in pinned host memory */
  const size_t npix = (size_t)w * h;
  uint16_t *raw = (uint16_t *)dt_hip_alloc_host_pinned(npix * sizeof(uint16_t));
  uint16_t *out = (uint16_t *)dt_hip_alloc_host_pinned(npix * 4 * sizeof(uint16_t));
  if(!raw || !out) return 1;
  uint64_t s = 0x9E3779B97F4A7C15ull;
  for(size_t k = 0; k < npix; k++)
  {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    raw[k] = (uint16_t)(512 + ((s >> 33) % 15000));
  }

  dt_hip_mem_t dev_raw = dt_hip_alloc_device(dev, w, h, 2), dev_out = dt_hip_alloc_device(dev, w, h, 8);
  if(!dev_raw || !dev_out) return 1;

  /* the node list, in pipe order, each with its piece view and committed data */
  const dt_hip_piece_t p_raw = piece_of(w, h, RGGB, 1, DT_HIP_TYPE_UINT16, ones);
  const dt_hip_piece_t p_cfa = piece_of(w, h, RGGB, 1, DT_HIP_TYPE_FLOAT, ones);
  const dt_hip_piece_t p_cfa_wb = piece_of(w, h, RGGB, 1, DT_HIP_TYPE_FLOAT, wb);
  const dt_hip_piece_t p_rgb = piece_of(w, h, 0, 4, DT_HIP_TYPE_FLOAT, wb);
  const dt_hip_rawprepare_data_t rawprepare = { 0, 0, 0, 0, { 512.f, 512.f, 512.f, 512.f }, { 15871.f, 15871.f, 15871.f, 15871.f } };
  const dt_hip_temperature_data_t temperature = { { wb[0], wb[1], wb[2], wb[3] } };
  const dt_hip_highlights_data_t highlights = { DT_HIP_HIGHLIGHTS_CLIP, 1.0f };
  const dt_hip_demosaic_data_t demosaic = { 0, 0, DT_HIP_DEMOSAIC_RCD, 0.0f };
  const dt_hip_exposure_data_t exposure = { -0.000244140625f, 1.6245047f };

  dt_hip_pipe_t *pipe = dt_hip_pipe_new(dev);
  CHECK(dt_hip_pipe_add_node(pipe, "rawprepare", &p_raw, &rawprepare, sizeof(rawprepare)));
  CHECK(dt_hip_pipe_add_node(pipe, "temperature", &p_cfa, &temperature, sizeof(temperature)));
  CHECK(dt_hip_pipe_add_node(pipe, "highlights", &p_cfa_wb, &highlights, sizeof(highlights)));
  CHECK(dt_hip_pipe_add_node(pipe, "demosaic", &p_cfa_wb, &demosaic, sizeof(demosaic)));
  CHECK(dt_hip_pipe_add_node(pipe, "exposure", &p_rgb, &exposure, sizeof(exposure)));
  CHECK(dt_hip_pipe_add_node(pipe, "export_u16", &p_rgb, NULL, 0));

  struct timespec t0, t1;
  double best = 1e30;
  if(nframes)
  {
    /* three slots; the pinned buffers go round with them: submit() hands a slot out again only when its frame's writer
     * has returned, so slot k's buffers are free to refill as soon as the wait for it comes back */
    enum { DEPTH = 3 };
    encoder_t enc = { 0, 0, 1 };
    uint16_t *pin_in[DEPTH], *pin_out[DEPTH];
    for(int k = 0; k < DEPTH; k++)
    {
      pin_in[k] = (uint16_t *)dt_hip_alloc_host_pinned(npix * sizeof(uint16_t));
      pin_out[k] = (uint16_t *)dt_hip_alloc_host_pinned(npix * 4 * sizeof(uint16_t));
      if(!pin_in[k] || !pin_out[k]) return 1;
    }
    dt_hip_batch_t *batch = dt_hip_batch_new(pipe, DEPTH, npix * sizeof(uint16_t), npix * 4 * sizeof(uint16_t));
    if(!batch) return 1;
    CHECK(dt_hip_batch_set_writer(batch, write_image, &enc));
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for(int f = 0; f < nframes; f++)
    {
      const int k = f % DEPTH;
      CHECK(dt_hip_batch_wait(batch, k));                   /* frame f - DEPTH has been written */
      memcpy(pin_in[k], raw, npix * sizeof(uint16_t));      /* "decode" frame f */
      if(dt_hip_batch_submit(batch, pin_in[k], pin_out[k]) != k) return 1;
    }
    CHECK(dt_hip_batch_drain(batch));
    clock_gettime(CLOCK_MONOTONIC, &t1);
    best = ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6) / nframes;
    memcpy(out, pin_out[(nframes - 1) % DEPTH], npix * 4 * sizeof(uint16_t));
    dt_hip_batch_free(batch);
    for(int k = 0; k < DEPTH; k++)
    {
      dt_hip_free_host_pinned(pin_in[k]);
      dt_hip_free_host_pinned(pin_out[k]);
    }
    printf("written=%ld all_equal=%d ", enc.frames, enc.all_equal);
  }
  else if(nbands == 1)
  {
    for(int pass = 0; pass < 3; pass++)
    {
      clock_gettime(CLOCK_MONOTONIC, &t0);
      CHECK(dt_hip_iop_basebuffer_process(dev, &p_raw, w, h, 2, raw, dev_raw)); /* the frame's upload */
      CHECK(dt_hip_pipe_process(pipe, dev_raw, dev_out));
      CHECK(dt_hip_read_host_from_device(dev, out, dev_out, w, h, 8));
      clock_gettime(CLOCK_MONOTONIC, &t1);
      const double ms = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
      if(ms < best) best = ms;
    }
  }
  else
  {
    /* one frame over `nbands` row bands: a pipe with the same nodes, an input and an output buffer per band, each on
     * its band's device; the library walks the bands in lockstep and moves the halo rows between them */
    const int ndev = dt_hip_get_num_devices();
    dt_hip_band_t bands[16];
    dt_hip_pipe_t *pipes[16];
    dt_hip_mem_t b_in[16], b_out[16];
    int b_dev[16];
    CHECK(dt_hip_plan_bands(w, h, DT_HIP_DEMOSAIC_RCD, nbands, bands));
    for(int k = 0; k < nbands; k++)
    {
      b_dev[k] = k % ndev;
      pipes[k] = dt_hip_pipe_new(b_dev[k]);
      CHECK(dt_hip_pipe_add_node(pipes[k], "rawprepare", &p_raw, &rawprepare, sizeof(rawprepare)));
      CHECK(dt_hip_pipe_add_node(pipes[k], "temperature", &p_cfa, &temperature, sizeof(temperature)));
      CHECK(dt_hip_pipe_add_node(pipes[k], "highlights", &p_cfa_wb, &highlights, sizeof(highlights)));
      CHECK(dt_hip_pipe_add_node(pipes[k], "demosaic", &p_cfa_wb, &demosaic, sizeof(demosaic)));
      CHECK(dt_hip_pipe_add_node(pipes[k], "exposure", &p_rgb, &exposure, sizeof(exposure)));
      CHECK(dt_hip_pipe_add_node(pipes[k], "export_u16", &p_rgb, NULL, 0));
      b_in[k] = dt_hip_alloc_device(b_dev[k], w, bands[k].rows, 2);
      b_out[k] = dt_hip_alloc_device(b_dev[k], w, bands[k].rows, 8);
      if(!b_in[k] || !b_out[k]) return 1;
    }
    for(int pass = 0; pass < 3; pass++)
    {
      clock_gettime(CLOCK_MONOTONIC, &t0);
      for(int k = 0; k < nbands; k++) /* each band's rows of the sensor buffer to its device */
        CHECK(dt_hip_write_host_to_device(b_dev[k], raw + (size_t)bands[k].row0 * w, b_in[k], w, bands[k].rows, 2));
      CHECK(dt_hip_pipe_process_bands(pipes, nbands, bands, b_in, b_out));
      for(int k = 0; k < nbands; k++)
        CHECK(dt_hip_read_host_from_device(b_dev[k], out + (size_t)bands[k].row0 * w * 4, b_out[k], w, bands[k].rows, 8));
      clock_gettime(CLOCK_MONOTONIC, &t1);
      const double ms = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
      if(ms < best) best = ms;
    }
    for(int k = 0; k < nbands; k++)
    {
      dt_hip_pipe_free(pipes[k]);
      dt_hip_release_mem_object(b_in[k]);
      dt_hip_release_mem_object(b_out[k]);
    }
  }

  uint64_t fnv = 0xcbf29ce484222325ull;
  const unsigned char *bytes = (const unsigned char *)out;
  for(size_t k = 0; k < npix * 8; k++) fnv = (fnv ^ bytes[k]) * 0x100000001b3ull;
  printf("%s %dx%d groups=%d bands=%d fnv1a=%016llx host_to_host_ms=%.3f\n", dt_hip_get_device_name(dev), w, h,
         dt_hip_pipe_num_groups(pipe), nbands, (unsigned long long)fnv, best);

  dt_hip_pipe_free(pipe);
  dt_hip_release_mem_object(dev_raw);
  dt_hip_release_mem_object(dev_out);
  dt_hip_free_host_pinned(raw);
  dt_hip_free_host_pinned(out);
  dt_hip_release_device(dev);
  dt_hip_cleanup();
  return 0;
}
