// placeholder until the filmic kernel lands (next commit)
#include "hip_common.h"
extern "C" int dt_hip_iop_filmicrgb_process(int devid, const dt_hip_piece_t *piece, const dt_hip_filmicrgb_data_t *d,
                                            dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  (void)devid; (void)piece; (void)d; (void)dev_in; (void)dev_out;
  ansel::set_last_error("filmicrgb: not implemented yet");
  return DT_HIP_INVALID_ARG;
}
