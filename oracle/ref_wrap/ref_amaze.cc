/* oracle/ref_wrap/ref_amaze.cc -- TEST INFRASTRUCTURE ONLY.
 * The reference's AMaZE demosaic (src/iop/demosaic/amaze.cc), compiled from where it lies; the one
 * type of src/common/image.h its includes need is declared here instead of pulling that header in. */
typedef int dt_image_orientation_t;
#include "iop/demosaic/amaze.cc"
