/* the device runtime header of the replaced build: src/common/opencl.h -> include/ansel_opencl_peer.h */
#include "ansel_opencl_peer.h"
#include "boundary_host.h"
