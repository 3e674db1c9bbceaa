/* TEST INFRASTRUCTURE: stands in for src/develop/imageop.h of the reference when its host files are compiled against libansel_hip */
#include "boundary_host.h"
