/* TEST INFRASTRUCTURE: stands in for src/develop/iop_order.h of the reference when its host files are compiled against libansel_hip */
#include "boundary_host.h"
