/* shim: see ref_host.h */
#include "ref_host.h"
