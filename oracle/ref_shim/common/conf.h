/* shim: see ref_host.h.  The one configuration lookup the hot path makes:
 * dt_interpolation_new(DT_INTERPOLATION_USERPREF) reads "plugins/lighttable/export/pixel_interpolator"
 * (src/pixel/interpolation.c:606-607). */
#pragma once
#include "ref_host.h"
const char *dt_conf_get_string_const(const char *name);
