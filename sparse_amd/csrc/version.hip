// Library identity entry points.
#include "common.h"

extern "C" int spamd_version(void) { return 100; /* 0.1.0 */ }

extern "C" const char* spamd_target_arch(void) { return "gfx950"; }
