#include "xv_common.h"
#ifndef XV_SRC_HASH
#define XV_SRC_HASH "unstamped"
#endif
extern "C" int xv_version(void) { return 602; }
extern "C" const char* xv_arch(void) { return "gfx950"; }
extern "C" const char* xv_source_hash(void) { return XV_SRC_HASH; }
