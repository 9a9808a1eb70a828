#include "xv_common.h"
extern "C" int xv_version(void) { return 100; }
extern "C" const char* xv_arch(void) { return "gfx950"; }
