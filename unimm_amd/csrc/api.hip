// ABI identification.
#include "common.h"

extern "C" int unimm_version(void) { return 18; }
extern "C" const char* unimm_arch(void) { return "gfx950"; }
