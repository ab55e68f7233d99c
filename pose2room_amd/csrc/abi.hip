// abi.hip -- library identification for loaders (include/p2r_hip.h).
#include "p2r_common.h"

extern "C" int p2r_abi_version(void) { return 3; }
extern "C" const char *p2r_build_arch(void) { return "gfx950"; }
