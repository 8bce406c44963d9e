#include "common.h"
extern "C" int tf_abi_version(void) { return 1; }
