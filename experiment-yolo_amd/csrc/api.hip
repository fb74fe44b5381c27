#include "common.h"
#include "dealyolo_hip.h"
extern "C" int dy_abi_version(void) { return 1; }
