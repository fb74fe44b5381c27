// LDConv sampling kernels (added after the training path)
#include "common.h"
