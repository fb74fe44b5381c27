// soft-NMS / inference decode kernels (added after the training path)
#include "common.h"
