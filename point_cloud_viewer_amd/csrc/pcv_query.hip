// pcv_query.hip — frustum/OBB transform-and-cull kernels (filled in below).
#include "pcv_internal.h"

