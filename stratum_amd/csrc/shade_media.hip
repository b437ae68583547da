// shade_media.hip — the k_shade instantiations STHIP_SHADE_MEDIA lists (kernel_instances.h), as a translation unit of their own
#define STHIP_TEMPLATE_INSTANCES_ONLY  // the non-template kernels of kernels.h are compiled once, in api.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/sthip.h"
#include "bvh_build.h"
#include "kernel_instances.h"
STHIP_SHADE_MEDIA(STHIP_SHADE_DEFINE)
