// trace_kernels.hip — every k_trace instantiation and the light pass's k_shade_light (kernel_instances.h), as a translation unit of their own
#define STHIP_TEMPLATE_INSTANCES_ONLY  // the non-template kernels of kernels.h are compiled once, in api.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/sthip.h"
#include "bvh_build.h"
#include "kernel_instances.h"
STHIP_TRACE_ALL(STHIP_TRACE_DEFINE)
STHIP_SHADE_LIGHT(STHIP_LIGHT_DEFINE)
