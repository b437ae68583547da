"""Wire structs of the boundary as numpy dtypes and ctypes structures.

Mirrors include/sthip_wire.h and include/sthip.h field for field (which in turn
restate src/Shaders/{scene,bdpt,transform,shading_data}.h of the reference).
"""
import ctypes as C

import numpy as np

# ---- numpy dtypes (array payloads) --------------------------------------------------------
InstanceData = np.dtype([("packed", "<u4", (4,))])
PackedVertexData = np.dtype([("position", "<f4", (3,)), ("u", "<f4"), ("normal", "<f4", (3,)), ("v", "<f4")])
TransformData = np.dtype([("m", "<f4", (3, 4))])
ProjectionData = np.dtype(
    [
        ("scale", "<f4", (2,)),
        ("offset", "<f4", (2,)),
        ("near_plane", "<f4"),
        ("far_plane", "<f4"),
        ("sensor_area", "<f4"),
        ("vertical_fov", "<f4"),
    ]
)
ViewData = np.dtype([("projection", ProjectionData), ("image_min", "<i4", (2,)), ("image_max", "<i4", (2,))])
VisibilityInfo = np.dtype([("instance_primitive_index", "<u4"), ("packed_normal", "<u4")])
DepthInfo = np.dtype([("z", "<f4"), ("prev_z", "<f4"), ("dz_dxy", "<f4", (2,))])
ShadingData = np.dtype(
    [
        ("position", "<f4", (3,)),
        ("flags", "<u4"),
        ("packed_geometry_normal", "<u4"),
        ("packed_shading_normal", "<u4"),
        ("packed_tangent", "<u4"),
        ("shape_area", "<f4"),
        ("uv", "<f4", (2,)),
        ("uv_screen_size", "<f4"),
        ("mean_curvature", "<f4"),
    ]
)
ImageValue4 = np.dtype([("value", "<f4", (4,)), ("image_index", "<u4")])
MaterialRecord = np.dtype(
    [("values", ImageValue4, (3,)), ("alpha_mask_index", "<u4"), ("bump_index", "<u4"), ("bump_strength", "<f4")]
)
Ray = np.dtype([("origin", "<f4", (3,)), ("tmin", "<f4"), ("direction", "<f4", (3,)), ("tmax", "<f4")])
Hit = np.dtype([("t", "<f4"), ("b1", "<f4"), ("b2", "<f4"), ("instance_primitive_index", "<u4")])

assert InstanceData.itemsize == 16 and PackedVertexData.itemsize == 32 and TransformData.itemsize == 48
assert ViewData.itemsize == 48 and VisibilityInfo.itemsize == 8 and DepthInfo.itemsize == 16
assert ShadingData.itemsize == 48 and MaterialRecord.itemsize == 72 and Ray.itemsize == 32 and Hit.itemsize == 16

# ---- BDPTFlagBits (bdpt.h:12-40) ----------------------------------------------------------
FLAG_NAMES = [
    "ePerformanceCounters",
    "eRemapThreads",
    "eCoherentRR",
    "eCoherentSampling",
    "eFlipTriangleUVs",
    "eFlipNormalMaps",
    "eAlphaTest",
    "eNormalMaps",
    "eShadingNormalShadowFix",
    "eRayCones",
    "eSampleBSDFs",
    "eNEE",
    "eNEEReservoirs",
    "eNEEReservoirReuse",
    "eMIS",
    "eSampleLightPower",
    "eUniformSphereSampling",
    "ePresampleLights",
    "eDeferShadowRays",
    "eConnectToViews",
    "eConnectToLightPaths",
    "eLVC",
    "eLVCReservoirs",
    "eLVCReservoirReuse",
    "eHashGridJitter",
    "eSampleEnvironmentMapDirectly",
]
FLAG = {n: i for i, n in enumerate(FLAG_NAMES)}

BDPT_FLAG_HAS_ENVIRONMENT = 1
BDPT_FLAG_HAS_EMISSIVES = 2
BDPT_FLAG_HAS_MEDIA = 4
BDPT_FLAG_TRACE_LIGHT = 8

INSTANCE_TYPE_TRIANGLES, INSTANCE_TYPE_SPHERE, INSTANCE_TYPE_VOLUME = 0, 1, 2
LAYOUT_IMAGE, LAYOUT_SHARD_TILES = 0, 1
INVALID_INSTANCE = 0xFFFF
MISS = 0xFFFFFFFF


def flag_mask(*names):
    m = 0
    for n in names:
        m |= 1 << FLAG[n]
    return m


# default sampling flags of the reference renderer (BDPT.cpp:55-62)
DEFAULT_SAMPLING_FLAGS = flag_mask(
    "eRemapThreads", "eRayCones", "eSampleBSDFs", "eCoherentRR", "eNormalMaps", "eNEE", "eMIS", "eDeferShadowRays"
)


# ---- ctypes structures (call descriptors) -------------------------------------------------
class BDPTPushConstants(C.Structure):
    _pack_ = 1
    _fields_ = [
        ("gOutputExtent", C.c_uint32 * 2),
        ("gViewCount", C.c_uint32),
        ("gLightCount", C.c_uint32),
        ("gLightDistributionPDF", C.c_uint32),
        ("gLightDistributionCDF", C.c_uint32),
        ("gEnvironmentMaterialAddress", C.c_uint32),
        ("gEnvironmentSampleProbability", C.c_float),
        ("gRandomSeed", C.c_uint32),
        ("gMinPathVertices", C.c_uint32),
        ("gMaxPathVertices", C.c_uint32),
        ("gMaxDiffuseVertices", C.c_uint32),
        ("gMaxNullCollisions", C.c_uint32),
        ("gLightPresampleTileSize", C.c_uint32),
        ("gLightPresampleTileCount", C.c_uint32),
        ("gLightPathCount", C.c_uint32),
        ("gReservoirM", C.c_uint32),
        ("gReservoirMaxM", C.c_uint32),
        ("gReservoirSpatialM", C.c_uint32),
        ("gHashGridBucketCount", C.c_uint32),
        ("gHashGridMinBucketRadius", C.c_float),
        ("gHashGridBucketPixelRadius", C.c_float),
        ("gDebugViewPathLength", C.c_uint32),
        ("gDebugLightPathLength", C.c_uint32),
    ]


assert C.sizeof(BDPTPushConstants) == 96


class SceneDesc(C.Structure):
    _fields_ = [
        ("gVertices", C.c_void_p),
        ("vertex_count", C.c_uint32),
        ("gIndices", C.c_void_p),
        ("indices_bytes", C.c_uint32),
        ("gInstances", C.c_void_p),
        ("instance_count", C.c_uint32),
        ("gInstanceTransforms", C.c_void_p),
        ("gInstanceInverseTransforms", C.c_void_p),
        ("gInstanceMotionTransforms", C.c_void_p),
        ("gMaterialData", C.c_void_p),
        ("material_bytes", C.c_uint32),
        ("gLightInstances", C.c_void_p),
        ("light_count", C.c_uint32),
        ("gImages", C.c_void_p),
        ("image_count", C.c_uint32),
        ("gDistributions", C.c_void_p),
        ("distribution_count", C.c_uint32),
        ("gImage1s", C.c_void_p),
        ("image1_count", C.c_uint32),
        ("gVolumes", C.c_void_p),
        ("volume_count", C.c_uint32),
    ]


class VolumeDesc(C.Structure):
    _fields_ = [("data", C.c_void_p), ("bytes", C.c_uint64)]


class ImageDesc(C.Structure):
    _fields_ = [("pixels", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


class FrameDesc(C.Structure):
    _fields_ = [
        ("gViews", C.c_void_p),
        ("gViewTransforms", C.c_void_p),
        ("gInverseViewTransforms", C.c_void_p),
        ("gPrevViews", C.c_void_p),
        ("gPrevInverseViewTransforms", C.c_void_p),
        ("view_count", C.c_uint32),
        ("gViewMediumInstances", C.c_void_p),
    ]


class Outputs(C.Structure):
    _fields_ = [
        ("device_ptrs", C.c_uint32),
        ("radiance_layout", C.c_uint32),  # LAYOUT_IMAGE / LAYOUT_SHARD_TILES
        ("gRadiance", C.c_void_p),
        ("gAlbedo", C.c_void_p),
        ("gVisibility", C.c_void_p),
        ("gDepth", C.c_void_p),
        ("gPrevUVs", C.c_void_p),
        ("gRayCount", C.c_void_p),
        ("debug_mode", C.c_uint32),  # BDPTDebugMode (DEBUG_*), with gDebugImage (RGBA32F, in / out)
        ("gDebugImage", C.c_void_p),
    ]


# BDPTDebugMode, bdpt.h:177-193 (include/sthip.h: STHIP_DEBUG_*)
(DEBUG_NONE, DEBUG_ALBEDO, DEBUG_SPECULAR, DEBUG_EMISSION, DEBUG_SHADING_NORMAL, DEBUG_GEOMETRY_NORMAL, DEBUG_DIR_OUT, DEBUG_PREV_UV, DEBUG_ENVIRONMENT_SAMPLE_TEST,
 DEBUG_ENVIRONMENT_SAMPLE_PDF, DEBUG_RESERVOIR_WEIGHT, DEBUG_PATH_LENGTH_CONTRIBUTION, DEBUG_LIGHT_TRACE_CONTRIBUTION, DEBUG_VIEW_TRACE_CONTRIBUTION, DEBUG_MODE_COUNT) = range(15)


class Stats(C.Structure):
    _fields_ = [
        ("rays_total", C.c_uint64),
        ("rays_path", C.c_uint64),
        ("rays_shadow", C.c_uint64),
        ("nodes_visited", C.c_uint64),
        ("tris_tested", C.c_uint64),
        ("nodes_visited_shadow", C.c_uint64),
        ("tris_tested_shadow", C.c_uint64),
        ("ms_trace", C.c_float),
        ("ms_shade", C.c_float),
        ("ms_total", C.c_float),
        ("launches_trace", C.c_uint32),
        ("bvh_node_bytes", C.c_uint32),
        ("bvh_tri_bytes", C.c_uint32),
        ("bvh_nodes", C.c_uint64),
        ("bvh_tris", C.c_uint64),
        ("bvh_build_ms", C.c_float),
        ("bvh_build_gpu_ms", C.c_float),
        ("inner_slots", C.c_uint64 * 2),
        ("tri_slots", C.c_uint64 * 2),
        ("round_slots", C.c_uint64 * 2),
        ("busy_rounds", C.c_uint64 * 2),
        ("rays_primary_packets", C.c_uint64),
        ("nodes_visited_primary", C.c_uint64),
        ("tris_tested_primary", C.c_uint64),
        ("ms_trace_primary", C.c_float),
        ("launches_primary", C.c_uint32),
        ("lane_states", C.c_uint64 * 8),
        ("rays_answered", C.c_uint64),
        ("paths_per_seed", C.c_uint32),
        ("seeds_in_flight", C.c_uint32),
        ("max_paths_in_flight", C.c_uint64),
        ("batch_halvings", C.c_uint32),
        ("full_rebuilds", C.c_uint32),
    ]


def ptr(a):
    """void* of a C-contiguous numpy array (None -> NULL)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def default_push_constants(width, height, light_count, view_count=1):
    """Defaults of the reference renderer (BDPT.cpp:63-76) plus the per-frame fields BDPT::render
    fills in (BDPT.cpp:469-503): no environment, no media."""
    pc = BDPTPushConstants()
    pc.gOutputExtent[0], pc.gOutputExtent[1] = width, height
    pc.gViewCount = view_count
    pc.gLightCount = light_count
    pc.gLightDistributionPDF = 0
    pc.gLightDistributionCDF = 0
    pc.gEnvironmentMaterialAddress = 0xFFFFFFFF
    pc.gEnvironmentSampleProbability = 0.5  # BDPT.cpp:67; BDPT::render forces 0 / 1 without an environment / without emitters (:488-496)
    pc.gRandomSeed = 0
    pc.gMinPathVertices = 4
    pc.gMaxPathVertices = 8
    pc.gMaxDiffuseVertices = 2
    pc.gMaxNullCollisions = 64  # BDPT.cpp:62; BDPT::render zeroes it for scenes without media (:497-500)
    pc.gLightPresampleTileSize = 1024
    pc.gLightPresampleTileCount = 128
    pc.gLightPathCount = width * height
    pc.gReservoirM = 16
    pc.gReservoirMaxM = 64
    pc.gReservoirSpatialM = 4
    pc.gHashGridBucketCount = 200000
    pc.gHashGridMinBucketRadius = 0.1
    pc.gHashGridBucketPixelRadius = 6
    return pc


# ---- after the path (include/sthip.h, "display transform, image metric, HDR export") ----
TONEMAP_MODES = [  # TonemapMode, tonemap.h:8-21
    "Raw",
    "Reinhard",
    "ReinhardExtended",
    "ReinhardLuminance",
    "ReinhardLuminanceExtended",
    "Uncharted2",
    "Filmic",
    "ACES",
    "ACESApprox",
    "ViridisR",
    "ViridisLengthRGB",
]
TONEMAP = {n: i for i, n in enumerate(TONEMAP_MODES)}
COMPARE_MODES = ["SMAPE", "MSE", "Average"]  # ImageCompareMode, image_compare.hlsl:5-9
COMPARE = {n: i for i, n in enumerate(COMPARE_MODES)}


class TonemapDesc(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("mode", C.c_uint32),
        ("modulate_albedo", C.c_uint32),
        ("gamma_correction", C.c_uint32),
        ("exposure", C.c_float),
        ("device_ptrs", C.c_uint32),
        ("exposure_alpha", C.c_float),
        ("gInput", C.c_void_p),
        ("gAlbedo", C.c_void_p),
        ("gOutput", C.c_void_p),
        ("out_max", C.c_void_p),
        ("exposure_state", C.c_void_p),
    ]


class AccumulateDesc(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("view_count", C.c_uint32),
        ("reprojection", C.c_uint32),
        ("demodulate_albedo", C.c_uint32),
        ("history_limit", C.c_float),
        ("device_ptrs", C.c_uint32),
        ("instance_count", C.c_uint32),
        ("gViews", C.c_void_p),
        ("gRadiance", C.c_void_p),
        ("gAlbedo", C.c_void_p),
        ("gVisibility", C.c_void_p),
        ("gDepth", C.c_void_p),
        ("gPrevUVs", C.c_void_p),
        ("gPrevVisibility", C.c_void_p),
        ("gPrevDepth", C.c_void_p),
        ("gPrevAccumColor", C.c_void_p),
        ("gPrevAccumMoments", C.c_void_p),
        ("gInstanceIndexMap", C.c_void_p),
        ("gAccumColor", C.c_void_p),
        ("gAccumMoments", C.c_void_p),
    ]

