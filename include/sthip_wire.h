/* sthip_wire.h — byte-exact wire structs of the Stratum path-tracing boundary.
 *
 * Every struct here is the plain-C restatement of a struct that Stratum shares
 * between its C++ host and its Slang/HLSL shaders. Layouts, field order and
 * bit packing follow the reference exactly (citations are relative to the
 * reference tree, src/Shaders/...). They are the data format of the C ABI in
 * sthip.h; both the HIP product and the CPU oracle consume them unchanged.
 */
#ifndef STHIP_WIRE_H
#define STHIP_WIRE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#pragma pack(push, 1)

/* scene.h:14-24 */
#define STHIP_INSTANCE_TYPE_TRIANGLES 0u
#define STHIP_INSTANCE_TYPE_SPHERE 1u
#define STHIP_INSTANCE_TYPE_VOLUME 2u
#define STHIP_INVALID_INSTANCE 0xFFFFu
#define STHIP_INVALID_PRIMITIVE 0xFFFFu
#define STHIP_IMAGE_COUNT 4096u /* scene.h:26; image_index >= this means "no texture" (image_value.h:187) */

/* W2 — InstanceData, scene.h:29-47 (makers :50-79).
 * packed[0] = type:4 | material_address:28
 * packed[1] = light_index:12 | prim_count:16 | index_stride:4
 * packed[2] = first_vertex (mesh) | radius as f32 bits (sphere) | volume_index
 * packed[3] = indices_byte_offset */
typedef struct sthip_InstanceData {
  uint32_t packed[4];
} sthip_InstanceData;

/* W3 — PackedVertexData, scene.h:81-94 */
typedef struct sthip_PackedVertexData {
  float position[3];
  float u;
  float normal[3];
  float v;
} sthip_PackedVertexData;

/* W5 — TransformData, transform.h:6-46: row-major 3x4 affine */
typedef struct sthip_TransformData {
  float m[3][4];
} sthip_TransformData;

/* W6 — ProjectionData, transform.h:109-148 */
typedef struct sthip_ProjectionData {
  float scale[2];
  float offset[2];
  float near_plane;
  float far_plane;
  float sensor_area;
  float vertical_fov;
} sthip_ProjectionData;

/* W6 — ViewData, scene.h:96-112 */
typedef struct sthip_ViewData {
  sthip_ProjectionData projection;
  int32_t image_min[2];
  int32_t image_max[2];
} sthip_ViewData;

/* W8 — VisibilityInfo / DepthInfo, scene.h:114-128.
 * instance_primitive_index = instance:16 (low) | primitive:16 (high), intersection.hlsli:13-18 */
typedef struct sthip_VisibilityInfo {
  uint32_t instance_primitive_index;
  uint32_t packed_normal;
} sthip_VisibilityInfo;

typedef struct sthip_DepthInfo {
  float z;
  float prev_z;
  float dz_dxy[2];
} sthip_DepthInfo;

/* W7 — ShadingData, shading_data.h:10-22 */
#define STHIP_SHADING_FLAG_FRONT_FACE 1u
#define STHIP_SHADING_FLAG_FLIP_BITANGENT 2u
typedef struct sthip_ShadingData {
  float position[3];
  uint32_t flags;
  uint32_t packed_geometry_normal;
  uint32_t packed_shading_normal;
  uint32_t packed_tangent;
  float shape_area;
  float uv[2];
  float uv_screen_size;
  float mean_curvature;
} sthip_ShadingData;

/* W9 — ShadowRayData, bdpt.h:83-90 */
typedef struct sthip_ShadowRayData {
  float contribution[3];
  uint32_t rng_offset;
  float ray_origin[3];
  uint32_t medium;
  float ray_direction[3];
  float ray_distance;
} sthip_ShadowRayData;

/* W1 — BDPTPushConstants, bdpt.h:51-81 (#pragma pack(1), 24 dwords) */
typedef struct sthip_BDPTPushConstants {
  uint32_t gOutputExtent[2];
  uint32_t gViewCount;
  uint32_t gLightCount;
  uint32_t gLightDistributionPDF;
  uint32_t gLightDistributionCDF;
  uint32_t gEnvironmentMaterialAddress; /* 0xFFFFFFFF = none */
  float gEnvironmentSampleProbability;
  uint32_t gRandomSeed;
  uint32_t gMinPathVertices;
  uint32_t gMaxPathVertices;
  uint32_t gMaxDiffuseVertices;
  uint32_t gMaxNullCollisions;
  uint32_t gLightPresampleTileSize;
  uint32_t gLightPresampleTileCount;
  uint32_t gLightPathCount;
  uint32_t gReservoirM;
  uint32_t gReservoirMaxM;
  uint32_t gReservoirSpatialM;
  uint32_t gHashGridBucketCount;
  float gHashGridMinBucketRadius;
  float gHashGridBucketPixelRadius;
  uint32_t gDebugViewPathLength;
  uint32_t gDebugLightPathLength;
} sthip_BDPTPushConstants;

/* S3 — material record as serialised by Material::store (Material.hpp:32-38,
 * image_value.h:183-207) and read by DisneyMaterial::load (disney_material.hlsli:46-79).
 * data[0] = base_color.rgb, emission; data[1] = metallic, roughness, anisotropic, subsurface;
 * data[2] = clearcoat, clearcoat_gloss, transmission, eta (disney_data.h:1-20). */
typedef struct sthip_ImageValue4 {
  float value[4];
  uint32_t image_index;
} sthip_ImageValue4;

typedef struct sthip_MaterialRecord {
  sthip_ImageValue4 values[3];
  uint32_t alpha_mask_index;
  uint32_t bump_index;
  float bump_strength;
} sthip_MaterialRecord;

#pragma pack(pop)

/* BDPTFlagBits, bdpt.h:12-40 (bit positions of sampling_flags) */
enum sthip_BDPTFlagBits {
  STHIP_ePerformanceCounters = 0,
  STHIP_eRemapThreads,
  STHIP_eCoherentRR,
  STHIP_eCoherentSampling,
  STHIP_eFlipTriangleUVs,
  STHIP_eFlipNormalMaps,
  STHIP_eAlphaTest,
  STHIP_eNormalMaps,
  STHIP_eShadingNormalShadowFix,
  STHIP_eRayCones,
  STHIP_eSampleBSDFs,
  STHIP_eNEE,
  STHIP_eNEEReservoirs,
  STHIP_eNEEReservoirReuse,
  STHIP_eMIS,
  STHIP_eSampleLightPower,
  STHIP_eUniformSphereSampling,
  STHIP_ePresampleLights,
  STHIP_eDeferShadowRays,
  STHIP_eConnectToViews,
  STHIP_eConnectToLightPaths,
  STHIP_eLVC,
  STHIP_eLVCReservoirs,
  STHIP_eLVCReservoirReuse,
  STHIP_eHashGridJitter,
  STHIP_eSampleEnvironmentMapDirectly,
  STHIP_eBDPTFlagCount
};

/* scene flags, bdpt.h:46-49 */
#define STHIP_BDPT_FLAG_HAS_ENVIRONMENT 1u
#define STHIP_BDPT_FLAG_HAS_EMISSIVES 2u
#define STHIP_BDPT_FLAG_HAS_MEDIA 4u
#define STHIP_BDPT_FLAG_TRACE_LIGHT 8u

#ifdef __cplusplus
} /* extern "C" */
static_assert(sizeof(sthip_InstanceData) == 16, "InstanceData");
static_assert(sizeof(sthip_PackedVertexData) == 32, "PackedVertexData");
static_assert(sizeof(sthip_TransformData) == 48, "TransformData");
static_assert(sizeof(sthip_ViewData) == 48, "ViewData");
static_assert(sizeof(sthip_VisibilityInfo) == 8, "VisibilityInfo");
static_assert(sizeof(sthip_DepthInfo) == 16, "DepthInfo");
static_assert(sizeof(sthip_ShadingData) == 48, "ShadingData");
static_assert(sizeof(sthip_ShadowRayData) == 48, "ShadowRayData");
static_assert(sizeof(sthip_BDPTPushConstants) == 96, "BDPTPushConstants");
static_assert(sizeof(sthip_MaterialRecord) == 72, "MaterialRecord");
#endif

#endif /* STHIP_WIRE_H */
