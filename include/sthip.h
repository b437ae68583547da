/* sthip.h — C ABI of libstratum_hip.so: the MI355X path-tracing inner loop of Stratum.
 *
 * This library replaces exactly two things in the reference:
 *   - the Slang/HLSL ray-trace pipeline in src/Shaders (entry points
 *     sample_visibility, bdpt.hlsl:149-300, and trace_shadows, bdpt.hlsl:302-326,
 *     and everything they call), and
 *   - the Vulkan dispatch sequence that BDPT::render records for them
 *     (src/Node/BDPT.cpp:423-838; dispatch_over, src/Core/CommandBuffer.hpp:183-197),
 *     together with the driver-side acceleration structure it relies on
 *     (src/Core/AccelerationStructure.cpp:5-27, src/Node/Scene.cpp:435-459,614-629).
 * Everything above it (Node graph, Scene::update, loaders, GUI) stays as it is and
 * hands over the same arrays it binds to the shaders today; field names below are the
 * reflected binding names the reference uses ("gSceneParams.gVertices", ...,
 * BDPT.cpp:401-417,611-629; parameter blocks bdpt.hlsl:19-62).
 *
 * Conventions: plain pointers and sizes, no exceptions across the ABI; every call
 * returns 0 on success or a negative sthip_status, and sthip_last_error() returns a
 * description. A context is single-threaded; one context per GPU. There is no CPU
 * backend: without a HIP device sthip_create fails.
 */
#ifndef STHIP_H
#define STHIP_H

#include <stdint.h>
#include "sthip_wire.h"

#ifdef __cplusplus
extern "C" {
#endif

#define STHIP_ABI_VERSION 11

typedef struct sthip_ctx sthip_ctx;

enum sthip_status {
  STHIP_OK = 0,
  STHIP_ERR_INVALID_ARGUMENT = -1,
  STHIP_ERR_NO_DEVICE = -2,
  STHIP_ERR_HIP = -3,
  STHIP_ERR_UNSUPPORTED = -4, /* a flag / scene feature outside the built hot path */
  STHIP_ERR_NO_SCENE = -5
};

typedef struct sthip_image_desc {
  const float* pixels; /* width * height * 4 floats */
  uint32_t width, height;
} sthip_image_desc;

typedef struct sthip_volume_desc {
  const void* data; /* the NanoVDB grid buffer */
  uint64_t bytes;
} sthip_volume_desc;

/* gSceneParams (bdpt.hlsl:19-35) as produced by Scene::update (Scene.cpp:299-684,
 * Scene.hpp:46-69). All pointers are host pointers, borrowed for the call and copied to HBM. */
typedef struct sthip_scene_desc {
  const sthip_PackedVertexData* gVertices; /* Scene.cpp:643-658 */
  uint32_t vertex_count;
  const void* gIndices; /* byte buffer, per-instance stride 2 or 4 (scene.h:139-161) */
  uint32_t indices_bytes;
  const sthip_InstanceData* gInstances; /* Scene.cpp:398-427 */
  uint32_t instance_count;
  const sthip_TransformData* gInstanceTransforms;
  const sthip_TransformData* gInstanceInverseTransforms;
  const sthip_TransformData* gInstanceMotionTransforms; /* may be NULL: identity */
  const void* gMaterialData; /* Material::store records, Material.hpp:32-38 */
  uint32_t material_bytes;
  const uint32_t* gLightInstances; /* Scene.cpp:406-409 */
  uint32_t light_count;
  /* Texture2D<float4> gImages[gImageCount] (bdpt.hlsl:33): what ImageValue::image_index refers to. RGBA32F,
   * row-major, row 0 first. The library builds the mip chain (2x2 box filter) and samples with repeat
   * addressing and trilinear filtering (the reference's gSamplerRepeat, BDPT.cpp:130-133, minus its 8x
   * anisotropy, which hardware-defined filtering cannot be restated). May be NULL / 0. */
  const struct sthip_image_desc* gImages;
  uint32_t image_count;
  /* StructuredBuffer<float> gDistributions (bdpt.hlsl:27): the concatenated distribution tables
   * MaterialResources::get_index(Buffer::View<float>) hands out offsets into (image_value.h:56-66, Scene.cpp:670-683);
   * here the four tables of the environment map (environment.h:17-22, built by dist2.h build_distributions).
   * May be NULL / 0. */
  const float* gDistributions;
  uint32_t distribution_count;
  /* Texture2D<float> gImage1s[] (bdpt.hlsl:34): what Material::alpha_mask refers to (MaterialRecord.alpha_mask_index,
   * Material.hpp:35). One float per texel (the reference stores R8Unorm coverage, Scene.cpp:182), row 0 first. With
   * eAlphaTest a triangle of a material that has a mask is hit only where the mask, sampled bilinearly with repeat
   * addressing at the hit's uv, is >= 0.75 (intersection.hlsli:117-131) — for closest-hit and shadow rays alike.
   * May be NULL / 0. */
  const struct sthip_image_desc* gImage1s;
  uint32_t image1_count;
  /* ByteAddressBuffer gVolumes[] (bdpt.hlsl:35): what InstanceData::volume_index (scene.h:46) and the Medium record's
   * density / albedo volume indices (Material.hpp:80-87) refer to. Each entry is one NanoVDB grid of type float exactly
   * as nanovdb::GridHandle holds it (format version 32.3, the NanoVDB the reference vendors). May be NULL / 0. */
  const struct sthip_volume_desc* gVolumes;
  uint32_t volume_count;
} sthip_scene_desc;

/* gFrameParams view arrays (bdpt.hlsl:37-43), filled by BDPT::render (BDPT.cpp:444-467).
 * gPrevViews / gPrevInverseViewTransforms may be NULL: static camera (previous = current). */
typedef struct sthip_frame_desc {
  const sthip_ViewData* gViews;
  const sthip_TransformData* gViewTransforms;
  const sthip_TransformData* gInverseViewTransforms;
  const sthip_ViewData* gPrevViews;
  const sthip_TransformData* gPrevInverseViewTransforms;
  uint32_t view_count;
  /* gViewMediumInstances (bdpt.hlsl:43, filled at BDPT.cpp:456-466): per view the volume instance the camera is inside
   * of, or 0xFFFF. May be NULL: every camera is outside every medium. */
  const uint32_t* gViewMediumInstances;
} sthip_frame_desc;

/* Output images/buffers of the two passes (bdpt.hlsl:44-49, BDPT.cpp:553-558).
 * gRadiance is required, the others may be NULL. With device_ptrs = 1 every non-NULL
 * pointer is a device pointer on the context's GPU (no copy; results are complete when
 * the call's stream work is complete, see sthip_set_stream). */
#define STHIP_LAYOUT_IMAGE 0u       /* gRadiance is the W x H image; pixels the shard does not own are zero */
#define STHIP_LAYOUT_SHARD_TILES 1u /* gRadiance holds only the shard's tiles, in slot order: sthip_shard_slot_count() float4 */
typedef struct sthip_outputs {
  uint32_t device_ptrs;
  uint32_t radiance_layout; /* STHIP_LAYOUT_*: the packed form is what ranks exchange (sthip_assemble_tiles) */
  float* gRadiance;                  /* RGBA32F, W*H*4: rgb = mean over the seeds of this call, a = sample count */
  float* gAlbedo;                    /* RGBA32F */
  sthip_VisibilityInfo* gVisibility; /* W*H */
  sthip_DepthInfo* gDepth;           /* W*H */
  float* gPrevUVs;                   /* RG32F, W*H*2 */
  uint64_t* gRayCount;               /* [2]: all trace_ray calls / path (non-shadow) rays; intersection.hlsli:66, path.hlsli:1006 */
  /* BDPTDebugMode (bdpt.h:177-193; a specialisation constant of the reference's pipeline, BDPT.cpp:526-541) and the image it
   * feeds, RGBA32F W x H (bdpt.hlsl:45), with gDebugViewPathLength / gDebugLightPathLength of the push constants. gDebugImage
   * is IN / OUT as upstream's is (it persists from frame to frame): ePathLengthContribution and eViewTraceContribution start
   * every pixel's sample at (0,0,0,1), the other modes overwrite a pixel where a path reaches their statement or add to what it
   * holds; pixels outside every view are not touched. The seeds of a call are the reference's successive frames. 0 or a NULL
   * image = off (the image is then neither read nor written). Host or device pointer as the others (device_ptrs). */
  uint32_t debug_mode;
  float* gDebugImage;
} sthip_outputs;

enum {
  STHIP_DEBUG_NONE = 0,
  STHIP_DEBUG_ALBEDO,                     /* = m.albedo() of the first hit */
  STHIP_DEBUG_SPECULAR,                   /* = m.is_specular() */
  STHIP_DEBUG_EMISSION,                   /* = m.Le() */
  STHIP_DEBUG_SHADING_NORMAL,             /* = n * .5 + .5 (after the normal map) */
  STHIP_DEBUG_GEOMETRY_NORMAL,
  STHIP_DEBUG_DIR_OUT,                    /* = the last sampled direction * .5 + .5 */
  STHIP_DEBUG_PREV_UV,                    /* = |prev uv - uv| * extent */
  STHIP_DEBUG_ENVIRONMENT_SAMPLE_TEST,    /* += eight environment samples as spots around the view direction; nothing is traced */
  STHIP_DEBUG_ENVIRONMENT_SAMPLE_PDF,     /* .rgb = the environment's pdf of the view direction; nothing is traced */
  STHIP_DEBUG_RESERVOIR_WEIGHT,           /* += W of the NEE reservoir's sample where it is unoccluded (inline shadow rays only, as upstream) */
  STHIP_DEBUG_PATH_LENGTH_CONTRIBUTION,   /* the unweighted contributions of (gDebugViewPathLength, gDebugLightPathLength) */
  STHIP_DEBUG_LIGHT_TRACE_CONTRIBUTION,   /* light tracing's splats with weight 1 */
  STHIP_DEBUG_VIEW_TRACE_CONTRIBUTION,    /* += the unweighted emission view paths find; light tracing is not added to gRadiance */
  STHIP_DEBUG_MODE_COUNT
};

/* ---- lifetime ---- */
int sthip_abi_version(void);
int sthip_create(int device, sthip_ctx** out_ctx); /* replaces Device/pipeline creation, BDPT.cpp:35-41,151-187 */
void sthip_destroy(sthip_ctx* ctx);
const char* sthip_last_error(const sthip_ctx* ctx); /* ctx may be NULL: error of a failed sthip_create */

/* Stream the kernels are enqueued on (a hipStream_t; NULL = default stream). With host
 * output pointers sthip_render synchronises before returning; with device pointers it only
 * enqueues. */
int sthip_set_stream(sthip_ctx* ctx, void* hip_stream);

/* ---- scene: replaces BLAS/TLAS build + descriptor writes (Scene.cpp:429-509,614-629; BDPT.cpp:341-421) ----
 * Both scene calls first wait for the work already enqueued on the context's stream (frames of the previous scene that
 * were rendered with device output pointers), then replace the resident arrays: a caller never has to synchronise
 * before re-uploading. */
int sthip_scene_upload(sthip_ctx* ctx, const sthip_scene_desc* scene);

/* Instances moved, nothing else changed (Scene::update with cached BLASes, Scene.cpp:435-459,614-629: only the TLAS is
 * rebuilt): new gInstanceTransforms / gInstanceInverseTransforms / gInstanceMotionTransforms (may be NULL: identity) for
 * the instance_count instances of the last sthip_scene_upload. The bottom levels stay in HBM, the top level is rebuilt
 * over the new world boxes. Instances whose transform was the identity at upload are part of one merged world-space
 * mesh: when one of THEM moves the resident tree cannot follow and the call builds the scene again, from the copy of the
 * arrays the context keeps since the upload ("keep_scene" = 1, the default) and the new transforms, with the configured
 * builder ("bvh_builder" = 1: ~10 ms per million triangles on the device) — sthip_stats::full_rebuilds counts these.
 * With "keep_scene" = 0 nothing is kept and such a call returns STHIP_ERR_UNSUPPORTED and changes nothing. */
int sthip_scene_update_transforms(sthip_ctx* ctx, const sthip_TransformData* gInstanceTransforms, const sthip_TransformData* gInstanceInverseTransforms,
                                  const sthip_TransformData* gInstanceMotionTransforms, uint32_t instance_count);

/* ---- frame: replaces the dispatch sequence of BDPT::render (BDPT.cpp:607-720) ----
 * Renders seeds seed_begin .. seed_begin+seed_count-1 (gRandomSeed = seed, BDPT.cpp:480), one
 * sample per pixel centre per seed (bdpt.hlsl:167), and averages them with the running mean of
 * temporal_accumulation.hlsl:118-131. push_constants->gRandomSeed is ignored.
 *
 * sampling_flags are BDPTFlagBits (bdpt.h:12-44), scene_flags BDPT_FLAG_HAS_* (bdpt.h:46-49) as BDPT::render
 * resolves them (BDPT.cpp:486-541). Built: the default view-path integrator and, per flag, eAlphaTest, eNormalMaps,
 * eRayCones, eFlip*, eShadingNormalShadowFix, eUniformSphereSampling, eSampleEnvironmentMapDirectly, ePresampleLights,
 * eNEEReservoirs, eConnectToViews (sample_photons + add_light_trace), eConnectToLightPaths, and BDPT_FLAG_HAS_MEDIA
 * (volume instances over gVolumes). The flags whose upstream result depends on the order threads run in are built
 * with ONE defined order each: eLVC / eLVCReservoirs (cache filled in light-path
 * index order), eNEEReservoirReuse / eLVCReservoirReuse (hash-grid appends in (path, vertex) order; the seeds of a call
 * are then traced one at a time, seed s reading the grid of seed s - 1; not on a pixel-tile shard), eCoherentRR and
 * eCoherentSampling (the wave = the 8x4 pixel group, its first lane = the lowest lane that executes the statement).
 * Rejected with STHIP_ERR_UNSUPPORTED, never ignored: eSampleLightPower (reads an uninitialised table upstream) and the
 * combinations DESIGN.md section 5 lists (media exclude eCoherentSampling; light
 * subpaths and reuse exclude environments). Media without eDeferShadowRays are traced with k_shade walking every NEE ray itself.
 * ePerformanceCounters does not change results here; eRemapThreads only through the path index
 * (map_pixel_coord, bdpt_util.hlsli:76-83) that ePresampleLights and the light subpaths key on. */
int sthip_render(sthip_ctx* ctx, const sthip_BDPTPushConstants* push_constants, uint32_t sampling_flags,
                 uint32_t scene_flags, const sthip_frame_desc* frame, uint32_t seed_begin, uint32_t seed_count,
                 const sthip_outputs* outputs);

/* Pixel-tile sharding for multi-GPU (one context per GPU): the frame is cut into
 * tile_w x tile_h tiles (multiples of the reference's 8x4 workgroup, bdpt.hlsl:11-12),
 * tile t is rendered iff t % shard_count == shard_rank; other pixels are written as zero
 * (including alpha) so that a sum-reduce over ranks assembles the frame.
 * shard_count = 1 (default) renders everything. */
int sthip_set_shard(sthip_ctx* ctx, uint32_t shard_rank, uint32_t shard_count, uint32_t tile_w, uint32_t tile_h);

/* The exchange step of a sharded frame without the zero padding: each rank renders with
 * outputs.radiance_layout = STHIP_LAYOUT_SHARD_TILES (sthip_shard_slot_count() float4 entries, 1/shard_count of the
 * frame), the ranks' buffers are gathered on one GPU (RCCL gather / all_gather, rank r at packed + r * rank_stride
 * float4 entries; rank_stride >= rank 0's slot count, which is the largest), and sthip_assemble_tiles scatters them into
 * the W x H image on that GPU's context (device pointers, enqueued on the context's stream). */
uint32_t sthip_shard_slot_count(uint32_t width, uint32_t height, uint32_t shard_rank, uint32_t shard_count, uint32_t tile_w,
                                uint32_t tile_h);
int sthip_assemble_tiles(sthip_ctx* ctx, const float* packed, uint64_t rank_stride, uint32_t shard_count, uint32_t tile_w,
                         uint32_t tile_h, uint32_t width, uint32_t height, float* frame);
/* The same for the other outputs of a sharded frame (the G-buffer: albedo 16 B, VisibilityInfo 8 B, DepthInfo 16 B, prev-uv
 * 8 B per pixel), which sthip_render writes as W x H images that are zero outside the shard's tiles:
 * sthip_pack_tiles gathers the context's own tiles (its sthip_set_shard) out of such an image into slot order —
 * sthip_shard_slot_count() entries of entry_bytes each, the padding slots of edge tiles zero — and
 * sthip_assemble_tiles_bytes scatters the gathered buffers of all ranks (rank r at packed + r * rank_stride entries) into the
 * image. entry_bytes: a multiple of 4, at most 64. Device pointers; enqueued on the context's stream. */
int sthip_pack_tiles(sthip_ctx* ctx, const void* image, uint32_t width, uint32_t height, uint32_t entry_bytes, void* packed);
int sthip_assemble_tiles_bytes(sthip_ctx* ctx, const void* packed, uint64_t rank_stride, uint32_t shard_count, uint32_t tile_w,
                               uint32_t tile_h, uint32_t width, uint32_t height, uint32_t entry_bytes, void* frame);

/* The other way to spread a render call over GPUs: every GPU renders the WHOLE frame for its own part of the
 * seed range (sthip_set_shard(ctx, 0, 1, ..)), and the images are added up — one ncclReduce(sum) of the accumulation buffer.
 * It serves the estimators that build whole-frame structures (light tracing's splats, the reservoir-reuse hash grids), which
 * a tile shard cannot; the mean of N seeds then depends on the order of a floating-point sum (~1e-7 relative against the
 * one-GPU frame, where a tile shard is bit-identical). A call's radiance output is (mean over its seeds, their number):
 * sthip_radiance_to_sums turns `entries` RGBA32F entries in place into (sum over the seeds, their number), which a
 * sum-reduce can add; with back != 0 it turns such sums back into (mean, number). Device pointer; enqueued on the stream. */
int sthip_radiance_to_sums(sthip_ctx* ctx, float* image, uint64_t entries, uint32_t back);

/* ---- the traversal contract on its own (T1/T2 of SURVEY.md §8a; intersection.hlsli:65-239) ---- */
typedef struct sthip_ray {
  float origin[3];
  float tmin;
  float direction[3];
  float tmax;
} sthip_ray;

typedef struct sthip_hit {
  float t; /* tmax of the ray on a miss */
  float b1, b2; /* barycentrics: P = v0 + b1 (v1-v0) + b2 (v2-v0), shading_data.hlsli:69-72 */
  uint32_t instance_primitive_index; /* 0xFFFFFFFF on a miss */
} sthip_hit;

/* any_hit bit 0 = 0: closest hit (trace_ray); 1: occlusion (RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH):
 * hits[i].instance_primitive_index is 0 when occluded, 0xFFFFFFFF when not; t,b1,b2 unspecified.
 * bit 1: alpha test on (gAlphaTest), bit 2: gFlipTriangleUVs for the mask lookup.
 * rays/hits are host pointers unless device_ptrs != 0. */
int sthip_trace_rays(sthip_ctx* ctx, const sthip_ray* rays, uint32_t ray_count, sthip_hit* hits, uint32_t any_hit,
                     uint32_t device_ptrs);

/* ---- measurement ---- */
typedef struct sthip_stats {
  uint64_t rays_total;     /* gRayCount[0] semantics, last render */
  uint64_t rays_path;      /* gRayCount[1] semantics */
  uint64_t rays_shadow;
  uint64_t nodes_visited;  /* closest-hit rays; only when collected (sthip_set_option "count_traversal") */
  uint64_t tris_tested;
  uint64_t nodes_visited_shadow; /* any-hit (shadow) rays */
  uint64_t tris_tested_shadow;
  float ms_trace;          /* hipEvent time spent in the traversal kernel (k_trace) during the last render ("time_kernels") */
  float ms_shade;
  float ms_total;          /* all kernels of the last render */
  uint32_t launches_trace; /* k_trace launches of the last render */
  uint32_t bvh_node_bytes; /* bytes of one BVH node as laid out in HBM */
  uint32_t bvh_tri_bytes;  /* bytes of one leaf triangle */
  uint64_t bvh_nodes;
  uint64_t bvh_tris;
  float bvh_build_ms;      /* wall time of the acceleration-structure build inside the last sthip_scene_upload */
  float bvh_build_gpu_ms;  /* of which device time of the GPU builder's kernels ("bvh_builder" = 1) */
  /* lane-occupancy diagnostics of the trace kernels, [0] closest-hit, [1] shadow rays (with "count_traversal"):
   * 64 per wave-level iteration of the node loop / the triangle loop / per scheduling round of a persistent wave, and
   * the number of lanes that held a ray summed over rounds: nodes_visited / inner_slots etc. are lane utilisations */
  uint64_t inner_slots[2];
  uint64_t tri_slots[2];
  uint64_t round_slots[2];
  uint64_t busy_rounds[2];
  /* the first bounce runs as wave packets in a kernel of its own (k_trace_primary, "packet_primary" = 1): its share of
   * rays_path / nodes_visited / tris_tested, its launches and its time, all of which are NOT part of ms_trace /
   * launches_trace (so those describe k_trace alone) */
  uint64_t rays_primary_packets;
  uint64_t nodes_visited_primary;
  uint64_t tris_tested_primary;
  float ms_trace_primary;
  uint32_t launches_primary;
  /* where the lanes of the persistent trace kernel are (with "count_traversal"; the 4-wide walk): summed over the wave-level
   * iterations of the node loop, the lanes [0] taking a node step, [1] waiting with a triangle leaf, [2] waiting with a
   * sentinel or an instance entry, [3] without a ray; [4] 64 per leaf phase, the lanes that [5] test a triangle and
   * [6] handle a sentinel or an entry in it; [7] 64 per refill stop of a wave */
  uint64_t lane_states[8];
  /* of rays_total / rays_path: last rays of paths — the path or diffuse budget ends at their far end, so they could only
   * still find an emitter — that missed the bounds of every emissive instance and were answered without a traversal
   * ("answer_last_rays" = 1, the default; scenes without images, spheres, environment or media). They are trace_ray calls of
   * the reference and are counted as such; nodes_visited etc. hold no visits for them */
  uint64_t rays_answered;
  /* the batch of the last render: paths of one seed on this shard, seeds traced together, the cap in force ("max_paths_in_flight":
   * sized at sthip_create from the device's FREE memory) and how often sthip_render has halved it because the device could not
   * hold a batch's state (a render retries with half the batch instead of failing with out-of-memory) */
  uint32_t paths_per_seed;
  uint32_t seeds_in_flight;
  uint64_t max_paths_in_flight;
  uint32_t batch_halvings;
  uint32_t full_rebuilds; /* sthip_scene_update_transforms calls that had to build the scene again (an instance of the merged mesh moved) */
} sthip_stats;
int sthip_get_stats(sthip_ctx* ctx, sthip_stats* out);

/* Measured memory-system ceilings for the roofline of the traversal kernel (bench.py): the rate, in GB/s, of
 *   STHIP_CEILING_TRIAD             a float4 stream triad over 3 x 512 MiB (the measured HBM figure SURVEY.md 8d asks for)
 *   STHIP_CEILING_NODE_GATHER_TABLE the traversal's own node fetch (64-byte BVH node per lane, 4 vector loads) at
 *                                   uniformly random nodes of the resident acceleration structure, with no dependence
 *                                   between fetches and no arithmetic: what L2 + Infinity Cache deliver to such gathers
 *   STHIP_CEILING_NODE_GATHER_L2    the same over a 2 MiB prefix of the node array (L2-hit rate)
 *   STHIP_CEILING_NODE_GATHER_L1    the same over a 16 KiB prefix (the vector-memory front end: nothing beats it)
 * counting 64 bytes per node fetch, as the algorithmic figure does. Synchronous; the gather kinds need a scene. */
enum { STHIP_CEILING_TRIAD = 0, STHIP_CEILING_NODE_GATHER_TABLE = 1, STHIP_CEILING_NODE_GATHER_L2 = 2, STHIP_CEILING_NODE_GATHER_L1 = 3 };
int sthip_measure_ceiling(sthip_ctx* ctx, uint32_t kind, double* gbytes_per_s);

/* named integer options: "count_traversal" (0/1), "time_kernels" (0/1); scheduler tuning of the persistent
 * trace kernels: "refill_idle" (1..64, default 16), "inner_min_lanes" (1..64, default 24),
 * "max_paths_in_flight" (how many seeds of the owned pixels are traced together; ~330 B of device memory per path at the default
 * flags; default: the largest power of two that leaves 4 KB of device memory per path, 2^26 on a 288 GB MI355X);
 * "bvh_builder": 0 = binned SAH on the host (default), 1 = the device-resident GPU builder (set before
 * sthip_scene_upload; a failed upload with it leaves the context without a scene), with "lbvh_algorithm" 1 = PLOC
 * (default) or 0 = Karras radix tree, "ploc_radius" (1..32, default 4) and "sah_top" (default 64: subtrees of at most that
 * many triangles keep their PLOC shape under a host-built SAH top; 0 = plain PLOC);
 * "wide_bvh" (the persistent trace kernel walks the tree collapsed into 4-wide nodes of 64 bytes with 8-bit child planes:
 * 1 = always (default) — host-built trees are collapsed on the host, the GPU builder's trees and the tree of a
 * transforms-only update on the device; 0 = never; 2 = only when the binary nodes exceed 4 MiB, one XCD's L2; 3 = the 8-wide
 * compressed form (80-byte nodes whose children are addressed by a base and a mask, 64-bit group stack) for host-built
 * trees, the 4-wide form for the others; not with "treetop" or "embed_leaves"), "tri_min_lanes" (1..64, default 1: the
 * 8-wide walk's leaf phase goes on while at least this many lanes hold a triangle), "hashgrid_serial" (0/1: build the reservoir-reuse hash grids with the one-thread serial probe
 * sequence instead of the parallel device build: the same grids, for tests), "reuse_grids_persist" (default 0: every call starts a
 * new chain, its first seed finds no grid — upstream's first frame, gReservoirSpatialM = 0, BDPT.cpp:482-483. 1: the grids the
 * last seed of a call leaves are what the first seed of the NEXT call looks into, as long as that call asks for the same reuse
 * flags, gHashGridBucketCount, extent and gMaxDiffuseVertices — a host that renders one frame per call; N calls of one seed are
 * then one call of N seeds. Setting the option, to either value, also drops the kept grids (upstream: a frame whose camera moved
 * without reprojection), and so does sthip_scene_upload), "cull_terminal" (default 1: in a round where the
 * path or diffuse budget can end, only the paths that still have something to do reach the shading kernel), "answer_last_rays"
 * (default 1: a path's last ray is traced only if it can reach the bounds of an emissive instance; sthip_stats::rays_answered;
 * identical results for rays that start within a few scene sizes of the scene — as every path ray does — which is also what
 * the hit contract itself needs),
 * "keep_scene" (default 1: a host copy of the uploaded arrays, see sthip_scene_update_transforms), "treetop" (default 0), "embed_leaves" (default 0), "lds_materials" (default 1), "lds_stack_levels" (4..150: LDS levels of the traversal stack; a higher
 * tree runs the bounded kernels, default: bounded at 32 levels beyond a height of 40): layout / scheduling options that
 * never change results, read at the next sthip_scene_upload / sthip_scene_update_transforms */
int sthip_set_option(sthip_ctx* ctx, const char* name, int64_t value);

/* ---- after the path: display transform, image metric, HDR export ---- */

/* Temporal accumulation of the renderer's output over frames, what Denoiser::denoise dispatches first
 * (src/Node/Denoiser.cpp:176-213 -> kernels/temporal_accumulation.hlsl:59-145): with `reprojection` the previous
 * frame's accumulated colour and luminance moments are fetched at gPrevUVs with a bilinear footprint whose taps must
 * pass the instance / normal / depth tests (:74-97), without it the same pixel is used (:102-109); then the new sample
 * is blended in with alpha = n_new / n, n clamped by history_limit (gHistoryLimit, 0 = unlimited). The binding names
 * are the shader's (denoiser.h). gViews is always a host pointer; the images are host pointers unless device_ptrs
 * (then, with up to 4 views, the call only enqueues its kernel on the context's stream: nothing is allocated or waited for).
 * (sthip_render's own N-seed mean is the same-pixel branch of this kernel applied seed by seed.) */
typedef struct sthip_accumulate_desc {
  uint32_t width, height;
  uint32_t view_count;        /* gViewCount */
  uint32_t reprojection;      /* specialisation constant gReprojection (Denoiser.cpp:76: on by default) */
  uint32_t demodulate_albedo; /* gDemodulateAlbedo: gRadiance.rgb /= 1e-2 + gAlbedo.rgb */
  float history_limit;        /* gHistoryLimit */
  uint32_t device_ptrs;
  uint32_t instance_count;    /* entries of gInstanceIndexMap */
  const sthip_ViewData* gViews;
  const float* gRadiance;                      /* RGBA32F: rgb = sample, a = its sample count */
  const float* gAlbedo;                        /* RGBA32F; may be NULL unless demodulate_albedo */
  const sthip_VisibilityInfo* gVisibility;     /* these five only with reprojection */
  const sthip_DepthInfo* gDepth;
  const float* gPrevUVs;                       /* RG32F */
  const sthip_VisibilityInfo* gPrevVisibility;
  const sthip_DepthInfo* gPrevDepth;
  const float* gPrevAccumColor;                /* RGBA32F: rgb = mean so far, a = sample count */
  const float* gPrevAccumMoments;              /* RG32F: mean luminance, mean squared luminance */
  const uint32_t* gInstanceIndexMap;           /* SceneData::mInstanceIndexMap (Scene.cpp:383-385,414-418); NULL = identity */
  float* gAccumColor;                          /* out, RGBA32F */
  float* gAccumMoments;                        /* out, RG32F */
} sthip_accumulate_desc;
int sthip_accumulate(sthip_ctx* ctx, const sthip_accumulate_desc* desc);

/* TonemapMode, src/Shaders/tonemap.h:8-21 */
enum {
  STHIP_TONEMAP_RAW = 0,
  STHIP_TONEMAP_REINHARD,
  STHIP_TONEMAP_REINHARD_EXTENDED,
  STHIP_TONEMAP_REINHARD_LUMINANCE,
  STHIP_TONEMAP_REINHARD_LUMINANCE_EXTENDED,
  STHIP_TONEMAP_UNCHARTED2,
  STHIP_TONEMAP_FILMIC,
  STHIP_TONEMAP_ACES,
  STHIP_TONEMAP_ACES_APPROX,
  STHIP_TONEMAP_VIRIDIS_R,
  STHIP_TONEMAP_VIRIDIS_LENGTH_RGB,
  STHIP_TONEMAP_MODE_COUNT
};

/* What BDPT::render binds and pushes for its "tone map" block (src/Node/BDPT.cpp:783-815, kernels/tonemap.hlsl:7-19):
 * specialisation constants gMode / gModulateAlbedo / gGammaCorrection, push constant gExposure, images gInput,
 * gAlbedo, gOutput (RGBA32F, width*height). The two dispatches of the reference (clear gMax + reduce_max, then main)
 * happen inside one call. gExposureAlpha (smoothing of the maxima over frames, default 0 = off) is not taken: every
 * call uses the maxima of its own input. out_max (optional, host memory, 4 floats) receives the rgb and luminance
 * maxima main() sees. */
typedef struct sthip_tonemap_desc {
  uint32_t width, height;
  uint32_t mode;
  uint32_t modulate_albedo;
  uint32_t gamma_correction;
  float exposure;
  uint32_t device_ptrs; /* gInput/gAlbedo/gOutput are device pointers */
  float exposure_alpha; /* gExposureAlpha (tonemap.hlsl:169-178): in (0, 1) the maxima the curves use are blended with the
                           previous frame's, lerp(prev, cur, alpha) (the luminance moments with sqrt(alpha)); needs exposure_state */
  const float* gInput;
  const float* gAlbedo; /* may be NULL when modulate_albedo == 0 */
  float* gOutput;
  float* out_max;
  /* host, in/out, may be NULL: what the reference keeps at bytes 16..39 of gMax / reads from gPrevMax — the (blended)
   * maxima r, g, b, luminance and the two luminance moments of the previous call; all zero before the first frame */
  float* exposure_state;
} sthip_tonemap_desc;
int sthip_tonemap(sthip_ctx* ctx, const sthip_tonemap_desc* desc);

/* ImageCompareMode, kernels/image_compare.hlsl:5-9 */
enum { STHIP_COMPARE_SMAPE = 0, STHIP_COMPARE_MSE = 1, STHIP_COMPARE_AVERAGE = 2 };

/* The metric ImageComparer shows (src/Node/ImageComparer.cpp:61-90 dispatching kernels/image_compare.hlsl:13-46):
 * per-pixel error over rgb, divided by 3*width*height, summed per group of 64 consecutive pixels, scaled by
 * `quantization` (gQuantization; the node's default is 1024, ImageComparer.hpp:18), truncated to uint and added atomically.
 * sum_out = the raw uint accumulator, overflow_out = the overflow flag; the displayed number is sum/quantization
 * (its square root for MSE, ImageComparer.cpp:88). image1/image2: RGBA32F, host unless device_ptrs. */
int sthip_image_compare(sthip_ctx* ctx, const float* image1, const float* image2, uint32_t width, uint32_t height, uint32_t metric, uint32_t quantization,
                        uint32_t device_ptrs, uint32_t* sum_out, uint32_t* overflow_out);

/* Radiance .hdr export of an RGBA32F host image, what BDPT's "Export HDR" does through stbi_write_hdr(path, w, h, 4,
 * pixels) (src/Node/BDPT.cpp:313-337): header "#?RADIANCE", FORMAT=32-bit_rle_rgbe, rows top to bottom, RGBE with
 * the shared exponent of the largest channel; rows of 8..32767 pixels are run-length coded per channel, others flat.
 * Host-only, needs no context. Returns STHIP_OK or STHIP_ERR_INVALID_ARGUMENT (bad arguments / cannot write). */
int sthip_write_hdr(const char* path, uint32_t width, uint32_t height, const float* rgba);

#ifdef __cplusplus
}
#endif
#endif /* STHIP_H */
