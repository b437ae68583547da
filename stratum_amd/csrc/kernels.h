// kernels.h — the wavefront path-tracing kernels that replace the reference's two compute passes
// sample_visibility (bdpt.hlsl:149-300) and trace_shadows (bdpt.hlsl:302-326).
//
// The reference runs one megakernel thread per pixel whose loop alternates next_vertex() and trace()
// (bdpt.hlsl:298-299). Here the same per-path state machine is cut at every trace() into stages that
// run as separate launches over compacted queues, so that the divergent BVH traversal runs with full
// waves and the shading stage with coherent memory access:
//     generate -> [ trace (closest hits of bounce d + shadow rays of bounce d - 1) -> shade ] x bounces -> resolve
// (with the section 8f estimators: presample_lights / a light pass of the same shape in front, k_shadow_media between
// trace and shade for scenes with volume instances)
// Results do not depend on scheduling: the RNG is counter based per (pixel, seed) (rng.hlsli:35-47),
// every pixel owns its outputs, and per-pixel sums keep the reference's order (emission in path
// order, then the deferred shadow-ray sum i = 1..gMaxDiffuseVertices, bdpt.hlsl:311-325).
#pragma once

#include "shading.h"
#include "hashgrid.h"
#include "medium.h"
#include "traverse.h"
#include <type_traits>

#define STHIP_BLOCK 256

// counter slots in FrameParams::counters: they accumulate over a render call (queue sizes live in FrameParams::qctl)
enum {
  CNT_RAYS_CLOSEST = 0,
  CNT_RAYS_SHADOW = 1,
  CNT_NODES = 2,  // +1: shadow rays
  CNT_TRIS = 4,   // +1: shadow rays
  CNT_INNER_SLOTS = 6,  // +1: shadow rays
  CNT_TRI_SLOTS = 8,    // +1: shadow rays
  CNT_ROUND_SLOTS = 10, // +1: shadow rays: 64 per round of a persistent wave
  CNT_BUSY_ROUNDS = 12, // +1: shadow rays: lanes holding a ray, summed over rounds
  CNT_NODES_PRIMARY = 14,  // the share of CNT_NODES / CNT_TRIS that k_trace_primary (first bounce as wave packets) counted
  CNT_TRIS_PRIMARY = 15,
  CNT_CROSSINGS = 16,  // media: closest-hit queries that only carried a path across a volume boundary (no new trace() call)
  CNT_RAYS_ANSWERED = 17,  // of CNT_RAYS_CLOSEST: last rays of paths that k_shade answered from the emitters' bounds (never queued)
  CNT_LANE_STATES = 20,  // 8 slots: TraverseCounters::st
  CNT_TOTAL = 28
};

struct FrameParams {
  DeviceScene scene;
  DeviceBvh bvh;
  sthip_BDPTPushConstants pc;
  uint32_t sampling_flags;
  uint32_t scene_flags;      // BDPT_FLAG_HAS_* after the host-side resolution (BDPT.cpp:486-496)
  uint32_t seed;             // gRandomSeed of the first seed in flight
  uint32_t seeds_in_flight;  // seeds traced together in one pass: slot = seed_index * paths_per_seed + pixel slot
  uint32_t paths_per_seed;   // owned tiles * tile_w * tile_h
  // views (device copies)
  const sthip_ViewData* views;
  const sthip_TransformData* view_xf;
  const sthip_ViewData* prev_views;
  const sthip_TransformData* prev_inv_view_xf;
  // sharding: tile t is owned iff t % shard_count == shard_rank
  uint32_t shard_rank, shard_count, tile_w, tile_h, tiles_x, tiles_y;
  uint32_t path_count;  // seeds_in_flight * paths_per_seed
  // path state, indexed by path slot
  float4* ray_o;       // xyz origin, w = bsdf_pdf
  float4* ray_d;       // xyz direction, w = eta_scale
  float4* hit;         // t, b1, b2, bits(instance_primitive_index)
  // BDPTDebugMode (sthip.h: STHIP_DEBUG_*; 0 = off): the pixel of gDebugImage as the path has left it so far, per path slot
  // (k_generate reads it from out_debug or clears it, the DEBUG instantiation of k_shade and the shadow rays update it, k_resolve
  // writes it back), and — inline shadow rays only — what an unoccluded shadow ray adds to it, beside the ray's record
  uint32_t debug_mode;
  float4* debug;
  float4* out_debug;
  float4* shadow_debug;
  uint32_t* hit_leaf;  // the hit triangle's index in the leaf-triangle array (RayHit::leaf): where k_shade reads its vertices
  float4* beta;        // xyz beta, w = bits(rng counter)
  uint32_t* meta;      // path_length | diffuse_vertices << 8
  float4* radiance;    // gRadiance[px].rgb of the seed in flight
  float4* shadow_sum;  // the sum `c` of trace_shadows
  float2* cone;        // RayDifferential (radius, spread), path.hlsli:224-244; only allocated for textured scenes
  float4* accum;       // running mean over seeds (temporal_accumulation.hlsl:118-131): rgb, n
  // queues
  uint32_t* queue[2];
  // light tracing (eConnectToViews)
  float4* bdpt;               // per path: path_pdf, path_pdf_rev, dVC, prev_cos_out (path.hlsli:262-267); prev_specular = meta bit 16
  uint32_t* light_trace;      // gLightTraceSamples: per seed in flight, uint4 per pixel = quantised rgb sums + overflow bits
  const sthip_TransformData* inv_view_xf;
  uint32_t light_pass;        // this pass traces light subpaths (sample_photons)
  uint32_t light_threads;     // threads of sample_photons' padded dispatch per seed
  uint32_t light_trace_quantization;
  uint32_t light_trace_empty;  // eConnectToViews with gMaxPathVertices <= 2: add_light_trace still runs (BDPT.cpp:753), over samples nobody wrote this frame — pinned to zero, as the oracle
  // light-subpath connections (eConnectToLightPaths)
  float4* light_vertices;     // gLightPathVertices: per seed in flight gLightPathCount * gMaxDiffuseVertices PathVertex records of 4 x float4 (bdpt.h:108-121)
  // the light vertex cache (eLVC, path.hlsli:523-527,683-800). Upstream hands out cache slots with an atomic counter; the
  // order defined here is light paths by path index, a path's vertices as it stores them (one order upstream's scheduler
  // may produce): k_shade_light stages a vertex at [path_index * (gMaxDiffuseVertices - 1) + diffuse_vertices - 1], a
  // scan compacts the stage into light_vertices (api.hip), lvc_count[seed in flight] = gLightPathVertexCount[0]
  float4* lvc_staging;
  const uint32_t* lvc_count;
  float4* path_contrib;       // per light path: path_contrib (path.hlsli:258,901,1043), what eLVCReservoirs store instead of beta
  // eNEEReservoirReuse (hashgrid.h): the previous seed's grid (hg_prev = 0 for the first seed of a call: nothing to look
  // up, BDPT.cpp:482-483) and the stage for this seed's appends, 4 x float4 per (path index, diffuse vertex):
  // (position, r.total_weight) (bits(r.M), bits(packed_geometry_normal), W, y.pdfA) (y.position, bits(cell_size)) (y.Le, bits(y.packed_geometry_normal))
  const uint32_t* hg_checksums;
  const uint32_t* hg_counters;
  const uint32_t* hg_indices;
  const float4* hg_data;      // NEEReservoir (bdpt.h:157-165), 3 x float4: (r.total_weight, bits(r.M), bits(packed_geometry_normal), W) (y.position, bits(y.packed_geometry_normal)) (y.Le, y.pdfA)
  uint32_t hg_prev;
  float4* hg_appends;
  // eLVCReservoirReuse: the same for connect_lvc's reservoirs. Stage: 6 x float4 per (path index, diffuse vertex):
  // (position, cell_size) (r.total_weight, bits(r.M), bits(packed_geometry_normal), W) + the PathVertex; grid data:
  // PathVertexReservoir (bdpt.h:166-174), 5 x float4: (r.total_weight, bits(r.M), bits(packed_geometry_normal), W) + the PathVertex
  const uint32_t* lg_checksums;
  const uint32_t* lg_counters;
  const uint32_t* lg_indices;
  const float4* lg_data;
  float4* lg_appends;
  // eCoherentRR (the reference's default, BDPT.cpp:58; path.hlsli:829-845): the survival probability of a Russian
  // roulette is WaveActiveMax(p) and the verdict WaveReadLaneFirst(rnd > p) over the lanes of the 8x4 workgroup that
  // execute the call in that iteration of the vertex loop. The paths of a group sit in 32 consecutive slots (rows 0-3
  // or 4-7 of a wave's 8x8 pixel block), but k_shade runs over compacted queues, so a round in which the roulette can
  // run is cut in two: k_shade<PROBE> stores (p, the random number the path would draw, 1, 0) here for every path that
  // reaches the call, k_rr_reduce turns each group's entries into (p_max, verdict, 1, 0), and k_shade proper reads
  // them. Null without the flag, with media (walks through volumes break the lockstep: the non-coherent form stays)
  // and in rounds in which no path can reach the call.
  float4* rr;
  // eCoherentSampling (path.hlsli:317,379-387,688,703): index = WaveReadLaneFirst(index) + WaveGetLaneIndex() over the 8x4
  // group, two sites per vertex (the NEE index of a presampled light; connect_lvc's). Same cut as for the roulette, one
  // probe per site because the draws between the sites depend on the first one's outcome: k_shade<PROBE> with probe_kind 2
  // stores (1, own draw) at the NEE site, k_cs_reduce turns a group's entries into (any, first lane's draw), probe_kind 3
  // runs on with that value up to connect_lvc's draw; the round proper reads both. Null without the flag.
  uint2* cs_nee;
  uint2* cs_lvc;
  uint32_t lds_material_bytes;  // bytes of gMaterialData a k_shade block stages in LDS (0: none)
  uint32_t probe_kind;  // of a k_shade<PROBE> launch: 1 = the roulette, 2 = the NEE index, 3 = connect_lvc's index
  // bounded LDS stacks (k_trace<., ., true>): the rays k_trace_deep traces again, 4 x float4 each, and their count
  float4* deep_rays;
  uint32_t* deep_count;
  float4* conn;               // per view path gMaxDiffuseVertices - 1 pending connection contributions of the last vertex shaded
  uint32_t shadow_stride;     // entries between the segments of shadow_rays (a vertex may queue gMaxDiffuseVertices records)
  // participating media (BDPT_FLAG_HAS_MEDIA): see the MEDIA instantiation of k_shade and k_shadow_media
  uint32_t media;
  uint32_t shadow_alt;        // entries: the shadow records of odd rounds start here in shadow_rays / shadow_ext (ping-pong)
  float4* media_state;        // per path 2 x float4: (vertex origin xyz, T_dir_pdf) (T_nee_pdf, bits(medium), bits(segments), 0)
  float4* shadow_hit;         // per shadow record of the round in flight: its closest hit (t, b1, b2, bits(ip))
  float4* shadow_ext;         // per shadow record: (bits(rng counter), nee_pdf, bits(result entry), 0)
  float4* shadow_result;      // per path gMaxDiffuseVertices entries: what the NEE ray of diffuse vertex i adds (trace_shadows' c += ...)
  const uint32_t* view_medium;  // gViewMediumInstances
  // media without eDeferShadowRays: an NEE ray draws from the path's own stream in the middle of its vertex (path.hlsli:329-332),
  // so k_shade walks it itself, then and there (visibility_walk_media); one full-height stack column per k_shade thread
  uint32_t inline_media;
  uint32_t* shade_stack;
  float4* presampled;   // gPresampledLights (ePresampleLights): per seed in flight, 2 x float4 per point: (position, bits(packed normal)) (Le, pdfA)
  float4* shadow_rays;  // 3 x float4 per entry: (origin, distance) (direction, bits(slot)) (contribution, 0)
  unsigned long long* counters;  // CNT_* (64-bit each)
  unsigned long long* qctl;      // queue control lines, queue_ctl(): [path | shadow][depth < 64][QUEUE_SEGMENTS] x 128 B
  uint32_t seg_stride;           // entries between the segments of queue[]
  uint32_t rounds;               // bounce rounds of this render (<= 63)
  // outputs (device pointers; may be null)
  float4* out_radiance;
  float4* out_albedo;
  sthip_VisibilityInfo* out_visibility;
  sthip_DepthInfo* out_depth;
  float2* out_prev_uv;
  uint32_t write_aov;
  uint32_t out_packed;       // out_radiance holds this shard's tiles in slot order instead of the W x H image
  uint32_t count_traversal;
  uint32_t refill_idle;      // persistent trace kernels: refill when this many lanes of a wave are idle
  uint32_t inner_min_lanes;  // leave the inner-node loop when fewer lanes than this are still walking
  uint32_t culled;           // k_shade: queue[depth & 1] is what k_cull_terminal kept of the round's queue (sizes in QCTL_KEPT)
  const uint8_t* inst_flags; // per instance: INST_FLAG_* of its material (scenes without images), for k_cull_terminal
  const struct EmitterBounds* emitters;  // the emissive triangle instances' bounds (aims_at_emitter), emitter_count of them;
  uint32_t emitter_count;                // 0: the last-ray filter of k_shade is off
  uint32_t no_specular;                  // no material of the scene is specular: the diffuse budget ends every path that reaches it
};
// One emissive triangle instance: the box of its vertices in the space the traversal tests its triangles in (world space for an
// instance with identity transforms, the instance's object space otherwise) and a sphere around it that sizes the per-ray
// padding of the slab test exactly as the traversal's own boxes are padded (setup_space).
struct EmitterBounds {
  float lo[3];
  uint32_t instance;
  float hi[3];
  uint32_t identity;
  float sphere[4];
};
#define STHIP_MAX_EMITTER_BOUNDS 16u
// DisneyMaterial::Le() > 0 somewhere / can_eval() / is_specular() of an instance's untextured material, evaluated on the host with
// the device's arithmetic at upload (api.hip); KEEP: not a triangle instance, no statement made
#define INST_FLAG_EMITS 1u
#define INST_FLAG_CAN_EVAL 2u
#define INST_FLAG_SPECULAR 4u
#define INST_FLAG_KEEP 0x80u

DEV bool flag(const FrameParams& p, int bit) { return (p.sampling_flags >> bit) & 1u; }

// one global atomic per wave (and none when the wave's sum is zero): per-thread atomics on one
// address serialise at the memory side at ~12 ns each
DEV void wave_add(unsigned long long* counter, uint32_t v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63u) == 0 && v) atomicAdd(counter, (unsigned long long)v);
}

// path slot -> pixel. Slots enumerate the owned tiles, and inside a tile 8x8 pixel blocks, so that a
// wave64 covers an 8x8 block of the image (the reference's eRemapThreads does the same with 8x4 groups
// of 32 threads, bdpt_util.hlsli:76-83). Returns false for slots that fall outside the image.
DEV bool slot_to_pixel(const FrameParams& p, uint32_t slot, uint32_t& px, uint32_t& py) {
  slot %= p.paths_per_seed;  // several seeds of the same pixel set may be in flight
  const uint32_t per_tile = p.tile_w * p.tile_h;
  const uint32_t local_tile = slot / per_tile;
  const uint32_t r = slot - local_tile * per_tile;
  const uint32_t tile = local_tile * p.shard_count + p.shard_rank;
  const uint32_t ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
  const uint32_t blocks_x = p.tile_w >> 3;
  const uint32_t b = r >> 6, lane = r & 63u;
  const uint32_t by = b / blocks_x, bx = b - by * blocks_x;
  px = tx * p.tile_w + (bx << 3) + (lane & 7u);
  py = ty * p.tile_h + (by << 3) + (lane >> 3);
  return px < p.pc.gOutputExtent[0] && py < p.pc.gOutputExtent[1] && tile < p.tiles_x * p.tiles_y;
}

// scene.h:132-137
DEV int get_view_index(const FrameParams& p, uint32_t x, uint32_t y) {
  for (uint32_t i = 0; i < p.pc.gViewCount; i++) {
    const sthip_ViewData& v = p.views[i];
    if ((int)x >= v.image_min[0] && (int)y >= v.image_min[1] && (int)x < v.image_max[0] && (int)y < v.image_max[1]) return (int)i;
  }
  return -1;
}
// transform.h:136-147
DEV f3 back_project(const sthip_ProjectionData& pr, float cx, float cy) {
  f3 r;
  if (pr.vertical_fov < 0) {
    r.x = (cx - pr.offset[0]) / pr.scale[0];
    r.y = (cy - pr.offset[1]) / pr.scale[1];
  } else {
    r.x = pr.near_plane * (cx * sgnf(pr.near_plane) - pr.offset[0]) / pr.scale[0];
    r.y = pr.near_plane * (cy * sgnf(pr.near_plane) - pr.offset[1]) / pr.scale[1];
  }
  r.z = pr.near_plane;
  return r;
}
// transform.h:118-135
DEV float4 project_point(const sthip_ProjectionData& pr, f3 v) {
  float4 r;
  if (pr.vertical_fov < 0) {
    r.x = v.x * pr.scale[0] + pr.offset[0];
    r.y = v.y * pr.scale[1] + pr.offset[1];
    r.z = (v.z - pr.far_plane) / (pr.near_plane - pr.far_plane);
    r.w = 1;
  } else {
    r.x = v.x * pr.scale[0] + v.z * pr.offset[0];
    r.y = v.y * pr.scale[1] + v.z * pr.offset[1];
    r.z = fabsf(pr.near_plane);
    r.w = v.z * sgnf(pr.near_plane);
  }
  return r;
}
// bdpt.hlsl:164-171: pixel centre -> world direction
DEV f3 primary_dir(const sthip_ViewData& view, const Xf& t, float fx, float fy, f3* local_out) {
  const float ex = (float)(view.image_max[0] - view.image_min[0]), ey = (float)(view.image_max[1] - view.image_min[1]);
  const float u = (fx + 0.5f - (float)view.image_min[0]) / ex;
  const float v = (fy + 0.5f - (float)view.image_min[1]) / ey;
  const float cx = 2 * u - 1;
  const float cy = -(2 * v - 1);
  const f3 local_dir = normalize3(back_project(view.projection, cx, cy));
  if (local_out) *local_out = local_dir;
  return normalize3(xf_vector(t, local_dir));
}

// ---------------------------------------------------------------------------------------------
// generate: PathIntegrator ctor (path.hlsli:285-298) + the prologue of sample_visibility
// (bdpt.hlsl:151-220): one primary ray per owned pixel
// ---------------------------------------------------------------------------------------------
#ifndef STHIP_TEMPLATE_INSTANCES_ONLY
inline __global__ void __launch_bounds__(STHIP_BLOCK) k_generate(FrameParams p) {
  for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < p.path_count; slot += gridDim.x * blockDim.x) {
    uint32_t px, py;
    const bool inside = slot_to_pixel(p, slot, px, py);
    const int view_index = inside ? get_view_index(p, px, py) : -1;
    p.shadow_sum[slot] = make_float4(0, 0, 0, 0);
    // (the radiance sum of a live path is first written by k_shade's first round; with media that round reads it)
    if (p.media || view_index < 0 || p.pc.gMaxPathVertices < 2) p.radiance[slot] = make_float4(0, 0, 0, 0);
    if (view_index < 0 || p.pc.gMaxPathVertices < 2) {
      // not traced: a ray that cannot hit anything and a dead path
      p.ray_o[slot] = make_float4(0, 0, 0, 1);
      p.ray_d[slot] = make_float4(0, 0, 1, 1);
      p.beta[slot] = make_float4(0, 0, 0, __uint_as_float(0u));
      p.meta[slot] = 0xFFFFFFFFu;  // marks "outside every view": resolve leaves the pixel untouched
      if (view_index >= 0) p.meta[slot] = 0xFFFFFFFEu;  // inside a view but gMaxPathVertices < 2: radiance stays 0
      if (p.debug_mode && view_index >= 0) {
        const bool clears = p.debug_mode == STHIP_DEBUG_PATH_LENGTH_CONTRIBUTION || p.debug_mode == STHIP_DEBUG_VIEW_TRACE_CONTRIBUTION;
        p.debug[slot] = clears ? make_float4(0, 0, 0, 1) : p.out_debug[(size_t)py * p.pc.gOutputExtent[0] + px];
      }
      continue;
    }
    const sthip_ViewData& view = p.views[view_index];
    const Xf t = load_xf(p.view_xf, (uint32_t)view_index);
    f3 local_dir;
    const f3 dir = primary_dir(view, t, (float)px, (float)py, &local_dir);
    if (p.cone) {  // bdpt.hlsl:176-189
      const float ex = (float)(view.image_max[0] - view.image_min[0]), ey = (float)(view.image_max[1] - view.image_min[1]);
      const float uvx = ((float)px + 0.5f - (float)view.image_min[0]) / ex, uvy = ((float)py + 0.5f - (float)view.image_min[1]) / ey;
      const float cxx = 2 * (((float)px + 1.0f + 0.5f - (float)view.image_min[0]) / ex) - 1, cxy = -(2 * uvy - 1);
      const float cyx = 2 * uvx - 1, cyy = -(2 * (((float)py + 1.0f + 0.5f - (float)view.image_min[1]) / ey) - 1);
      const f3 dir_dx = back_project(view.projection, cxx, cxy), dir_dy = back_project(view.projection, cyx, cyy);
      const f3 l = local_dir / local_dir.z;
      float spread = 0.0f;
      if (flag(p, STHIP_eRayCones)) spread = fminf(length3(dir_dx / dir_dx.z - l), length3(dir_dy / dir_dy.z - l));
      p.cone[slot] = make_float2(0.0f, spread);
    }
    p.ray_o[slot] = make_float4(t.r0.w, t.r1.w, t.r2.w, 1.0f);
    p.ray_d[slot] = make_float4(dir.x, dir.y, dir.z, 1.0f);
    p.beta[slot] = make_float4(1, 1, 1, __uint_as_float(0u));
    p.meta[slot] = 1u;  // path_length = 1, diffuse_vertices = 0
    if (p.debug_mode) {  // bdpt.hlsl:161-162: two modes start from (0,0,0,1), the others from what the image holds
      const bool clears = p.debug_mode == STHIP_DEBUG_PATH_LENGTH_CONTRIBUTION || p.debug_mode == STHIP_DEBUG_VIEW_TRACE_CONTRIBUTION;
      p.debug[slot] = clears ? make_float4(0, 0, 0, 1) : p.out_debug[(size_t)py * p.pc.gOutputExtent[0] + px];
    }
    if (p.media) {  // bdpt.hlsl:208: the medium the camera sits in
      p.media_state[2 * (size_t)slot] = make_float4(t.r0.w, t.r1.w, t.r2.w, 1.0f);
      p.media_state[2 * (size_t)slot + 1] = make_float4(1.0f, __uint_as_float(p.view_medium ? p.view_medium[view_index] : 0xFFFFu), __uint_as_float(0u), 0.0f);
      for (uint32_t k = 0; k < p.pc.gMaxDiffuseVertices; k++) p.shadow_result[(size_t)slot * p.pc.gMaxDiffuseVertices + k] = make_float4(0, 0, 0, 0);
    }
    if (p.bdpt) p.bdpt[slot] = make_float4(1, 1, 1, fabsf(local_dir.z));  // bdpt.hlsl:172,213-220: path_pdf, path_pdf_rev, dVC, prev_cos_out
  }
}
#endif

// ---------------------------------------------------------------------------------------------
// trace: trace_ray (intersection.hlsli:65-191) for every queued path, trace_visibility_ray for every shadow record.
// Persistent waves: the grid only fills the machine; each wave keeps pulling rays from the queue and
// re-packs its lanes — whenever REFILL_IDLE or more lanes have finished their ray, those lanes fetch
// new rays (ballot + popcount ranking, WaveWork) while the others keep their traversal state.
// ---------------------------------------------------------------------------------------------


#define TRACE_NONE 0xFFFFFFFFu
#ifndef SHADE_BLOCKS
#define SHADE_BLOCKS 3
#endif
#ifndef PRIMARY_BLOCKS
#define PRIMARY_BLOCKS 6  // measured: 0.418 ms at 4 blocks per CU, 0.380 at 5, 0.373 at 6, 0.378 at 8 (latency bound: occupancy pays, spills do not hurt)
#endif
#define UNIFORM_CONST __attribute__((address_space(4)))
typedef float f4v __attribute__((ext_vector_type(4)));
typedef uint32_t u2v __attribute__((ext_vector_type(2)));
// 16 bytes from a wave-uniform address through the scalar cache (the constant address space selects s_load)
DEV float4 uniform_load4(const void* base, size_t byte) {
  const f4v v = *(const UNIFORM_CONST f4v*)((const char*)base + byte);
  return make_float4(v.x, v.y, v.z, v.w);
}
DEV uint2 uniform_load2u(const void* base, size_t byte) {
  const u2v v = *(const UNIFORM_CONST u2v*)((const char*)base + byte);
  return make_uint2(v.x, v.y);
}
// One persistent kernel serves both ray kinds. A wave first feeds its idle lanes from the path queue of bounce
// `depth_closest` (trace_ray, intersection.hlsli:65-191) and, once that queue is dry, from the shadow-ray queue of
// bounce `depth_shadow` (trace_visibility_ray without media, intersection.hlsli:192-239, and the per-pixel sum of
// trace_shadows, bdpt.hlsl:311-325) — the two are independent, and the shadow rays of the previous bounce fill the
// lanes that would otherwise idle while the last, longest closest-hit rays of a launch finish. Either depth may be
// TRACE_NONE.
// What a trace kernel does with a finished ray: a closest-hit query stores the hit for k_shade (or k_shadow_media, for a
// segment of a walk through media); an unoccluded visibility ray adds what it carries where it belongs.
template <bool ALPHA>
DEV void finish_ray(const FrameParams& p, float4* target, uint32_t slot, bool shadow_lane, bool any, const RayHit& hit, f3 contribution) {
  if (ALPHA && p.media && shadow_lane) {
    p.shadow_hit[slot] = make_float4(hit.t, hit.b1, hit.b2, __uint_as_float(hit.ip));
  } else if (!any) {
    p.hit[slot] = make_float4(hit.t, hit.b1, hit.b2, __uint_as_float(hit.ip));
    p.hit_leaf[slot] = hit.leaf;
  } else if (hit.ip == 0xFFFFFFFFu) {  // unoccluded
    if (slot & 0x80000000u) {
      // a light-path vertex seen by the camera: accumulate_light_contribution, path.hlsli:47-60 — quantised
      // integer sums (order-independent, so the image does not depend on scheduling) + overflow bits
      uint32_t* lt = p.light_trace + 4 * (size_t)(slot & 0x7FFFFFFFu);
      const float q = (float)p.light_trace_quantization;
      const float cf[3] = {fmaxf(0.0f, contribution.x) * q, fmaxf(0.0f, contribution.y) * q, fmaxf(0.0f, contribution.z) * q};
      uint32_t overflow = 0;
      for (int k = 0; k < 3; k++) {
        const uint32_t ci = cf[k] >= 4294967296.0f ? 0xFFFFFFFFu : (cf[k] == cf[k] ? (uint32_t)cf[k] : 0u);
        if (ci) {
          const uint32_t prev = atomicAdd(&lt[k], ci);
          if (ci > 0xFFFFFFFFu - prev) overflow |= 1u << k;
        }
      }
      if (overflow) atomicOr(&lt[3], overflow);
    } else if (slot & 0x40000000u) {  // a light-subpath connection: its own entry, folded into gRadiance in order later
      p.conn[slot & 0x3FFFFFFFu] = make_float4(contribution.x, contribution.y, contribution.z, 0.0f);
    } else {  // NEE: each pixel has at most one shadow ray per bounce
      float4 c = target[slot];
      c.x = c.x + contribution.x;
      c.y = c.y + contribution.y;
      c.z = c.z + contribution.z;
      target[slot] = c;
    }
  }
}

// BOUNDED: the LDS stack has p.bvh.lds_levels < p.bvh.stack_depth levels (a tree too high for the LDS at full occupancy); a
// ray that overflows it is handed to k_trace_deep, which traces it again with a full stack in global memory (p.bvh.spill) —
// hits do not depend on the traversal order, so the result is the same as with an LDS stack of full height.
// TOP: the treetop (DeviceBvh::top_nodes) is held in LDS and followed ("treetop" = 1). Off by default since the loop lost
// its other exec-mask regions: the LDS-or-global branch it needs in every step now costs more than the LDS reads save.
// WIDE: the walk goes over the 4-wide form of the tree (DeviceBvh::wide_nodes, "wide_bvh" = 1): fewer, fatter dependent
// steps. Its stack has three levels from the last usable one on: every step writes top, top + 1 and top + 2.
// WIDE = 2: the walk goes over the 8-wide compressed form (DeviceBvh::wide8_nodes, "wide_bvh" = 3; traverse.h: Traversal8):
// the stack holds 64-bit groups, one push per step at most.
#ifndef STHIP_TRACE_ATTR
#define STHIP_TRACE_ATTR  // (experiments: e.g. __attribute__((amdgpu_waves_per_eu(5))) — EXPERIMENTS.md, round 4)
#endif
template <bool COUNT, bool ALPHA, bool BOUNDED = false, bool TOP = false, int WIDE = 0>
__global__ void __launch_bounds__(STHIP_BLOCK) STHIP_TRACE_ATTR k_trace(FrameParams p, uint32_t depth_closest, uint32_t depth_shadow) {
  extern __shared__ uint32_t lds_stack[];
  // the treetop behind the stacks: copied once per (persistent) block
  float4* top_lds = reinterpret_cast<float4*>(lds_stack + (size_t)p.bvh.lds_levels * STHIP_BLOCK);
  DeviceBvh bvh = p.bvh;  // k_trace's view: with TOP the entry table whose roots point into the treetop, and the treetop's root
  if (TOP) {
    for (uint32_t i = threadIdx.x; i < p.bvh.top_count * 3u; i += STHIP_BLOCK) top_lds[i] = p.bvh.top_nodes[i];
    __syncthreads();
    if (p.bvh.top_count) {
      bvh.entries = p.bvh.top_entries;
      bvh.root_ref = p.bvh.top_root_ref;
    }
  }
  const bool first = depth_closest == 0;        // first bounce: every slot, no queue
  const uint32_t* queue = p.queue[depth_closest & 1u];
  unsigned long long* ctl_c = queue_ctl(p.qctl, 0, depth_closest == TRACE_NONE ? 0u : depth_closest, 0);
  unsigned long long* ctl_s = queue_ctl(p.qctl, 1, depth_shadow == TRACE_NONE ? 0u : depth_shadow, 0);
  uint32_t* stack = lds_stack + (WIDE == 2 ? 2u * threadIdx.x : threadIdx.x);  // (8-wide: this lane's column of 64-bit entries)
  float4* target = flag(p, STHIP_eDeferShadowRays) ? p.shadow_sum : p.radiance;
  TraverseCounters cnt[2];
  cnt[0].clear();
  cnt[1].clear();
  uint32_t round_slots[2] = {0, 0}, busy_rounds[2] = {0, 0};
  if (WIDE == 1) {
    bvh.entries = p.bvh.wide_entries;
    bvh.root_ref = p.bvh.wide_root_ref;
  }
  if (WIDE == 2) {
    bvh.entries = p.bvh.wide8_entries;
    bvh.root_ref = p.bvh.wide8_root;
  }
  typename std::conditional<WIDE == 2, Traversal8<COUNT, STHIP_BLOCK, ALPHA, BOUNDED, STHIP_ENTRY_BATCH>,
                            Traversal<TRAV_MIXED, COUNT, STHIP_BLOCK, ALPHA, TOP, BOUNDED, true, STHIP_ENTRY_BATCH, WIDE == 1>>::type tr;  // SAVE_WORLD: 13 more registers, no occupancy step crossed (118 of 128; the ALPHA instantiations sit between 128 and 168 either way)
  if constexpr (WIDE == 2) {
    tr.tri_min = p.bvh.wide8_tri_min;
    if (BOUNDED) tr.limit = (p.bvh.lds_levels - 1u) * STHIP_BLOCK;
  } else {
    tr.top_lds = (const LdsFloat4*)top_lds;
    if (BOUNDED) tr.limit = (p.bvh.lds_levels - (WIDE ? 3u : 1u)) * STHIP_BLOCK;  // (the wide step writes three levels from `top` on)
  }
  tr.reset();
  tr.any = false;
  WaveWork work_c, work_s;
  work_c.init();
  work_s.init();
  work_c.exhausted = depth_closest == TRACE_NONE;
  work_s.exhausted = depth_shadow == TRACE_NONE;
  uint32_t slot = 0;
  f3 contribution = F3s(0.0f);
  bool busy = false;  // this lane holds a ray whose result is not stored yet
  bool shadow_lane = false;  // ... and it came from the shadow queue
  for (;;) {
    // A lane whose ray has finished counts as idle, but its result is stored only here, when the wave stops to refill: the
    // stores (and, for an unoccluded shadow ray, a read-modify-write of the pixel's sum with its load latency) then run for
    // all the lanes that finished since the last stop at once, instead of for two or three lanes in almost every round.
    const unsigned long long idle = __ballot(!busy || !tr.active());
    if ((uint32_t)__popcll(idle) >= p.refill_idle || idle == ~0ull) {
      if (COUNT && (threadIdx.x & 63u) == 0) cnt[0].st[7] += 64;
      if (busy && !tr.active()) {
        if (BOUNDED && tr.overflowed(stack)) {
          // the LDS stack was too short for this ray: its result is void. k_trace_deep traces it again with a stack as high as
          // the tree (global memory) and does what would have been done here; hits do not depend on the traversal order
          const uint32_t e = atomicAdd(p.deep_count, 1u);
          float4* rec = p.deep_rays + 4 * (size_t)e;
          rec[0] = make_float4(tr.o.x, tr.o.y, tr.o.z, tr.tmin);
          rec[1] = make_float4(tr.d.x, tr.d.y, tr.d.z, tr.tmax);
          rec[2] = make_float4(contribution.x, contribution.y, contribution.z, __uint_as_float(slot));
          rec[3] = make_float4(__uint_as_float((shadow_lane ? 1u : 0u) | (tr.any ? 2u : 0u)), 0.0f, 0.0f, 0.0f);
        } else {
          finish_ray<ALPHA>(p, target, slot, shadow_lane, tr.any, tr.hit, contribution);
        }
        busy = false;
      }
      if (!work_c.exhausted) {
        const uint32_t idx = work_c.take(!busy, ctl_c, first ? 0u : p.seg_stride, p.path_count);
        if (idx != 0xFFFFFFFFu) {
          slot = first ? idx : queue[idx];
          if (first && p.meta[slot] >= 0xFFFFFFFEu) {
            p.hit[slot] = make_float4(__builtin_inff(), 0, 0, __uint_as_float(0xFFFFFFFFu));
          } else {
            const float4 ro = p.ray_o[slot], rd = p.ray_d[slot];
            tr.any = false;
            tr.start(bvh, stack, xyz(ro), xyz(rd), 0.0f, __builtin_inff());
            busy = true;
            shadow_lane = false;
          }
        }
      }
      if (work_c.exhausted && !work_s.exhausted) {
        uint32_t idx = work_s.take(!busy, ctl_s, p.shadow_stride, 0);
        if (idx != 0xFFFFFFFFu) {
          if (ALPHA && p.media && (depth_shadow & 1u)) idx += p.shadow_alt;
          const float4 s0 = p.shadow_rays[3 * (size_t)idx], s1 = p.shadow_rays[3 * (size_t)idx + 1];
          // the record's index stands in for the slot while the ray is traced: what an unoccluded ray adds (and where) is
          // read again when it finishes — three registers per lane for the whole traversal are worth two loads at its end
          slot = __float_as_uint(s1.w);
          contribution = xyz(p.shadow_rays[3 * (size_t)idx + 2]);
          if (ALPHA && p.media) slot = idx;
          // with media a visibility ray is a walk from volume boundary to volume boundary (trace_visibility_ray,
          // intersection.hlsli:192-239): each segment is a closest-hit query whose result k_shadow_media consumes
          tr.any = !(ALPHA && p.media);
          tr.start(bvh, stack, xyz(s0), xyz(s1), 0.0f, s0.w);
          busy = true;
          shadow_lane = true;
        }
      }
      if (!__any(busy)) {
        if (work_c.exhausted && work_s.exhausted) break;
        continue;
      }
    }
    if (COUNT) {
      // a round belongs to the kind most of the wave is working on (the kinds only mix while the path queue drains)
      const uint32_t k = work_c.exhausted && (uint32_t)__popcll(__ballot(busy && tr.active() && tr.any)) * 2u >= (uint32_t)__popcll(__ballot(busy && tr.active())) ? 1u : 0u;
      if ((threadIdx.x & 63u) == 0) round_slots[k] += 64;
      if (busy && tr.active()) busy_rounds[k]++;
      const uint32_t n0 = cnt[k].inner_slots, t0 = cnt[k].tri_slots;
      TraverseCounters c;
      c.clear();
      tr.round(bvh, stack, p.inner_min_lanes, c);
      cnt[tr.any ? 1 : 0].nodes += c.nodes;
      cnt[tr.any ? 1 : 0].tris += c.tris;
      cnt[k].inner_slots = n0 + c.inner_slots;
      cnt[k].tri_slots = t0 + c.tri_slots;
      for (int i = 0; i < 8; i++) cnt[0].st[i] += c.st[i];
    } else {
      tr.round(bvh, stack, p.inner_min_lanes, cnt[0]);
    }
  }
  if (COUNT) {
    for (uint32_t k = 0; k < 2; k++) {
      wave_add(&p.counters[CNT_NODES + k], cnt[k].nodes);
      wave_add(&p.counters[CNT_TRIS + k], cnt[k].tris);
      wave_add(&p.counters[CNT_INNER_SLOTS + k], cnt[k].inner_slots);
      wave_add(&p.counters[CNT_TRI_SLOTS + k], cnt[k].tri_slots);
      wave_add(&p.counters[CNT_ROUND_SLOTS + k], round_slots[k]);
      wave_add(&p.counters[CNT_BUSY_ROUNDS + k], busy_rounds[k]);
    }
    for (int i = 0; i < 8; i++) wave_add(&p.counters[CNT_LANE_STATES + i], cnt[0].st[i]);
  }
}

// The rays that overflowed the bounded LDS stacks of the k_trace launch before it (rare: a stack of 32 levels holds all
// but pathological rays; tests force it with lds_stack_levels). A fixed grid, one full-height stack column per thread.
template <bool COUNT, bool ALPHA>
__global__ void __launch_bounds__(STHIP_BLOCK) k_trace_deep(FrameParams p) {
  const uint32_t count = *p.deep_count;
  const uint32_t tid = blockIdx.x * STHIP_BLOCK + threadIdx.x, threads = gridDim.x * STHIP_BLOCK;
  uint32_t* column = p.bvh.spill + (size_t)tid * p.bvh.stack_depth;
  float4* target = flag(p, STHIP_eDeferShadowRays) ? p.shadow_sum : p.radiance;
  TraverseCounters cnt[2];
  cnt[0].clear();
  cnt[1].clear();
  for (uint32_t e = tid; e < count; e += threads) {
    const float4* rec = p.deep_rays + 4 * (size_t)e;
    const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
    const uint32_t bits = __float_as_uint(rec[3].x);
    const bool any = (bits & 2u) != 0;
    RayHit h;
    if (any)
      traverse<TRAV_ANY, COUNT, 1, ALPHA>(p.bvh, xyz(r0), xyz(r1), r0.w, r1.w, column, h, cnt[1]);
    else
      traverse<TRAV_CLOSEST, COUNT, 1, ALPHA>(p.bvh, xyz(r0), xyz(r1), r0.w, r1.w, column, h, cnt[0]);
    finish_ray<ALPHA>(p, target, __float_as_uint(r2.w), (bits & 1u) != 0, any, h, xyz(r2));
  }
  if (COUNT) {
    for (uint32_t k = 0; k < 2; k++) {
      if (cnt[k].nodes) atomicAdd(&p.counters[CNT_NODES + k], (unsigned long long)cnt[k].nodes);
      if (cnt[k].tris) atomicAdd(&p.counters[CNT_TRIS + k], (unsigned long long)cnt[k].tris);
    }
  }
  // The last block to get here leaves the count at zero for the next k_trace launch (no memset between launches): every
  // block has read the count before it arrives, and nothing else touches it until this kernel has ended. With no ray in
  // the queue (nearly every launch) there is nothing to reset, and the 2048 blocks do not line up on one atomic (~11 ns each).
  if (count == 0) return;
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(p.deep_count + 1, 1u) == gridDim.x - 1u) {
    p.deep_count[0] = 0u;
    p.deep_count[1] = 0u;
  }
}

// ---------------------------------------------------------------------------------------------
// trace_primary: the first bounce as wave packets. The 64 slots of a wave are one 8x8 pixel block (slot_to_pixel),
// i.e. 64 rays from one origin through neighbouring pixels, so the wave walks the tree ONCE for all of them: one
// shared stack, every node and triangle fetched at a wave-uniform address (one cache line for the wave instead of 64),
// no lane divergence; a subtree is entered when any lane's ray hits its box. Each lane still tests its own ray with
// the contract's arithmetic against every triangle the packet reaches, and the closest hit is a minimum over all
// triangles (ties by id), so visiting more leaves than a single ray would cannot change the answer: results are
// bit-identical to k_trace's.
// ---------------------------------------------------------------------------------------------
template <bool COUNT, bool ALPHA>
__global__ void __launch_bounds__(STHIP_BLOCK, PRIMARY_BLOCKS) k_trace_primary(FrameParams p) {
  extern __shared__ uint32_t lds_stack[];
  const uint32_t wave_in_block = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  uint32_t* stack = lds_stack + wave_in_block * p.bvh.stack_depth;  // one stack per wave
  // behind the stacks: every lane's world-space ray constants (13 words, [word][thread]), kept while the packet is inside an
  // instance — leaving it reads them back instead of running setup_space again (this kernel has no registers to spare)
  float* world_save = reinterpret_cast<float*>(lds_stack + (STHIP_BLOCK / 64) * p.bvh.stack_depth) + threadIdx.x;
  const uint32_t packets = (p.path_count + 63u) >> 6;
  TraverseCounters cnt;
  cnt.clear();
  for (uint32_t packet = blockIdx.x * (blockDim.x >> 6) + wave_in_block; packet < packets; packet += gridDim.x * (blockDim.x >> 6)) {
    const uint32_t slot = packet * 64u + lane;
    bool live = slot < p.path_count && p.meta[slot] < 0xFFFFFFFEu;
    f3 o = F3s(0.0f), d = F3(0.0f, 0.0f, 1.0f);
    if (live) {
      o = xyz(p.ray_o[slot]);
      d = xyz(p.ray_d[slot]);
    }
    RayHit hit;
    hit.t = __builtin_inff();
    hit.b1 = hit.b2 = 0.0f;
    hit.ip = 0xFFFFFFFFu;
    hit.leaf = 0xFFFFFFFFu;
    if (!__any(live)) {
      if (slot < p.path_count) p.hit[slot] = make_float4(hit.t, 0, 0, __uint_as_float(hit.ip));
      continue;
    }
    RaySpace sp;
    setup_space(sp, o, d, p.bvh.scene_cx, p.bvh.scene_cy, p.bvh.scene_cz, p.bvh.scene_radius);
    {
      const float w[13] = {sp.idir.x, sp.idir.y, sp.idir.z, sp.noodL.x, sp.noodL.y, sp.noodL.z, sp.noodH.x, sp.noodH.y, sp.noodH.z, sp.Sx, sp.Sy, sp.Sz, __int_as_float(sp.k)};
      for (int k = 0; k < 13; k++) world_save[k * STHIP_BLOCK] = w[k];
    }
    uint32_t id_bits = 0;
    uint32_t top = 0;
    uint32_t ref = p.bvh.root_ref;  // wave-uniform throughout
    const char* nbase = reinterpret_cast<const char*>(p.bvh.nodes);
    const char* tbase = reinterpret_cast<const char*>(p.bvh.tris);
    while (ref != TRAV_DONE) {
      ref = (uint32_t)__builtin_amdgcn_readfirstlane((int)ref);
      if (!(ref & BVH_LEAF_BIT)) {
        // wave-uniform address in the constant address space: scalar loads (the node lands in SGPRs through the scalar
        // cache instead of occupying 14 VGPRs of all 64 lanes and a texture-addresser slot per lane)
        const size_t nb = (size_t)ref * BVH_NODE_BYTES;
        const float4 n0 = uniform_load4(nbase, nb), n1 = uniform_load4(nbase, nb + 16), nz = uniform_load4(nbase, nb + 32);
        // the child references: low bytes of the x / y planes (bvh.h); wave-uniform, so this is scalar arithmetic
        const uint2 cr = make_uint2((__float_as_uint(n0.x) & 0xFFu) | ((__float_as_uint(n0.y) & 0xFFu) << 8) | ((__float_as_uint(n0.z) & 0xFFu) << 16) | (__float_as_uint(n0.w) << 24),
                                    (__float_as_uint(n1.x) & 0xFFu) | ((__float_as_uint(n1.y) & 0xFFu) << 8) | ((__float_as_uint(n1.z) & 0xFFu) << 16) | (__float_as_uint(n1.w) << 24));
        if (COUNT && lane == 0) {
          cnt.nodes++;
          cnt.inner_slots += 64;
        }
        const float tbest = hit.t;
        const float a0x = fmaf(n0.x, sp.idir.x, sp.noodL.x), b0x = fmaf(n0.y, sp.idir.x, sp.noodH.x);
        const float a0y = fmaf(n0.z, sp.idir.y, sp.noodL.y), b0y = fmaf(n0.w, sp.idir.y, sp.noodH.y);
        const float a0z = fmaf(nz.x, sp.idir.z, sp.noodL.z), b0z = fmaf(nz.y, sp.idir.z, sp.noodH.z);
        const float a1x = fmaf(n1.x, sp.idir.x, sp.noodL.x), b1x = fmaf(n1.y, sp.idir.x, sp.noodH.x);
        const float a1y = fmaf(n1.z, sp.idir.y, sp.noodL.y), b1y = fmaf(n1.w, sp.idir.y, sp.noodH.y);
        const float a1z = fmaf(nz.z, sp.idir.z, sp.noodL.z), b1z = fmaf(nz.w, sp.idir.z, sp.noodH.z);
        const float tn0 = fmaxf(fmaxf(fminf(a0x, b0x), fminf(a0y, b0y)), fmaxf(fminf(a0z, b0z), 0.0f));
        const float tf0 = fminf(fminf(fmaxf(a0x, b0x), fmaxf(a0y, b0y)), fminf(fmaxf(a0z, b0z), tbest));
        const float tn1 = fmaxf(fmaxf(fminf(a1x, b1x), fminf(a1y, b1y)), fmaxf(fminf(a1z, b1z), 0.0f));
        const float tf1 = fminf(fminf(fmaxf(a1x, b1x), fmaxf(a1y, b1y)), fminf(fmaxf(a1z, b1z), tbest));
        const bool h0 = live & (tn0 <= tf0);
        const bool h1 = live & (tn1 <= tf1);
        const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1);
        if (m0 | m1) {
          // nearer child first by vote: lanes that hit child 1 and see it nearer (or do not hit child 0 at all)
          const unsigned long long prefer1 = __ballot(h1 & (!h0 | (tn1 < tn0)));
          const bool first1 = m1 && (!m0 || __popcll(prefer1) * 2 > __popcll(m0 | m1));
          ref = first1 ? cr.y : cr.x;
          if (m0 && m1) {
            stack[top] = first1 ? cr.x : cr.y;
            top++;
          }
        } else {
          ref = top ? stack[--top] : TRAV_DONE;
        }
        continue;
      }
      // leaf bit: exit sentinel, instance entry, or triangles
      if (ref == TRAV_EXIT_INSTANCE) {
        sp.o = o;
        sp.idir = F3(world_save[0], world_save[STHIP_BLOCK], world_save[2 * STHIP_BLOCK]);
        sp.noodL = F3(world_save[3 * STHIP_BLOCK], world_save[4 * STHIP_BLOCK], world_save[5 * STHIP_BLOCK]);
        sp.noodH = F3(world_save[6 * STHIP_BLOCK], world_save[7 * STHIP_BLOCK], world_save[8 * STHIP_BLOCK]);
        sp.Sx = world_save[9 * STHIP_BLOCK];
        sp.Sy = world_save[10 * STHIP_BLOCK];
        sp.Sz = world_save[11 * STHIP_BLOCK];
        sp.k = __float_as_int(world_save[12 * STHIP_BLOCK]);
        id_bits = 0;
        ref = top ? stack[--top] : TRAV_DONE;
        continue;
      }
      if (ref & BVH_INST_BIT) {
        const TlasEntry* e = p.bvh.entries + (ref & 0xFFFFu);
        const float4* ev = reinterpret_cast<const float4*>(e);
        const uint4 info = *reinterpret_cast<const uint4*>(ev + 3);
        if (info.z != TLAS_ENTRY_IDENTITY) {
          const float4 r0 = ev[0], r1 = ev[1], r2 = ev[2];
          const float4 sph = ev[4];
          const float m[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
          if (info.z == TLAS_ENTRY_SPHERE) {
            if (COUNT && lane == 0) {
              cnt.tris++;
              cnt.tri_slots += 64;
            }
            float t;
            if (live && sphere_test(obj_point(m, o), obj_vector(m, d), sph.w, 0.0f, __builtin_inff(), t)) {
              const uint32_t ip = info.y | 0xFFFF0000u;
              if (t < hit.t || (t == hit.t && hit.ip != 0xFFFFFFFFu && hit_key(ip) < hit_key(hit.ip))) {
                hit.t = t;
                hit.b1 = hit.b2 = 0.0f;
                hit.ip = ip;
              }
            }
            ref = top ? stack[--top] : TRAV_DONE;
            continue;
          }
          setup_space(sp, obj_point(m, o), obj_vector(m, d), sph.x, sph.y, sph.z, sph.w);
          id_bits = info.y;
          stack[top] = TRAV_EXIT_INSTANCE;
          top++;
        }
        ref = info.x;
        continue;
      }
      const uint32_t first = (ref & 0x3FFFFFFFu) >> 2;
      const uint32_t count = (ref & 3u) + 1u;
      for (uint32_t i = 0; i < count; i++) {
        const size_t tb = (size_t)((first + i) * 48u);
        const float4 v0 = uniform_load4(tbase, tb), v1 = uniform_load4(tbase, tb + 16), v2 = uniform_load4(tbase, tb + 32);
        if (COUNT && lane == 0) {
          cnt.tris++;
          cnt.tri_slots += 64;
        }
        float t, b1, b2;
        // (no exec-mask regions: see Traversal::leaf_step)
        bool candidate = live & tri_test(sp, xyz(v0), xyz(v1), xyz(v2), 0.0f, __builtin_inff(), t, b1, b2);
        if (ALPHA && candidate) {
          const uint32_t mask = p.bvh.alpha_test ? p.bvh.inst_alpha[(__float_as_uint(v0.w) | id_bits) & 0xFFFFu] : BVH_NO_ALPHA;
          if (mask != BVH_NO_ALPHA) {
            const float2* q = p.bvh.tri_uv + (size_t)(first + i) * 3u;
            const float2 u0 = q[0], u1 = q[1], u2 = q[2];
            const float u = u0.x + (u1.x - u0.x) * b1 + (u2.x - u0.x) * b2;
            float v = u0.y + (u1.y - u0.y) * b1 + (u2.y - u0.y) * b2;
            if (p.bvh.flip_uvs) v = 1 - v;
            candidate = sample_image1(p.bvh, mask, u, v) >= 0.75f;
          }
        }
        const uint32_t ip = __float_as_uint(v0.w) | id_bits;
        const bool closer = candidate & ((t < hit.t) | ((t == hit.t) & (hit.ip != 0xFFFFFFFFu) & (hit_key(ip) < hit_key(hit.ip))));
        hit.t = closer ? t : hit.t;
        hit.b1 = closer ? b1 : hit.b1;
        hit.b2 = closer ? b2 : hit.b2;
        hit.ip = closer ? ip : hit.ip;
        hit.leaf = closer ? first + i : hit.leaf;
      }
      ref = top ? stack[--top] : TRAV_DONE;
    }
    if (slot < p.path_count) {
      p.hit[slot] = make_float4(hit.t, hit.b1, hit.b2, __uint_as_float(hit.ip));
      p.hit_leaf[slot] = hit.leaf;
    }
  }
  if (COUNT) {
    wave_add(&p.counters[CNT_NODES], cnt.nodes);
    wave_add(&p.counters[CNT_TRIS], cnt.tris);
    wave_add(&p.counters[CNT_NODES_PRIMARY], cnt.nodes);
    wave_add(&p.counters[CNT_TRIS_PRIMARY], cnt.tris);
    wave_add(&p.counters[CNT_INNER_SLOTS], cnt.inner_slots);
    wave_add(&p.counters[CNT_TRI_SLOTS], cnt.tri_slots);
  }
}

// ---------------------------------------------------------------------------------------------
// The last ray of a path. A vertex at which the path or diffuse budget ends can only add the emission of what the ray hits
// (next_vertex, path.hlsli:955-966 in front of :975; eval_emission :847-894), so when k_shade sends a path into its last
// vertex the ray matters only if its closest hit lies on an emissive triangle — and it cannot if the ray misses the bounds
// of every emissive instance. Such a ray is answered on the spot: counted as the trace_ray call it stands for, never
// queued, and the path ends with what it has. The test is the traversal's slab ARITHMETIC (setup_space's padded form, in the
// space the instance's triangles are tested in) on the box of the instance's vertices widened by 2^-15 of their magnitude.
// That box is not one of the walk's: the tree's ancestors of a triangle are looser (outward-rounded planes, 8-bit grids,
// the top level's boxes), so the filter asks MORE of a ray than the walk does. It drops nothing the walk would find as long
// as the triangle test only accepts points the padded box holds: the test's own tolerance is ~3e-7 x distance, the padding
// 4e-6 x (distance to the scene + its radius) + 3e-7 |origin| — true while the ray's origin lies within a few scene sizes of
// the scene, the same precondition under which any tree finds the contract's hits at all (tests/test_oracle.py pins the far-origin hole of the
// contract; paths start on surfaces of the scene, so they meet it). Under it frames and ray counts are those of the
// unfiltered pipeline bit for bit ("answer_last_rays" = 0): tested on degenerate / sliver / tiny emitters and on scenes at
// large coordinates (tests/test_gpu_parity.py), and by tools/stress_last_ray_filter.py at 50 M rays per scene.
// ---------------------------------------------------------------------------------------------
DEV bool aims_at_emitter(const FrameParams& p, f3 o, f3 d) {
  // the table and the transforms are read at wave-uniform addresses (scalar loads); no early exit: every lane tests every box
  RaySpace world;
  setup_space(world, o, d, p.bvh.scene_cx, p.bvh.scene_cy, p.bvh.scene_cz, p.bvh.scene_radius);
  bool aims = false;
  for (uint32_t e = 0; e < p.emitter_count; e++) {
    const size_t at = (size_t)e * sizeof(EmitterBounds);
    const float4 b0 = uniform_load4(p.emitters, at), b1 = uniform_load4(p.emitters, at + 16), b2 = uniform_load4(p.emitters, at + 32);  // lo | instance, hi | identity, sphere
    RaySpace sp = world;
    if (__float_as_uint(b1.w) == 0u) {  // (wave-uniform) an instance with a transform: its triangles are tested in object space
      const size_t xat = (size_t)__float_as_uint(b0.w) * sizeof(sthip_TransformData);
      const float4 r0 = uniform_load4(p.scene.inv_xf, xat), r1 = uniform_load4(p.scene.inv_xf, xat + 16), r2 = uniform_load4(p.scene.inv_xf, xat + 32);
      const float m[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
      setup_space(sp, obj_point(m, o), obj_vector(m, d), b2.x, b2.y, b2.z, b2.w);
    }
    const float ax = fmaf(b0.x, sp.idir.x, sp.noodL.x), bx = fmaf(b1.x, sp.idir.x, sp.noodH.x);
    const float ay = fmaf(b0.y, sp.idir.y, sp.noodL.y), by = fmaf(b1.y, sp.idir.y, sp.noodH.y);
    const float az = fmaf(b0.z, sp.idir.z, sp.noodL.z), bz = fmaf(b1.z, sp.idir.z, sp.noodH.z);
    const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    aims = aims | (tn <= tf);
  }
  return aims;
}

// sample_point_on_light, light.hlsli:37-152 (uniform light choice), as one function for connect_light and presample_lights
struct LightSample {
  f3 Le, to_light, normal, position;
  float pdf, dist;
  bool is_env, area_measure;
};
template <bool TEXTURED, bool EXT>
DEV void sample_point_on_light(const FrameParams& p, bool has_env, bool has_emissives, float r0, float r1, float r2, float r3, f3 ref_pos, LightSample& ls) {
  ls.Le = F3s(0.0f);
  ls.to_light = F3s(0.0f);
  ls.normal = F3s(0.0f);
  ls.position = F3s(0.0f);
  ls.pdf = 0;
  ls.dist = 0;
  ls.is_env = false;
  ls.area_measure = true;
  if (has_env && (!has_emissives || r3 <= p.pc.gEnvironmentSampleProbability)) {  // light.hlsli:38-48
    Environment env;
    env.load(p.scene, p.pc.gEnvironmentMaterialAddress);
    ls.Le = env.sample(p.scene, r0, r1, ls.to_light, ls.pdf, flag(p, STHIP_eSampleEnvironmentMapDirectly));
    if (has_emissives) ls.pdf *= p.pc.gEnvironmentSampleProbability;
    ls.is_env = true;
    ls.dist = __builtin_inff();
    ls.area_measure = false;
  } else if (has_emissives) {
    const float rw = has_env ? (r3 - p.pc.gEnvironmentSampleProbability) / (1 - p.pc.gEnvironmentSampleProbability) : r3;
    const int li = (int)(rw * ((float)p.pc.gLightCount * .9999f));
    ls.pdf = 1 / (float)p.pc.gLightCount;
    const uint32_t light_instance_index = p.scene.lights[li];
    if (has_env) ls.pdf *= 1 - p.pc.gEnvironmentSampleProbability;
    const Inst lin = load_inst(p.scene, light_instance_index);
    float lu, lv;
    if (EXT && lin.type() == STHIP_INSTANCE_TYPE_SPHERE) {  // light.hlsli:58-121
      const float r = lin.radius();
      const Xf t = load_xf(p.scene.xf, light_instance_index);

      if (flag(p, STHIP_eUniformSphereSampling)) {
        ls.pdf /= 4 * DET_PI * r * r;
        const float z = 1 - 2 * r0;
        const float r_ = sqrtf(fmaxf(0.0f, 1 - z * z));
        const float phi = DET_2PI * r1;
        float sp, cp;
        det_sincosf(phi, &sp, &cp);
        const f3 local_normal = F3(r_ * cp, z, r_ * sp);
        cartesian_to_spherical_uv(local_normal, lu, lv);
        ls.position = xf_point(t, r * local_normal);
        ls.normal = normalize3(xf_vector(t, local_normal));
      } else {
        const f3 center = F3(t.r0.w, t.r1.w, t.r2.w);
        f3 to_center = center - ref_pos;
        const float dist = length3(to_center);
        to_center = to_center / dist;
        const float sinThetaMax = r / dist;
        const float sinThetaMax2 = sinThetaMax * sinThetaMax;
        const float invSinThetaMax = 1 / sinThetaMax;
        const float cosThetaMax = sqrtf(fmaxf(0.0f, 1 - sinThetaMax2));
        ls.pdf /= DET_2PI * (1 - cosThetaMax);
        ls.area_measure = false;
        float cosTheta = (cosThetaMax - 1) * r0 + 1;
        float sinTheta2 = 1 - cosTheta * cosTheta;
        if (sinThetaMax2 < 0.00068523f) {
          sinTheta2 = sinThetaMax2 * r0;
          cosTheta = sqrtf(1 - sinTheta2);
        }
        const float cosAlpha = sinTheta2 * invSinThetaMax + cosTheta * sqrtf(fmaxf(0.0f, 1 - sinTheta2 * invSinThetaMax * invSinThetaMax));
        const float sinAlpha = sqrtf(fmaxf(0.0f, 1 - cosAlpha * cosAlpha));
        const float phi = r1 * 2 * DET_PI;
        float sp, cp;
        det_sincosf(phi, &sp, &cp);
        f3 T, B;
        make_orthonormal(to_center, T, B);
        ls.normal = -(T * sinAlpha * cp + B * sinAlpha * sp + to_center * cosAlpha);
        ls.position = center + r * ls.normal;
        const f3 local_normal = xf_vector(load_xf(p.scene.inv_xf, light_instance_index), ls.normal);
        cartesian_to_spherical_uv(local_normal, lu, lv);
      }
      ls.to_light = ls.position - ref_pos;
      ls.dist = length3(ls.to_light);
      ls.to_light = ls.to_light / ls.dist;
    } else {
      const uint32_t lpc = lin.prim_count();
      const uint32_t lprim = (uint32_t)fminf(r2 * (float)lpc, (float)(lpc - 1));
      const float a = sqrtf(r0);
      ShadingData lsd;
      make_triangle_shading_data(p.scene, lsd, light_instance_index, lin, lprim, 1 - a, a * r1, TEXTURED && flag(p, STHIP_eFlipTriangleUVs));
      lu = lsd.u;
      lv = lsd.v;
      ls.normal = lsd.geometry_normal();
      ls.position = lsd.position;
      ls.to_light = lsd.position - ref_pos;
      ls.dist = length3(ls.to_light);
      ls.to_light = ls.to_light / ls.dist;
      ls.pdf /= lsd.shape_area * (float)lpc;
    }
    if (ls.pdf > 0) {
      DisneyMaterial lm;
      if (TEXTURED) {  // light.hlsli:143-150: uv of the light sample, uv_screen_size = 0, no normal map
        uint32_t dn = 0, dt = 0;
        lm.load_textured(p.scene, lin.material_address(), lu, lv, 0.0f, dn, dt, p.sampling_flags & ~(1u << STHIP_eNormalMaps));
      } else {
        lm.load(p.scene, lin.material_address());
      }
      ls.Le = lm.Le();
    }
  }
}

// presample_lights, bdpt.hlsl:84-99 (dispatched once per frame, BDPT.cpp:644-651): gLightPresampleTileSize x TileCount
// light points per seed, drawn with rng_init(-1, index) and the reference point 0
template <bool TEXTURED, bool EXT>
__global__ void __launch_bounds__(STHIP_BLOCK) k_presample_lights(FrameParams p) {
  const uint32_t n = p.pc.gLightPresampleTileSize * p.pc.gLightPresampleTileCount;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * p.seeds_in_flight) return;
  const uint32_t seed_index = i / n, index = i - seed_index * n;
  Rng rng;
  rng.x = rng.y = 0xFFFFFFFFu;
  rng.seed = p.seed + seed_index;
  rng.counter = index;
  const float r0 = rng.next_float(), r1 = rng.next_float(), r2 = rng.next_float(), r3 = rng.next_float();
  const bool has_env = EXT && (p.scene_flags & STHIP_BDPT_FLAG_HAS_ENVIRONMENT) != 0;
  const bool has_emissives = !EXT || (p.scene_flags & STHIP_BDPT_FLAG_HAS_EMISSIVES) != 0;
  LightSample ls;
  sample_point_on_light<TEXTURED, EXT>(p, has_env, has_emissives, r0, r1, r2, r3, F3s(0.0f), ls);
  p.presampled[2 * (size_t)i] = make_float4(ls.position.x, ls.position.y, ls.position.z, __uint_as_float(pack_normal_octahedron(ls.normal)));
  p.presampled[2 * (size_t)i + 1] = make_float4(ls.Le.x, ls.Le.y, ls.Le.z, ls.is_env ? -ls.pdf : ls.pdf);
}

// ---------------------------------------------------------------------------------------------
// Light tracing (eConnectToViews): sample_photons (bdpt.hlsl:101-147) as a second kind of path through the same
// wavefront pipeline. k_generate_light starts one light subpath per thread of the reference's padded dispatch,
// k_trace extends it, k_shade_light runs next_vertex() for light paths: connect_view (a visibility ray to the camera
// that splats into gLightTraceSamples when it arrives) and the adjoint bounce.
// ---------------------------------------------------------------------------------------------
// thread (x, y) of dispatch_over(W, ceil(gLightPathCount / W)) in 8x4 groups for light slot e, and its path index
DEV void light_thread(const FrameParams& p, uint32_t e, uint32_t& x, uint32_t& y, uint32_t& path_index) {
  const uint32_t W = p.pc.gOutputExtent[0];
  const uint32_t gw = (W + 7u) >> 3;
  const uint32_t group = e >> 5, local = e & 31u;
  x = (group % gw) * 8u + (local & 7u);
  y = (group / gw) * 4u + (local >> 3);
  path_index = flag(p, STHIP_eRemapThreads) ? e : y * W + x;  // map_pixel_coord, bdpt_util.hlsli:76-83
}

template <bool TEXTURED, bool EXT>
__global__ void __launch_bounds__(STHIP_BLOCK) k_generate_light(FrameParams p) {
  const bool has_env = EXT && (p.scene_flags & STHIP_BDPT_FLAG_HAS_ENVIRONMENT) != 0;
  const bool has_emissives = !EXT || (p.scene_flags & STHIP_BDPT_FLAG_HAS_EMISSIVES) != 0;
  uint32_t live = 0;
  for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < p.path_count; slot += gridDim.x * blockDim.x) {
    const uint32_t seed_index = slot / p.light_threads, e = slot - seed_index * p.light_threads;
    uint32_t x, y, path_index;
    light_thread(p, e, x, y, path_index);
    p.meta[slot] = 0xFFFFFFFEu;  // no path unless everything below succeeds
    p.beta[slot] = make_float4(0, 0, 0, 0);
    if (path_index >= p.pc.gLightPathCount) continue;
    Rng rng;
    rng.x = x;
    rng.y = y;
    rng.seed = p.seed + seed_index;
    rng.counter = 0xFFFFFFu;  // path.hlsli:295-297 with gTraceLight
    const float r0 = rng.next_float(), r1 = rng.next_float(), r2 = rng.next_float(), r3 = rng.next_float();
    LightSample ls;
    sample_point_on_light<TEXTURED, EXT>(p, has_env, has_emissives, r0, r1, r2, r3, F3s(0.0f), ls);
    if (ls.pdf <= 0 || all_le0(ls.Le)) continue;
    f3 beta = ls.Le / ls.pdf;
    const float u1 = rng.next_float(), u2 = rng.next_float();
    const f3 local_dir_out = sample_cos_hemisphere(u1, u2);
    const float bsdf_pdf = cosine_hemisphere_pdfW(local_dir_out.z);
    beta = beta * (local_dir_out.z / bsdf_pdf);
    f3 T, B;
    make_orthonormal(ls.normal, T, B);
    const f3 direction = T * local_dir_out.x + B * local_dir_out.y + ls.normal * local_dir_out.z;
    const f3 origin = ray_offset(ls.position, ls.normal);
    p.ray_o[slot] = make_float4(origin.x, origin.y, origin.z, bsdf_pdf);
    p.ray_d[slot] = make_float4(direction.x, direction.y, direction.z, 1.0f);
    p.beta[slot] = make_float4(beta.x, beta.y, beta.z, __uint_as_float(rng.counter));
    p.bdpt[slot] = make_float4(ls.pdf, 1.0f, 1 / ls.pdf, local_dir_out.z);  // path_pdf, path_pdf_rev, dVC, prev_cos_out
    if (p.path_contrib) {  // bdpt.hlsl:120,135
      const f3 pcb = ls.Le * local_dir_out.z;
      p.path_contrib[slot] = make_float4(pcb.x, pcb.y, pcb.z, 0.0f);
    }
    p.meta[slot] = 1u;
    if (TEXTURED) p.cone[slot] = make_float2(0.0f, 0.0f);
    if (p.media) {  // bdpt.hlsl:114: a light path starts outside every medium; T_dir_pdf = T_nee_pdf = 1, no segment walked yet
      p.media_state[2 * (size_t)slot] = make_float4(origin.x, origin.y, origin.z, 1.0f);
      p.media_state[2 * (size_t)slot + 1] = make_float4(1.0f, __uint_as_float(0xFFFFu), __uint_as_float(0u), 0.0f);
    }
    live++;
  }
  wave_add(&p.counters[CNT_RAYS_CLOSEST], live);  // every live light path traces its first ray (path.hlsli:1006)
}

// ray counts of a finished pass: every queued path / shadow record was traced exactly once
#ifndef STHIP_TEMPLATE_INSTANCES_ONLY
inline __global__ void k_count_rays(FrameParams p) {
  unsigned long long closest = 0, shadow = 0, answered = 0;
  for (uint32_t d = 0; d < p.rounds; d++)
    for (uint32_t s = 0; s < QUEUE_SEGMENTS; s++) {
      if (d) closest += queue_ctl(p.qctl, 0, d, s)[QCTL_SIZE];
      if (d) answered += queue_ctl(p.qctl, 0, d, s)[QCTL_ANSWERED];
      shadow += queue_ctl(p.qctl, 1, d, s)[QCTL_SIZE];
    }
  p.counters[CNT_RAYS_CLOSEST] += closest + answered;
  p.counters[CNT_RAYS_ANSWERED] += answered;
  p.counters[CNT_RAYS_SHADOW] += shadow;
}
#endif

// One step of trace_visibility_ray with media (intersection.hlsli:192-239), after the closest hit (dt, ip) of the walk's
// current segment is known: a surface ends the walk with nothing; a volume boundary is crossed — delta tracking that cannot
// scatter over the segment if it ran inside a medium — and the walk goes on from the other side. Returns true when the walk
// has ended (the miss / end of the ray included).
DEV bool visibility_step_media(const FrameParams& p, Rng& rng, f3& o, f3 d, float& t_max, uint32_t& cur_medium, f3& contribution, float& T_dir, float& T_nee, float dt, uint32_t ip) {
  if (!isinf(t_max)) t_max -= dt;
  if (ip == 0xFFFFFFFFu) return true;
  const uint32_t hit_inst = ip & 0xFFFFu;
  const Inst hin = load_inst(p.scene, hit_inst);
  if (hin.type() != STHIP_INSTANCE_TYPE_VOLUME) {  // a surface: occluded
    contribution = F3s(0.0f);
    T_dir = 0;
    T_nee = 0;
    return true;
  }
  if (cur_medium != 0xFFFFu) {
    Medium mm;
    mm.load(p.scene, load_inst(p.scene, cur_medium).material_address());
    const Xf inv = load_xf(p.scene.inv_xf, cur_medium);
    f3 dir_pdf = F3s(1.0f), nee3 = F3s(1.0f), scatter_p;
    mm.delta_track(p.scene, rng, xf_point(inv, o), xf_vector(inv, d), dt, contribution, dir_pdf, nee3, false, p.pc.gMaxNullCollisions, scatter_p);
    T_dir *= average3(dir_pdf);
    T_nee *= average3(nee3);
  }
  f3 bpos;
  uint32_t bn;
  volume_boundary(p.scene, hit_inst, hin, o, d, dt, bpos, bn);
  const f3 bgn = unpack_normal_octahedron(bn);
  if (dot3(d, bgn) < 0) {  // entering
    cur_medium = hit_inst;
    o = ray_offset(bpos, -bgn);
  } else {
    cur_medium = 0xFFFFu;
    o = ray_offset(bpos, bgn);
  }
  return !(t_max > 1e-6f);
}
// The whole walk at once, by the calling thread, with `column` as its traversal stack (global memory, as k_trace_deep's): the
// form media take WITHOUT eDeferShadowRays, where the walk draws from the path's own stream between the light sample and the
// BSDF sample of a vertex (path.hlsli:329-332). Returns the closest-hit queries it made (gRayCount[0] counts them).
DEV uint32_t visibility_walk_media(const FrameParams& p, Rng& rng, f3 o, f3 d, float t_max, uint32_t cur_medium, f3& contribution, float& T_dir, float& T_nee, uint32_t* column) {
  uint32_t segments = 0;
  TraverseCounters cnt;
  cnt.clear();
  // (upstream's loop has no bound; 64 segments — the bound of its path walk, intersection.hlsli:247 — pins the end of a walk
  // that never gets anywhere, on both sides: a thread of k_shade must not spin on a degenerate boundary)
  while (t_max > 1e-6f && segments < 64u) {
    RayHit h;
    traverse<TRAV_CLOSEST, false, 1, true>(p.bvh, o, d, 0.0f, t_max, column, h, cnt);
    segments++;
    if (visibility_step_media(p, rng, o, d, t_max, cur_medium, contribution, T_dir, T_nee, h.t, h.ip)) break;
  }
  return segments;
}

// path_weight, path.hlsli:16-28: one over the number of ways upstream counts for a path of this many vertices
DEV float path_weight(const FrameParams& p, uint32_t view_length, uint32_t light_length) {
  const uint32_t nv = view_length + light_length;
  if (nv <= 2) return 1;
  uint32_t ways = 1;
  if (flag(p, STHIP_eNEE)) ways++;
  if (flag(p, STHIP_eConnectToViews) && nv <= p.pc.gMaxPathVertices + 1) ways++;
  if (flag(p, STHIP_eConnectToLightPaths)) ways += min(p.pc.gMaxPathVertices, nv - 2);
  return 1.f / (float)ways;
}

// MEDIA: the scene has volume instances: a light path's trace() is the same walk from volume boundary to volume boundary as a view
// path's (k_shade), it scatters inside media (phase function, connect_view from there), and every connect_view walks its
// visibility ray through the media itself, in the light path's own stream (path.hlsli:577-581: trace_visibility_ray is inline
// there whatever eDeferShadowRays says) — visibility_walk_media, then the splat at once.
template <bool TEXTURED, bool EXT, bool MEDIA = false>
__global__ void __launch_bounds__(STHIP_BLOCK, MEDIA ? 2 : 3) k_shade_light(FrameParams p, uint32_t depth) {
  const uint32_t seg = blockIdx.x % QUEUE_SEGMENTS;
  const uint32_t seg_base = seg * p.seg_stride;
  uint32_t n, slot0 = 0;
  if (depth == 0) {
    const uint32_t per = (((p.path_count + QUEUE_SEGMENTS - 1u) / QUEUE_SEGMENTS) + 63u) & ~63u;
    slot0 = seg * per;
    n = slot0 < p.path_count ? (slot0 + per < p.path_count ? per : p.path_count - slot0) : 0u;
  } else {
    n = (uint32_t)queue_ctl(p.qctl, 0, depth, seg)[QCTL_SIZE];
  }
  const uint32_t first = (blockIdx.x / QUEUE_SEGMENTS) * blockDim.x + threadIdx.x;
  const uint32_t step = ((gridDim.x - seg + QUEUE_SEGMENTS - 1u) / QUEUE_SEGMENTS) * blockDim.x;
  const uint32_t* queue_in = p.queue[depth & 1u] + seg_base;
  uint32_t* queue_out = p.queue[(depth + 1) & 1u] + seg_base;
  float4* shadow_out = p.shadow_rays + 3 * (size_t)seg * p.shadow_stride;
  unsigned long long* queue_size = &queue_ctl(p.qctl, 0, depth + 1, seg)[QCTL_SIZE];
  unsigned long long* shadow_size = &queue_ctl(p.qctl, 1, depth, seg)[QCTL_SIZE];
  const uint32_t W = p.pc.gOutputExtent[0], H = p.pc.gOutputExtent[1];
  const bool connect_views = flag(p, STHIP_eConnectToViews), connect_paths = flag(p, STHIP_eConnectToLightPaths);
  for (uint32_t i = first; i < n; i += step) {
    const uint32_t slot = depth == 0 ? slot0 + i : queue_in[i];
    const uint32_t meta = p.meta[slot];
    if (meta >= 0xFFFFFFFEu) continue;
    const uint32_t seed_index = slot / p.light_threads;
    uint32_t tx, ty, path_index;
    light_thread(p, slot - seed_index * p.light_threads, tx, ty, path_index);
    const float4 ro = p.ray_o[slot], rd = p.ray_d[slot], hh = p.hit[slot], bb = p.beta[slot], bd = p.bdpt[slot];
    const uint32_t hit_leaf = p.hit_leaf[slot];
    const f3 seg_origin = xyz(ro), direction = xyz(rd);  // the ray k_trace traced
    f3 origin = seg_origin;                              // the previous vertex (differs from seg_origin only with MEDIA)
    float bsdf_pdf = ro.w, eta_scale = rd.w;
    f3 beta = xyz(bb);
    float path_pdf = bd.x, path_pdf_rev = bd.y, dVC = bd.z, prev_cos_out = bd.w;
    bool prev_specular = (meta >> 16) & 1u;
    f3 path_contrib = p.path_contrib ? xyz(p.path_contrib[slot]) : F3s(0.0f);
    Rng rng;
    rng.x = tx;
    rng.y = ty;
    rng.seed = p.seed + seed_index;
    rng.counter = __float_as_uint(bb.w);
    uint32_t path_length = meta & 0xFFu, diffuse_vertices = (meta >> 8) & 0xFFu;
    const uint32_t ip = __float_as_uint(hh.w);
    uint32_t medium = 0xFFFFu;
    float T_dir_pdf = 1;
    bool medium_vertex = false;
    f3 scatter_p = F3s(0.0f);
    uint32_t walk_segments = 0;
    uint32_t* column = MEDIA ? p.shade_stack + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * p.bvh.stack_depth : nullptr;
    if (MEDIA) {
      // one step of the medium-aware trace_ray, intersection.hlsli:246-283 (as k_shade's)
      const float4 m0 = p.media_state[2 * (size_t)slot], m1 = p.media_state[2 * (size_t)slot + 1];
      origin = xyz(m0);
      T_dir_pdf = m0.w;
      float T_nee_pdf = m1.x;
      medium = __float_as_uint(m1.y);
      const uint32_t segments = __float_as_uint(m1.z);
      if (medium != 0xFFFFu) {
        Medium mm;
        mm.load(p.scene, load_inst(p.scene, medium).material_address());
        const Xf inv = load_xf(p.scene.inv_xf, medium);
        f3 dir_pdf = F3s(1.0f), nee_pdf = F3s(1.0f);
        const bool scattered = mm.delta_track(p.scene, rng, xf_point(inv, seg_origin), xf_vector(inv, direction), hh.x, beta, dir_pdf, nee_pdf, true, p.pc.gMaxNullCollisions, scatter_p);
        T_dir_pdf *= average3(dir_pdf);
        T_nee_pdf *= average3(nee_pdf);
        medium_vertex = scattered && isfinite(scatter_p.x) && isfinite(scatter_p.y) && isfinite(scatter_p.z);
      }
      if (!medium_vertex && ip != 0xFFFFFFFFu) {
        const uint32_t hit_inst = ip & 0xFFFFu;
        const Inst hin = load_inst(p.scene, hit_inst);
        if (hin.type() == STHIP_INSTANCE_TYPE_VOLUME) {  // a volume boundary: enter or leave, walk on from the other side; no vertex
          if (segments >= 62u) continue;
          f3 bpos;
          uint32_t bn;
          volume_boundary(p.scene, hit_inst, hin, seg_origin, direction, hh.x, bpos, bn);
          const f3 bgn = unpack_normal_octahedron(bn);
          f3 next_origin;
          if (dot3(direction, bgn) < 0) {
            medium = hit_inst;
            next_origin = ray_offset(bpos, -bgn);
          } else {
            medium = 0xFFFFu;
            next_origin = ray_offset(bpos, bgn);
          }
          p.ray_o[slot] = make_float4(next_origin.x, next_origin.y, next_origin.z, bsdf_pdf);
          p.beta[slot] = make_float4(beta.x, beta.y, beta.z, __uint_as_float(rng.counter));
          p.media_state[2 * (size_t)slot] = make_float4(origin.x, origin.y, origin.z, T_dir_pdf);
          p.media_state[2 * (size_t)slot + 1] = make_float4(T_nee_pdf, __uint_as_float(medium), __uint_as_float(segments + 1u), 0.0f);
          const uint32_t k = (uint32_t)atomicAdd(queue_size, 1ull);
          queue_out[k] = slot;
          atomicAdd(&p.counters[CNT_CROSSINGS], 1ull);
          continue;
        }
      }
    }
    bool alive = false;
    f3 new_origin = origin, new_direction = direction;
    float2 cone_out = make_float2(0.0f, 0.0f);
    // the splat of a connect_view whose visibility is known at once (accumulate_light_contribution, path.hlsli:47-60)
    auto splat = [&](uint32_t ix, uint32_t iy, f3 c) {
      uint32_t* lt = p.light_trace + 4 * ((size_t)seed_index * W * H + (size_t)iy * W + ix);
      const float q = (float)p.light_trace_quantization;
      const float cf[3] = {fmaxf(0.0f, c.x) * q, fmaxf(0.0f, c.y) * q, fmaxf(0.0f, c.z) * q};
      uint32_t overflow = 0;
      for (int k = 0; k < 3; k++) {
        const uint32_t ci = cf[k] >= 4294967296.0f ? 0xFFFFFFFFu : (cf[k] == cf[k] ? (uint32_t)cf[k] : 0u);
        if (ci) {
          const uint32_t prev = atomicAdd(&lt[k], ci);
          if (ci > 0xFFFFFFFFu - prev) overflow |= 1u << k;
        }
      }
      if (overflow) atomicOr(&lt[3], overflow);
    };
    // connect_view up to the material's part (path.hlsli:533-560): the pixel, the direction, the sensor's terms. False: no connection.
    struct ViewLink {
      int ix, iy;
      f3 to_view, contribution;
      float dist, G_rev;
    };
    auto view_link = [&](f3 position_v, ViewLink& L) -> bool {
      uint32_t view_index = 0;
      if (p.pc.gViewCount > 1) view_index = (uint32_t)fminf(rng.next_float() * (float)p.pc.gViewCount, (float)(p.pc.gViewCount - 1));
      const sthip_ViewData& view = p.views[view_index];
      float4 sp = project_point(view.projection, xf_point(load_xf(p.inv_view_xf, view_index), position_v));
      sp.y = -sp.y;
      sp.x = sp.x / sp.w;
      sp.y = sp.y / sp.w;
      sp.z = sp.z / sp.w;
      if (fabsf(sp.x) >= 1 || fabsf(sp.y) >= 1 || fabsf(sp.z) >= 1 || sp.z <= 0) return false;
      const float u = sp.x * .5f + .5f, v = sp.y * .5f + .5f;
      L.ix = view.image_min[0] + (int)((float)(view.image_max[0] - view.image_min[0]) * u);
      L.iy = view.image_min[1] + (int)((float)(view.image_max[1] - view.image_min[1]) * v);
      const Xf t = load_xf(p.view_xf, view_index);
      const f3 position = F3(t.r0.w, t.r1.w, t.r2.w);
      const f3 view_normal = normalize3(xf_vector(t, F3(0, 0, 1)));
      L.to_view = position - position_v;
      L.dist = length3(L.to_view);
      L.to_view = L.to_view / L.dist;
      const float sensor_cos_theta = fabsf(dot3(L.to_view, view_normal));
      const float sensor_importance = 1 / (view.projection.sensor_area * 1.0f * (pow2f(sensor_cos_theta) * pow2f(sensor_cos_theta)));
      L.contribution = beta * sensor_importance / (1.0f / (sensor_cos_theta / pow2f(L.dist)));
      L.G_rev = fabsf(prev_cos_out) / len_sqr(origin - position_v);
      return true;
    };
    // ... and its weight (path.hlsli:583-613)
    auto view_weight = [&](float pdf_rev, float G_rev, float G_here) {
      float weight;
      if (flag(p, STHIP_eMIS)) {
        if (connect_paths)  // dL_1, path.hlsli:591-596
          weight = 1 / (1 + connection_dVC(dVC, pdf_rev * G_rev, bsdf_pdf * G_here, prev_specular) * pow2f(1.0f));
        else
          weight = prev_specular ? 1.0f : mis2(true, path_pdf, 1.0f * path_pdf_rev * (pdf_rev * G_rev));
      } else {
        weight = path_weight(p, 1, path_length);
      }
      if (p.debug_mode == STHIP_DEBUG_LIGHT_TRACE_CONTRIBUTION) weight = 1;  // path.hlsli:608-609
      if (p.debug_mode == STHIP_DEBUG_PATH_LENGTH_CONTRIBUTION && p.pc.gDebugViewPathLength == 1)  // :611-613
        weight = p.pc.gDebugLightPathLength == path_length ? 1.0f : 0.0f;
      return weight;
    };
    do {
      // trace(), path.hlsli:1009-1012
      if ((MEDIA && T_dir_pdf <= 0) || all_le0(beta)) break;
      if (MEDIA) {
        beta = beta / T_dir_pdf;
        if (!flag(p, STHIP_eDeferShadowRays)) bsdf_pdf *= T_dir_pdf;
        path_pdf *= T_dir_pdf;
      }
      path_length++;
      if (MEDIA && medium_vertex) {
        // ---- a vertex inside a medium: trace()'s tail (path.hlsli:1033-1043) and next_vertex(Medium), light branch (:955-998) ----
        Medium mm;
        mm.load(p.scene, load_inst(p.scene, medium).material_address());
        const float dist2 = len_sqr(scatter_p - origin);
        const float G = 1 / dist2;  // no cosine at a medium vertex (ngdotin = 1)
        path_pdf *= bsdf_pdf * G;
        if (p.path_contrib) path_contrib = path_contrib * G;
        const f3 local_dir_in = -direction;
        if (!mm.can_eval() || path_length >= p.pc.gMaxPathVertices) break;
        if (!mm.is_specular()) {
          diffuse_vertices++;
          if (diffuse_vertices > p.pc.gMaxDiffuseVertices) break;
          if (connect_paths && path_length + 2 <= p.pc.gMaxPathVertices && diffuse_vertices < p.pc.gMaxDiffuseVertices) {
            // vertex() / store_light_vertex() at a vertex inside a medium (path.hlsli:491-531): PATH_VERTEX_FLAG_IS_MEDIUM; the normals,
            // tangent and uv upstream stores are the stale ones of the last surface query and nothing reads them: pinned to 0
            const size_t per_seed = (size_t)p.pc.gLightPathCount * p.pc.gMaxDiffuseVertices;
            const size_t idx = (size_t)W * H * (diffuse_vertices - 1) + path_index;
            if (p.lvc_staging || idx < per_seed) {
              float4* lv = p.lvc_staging ? p.lvc_staging + 4 * (((size_t)seed_index * p.pc.gLightPathCount + path_index) * (p.pc.gMaxDiffuseVertices - 1) + (diffuse_vertices - 1))
                                         : p.light_vertices + 4 * ((size_t)seed_index * per_seed + idx);
              const uint32_t vflags = 2u | 4u | (prev_specular ? 8u : 0u);
              const f3 stored = flag(p, STHIP_eLVCReservoirs) ? path_contrib : beta;  // path.hlsli:513
              const uint32_t pb0 = det_f32tof16(stored.x) | (det_f32tof16(stored.y) << 16);
              const uint32_t pb1 = det_f32tof16(stored.z) | ((path_length & 0x7Fu) << 16) | ((diffuse_vertices & 0x1Fu) << 23) | (vflags << 28);
              lv[0] = make_float4(scatter_p.x, scatter_p.y, scatter_p.z, __uint_as_float(0u));
              lv[1] = make_float4(__uint_as_float(load_inst(p.scene, medium).material_address()), __uint_as_float(pack_normal_octahedron(local_dir_in)), __uint_as_float(0u), __uint_as_float(0u));
              lv[2] = make_float4(0.0f, 0.0f, __uint_as_float(pb0), __uint_as_float(pb1));
              lv[3] = make_float4(dVC, prev_cos_out / len_sqr(origin - scatter_p), bsdf_pdf * G, path_pdf);
            }
          }
          if (connect_views) do {  // connect_view at a medium vertex: no geometry (path.hlsli:562-565), the phase function as f and both pdfs
            ViewLink L;
            if (!view_link(scatter_p, L)) break;
            const float v = mm.phase(local_dir_in, L.to_view);
            if (v < 1e-6f) break;
            f3 contribution = L.contribution * v;
            if (all_le0(contribution)) break;
            float nee_pdf = 1, dir_pdf = 1;
            walk_segments += visibility_walk_media(p, rng, scatter_p, L.to_view, L.dist, medium, contribution, dir_pdf, nee_pdf, column);
            if (nee_pdf > 0) contribution = contribution / nee_pdf;
            const float weight = view_weight(v, L.G_rev, G);
            const f3 c = contribution * weight;
            if (L.ix < 0 || L.iy < 0 || (uint32_t)L.ix >= W || (uint32_t)L.iy >= H) break;
            if (p.shard_count > 1 && (((uint32_t)L.iy / p.tile_h) * p.tiles_x + (uint32_t)L.ix / p.tile_w) % p.shard_count != p.shard_rank) break;
            splat((uint32_t)L.ix, (uint32_t)L.iy, c);
          } while (0);
        }
        // sample_direction with the phase function (path.hlsli:898-952, medium branch)
        const float s0 = rng.next_float(), s1 = rng.next_float(), s2 = rng.next_float();
        (void)s2;
        float ppdf, proughness;
        const f3 dir_out = mm.sample(s0, s1, local_dir_in, ppdf, proughness);
        if (p.path_contrib) path_contrib = path_contrib * ppdf;  // path.hlsli:901 (Medium::sample returns the phase value)
        if (ppdf < 1e-6f) break;
        if (TEXTURED) {
          const float2 cone = p.cone[slot];
          float rd_radius = cone.x, rd_spread = cone.y;
          if (flag(p, STHIP_eRayCones)) {
            rd_radius += rd_spread * sqrtf(dist2);
            rd_spread = fmaxf(0.0f, lerp1((rd_spread + 2 * 0.0f * rd_radius) / -1.0f, 0.2f, proughness));
          }
          cone_out = make_float2(rd_radius, rd_spread);
        }
        {
          const float G_rev = prev_cos_out / len_sqr(origin - scatter_p);
          path_pdf_rev *= ppdf * G_rev;
          dVC = connection_dVC(dVC, ppdf * G_rev, bsdf_pdf * G, mm.is_specular());
          prev_specular = mm.is_specular();
        }
        bsdf_pdf = ppdf;
        prev_cos_out = 1;
        new_origin = scatter_p;
        new_direction = dir_out;
        alive = true;
        break;
      }
      if (ip == 0xFFFFFFFFu) break;  // light paths that leave the scene end (no environment with light tracing)
      const uint32_t inst_index = ip & 0xFFFFu;
      const Inst in = load_inst(p.scene, inst_index);
      ShadingData sd;
      if (EXT && in.type() == STHIP_INSTANCE_TYPE_SPHERE) {
        const Xf inv = load_xf(p.scene.inv_xf, inst_index);
        const float im[12] = {inv.r0.x, inv.r0.y, inv.r0.z, inv.r0.w, inv.r1.x, inv.r1.y, inv.r1.z, inv.r1.w, inv.r2.x, inv.r2.y, inv.r2.z, inv.r2.w};
        make_sphere_shading_data(p.scene, sd, inst_index, in, obj_point(im, seg_origin) + obj_vector(im, direction) * hh.x);
      } else {
        make_hit_shading_data(p.scene, sd, inst_index, hit_leaf, hh.y, hh.z, TEXTURED && flag(p, STHIP_eFlipTriangleUVs));
      }
      const f3 gn = sd.geometry_normal();
      const float dist2 = len_sqr(sd.position - origin);
      float G = 1 / dist2;
      const float ngdotin = -dot3(direction, gn);
      G *= fabsf(ngdotin);
      path_pdf *= bsdf_pdf * G;  // path.hlsli:1042
      if (p.path_contrib) path_contrib = path_contrib * G;  // path.hlsli:1043
      DisneyMaterial m;
      float rd_radius = 0, rd_spread = 0;  // RayDifferential of the light path: starts at (0, 0), bounces spread it (path.hlsli:911-916)
      if (TEXTURED) {
        const float2 cone = p.cone[slot];
        rd_radius = cone.x;
        rd_spread = cone.y;
        if (flag(p, STHIP_eRayCones)) {  // path.hlsli:1026-1029
          rd_radius += rd_spread * sqrtf(dist2);
          sd.uv_screen_size *= rd_radius;
        }
        m.load_textured(p.scene, in.material_address(), sd.u, sd.v, sd.uv_screen_size, sd.packed_shading_normal, sd.packed_tangent, p.sampling_flags);
      } else {
        m.load(p.scene, in.material_address());
      }
      const Frame3 frame = make_frame(sd);
      const f3 local_dir_in = normalize3(frame.to_local(-direction));
      // next_vertex(BSDF), path.hlsli:955-998, light branch
      if (!m.can_eval() || path_length >= p.pc.gMaxPathVertices) break;
      const float ngdotns = dot3(gn, sd.shading_normal());
      const bool fix = flag(p, STHIP_eShadingNormalShadowFix);
      if (!m.is_specular()) {
        diffuse_vertices++;
        if (diffuse_vertices > p.pc.gMaxDiffuseVertices) break;
        if (connect_paths && path_length + 2 <= p.pc.gMaxPathVertices && diffuse_vertices < p.pc.gMaxDiffuseVertices) {
          // vertex() / store_light_vertex(), path.hlsli:491-531: slot light_vertex_index(path_index, diffuse_vertices) (:64)
          const size_t per_seed = (size_t)p.pc.gLightPathCount * p.pc.gMaxDiffuseVertices;
          const size_t idx = (size_t)W * H * (diffuse_vertices - 1) + path_index;
          if (p.lvc_staging || idx < per_seed) {
            // eLVC: staged per (path, vertex) and compacted in that order afterwards (see FrameParams::lvc_staging)
            float4* lv = p.lvc_staging ? p.lvc_staging + 4 * (((size_t)seed_index * p.pc.gLightPathCount + path_index) * (p.pc.gMaxDiffuseVertices - 1) + (diffuse_vertices - 1))
                                       : p.light_vertices + 4 * ((size_t)seed_index * per_seed + idx);
            const uint32_t vflags = 2u | (prev_specular ? 8u : 0u);  // IS_BACKGROUND as upstream sets it, IS_PREV_DELTA
            const f3 stored = flag(p, STHIP_eLVCReservoirs) ? path_contrib : beta;  // path.hlsli:513
            const uint32_t pb0 = det_f32tof16(stored.x) | (det_f32tof16(stored.y) << 16);
            const uint32_t pb1 = det_f32tof16(stored.z) | ((path_length & 0x7Fu) << 16) | ((diffuse_vertices & 0x1Fu) << 23) | (vflags << 28);
            lv[0] = make_float4(sd.position.x, sd.position.y, sd.position.z, __uint_as_float(sd.packed_geometry_normal));
            lv[1] = make_float4(__uint_as_float(in.material_address()), __uint_as_float(pack_normal_octahedron(local_dir_in)), __uint_as_float(sd.packed_shading_normal),
                                __uint_as_float(sd.packed_tangent));
            lv[2] = make_float4(sd.u, sd.v, __uint_as_float(pb0), __uint_as_float(pb1));
            lv[3] = make_float4(dVC, prev_cos_out / len_sqr(origin - sd.position), bsdf_pdf * G, path_pdf);
          }
        }
        if (connect_views) do {  // connect_view, path.hlsli:533-613
          ViewLink L;
          if (!view_link(sd.position, L)) break;
          const f3 to_view = L.to_view;
          const float dist = L.dist;
          f3 contribution = L.contribution;
          const float ngdotout = dot3(to_view, gn);
          const f3 ray_origin = ray_offset(sd.position, ngdotout > 0 ? gn : -gn);
          const f3 local_to_view = normalize3(frame.to_local(to_view));
          contribution = contribution * shading_normal_correction(local_dir_in.z, local_to_view.z, ngdotin, ngdotout, ngdotns, fix, true);
          MaterialEvalRecord ev;
          m.eval(ev, local_dir_in, local_to_view, true);
          if (ev.pdf_fwd < 1e-6f) break;
          contribution = contribution * ev.f;
          if (all_le0(contribution)) break;
          if (MEDIA) {  // the visibility ray walks through the media now, in this path's stream (path.hlsli:577-581)
            float nee_pdf = 1, dir_pdf = 1;
            walk_segments += visibility_walk_media(p, rng, ray_origin, to_view, dist, medium, contribution, dir_pdf, nee_pdf, column);
            if (nee_pdf > 0) contribution = contribution / nee_pdf;
          }
          const float weight = view_weight(ev.pdf_rev, L.G_rev, G);
          const f3 c = contribution * weight;
          const int ix = L.ix, iy = L.iy;
          // the splat lands on pixel (ix, iy) of this seed's image; a sharded renderer keeps only its own pixels
          if (ix < 0 || iy < 0 || (uint32_t)ix >= W || (uint32_t)iy >= H) break;
          if (p.shard_count > 1 && (((uint32_t)iy / p.tile_h) * p.tiles_x + (uint32_t)ix / p.tile_w) % p.shard_count != p.shard_rank) break;
          if (MEDIA || !(dist > 1e-6f)) {  // visibility known: with media the walk above; a zero-length ray never runs trace_visibility_ray's loop
            splat((uint32_t)ix, (uint32_t)iy, c);
            break;
          }
          const uint32_t k = (uint32_t)atomicAdd(shadow_size, 1ull);
          shadow_out[3 * (size_t)k] = make_float4(ray_origin.x, ray_origin.y, ray_origin.z, dist);
          shadow_out[3 * (size_t)k + 1] = make_float4(to_view.x, to_view.y, to_view.z, __uint_as_float(0x80000000u | (uint32_t)((size_t)seed_index * W * H + (size_t)iy * W + ix)));
          shadow_out[3 * (size_t)k + 2] = make_float4(c.x, c.y, c.z, 0.0f);
        } while (0);
      }
      // sample_direction with the adjoint BSDF, path.hlsli:898-952
      const float s0 = rng.next_float(), s1 = rng.next_float(), s2 = rng.next_float();
      MaterialSampleRecord ms;
      const f3 sampled_f = m.sample(ms, F3(s0, s1, s2), local_dir_in, beta, true);
      if (p.path_contrib) path_contrib = path_contrib * sampled_f;  // path.hlsli:901
      if (ms.pdf_fwd < 1e-6f) break;
      if (ms.eta != 0) eta_scale /= pow2f(ms.eta);
      if (TEXTURED && flag(p, STHIP_eRayCones)) {
        float spec_spread = rd_spread + 2 * sd.mean_curvature * rd_radius;
        if (ms.eta != 0) spec_spread = spec_spread / ms.eta;
        rd_spread = fmaxf(0.0f, lerp1(spec_spread, 0.2f, ms.roughness));
      }
      cone_out = make_float2(rd_radius, rd_spread);
      {
        const float G_rev = prev_cos_out / len_sqr(origin - sd.position);
        path_pdf_rev *= ms.pdf_rev * G_rev;
        dVC = connection_dVC(dVC, ms.pdf_rev * G_rev, bsdf_pdf * G, m.is_specular());
        prev_specular = m.is_specular();
      }
      bsdf_pdf = ms.pdf_fwd;
      const float ndotout = ms.dir_out.z;
      const f3 dir_out = normalize3(frame.to_world(ms.dir_out));
      const float ngdotout = dot3(gn, dir_out);
      new_origin = ray_offset(sd.position, ngdotout > 0 ? gn : -gn);
      beta = beta * shading_normal_correction(local_dir_in.z, ndotout, ngdotin, ngdotout, ngdotns, fix, true);
      prev_cos_out = ngdotout;
      if (all_le0(beta)) break;
      new_direction = dir_out;
      alive = true;
    } while (0);
    if (MEDIA && walk_segments) atomicAdd(&p.counters[CNT_RAYS_SHADOW], (unsigned long long)walk_segments);  // (inline walks: not in any queue)
    if (alive) {
      p.ray_o[slot] = make_float4(new_origin.x, new_origin.y, new_origin.z, bsdf_pdf);
      p.ray_d[slot] = make_float4(new_direction.x, new_direction.y, new_direction.z, eta_scale);
      p.beta[slot] = make_float4(beta.x, beta.y, beta.z, __uint_as_float(rng.counter));
      p.bdpt[slot] = make_float4(path_pdf, path_pdf_rev, dVC, prev_cos_out);
      if (p.path_contrib) p.path_contrib[slot] = make_float4(path_contrib.x, path_contrib.y, path_contrib.z, 0.0f);
      if (TEXTURED) p.cone[slot] = cone_out;
      if (MEDIA) {  // a new trace() starts at this vertex
        p.media_state[2 * (size_t)slot] = make_float4(new_origin.x, new_origin.y, new_origin.z, 1.0f);
        p.media_state[2 * (size_t)slot + 1] = make_float4(1.0f, __uint_as_float(medium), __uint_as_float(0u), 0.0f);
      }
      p.meta[slot] = path_length | (diffuse_vertices << 8) | (prev_specular ? 1u << 16 : 0u);
      const uint32_t k = (uint32_t)atomicAdd(queue_size, 1ull);
      queue_out[k] = slot;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// shade: the tail of trace() (path.hlsli:1012-1043), the first-hit block of sample_visibility
// (bdpt.hlsl:222-296) at depth 0, then next_vertex() (path.hlsli:955-998,1048-1075) up to the point
// where the next ray is known
// ---------------------------------------------------------------------------------------------
// TEXTURED: the scene binds images; ray cones, image values and normal maps are evaluated
// EXT: the scene has sphere instances or an environment: sphere hits, sphere lights, environment
// emission and environment light sampling. Scenes without them run the instantiation that carries none of it.
// LT: eConnectToViews is on: the view path carries the BDPT quantities (path_pdf, path_pdf_rev, dVC, prev_specular) and
// uses the weights of path.hlsli:341-351,870-880 (only instantiated with EXT).
// MEDIA: the scene has volume instances (BDPT_FLAG_HAS_MEDIA). A path's trace() is then a walk from volume boundary to
// volume boundary (the medium-aware trace_ray, intersection.hlsli:240-285): every k_trace round delivers one segment's
// closest hit, and this kernel first runs that segment's delta tracking and boundary logic — the path either crosses a
// boundary (it is queued again with its new origin and medium, no vertex), scatters inside the medium (a medium vertex:
// phase function instead of a material), or arrives at a surface / leaves the scene as before. Only instantiated with
// EXT and without LT.
// PROBE: the first half of a round with eCoherentRR (see FrameParams::rr): the same code up to the Russian roulette, where a
// path records its p and the random number it would draw, and nothing else is written.
// DEBUG: BDPTDebugMode is live (FrameParams::debug_mode): the statements that feed gDebugImage are compiled in (the general
// instantiations only: TEXTURED and EXT)
template <bool TEXTURED, bool EXT, bool LT = false, int MEDIA = 0, bool PROBE = false, bool DEBUG = false>  // MEDIA: 1 = volumes, NEE walks deferred (k_shadow_media); 2 = volumes, NEE walks inline (visibility_walk_media: an instantiation of its own, its registers are not the deferred form's business)
// Resident blocks per CU: the plain instantiations (the bench path; textured or not) run best at SHADE_BLOCKS = 3 (168 registers,
// a handful of spills); the extended ones (EXT: spheres, environments, reservoirs, media; LT: light subpaths) spill 200-950
// registers there and are 4-8 % faster at 2 (256 registers): EXPERIMENTS.md round 4. The probes without light subpaths do not spill: 3.
__global__ void __launch_bounds__(STHIP_BLOCK, (LT || (EXT && !PROBE)) ? 2 : SHADE_BLOCKS) k_shade(FrameParams p, uint32_t depth) {
  // gMaterialData staged in LDS (untextured instantiations; the table of a scene is a few KB): a vertex then reads its 72-byte
  // record with LDS reads instead of a divergent gather. p.lds_material_bytes = 0 turns it off (table too large, option).
  extern __shared__ uint32_t shade_lds[];
  if (!TEXTURED && p.lds_material_bytes) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(p.scene.materials);
    for (uint32_t i = threadIdx.x; i < (p.lds_material_bytes >> 2); i += STHIP_BLOCK) shade_lds[i] = src[i];
    __syncthreads();
  }
  // Workgroup b works on segment b % 8 (its XCD's) of the incoming queue and appends to the same segment of the
  // outgoing queues, so a segment never grows. In the first bounce a segment is a contiguous eighth of the slots
  // (the same split WaveWork uses): an XCD keeps one part of the image, and the part of the scene seen from it,
  // from kernel to kernel.
  const uint32_t seg = blockIdx.x % QUEUE_SEGMENTS;
  const uint32_t seg_base = seg * p.seg_stride;
  uint32_t n, slot0 = 0;
  if (depth == 0) {
    const uint32_t per = (((p.path_count + QUEUE_SEGMENTS - 1u) / QUEUE_SEGMENTS) + 63u) & ~63u;
    slot0 = seg * per;
    n = slot0 < p.path_count ? (slot0 + per < p.path_count ? per : p.path_count - slot0) : 0u;
  } else {
    n = (uint32_t)queue_ctl(p.qctl, 0, depth, seg)[p.culled ? QCTL_KEPT : QCTL_SIZE];
  }
  const uint32_t first = (blockIdx.x / QUEUE_SEGMENTS) * blockDim.x + threadIdx.x;
  const uint32_t step = ((gridDim.x - seg + QUEUE_SEGMENTS - 1u) / QUEUE_SEGMENTS) * blockDim.x;
  const uint32_t* queue_in = p.queue[depth & 1u] + seg_base;
  uint32_t* queue_out = p.queue[(depth + 1) & 1u] + seg_base;
  const size_t shadow_base = (size_t)seg * p.shadow_stride + ((MEDIA && (depth & 1u)) ? p.shadow_alt : 0u);
  float4* shadow_out = p.shadow_rays + 3 * shadow_base;
  unsigned long long* queue_size = &queue_ctl(p.qctl, 0, depth + 1, seg)[QCTL_SIZE];
  unsigned long long* shadow_size = &queue_ctl(p.qctl, 1, depth, seg)[QCTL_SIZE];
  const bool use_nee = flag(p, STHIP_eNEE);
  const bool use_mis = flag(p, STHIP_eMIS);
  const bool sample_bsdfs = flag(p, STHIP_eSampleBSDFs);
  const bool has_env = EXT && (p.scene_flags & STHIP_BDPT_FLAG_HAS_ENVIRONMENT) != 0;
  const bool has_emissives = !EXT || (p.scene_flags & STHIP_BDPT_FLAG_HAS_EMISSIVES) != 0;
  for (uint32_t i = first; i < n; i += step) {
    const uint32_t slot = depth == 0 ? slot0 + i : queue_in[i];
    uint32_t meta = p.meta[slot];
    if (meta >= 0xFFFFFFFEu) continue;
    uint32_t px, py;
    slot_to_pixel(p, slot, px, py);
    const float4 ro = p.ray_o[slot], rd = p.ray_d[slot], hh = p.hit[slot], bb = p.beta[slot];
    const uint32_t hit_leaf = p.hit_leaf[slot];  // (the slots of misses hold stale values: only read behind `ip != miss`)
    const f3 seg_origin = xyz(ro), direction = xyz(rd);  // the ray k_trace traced
    f3 origin = seg_origin;                              // the previous vertex (differs from seg_origin only with MEDIA)
    float bsdf_pdf = ro.w;
    float eta_scale = rd.w;
    f3 beta = xyz(bb);
    Rng rng;
    rng.x = px;
    rng.y = py;
    const uint32_t seed_index = slot / p.paths_per_seed;
    rng.seed = p.seed + seed_index;
    rng.counter = __float_as_uint(bb.w);
    uint32_t path_length = meta & 0xFFu, diffuse_vertices = (meta >> 8) & 0xFFu;
    float path_pdf = 1, path_pdf_rev = 1, dVC = 1, prev_cos_out = 1;
    bool prev_specular = false;
    if (LT) {
      const float4 bd = p.bdpt[slot];
      path_pdf = bd.x;
      path_pdf_rev = bd.y;
      dVC = bd.z;
      prev_cos_out = bd.w;
      prev_specular = (meta >> 16) & 1u;
    }
    const uint32_t ip = __float_as_uint(hh.w);
    const bool primary = MEDIA ? path_length == 1 : depth == 0;  // this vertex is the first one of the path
    uint32_t medium = 0xFFFFu;   // _medium: the volume instance the path is inside of
    float T_dir_pdf = 1, T_nee_pdf = 1;
    uint32_t walk_segments = 0;  // closest-hit queries of this vertex's inline visibility walk (MEDIA without eDeferShadowRays)
    bool medium_vertex = false;  // the vertex is a scattering event inside `medium`
    f3 scatter_p = F3s(0.0f);
    if (MEDIA) {
      // one step of the medium-aware trace_ray, intersection.hlsli:246-283
      const float4 m0 = p.media_state[2 * (size_t)slot], m1 = p.media_state[2 * (size_t)slot + 1];
      origin = xyz(m0);
      T_dir_pdf = m0.w;
      T_nee_pdf = m1.x;
      medium = __float_as_uint(m1.y);
      const uint32_t segments = __float_as_uint(m1.z);
      if (medium != 0xFFFFu) {
        Medium mm;
        mm.load(p.scene, load_inst(p.scene, medium).material_address());
        const Xf inv = load_xf(p.scene.inv_xf, medium);
        const float im[12] = {inv.r0.x, inv.r0.y, inv.r0.z, inv.r0.w, inv.r1.x, inv.r1.y, inv.r1.z, inv.r1.w, inv.r2.x, inv.r2.y, inv.r2.z, inv.r2.w};
        f3 dir_pdf = F3s(1.0f), nee_pdf = F3s(1.0f);
        // transform_point / transform_vector of the inverse instance transform, as the reference writes them
        const bool scattered = mm.delta_track(p.scene, rng, xf_point(inv, seg_origin), xf_vector(inv, direction), hh.x, beta, dir_pdf, nee_pdf, true, p.pc.gMaxNullCollisions, scatter_p);
        (void)im;
        T_dir_pdf *= average3(dir_pdf);
        T_nee_pdf *= average3(nee_pdf);
        medium_vertex = scattered && isfinite(scatter_p.x) && isfinite(scatter_p.y) && isfinite(scatter_p.z);
      }
      if (!medium_vertex && ip != 0xFFFFFFFFu) {
        const uint32_t hit_inst = ip & 0xFFFFu;
        const Inst hin = load_inst(p.scene, hit_inst);
        if (hin.type() == STHIP_INSTANCE_TYPE_VOLUME) {
          // a volume boundary: enter or leave, and walk on from the other side of it; no vertex
          if (segments >= 62u) continue;  // the reference gives up after 64 segments; the rounds of a render end before that
          f3 bpos;
          uint32_t bn;
          volume_boundary(p.scene, hit_inst, hin, seg_origin, direction, hh.x, bpos, bn);
          const f3 bgn = unpack_normal_octahedron(bn);
          f3 next_origin;
          if (dot3(direction, bgn) < 0) {  // SHADING_FLAG_FRONT_FACE: entering
            medium = hit_inst;
            next_origin = ray_offset(bpos, -bgn);
          } else {
            medium = 0xFFFFu;
            next_origin = ray_offset(bpos, bgn);
          }
          p.ray_o[slot] = make_float4(next_origin.x, next_origin.y, next_origin.z, bsdf_pdf);
          p.beta[slot] = make_float4(beta.x, beta.y, beta.z, __uint_as_float(rng.counter));
          p.media_state[2 * (size_t)slot] = make_float4(origin.x, origin.y, origin.z, T_dir_pdf);
          p.media_state[2 * (size_t)slot + 1] = make_float4(T_nee_pdf, __uint_as_float(medium), __uint_as_float(segments + 1u), 0.0f);
          const uint32_t k = (uint32_t)atomicAdd(queue_size, 1ull);
          queue_out[k] = slot;
          atomicAdd(&p.counters[CNT_CROSSINGS], 1ull);
          continue;
        }
      }
    }
    f3 radiance = (!MEDIA && depth == 0) ? F3s(0.0f) : xyz(p.radiance[slot]);
    bool radiance_dirty = !MEDIA && depth == 0;  // the first round writes the sum (k_generate leaves it unwritten), later ones only a changed one
    const bool connect_paths = LT && flag(p, STHIP_eConnectToLightPaths);
    if (connect_paths && depth > 0 && !PROBE) {
      // connect_light_subpath's accumulate_contribution calls of the previous vertex (path.hlsli:802-822), whose
      // visibility rays k_trace has resolved by now: added here, in i order, before anything this vertex adds
      float4* cn = p.conn + (size_t)slot * (p.pc.gMaxDiffuseVertices - 1);
      for (uint32_t k = 0; k + 1 < p.pc.gMaxDiffuseVertices; k++) {
        const float4 c = cn[k];
        if (c.x != 0 || c.y != 0 || c.z != 0) {
          radiance = radiance + xyz(c);
          radiance_dirty = true;
          cn[k] = make_float4(0, 0, 0, 0);
        }
      }
    }
    const size_t pixel = (size_t)py * p.pc.gOutputExtent[0] + px;
    // gDebugImage[pixel_coord] as this path has left it (DEBUG)
    float4 dbg = make_float4(0, 0, 0, 0);
    bool dbg_dirty = false;
    if (DEBUG) dbg = p.debug[slot];
    auto debug_set = [&](f3 c) {
      dbg = make_float4(c.x, c.y, c.z, 1.0f);
      dbg_dirty = true;
    };
    auto debug_add = [&](f3 c) {
      dbg.x = dbg.x + c.x;
      dbg.y = dbg.y + c.y;
      dbg.z = dbg.z + c.z;
      dbg_dirty = true;
    };
    auto debug_is = [&](uint32_t mode) { return DEBUG && p.debug_mode == mode; };
    // bdpt.hlsl:294-295: how far the first vertex moved on the screen since the previous frame, in pixels
    auto debug_prev_uv = [&](uint32_t instance, f3 position) {
      const int view_index = get_view_index(p, px, py);
      const sthip_ViewData& view = p.views[view_index];
      const float ex = (float)(view.image_max[0] - view.image_min[0]), ey = (float)(view.image_max[1] - view.image_min[1]);
      const float uvx = ((float)px + 0.5f - (float)view.image_min[0]) / ex, uvy = ((float)py + 0.5f - (float)view.image_min[1]) / ey;
      const Xf prev_inv_view = load_xf(p.prev_inv_view_xf, (uint32_t)view_index);
      const f3 prev_cam_pos = xf_point(xf_mul(prev_inv_view, load_xf(p.scene.motion_xf, instance)), position);
      float4 pc4 = project_point(p.prev_views[view_index].projection, prev_cam_pos);
      pc4.y = -pc4.y;
      pc4.x = pc4.x / pc4.w;
      pc4.y = pc4.y / pc4.w;
      debug_set(F3(fabsf((pc4.x * .5f + .5f) - uvx) * (float)p.pc.gOutputExtent[0], fabsf((pc4.y * .5f + .5f) - uvy) * (float)p.pc.gOutputExtent[1], 0.0f));
    };
    // eval_bsdf(PathVertex, dir_out, adjoint), path.hlsli:100-123, of a stored light vertex (4 x float4, bdpt.h:108-156) towards
    // dir_out: a surface vertex loads its material again at the stored uv with a zero footprint (a normal map perturbs the stored,
    // already perturbed frame once more, as upstream); a vertex inside a medium (PATH_VERTEX_FLAG_IS_MEDIUM) is its phase function
    // with the stored direction, no geometry. False: f = 0.
    auto eval_light_vertex = [&](const float4* lvp, f3 dir_out, MaterialEvalRecord& lev, float& cos_theta_light) -> bool {
      const float4 v0 = lvp[0], v1 = lvp[1], v2 = lvp[2];
      const uint32_t pb1 = __float_as_uint(v2.w);
      const f3 lv_dir_in = unpack_normal_octahedron(__float_as_uint(v1.y));
      if (MEDIA && ((pb1 >> 28) & 4u)) {
        Medium lmm;
        lmm.load(p.scene, __float_as_uint(v1.x));
        if (lmm.is_specular()) return false;
        const float ph = lmm.phase(lv_dir_in, dir_out);
        lev.f = F3s(ph);
        lev.pdf_fwd = lev.pdf_rev = ph;
        cos_theta_light = 1;
        return true;
      }
      uint32_t lv_ns = __float_as_uint(v1.z), lv_tg = __float_as_uint(v1.w);
      DisneyMaterial lm;
      if (TEXTURED)
        lm.load_textured(p.scene, __float_as_uint(v1.x), v2.x, v2.y, 0.0f, lv_ns, lv_tg, p.sampling_flags);
      else
        lm.load(p.scene, __float_as_uint(v1.x));
      if (lm.is_specular()) return false;
      Frame3 lf;
      lf.n = unpack_normal_octahedron(lv_ns);
      lf.t = unpack_normal_octahedron(lv_tg);
      lf.b = cross3(lf.n, lf.t) * (((pb1 >> 28) & 1u) ? -1.0f : 1.0f);
      const f3 lv_dir_out = normalize3(lf.to_local(dir_out));
      lm.eval(lev, lv_dir_in, lv_dir_out, true);
      if (lev.pdf_fwd < 1e-6f) return false;
      const f3 lv_ng = unpack_normal_octahedron(__float_as_uint(v0.w));
      cos_theta_light = dot3(lv_ng, dir_out);
      lev.f = lev.f * shading_normal_correction(lv_dir_in.z, lv_dir_out.z, dot3(lv_ng, normalize3(lf.to_world(lv_dir_in))), cos_theta_light, dot3(lv_ng, lf.n),
                                                flag(p, STHIP_eShadingNormalShadowFix), true);
      return true;
    };
    // accumulate_contribution's debug half (path.hlsli:302-303): the unweighted contribution of one (view, light) length pair
    auto debug_path_length = [&](f3 contrib, uint32_t view_length, uint32_t light_length) {
      if (debug_is(STHIP_DEBUG_PATH_LENGTH_CONTRIBUTION) && p.pc.gDebugLightPathLength == light_length && view_length == p.pc.gDebugViewPathLength) debug_add(contrib);
    };
    bool alive = false;
    float rd_radius = 0, rd_spread = 0;  // RayDifferential of this path (TEXTURED only)
    f3 new_origin = origin, new_direction = direction;

    do {
      // trace(), path.hlsli:1009-1010: the ray was traced (and counted) even if beta died meanwhile
      if ((MEDIA && T_dir_pdf <= 0) || all_le0(beta)) break;
      if (MEDIA) beta = beta / T_dir_pdf;  // path.hlsli:1009 (bsdf_pdf keeps its value with eDeferShadowRays, :1010)
      if (MEDIA && !flag(p, STHIP_eDeferShadowRays)) bsdf_pdf *= T_dir_pdf;  // (the RESOLVED flag: without eNEE and eLVC the host clears it, BDPT.cpp:522-523 — so also where no NEE ray exists to walk inline)
      if (LT && MEDIA) path_pdf *= T_dir_pdf;  // path.hlsli:1012
      path_length++;
      if (MEDIA && medium_vertex) {
        // ---- a vertex inside a medium: trace()'s tail (path.hlsli:1033-1043) and next_vertex(Medium) (:955-998,1062-1066) ----
        const uint32_t maddr = load_inst(p.scene, medium).material_address();
        Medium mm;
        mm.load(p.scene, maddr);
        const float dist2 = len_sqr(scatter_p - origin);
        float G = 1 / dist2;  // no cosine at a medium vertex (path.hlsli:1035-1036)
        if (LT) path_pdf *= bsdf_pdf * G;  // path.hlsli:1042
        if (primary) {  // bdpt.hlsl:213-220,245-296: no albedo / emission for a medium vertex; the visibility normal and
          bsdf_pdf = 1;   // the depth derivatives read the stale surface of the last query upstream and are pinned to 0
          G = 1;
          if (LT) path_pdf = path_pdf_rev = dVC = 1;
          if (DEBUG) {    // (bdpt.hlsl:222-223: the stale normals again, pinned likewise; :294-295)
            if (p.debug_mode == STHIP_DEBUG_GEOMETRY_NORMAL || p.debug_mode == STHIP_DEBUG_SHADING_NORMAL) debug_set(unpack_normal_octahedron(0u) * .5f + F3s(.5f));
            else if (p.debug_mode == STHIP_DEBUG_PREV_UV) debug_prev_uv(medium, scatter_p);
          }
          if (p.write_aov && seed_index == 0) {
            const int view_index = get_view_index(p, px, py);
            if (p.out_visibility) {
              sthip_VisibilityInfo vis;
              vis.instance_primitive_index = medium | 0xFFFF0000u;
              vis.packed_normal = 0;
              p.out_visibility[pixel] = vis;
            }
            const Xf prev_inv_view = load_xf(p.prev_inv_view_xf, (uint32_t)view_index);
            const f3 prev_cam_pos = xf_point(xf_mul(prev_inv_view, load_xf(p.scene.motion_xf, medium)), scatter_p);
            if (p.out_depth) {
              sthip_DepthInfo dpt;
              dpt.z = length3(scatter_p - origin);
              dpt.prev_z = length3(prev_cam_pos);
              dpt.dz_dxy[0] = dpt.dz_dxy[1] = 0;
              p.out_depth[pixel] = dpt;
            }
            if (p.out_prev_uv) {
              float4 pc4 = project_point(p.prev_views[view_index].projection, prev_cam_pos);
              pc4.y = -pc4.y;
              pc4.x = pc4.x / pc4.w;
              pc4.y = pc4.y / pc4.w;
              p.out_prev_uv[pixel] = make_float2(pc4.x * .5f + .5f, pc4.y * .5f + .5f);
            }
          }
        }
        if (!(any_gt0(beta) && !any_nan(beta))) break;
        const f3 local_dir_in = -direction;
        if (!mm.can_eval() || path_length >= p.pc.gMaxPathVertices) break;
        if (!mm.is_specular()) {
          diffuse_vertices++;
          if (diffuse_vertices > p.pc.gMaxDiffuseVertices) break;
          if (path_length >= p.pc.gMinPathVertices) {
            const float rr = luminance3(beta) / eta_scale * 0.95f;
            if (!(rr >= 1)) {
              if (rng.next_float() > rr) break;
              beta = beta / rr;
              if (LT) path_pdf *= rr;  // path.hlsli:842
            }
          }
          if (use_nee) do {
            // connect_light at a medium vertex, path.hlsli:311-366 with DirectLightSample::setup's medium branch (:207-212):
            // no ray offset, no distance epsilon, no shading-normal terms; the phase function is f and both pdfs
            f3 cLe = F3s(0.0f), c_dir = F3s(0.0f);
            float c_pdfA = 0, c_dist = 0, c_G = 0;
            const bool presampled = flag(p, STHIP_ePresampleLights);
            auto medium_candidate = [&](uint32_t ti, f3& cLe, float& c_pdfA, f3& c_dir, float& c_dist, float& c_G, f3& c_pos, uint32_t& c_pgn) {
            if (presampled) {
              uint32_t path_index;
              if (flag(p, STHIP_eRemapThreads))
                path_index = ((py >> 2) * ((p.pc.gOutputExtent[0] + 7u) >> 3) + (px >> 3)) * 32u + (py & 3u) * 8u + (px & 7u);
              else
                path_index = py * p.pc.gOutputExtent[0] + px;
              const uint32_t tile_size = p.pc.gLightPresampleTileSize;
              const uint32_t tile_offset = ((path_index / tile_size) % p.pc.gLightPresampleTileCount) * tile_size;
              const float4* lp = p.presampled + 2 * ((size_t)seed_index * tile_size * p.pc.gLightPresampleTileCount + tile_offset + ti % tile_size);
              const float4 l0 = lp[0], l1 = lp[1];
              cLe = xyz(l1);
              c_pdfA = l1.w;
              c_pos = xyz(l0);
              c_pgn = __float_as_uint(l0.w);
              c_dir = xyz(l0) - scatter_p;
              const float d2 = len_sqr(c_dir);
              c_dist = sqrtf(d2);
              c_dir = c_dir / c_dist;
              c_G = fabsf(dot3(c_dir, unpack_normal_octahedron(__float_as_uint(l0.w)))) / d2;
            } else {
              const float r0 = rng.next_float(), r1 = rng.next_float(), r2 = rng.next_float(), r3 = rng.next_float();
              LightSample ls;
              sample_point_on_light<TEXTURED, EXT>(p, has_env, has_emissives, r0, r1, r2, r3, scatter_p, ls);
              cLe = ls.Le;
              c_dir = ls.to_light;
              c_dist = ls.dist;
              c_pdfA = ls.pdf;
              c_pos = ls.position;
              c_pgn = pack_normal_octahedron(ls.normal);
              if (ls.is_env) {
                c_G = 1;
              } else {
                c_G = fabsf(dot3(c_dir, ls.normal)) / pow2f(c_dist);
                if (!ls.area_measure) c_pdfA = c_pdfA * c_G;
              }
            }
            };
            uint32_t* column = (MEDIA == 2) ? p.shade_stack + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * p.bvh.stack_depth : nullptr;
            if (EXT && flag(p, STHIP_eNEEReservoirs)) {
              // connect_light_reservoir at a medium vertex (path.hlsli:368-486 with setup()'s medium branch, :207-212: local_to_light
              // is the WORLD direction, so the target is luminance(Le) G |direction.z|). Spatial reuse as at a surface (:402-439);
              // the geometry normal upstream takes the jitter's tangent plane from, and stores, is the stale one of the last
              // surface query here: pinned to the packed value 0 on both sides
              float total_weight = 0, r_target_pdf = 0;
              uint32_t M = 0;
              f3 y_pos = F3s(0.0f);  // the chosen candidate's PresampledLightPoint (what a reservoir stores)
              uint32_t y_pgn = 0;
              uint32_t ti = rng.next_uint();
              for (uint32_t k = 0; k < p.pc.gReservoirM; k++) {
                if (presampled) ti = rng.next_uint();
                f3 iLe = F3s(0.0f), i_dir = F3s(0.0f), i_pos = F3s(0.0f);
                float i_pdfA = 0, i_dist = 0, i_G = 0;
                uint32_t i_pgn = 0;
                medium_candidate(ti, iLe, i_pdfA, i_dir, i_dist, i_G, i_pos, i_pgn);
                if (i_pdfA <= 0 || all_le0(iLe)) continue;
                const float target_pdf = luminance3(iLe) * i_G * fabsf(i_dir.z);
                const float w = target_pdf / i_pdfA;
                M++;
                total_weight += w;
                if (rng.next_float() * total_weight <= w) {
                  r_target_pdf = target_pdf;
                  cLe = iLe;
                  c_dir = i_dir;
                  c_pdfA = i_pdfA;
                  c_dist = i_dist;
                  c_G = i_G;
                  y_pos = i_pos;
                  y_pgn = i_pgn;
                }
              }
              const bool reuse = p.hg_appends != nullptr;
              f3 hg_t = F3s(0.0f), hg_b = F3s(0.0f);
              float cell_size = 0;
              auto jittered = [&]() {
                const float phi = rng.next_float() * 2 * DET_PI;
                if (!flag(p, STHIP_eHashGridJitter)) return scatter_p;
                const float radius = cell_size * rng.next_float();
                float sn, cs;
                det_sincosf(phi, &sn, &cs);
                return scatter_p + (hg_t * cs + hg_b * sn) * radius;
              };
              if (reuse) {
                make_orthonormal(unpack_normal_octahedron(0u), hg_t, hg_b);
                const Xf vt = load_xf(p.view_xf, 0);
                cell_size = hashgrid_cell_size(p.pc, p.views[0], F3(vt.r0.w, vt.r1.w, vt.r2.w), scatter_p);
                if (p.hg_prev && p.pc.gReservoirSpatialM > 0) {
                  const f3 at = jittered();
                  const uint32_t bucket = hashgrid_find(p.hg_checksums, p.pc.gHashGridBucketCount, at, cell_size);
                  if (bucket != 0xFFFFFFFFu) {
                    const uint32_t bucket_start = p.hg_indices[bucket], bucket_size = p.hg_counters[bucket];
                    uint32_t Msum = M;
                    for (uint32_t k = 0; k < p.pc.gReservoirSpatialM; k++) {
                      const float4* pr = p.hg_data + 3 * (size_t)(bucket_start + rng.next_uint() % bucket_size);
                      const float4 q0 = pr[0], q1 = pr[1], q2 = pr[2];
                      const f3 iLe = xyz(q2);
                      const float i_pdfA = q2.w;
                      f3 i_dir = xyz(q1) - scatter_p;
                      const float d2 = len_sqr(i_dir);
                      const float i_dist = sqrtf(d2);
                      i_dir = i_dir / i_dist;
                      const float i_G = fabsf(dot3(i_dir, unpack_normal_octahedron(__float_as_uint(q1.w)))) / d2;
                      if (i_pdfA <= 0 || all_le0(iLe)) continue;
                      const uint32_t prev_M = __float_as_uint(q0.y);
                      Msum += prev_M;
                      const float target_pdf = luminance3(iLe) * i_G * fabsf(i_dir.z);
                      const float w = target_pdf * q0.w * (float)prev_M;
                      M++;
                      total_weight += w;
                      if (rng.next_float() * total_weight <= w) {
                        r_target_pdf = target_pdf;
                        cLe = iLe;
                        c_dir = i_dir;
                        c_pdfA = i_pdfA;
                        c_dist = i_dist;
                        c_G = i_G;
                        y_pos = xyz(q1);
                        y_pgn = __float_as_uint(q1.w);
                      }
                    }
                    M = Msum;
                  }
                }
              }
              const float W = (r_target_pdf > 0 && M > 0) ? total_weight / ((float)M * r_target_pdf) : 0;
              if (W <= 1e-6f || W != W) break;
              if (reuse) {  // path.hlsli:434-439
                const f3 at = jittered();
                uint32_t path_index;  // map_pixel_coord, bdpt_util.hlsli:76-83
                if (flag(p, STHIP_eRemapThreads))
                  path_index = ((py >> 2) * ((p.pc.gOutputExtent[0] + 7u) >> 3) + (px >> 3)) * 32u + (py & 3u) * 8u + (px & 7u);
                else
                  path_index = py * p.pc.gOutputExtent[0] + px;
                float4* a = p.hg_appends + 4 * ((size_t)path_index * p.pc.gMaxDiffuseVertices + (diffuse_vertices - 1));
                a[0] = make_float4(at.x, at.y, at.z, total_weight);
                a[1] = make_float4(__uint_as_float(min(M, p.pc.gReservoirMaxM)), __uint_as_float(0u), W, c_pdfA);
                a[2] = make_float4(y_pos.x, y_pos.y, y_pos.z, cell_size);
                a[3] = make_float4(cLe.x, cLe.y, cLe.z, __uint_as_float(y_pgn));
              }
              const float f = mm.phase(local_dir_in, c_dir);
              f3 contrib = cLe * f * c_G * W;
              if (all_le0(contrib) || c_pdfA < 1e-6f) break;
              float weight = sample_bsdfs ? 1 - 0.5f : 1.0f;
              if (LT) {  // path.hlsli:458-465; setup()'s medium branch leaves emission_pdfA = 0 (:211), the phase function is pdf_rev
                const float dL = connection_dVC(W, 0.0f, 1 / W, false);
                const float G_rev = prev_cos_out / len_sqr(origin - scatter_p);
                const float dE = connection_dVC(dVC, f * G_rev, bsdf_pdf * G, prev_specular);
                weight = 1 / (1 + dE * pow2f(0.0f) + dL * pow2f(f * c_G));
              }
              if (MEDIA == 2) {  // :474-485
                float dir_pdf = 1, nee_pdf = 1;
                walk_segments += visibility_walk_media(p, rng, scatter_p, c_dir, c_dist, medium, contrib, dir_pdf, nee_pdf, column);
                if (nee_pdf <= 0) break;
                contrib = contrib / nee_pdf;
                if (all_le0(contrib)) break;
                if (debug_is(STHIP_DEBUG_RESERVOIR_WEIGHT)) debug_add(F3s(W));
                debug_path_length(beta * contrib, path_length, 1);
                radiance = radiance + (beta * contrib) * weight;
                radiance_dirty = true;
                break;
              }
              const f3 c = beta * contrib * weight;
              if (all_le0(c)) break;
              if (LT && flag(p, STHIP_eLVC) && connect_paths) break;  // (connect_lvc's record takes the slot, as above)
              const uint32_t entry = slot * p.pc.gMaxDiffuseVertices + (diffuse_vertices - 1);
              if (!(c_dist > 1e-6f)) {
                p.shadow_result[entry] = make_float4(c.x, c.y, c.z, 0.0f);
                break;
              }
              const uint32_t k = (uint32_t)atomicAdd(shadow_size, 1ull);
              shadow_out[3 * (size_t)k] = make_float4(scatter_p.x, scatter_p.y, scatter_p.z, c_dist);
              shadow_out[3 * (size_t)k + 1] = make_float4(c_dir.x, c_dir.y, c_dir.z, __uint_as_float(slot));
              shadow_out[3 * (size_t)k + 2] = make_float4(c.x, c.y, c.z, __uint_as_float(medium));
              p.shadow_ext[shadow_base + k] = make_float4(__uint_as_float(rng.counter), 1.0f, __uint_as_float(entry), 0.0f);
              break;
            }
            f3 unused_pos = F3s(0.0f);
            uint32_t unused_pgn = 0;
            medium_candidate(presampled ? rng.next_uint() : 0u, cLe, c_pdfA, c_dir, c_dist, c_G, unused_pos, unused_pgn);
            if (all_le0(cLe) && c_pdfA < 1e-6f) break;
            const float f = mm.phase(local_dir_in, c_dir);
            const float pdfA_fwd = f * c_G;
            if (pdfA_fwd < 1e-6f) break;
            if (MEDIA == 2) {  // path.hlsli:329-365 at a medium vertex (no shading-normal term, :334)
              f3 wLe = cLe;
              float w_fwd = pdfA_fwd, w_pdfA = c_pdfA;
              walk_segments += visibility_walk_media(p, rng, scatter_p, c_dir, c_dist, medium, wLe, w_fwd, w_pdfA, column);
              if (all_le0(wLe)) break;
              const f3 contrib = wLe * f * c_G / w_pdfA;
              if (all_le0(contrib)) break;
              float weight = 1;
              if (LT) {  // BDPT MIS, path.hlsli:341-351 (emission_pdfA = 0 at a medium vertex)
                if (use_mis) {
                  const float dL = connection_dVC(1 / w_pdfA, 0.0f, w_pdfA, false);
                  const float G_rev = prev_cos_out / len_sqr(origin - scatter_p);
                  const float dE = connection_dVC(dVC, f * G_rev, bsdf_pdf * G, prev_specular);
                  weight = 1 / (1 + dE * pow2f(0.0f) + dL * pow2f(w_fwd));
                } else {
                  weight = path_weight(p, path_length, 1);
                }
              } else if (sample_bsdfs)
                weight = mis2(use_mis, w_pdfA, w_fwd);
              debug_path_length(beta * contrib, path_length, 1);
              radiance = radiance + (beta * contrib) * weight;
              radiance_dirty = true;
              break;
            }
            const f3 contrib = cLe * f * c_G / c_pdfA;
            if (all_le0(contrib)) break;
            float weight = 1;
            if (LT) {
              if (use_mis) {
                const float dL = connection_dVC(1 / c_pdfA, 0.0f, c_pdfA, false);
                const float G_rev = prev_cos_out / len_sqr(origin - scatter_p);
                const float dE = connection_dVC(dVC, f * G_rev, bsdf_pdf * G, prev_specular);
                weight = 1 / (1 + dE * pow2f(0.0f) + dL * pow2f(pdfA_fwd));
              } else {
                weight = path_weight(p, path_length, 1);
              }
            } else if (sample_bsdfs)
              weight = mis2(use_mis, c_pdfA, pdfA_fwd);
            const f3 c = beta * contrib * weight;
            if (all_le0(c)) break;
            if (LT && flag(p, STHIP_eLVC) && connect_paths) break;  // (eDeferShadowRays here: connect_lvc's record takes this vertex's slot right after, path.hlsli:775-783)
            const uint32_t entry = slot * p.pc.gMaxDiffuseVertices + (diffuse_vertices - 1);
            if (!(c_dist > 1e-6f)) {  // trace_visibility_ray's loop never runs
              p.shadow_result[entry] = make_float4(c.x, c.y, c.z, 0.0f);
              break;
            }
            const uint32_t k = (uint32_t)atomicAdd(shadow_size, 1ull);
            shadow_out[3 * (size_t)k] = make_float4(scatter_p.x, scatter_p.y, scatter_p.z, c_dist);
            shadow_out[3 * (size_t)k + 1] = make_float4(c_dir.x, c_dir.y, c_dir.z, __uint_as_float(slot));
            shadow_out[3 * (size_t)k + 2] = make_float4(c.x, c.y, c.z, __uint_as_float(medium));
            p.shadow_ext[shadow_base + k] = make_float4(__uint_as_float(rng.counter), 1.0f, __uint_as_float(entry), 0.0f);
          } while (0);
        }
        if (LT && connect_paths && !mm.is_specular()) {
          // connect_light_subpath from a vertex inside a medium (path.hlsli:802-822 with connect_light_vertex's medium branch,
          // :649-650): the direction itself as local_to_light, no ray offset, no cosine in the reverse pdf, the phase function
          // as f and both pdfs; every connection walks its visibility ray at once
          uint32_t path_index;  // map_pixel_coord, bdpt_util.hlsli:76-83
          if (flag(p, STHIP_eRemapThreads))
            path_index = ((py >> 2) * ((p.pc.gOutputExtent[0] + 7u) >> 3) + (px >> 3)) * 32u + (py & 3u) * 8u + (px & 7u);
          else
            path_index = py * p.pc.gOutputExtent[0] + px;
          const size_t per_seed = (size_t)p.pc.gLightPathCount * p.pc.gMaxDiffuseVertices;
          const size_t level = (size_t)p.pc.gOutputExtent[0] * p.pc.gOutputExtent[1];
          // fits(): the three tests in front of every connection; connect_light_vertex (path.hlsli:618-680) from this medium vertex:
          // the contribution without beta, zero when the connection fails; weight, direction and distance come out with it
          auto vertex_fits = [&](const float4* lvp) {
            const float4 v2 = lvp[2];
            const uint32_t pb0 = __float_as_uint(v2.z), pb1 = __float_as_uint(v2.w);
            const f3 lv_beta = F3(det_f16tof32(pb0 & 0xFFFFu), det_f16tof32(pb0 >> 16), det_f16tof32(pb1 & 0xFFFFu));
            const uint32_t lv_length = (pb1 >> 16) & 0x7Fu, lv_diffuse = (pb1 >> 23) & 0x1Fu;
            return !(lv_length + path_length > p.pc.gMaxPathVertices || lv_diffuse + diffuse_vertices > p.pc.gMaxDiffuseVertices || all_le0(lv_beta));
          };
          auto connect_from_medium = [&](const float4* lvp, float& weight, f3& ray_direction, float& ray_distance) -> f3 {
            const float4 v0 = lvp[0], v2 = lvp[2], v3 = lvp[3];
            const uint32_t pb0 = __float_as_uint(v2.z), pb1 = __float_as_uint(v2.w);
            const uint32_t lv_length = (pb1 >> 16) & 0x7Fu;
            const f3 none = F3s(0.0f);
            f3 contrib = F3(det_f16tof32(pb0 & 0xFFFFu), det_f16tof32(pb0 >> 16), det_f16tof32(pb1 & 0xFFFFu));
            if (all_le0(contrib) || any_nan(contrib)) return none;
            ray_direction = xyz(v0) - scatter_p;
            ray_distance = length3(ray_direction);
            const float rcp_dist = 1 / ray_distance;
            ray_direction = ray_direction * rcp_dist;
            const float rcp_dist2 = pow2f(rcp_dist);
            contrib = contrib * rcp_dist2;
            float connection_G_fwd = rcp_dist2;
            if (!((pb1 >> 28) & 4u)) ray_distance = ray_distance * 0.999f;
            MaterialEvalRecord lev;
            float cos_theta_light = 0;
            if (!eval_light_vertex(lvp, -ray_direction, lev, cos_theta_light)) return none;
            contrib = contrib * lev.f;
            connection_G_fwd *= fabsf(cos_theta_light);
            const float dL = connection_dVC(v3.x, lev.pdf_rev * v3.y, v3.z, ((pb1 >> 28) & 8u) != 0);
            const float pdfA_rev = lev.pdf_fwd * rcp_dist2;
            if (all_le0(contrib) || any_nan(contrib)) return none;
            const float ph = mm.phase(local_dir_in, ray_direction);
            if (ph < 1e-6f) return none;
            contrib = contrib * ph;
            if (all_le0(contrib)) return none;
            if (use_mis) {
              const float G_rev = prev_cos_out / len_sqr(origin - scatter_p);
              const float dE = connection_dVC(dVC, ph * G_rev, bsdf_pdf * G, prev_specular);
              weight = 1 / (1 + dE * pow2f(pdfA_rev) + dL * pow2f(ph * connection_G_fwd));
            } else
              weight = path_weight(p, path_length, lv_length);
            return contrib;
          };
          uint32_t* column = p.shade_stack + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * p.bvh.stack_depth;
          if (flag(p, STHIP_eLVC)) {
            // connect_lvc, path.hlsli:683-800, from a vertex inside a medium: ONE vertex of the cache, picked uniformly or, with
            // eLVCReservoirs, by resampled importance sampling; reuse through the hash grid as at a surface, with the stale
            // geometry normal pinned to the packed value 0
            const float4* cache = p.light_vertices + 4 * (size_t)seed_index * per_seed;
            const uint32_t n = (uint32_t)min((unsigned long long)p.lvc_count[seed_index], (unsigned long long)per_seed);
            const uint32_t li0 = rng.next_uint();
            f3 contrib = F3s(0.0f), ray_direction = F3s(0.0f);
            float weight = 1, ray_distance = 0;
            uint32_t lvc_length = n ? (__float_as_uint(cache[4 * (size_t)(li0 % n) + 2].w) >> 16) & 0x7Fu : 0u;
            if (flag(p, STHIP_eLVCReservoirs)) {
              float total_weight = 0, r_target_pdf = 0;
              uint32_t M = 0;
              const float4* chosen = n ? cache + 4 * (size_t)(li0 % n) : nullptr;
              auto resample = [&](const float4* lvp) {
                f3 rd_i = F3s(0.0f);
                float dist_i = 0, weight_i = 0;
                const f3 contrib_i = connect_from_medium(lvp, weight_i, rd_i, dist_i);
                const float target_pdf_i = luminance3(contrib_i);
                const float w = target_pdf_i / lvp[3].w;
                M++;
                total_weight += w;
                if (rng.next_float() * total_weight <= w) {
                  contrib = contrib_i;
                  weight = weight_i;
                  ray_direction = rd_i;
                  ray_distance = dist_i;
                  r_target_pdf = target_pdf_i;
                  chosen = lvp;
                  lvc_length = (__float_as_uint(lvp[2].w) >> 16) & 0x7Fu;
                }
              };
              for (uint32_t ri = 0; ri < p.pc.gReservoirM; ri++) {
                const uint32_t pick = rng.next_uint();
                if (!n) continue;
                const float4* lvp = cache + 4 * (size_t)(pick % n);
                if (!vertex_fits(lvp)) continue;
                resample(lvp);
              }
              if (p.lg_appends) {  // eLVCReservoirReuse, path.hlsli:727-768
                f3 hg_t, hg_b;
                make_orthonormal(unpack_normal_octahedron(0u), hg_t, hg_b);
                const Xf vt = load_xf(p.view_xf, 0);
                const float cell_size = hashgrid_cell_size(p.pc, p.views[0], F3(vt.r0.w, vt.r1.w, vt.r2.w), scatter_p);
                auto jittered = [&]() {
                  const float phi = rng.next_float() * 2 * DET_PI;
                  if (!flag(p, STHIP_eHashGridJitter)) return scatter_p;
                  const float radius = cell_size * rng.next_float();
                  float sn, cs;
                  det_sincosf(phi, &sn, &cs);
                  return scatter_p + (hg_t * cs + hg_b * sn) * radius;
                };
                if (p.hg_prev && p.pc.gReservoirSpatialM > 0) {
                  const f3 at = jittered();
                  const uint32_t bucket = hashgrid_find(p.lg_checksums, p.pc.gHashGridBucketCount, at, cell_size);
                  if (bucket != 0xFFFFFFFFu) {
                    const uint32_t bucket_start = p.lg_indices[bucket], bucket_size = p.lg_counters[bucket];
                    uint32_t Msum = M;
                    for (uint32_t k = 0; k < p.pc.gReservoirSpatialM; k++) {
                      const float4* pr = p.lg_data + 5 * (size_t)(bucket_start + rng.next_uint() % bucket_size);
                      if (!vertex_fits(pr + 1)) continue;
                      Msum += __float_as_uint(pr[0].y);
                      resample(pr + 1);
                    }
                    M = Msum;
                  }
                }
                const float W = (r_target_pdf > 0 && M > 0) ? total_weight / ((float)M * r_target_pdf) : 0.0f;
                const f3 at = jittered();
                M = min(M, p.pc.gReservoirMaxM);
                float4* a = p.lg_appends + 6 * ((size_t)path_index * p.pc.gMaxDiffuseVertices + (diffuse_vertices - 1));
                a[0] = make_float4(at.x, at.y, at.z, cell_size);
                a[1] = make_float4(total_weight, __uint_as_float(M), __uint_as_float(0u), W);
                for (int q = 0; q < 4; q++) a[2 + q] = chosen ? chosen[q] : make_float4(0, 0, 0, 0);
              }
              contrib = contrib * ((r_target_pdf > 0 && M > 0) ? total_weight / ((float)M * r_target_pdf) : 0.0f);
            } else if (n) {
              const float4* lvp = cache + 4 * (size_t)(li0 % n);
              if (vertex_fits(lvp)) contrib = connect_from_medium(lvp, weight, ray_direction, ray_distance);
            }
            contrib = contrib * (float)(p.pc.gMaxDiffuseVertices - 1);
            contrib = contrib * beta;
            if (flag(p, STHIP_eDeferShadowRays)) {  // the record takes this vertex's entry (the one connect_light's record would have taken)
              const f3 c = contrib * weight;
              if (!all_le0(c)) {
                const uint32_t entry = slot * p.pc.gMaxDiffuseVertices + (diffuse_vertices - 1);
                if (!(ray_distance > 1e-6f)) {
                  p.shadow_result[entry] = make_float4(c.x, c.y, c.z, 0.0f);
                } else {
                  const uint32_t k = (uint32_t)atomicAdd(shadow_size, 1ull);
                  shadow_out[3 * (size_t)k] = make_float4(scatter_p.x, scatter_p.y, scatter_p.z, ray_distance);
                  shadow_out[3 * (size_t)k + 1] = make_float4(ray_direction.x, ray_direction.y, ray_direction.z, __uint_as_float(slot));
                  shadow_out[3 * (size_t)k + 2] = make_float4(c.x, c.y, c.z, __uint_as_float(medium));
                  p.shadow_ext[shadow_base + k] = make_float4(__uint_as_float(rng.counter), 1.0f, __uint_as_float(entry), 0.0f);
                }
              }
            } else if (any_gt0(contrib) && weight > 0) {
              float dir_pdf = 1, nee_pdf = 1;
              walk_segments += visibility_walk_media(p, rng, scatter_p, ray_direction, ray_distance, medium, contrib, dir_pdf, nee_pdf, column);
              if (!(all_le0(contrib) || nee_pdf <= 0)) {
                contrib = contrib / nee_pdf;
                debug_path_length(contrib, path_length, lvc_length);
                radiance = radiance + contrib * weight;
                radiance_dirty = true;
              }
            }
          } else
          for (uint32_t li = 1; li < p.pc.gMaxDiffuseVertices; li++) {
            const size_t idx = level * (li - 1) + path_index;
            if (idx >= per_seed) break;
            const float4* lvp = p.light_vertices + 4 * ((size_t)seed_index * per_seed + idx);
            if (!vertex_fits(lvp)) break;
            f3 ray_direction = F3s(0.0f);
            float ray_distance = 0, weight = 0;
            f3 contrib = beta * connect_from_medium(lvp, weight, ray_direction, ray_distance);
            if (all_le0(contrib) || weight <= 0) continue;
            float dir_pdf = 1, nee_pdf = 1;
            walk_segments += visibility_walk_media(p, rng, scatter_p, ray_direction, ray_distance, medium, contrib, dir_pdf, nee_pdf, column);
            if (all_le0(contrib) || nee_pdf <= 0) continue;
            contrib = contrib / nee_pdf;
            debug_path_length(contrib, path_length, (__float_as_uint(lvp[2].w) >> 16) & 0x7Fu);
            radiance = radiance + contrib * weight;
            radiance_dirty = true;
          }
        }
        if (!sample_bsdfs) break;
        // sample_direction, path.hlsli:898-952, with the phase function
        const float s0 = rng.next_float(), s1 = rng.next_float(), s2 = rng.next_float();
        (void)s2;
        float ppdf, proughness;
        const f3 dir_out = mm.sample(s0, s1, local_dir_in, ppdf, proughness);
        if (ppdf < 1e-6f) break;
        // eta = -1: eta_scale /= 1; RayDifferential::refract with eta = -1 (mean curvature pinned to 0 for a medium vertex)
        if (TEXTURED) {
          const float2 cone = p.cone[slot];
          rd_radius = cone.x;
          rd_spread = cone.y;
          if (flag(p, STHIP_eRayCones)) {
            rd_radius += rd_spread * sqrtf(dist2);  // path.hlsli:1026-1029
            rd_spread = fmaxf(0.0f, lerp1((rd_spread + 2 * 0.0f * rd_radius) / -1.0f, 0.2f, proughness));
          }
        }
        if (LT) {  // path.hlsli:918-924,927-929
          const float G_rev = prev_cos_out / len_sqr(origin - scatter_p);
          if (path_length > 2) path_pdf_rev *= ppdf * G_rev;
          dVC = connection_dVC(dVC, ppdf * G_rev, bsdf_pdf * G, mm.is_specular());
          prev_specular = mm.is_specular();
          prev_cos_out = 1;
        }
        bsdf_pdf = ppdf;
        new_origin = scatter_p;
        new_direction = dir_out;
        if (debug_is(STHIP_DEBUG_DIR_OUT)) debug_set(dir_out * .5f + F3s(.5f));  // path.hlsli:950
        alive = true;
        break;
      }
      if (ip == 0xFFFFFFFFu) {
        // miss: bdpt.hlsl:231-242 at depth 0, path.hlsli:1049-1058 later (no environment)
        if (DEBUG && primary && (p.debug_mode == STHIP_DEBUG_GEOMETRY_NORMAL || p.debug_mode == STHIP_DEBUG_SHADING_NORMAL))
          debug_set(unpack_normal_octahedron(0u) * .5f + F3s(.5f));  // bdpt.hlsl:222-223 reads normals nobody wrote on a miss: pinned to the packed value 0, as the visibility output
        if (primary && p.write_aov && seed_index == 0) {
          const int view_index = get_view_index(p, px, py);
          const sthip_ViewData& view = p.views[view_index];
          const float ex = (float)(view.image_max[0] - view.image_min[0]), ey = (float)(view.image_max[1] - view.image_min[1]);
          if (p.out_albedo) p.out_albedo[pixel] = make_float4(1, 1, 1, 1);
          if (p.out_visibility) {
            sthip_VisibilityInfo vis;
            vis.instance_primitive_index = 0xFFFFFFFFu;
            vis.packed_normal = 0;
            p.out_visibility[pixel] = vis;
          }
          if (p.out_depth) {
            sthip_DepthInfo dpt;
            dpt.z = __builtin_inff();
            dpt.prev_z = __builtin_inff();
            dpt.dz_dxy[0] = dpt.dz_dxy[1] = 0;
            p.out_depth[pixel] = dpt;
          }
          if (p.out_prev_uv) p.out_prev_uv[pixel] = make_float2(((float)px + 0.5f - (float)view.image_min[0]) / ex, ((float)py + 0.5f - (float)view.image_min[1]) / ey);
        }
        if (has_env) {
          // bdpt.hlsl:236-240 / path.hlsli:1051-1055 -> eval_emission (path.hlsli:847-894) for the background: cos = 1,
          // G = 1 (trace(), path.hlsli:1015-1019), pdf of the direction from Environment::eval_pdf (light.hlsli:155-162)
          Environment env;
          env.load(p.scene, p.pc.gEnvironmentMaterialAddress);
          const f3 eLe = env.eval(p.scene, direction);
          if (!all_le0(eLe)) {
            float light_pdf = env.eval_pdf(p.scene, direction, flag(p, STHIP_eSampleEnvironmentMapDirectly));
            if (has_emissives) light_pdf *= p.pc.gEnvironmentSampleProbability;
            if (MEDIA && !flag(p, STHIP_eDeferShadowRays)) light_pdf *= T_nee_pdf;  // path.hlsli:866
            float weight = 1;
            if (path_length > 2 && use_nee) weight = flag(p, STHIP_eNEEReservoirs) ? 0.5f : mis2(use_mis, bsdf_pdf, light_pdf);
            if (debug_is(STHIP_DEBUG_VIEW_TRACE_CONTRIBUTION)) debug_add(beta * eLe);  // path.hlsli:890-891
            debug_path_length(beta * eLe, path_length, 0);
            radiance = radiance + (beta * eLe) * weight;
            radiance_dirty = true;
          }
        }
        break;
      }
      const uint32_t inst_index = ip & 0xFFFFu;
      const Inst in = load_inst(p.scene, inst_index);
      ShadingData sd;
      float shape_pdf;
      bool shape_pdf_area_measure = true;
      if (EXT && in.type() == STHIP_INSTANCE_TYPE_SPHERE) {  // intersection.hlsli:140-159
        const Xf inv = load_xf(p.scene.inv_xf, inst_index);
        const float im[12] = {inv.r0.x, inv.r0.y, inv.r0.z, inv.r0.w, inv.r1.x, inv.r1.y, inv.r1.z, inv.r1.w, inv.r2.x, inv.r2.y, inv.r2.z, inv.r2.w};
        const f3 local_hit_pos = obj_point(im, seg_origin) + obj_vector(im, direction) * hh.x;
        make_sphere_shading_data(p.scene, sd, inst_index, in, local_hit_pos);
        if (flag(p, STHIP_eUniformSphereSampling)) {
          shape_pdf = 1 / sd.shape_area;
        } else {
          const Xf t = load_xf(p.scene.xf, inst_index);
          const f3 to_center = F3(t.r0.w, t.r1.w, t.r2.w) - origin;
          const float sin_elevation_max_sq = pow2f(in.radius()) / dot3(to_center, to_center);
          const float cos_elevation_max = sqrtf(fmaxf(0.0f, 1 - sin_elevation_max_sq));
          shape_pdf = 1 / (DET_2PI * (1 - cos_elevation_max));
          shape_pdf_area_measure = false;
        }
      } else {
        make_hit_shading_data(p.scene, sd, inst_index, hit_leaf, hh.y, hh.z, TEXTURED && flag(p, STHIP_eFlipTriangleUVs));
        shape_pdf = 1 / (sd.shape_area * (float)in.prim_count());  // intersection.hlsli:172
      }
      const f3 gn = sd.geometry_normal();
      // path.hlsli:1023-1040
      const float dist2 = len_sqr(sd.position - origin);
      float G = 1 / dist2;
      const float ngdotin = -dot3(direction, gn);
      G *= fabsf(ngdotin);
      if (LT) path_pdf *= bsdf_pdf * G;  // path.hlsli:1042
      if (primary) {  // bdpt.hlsl:213-220
        bsdf_pdf = 1;
        G = 1;
        if (LT) path_pdf = path_pdf_rev = dVC = 1;
      }
      if (DEBUG && primary) {  // bdpt.hlsl:222-223: the normals of the intersection (before the material's normal map)
        if (p.debug_mode == STHIP_DEBUG_GEOMETRY_NORMAL) debug_set(gn * .5f + F3s(.5f));
        if (p.debug_mode == STHIP_DEBUG_SHADING_NORMAL) debug_set(sd.shading_normal() * .5f + F3s(.5f));
      }
      DisneyMaterial m;
      uint32_t first_hit_normal = sd.packed_shading_normal;
      if (TEXTURED) {
        const float2 cone = p.cone[slot];
        rd_radius = cone.x;
        rd_spread = cone.y;
        if (flag(p, STHIP_eRayCones)) {  // path.hlsli:1026-1029
          rd_radius += rd_spread * sqrtf(dist2);
          sd.uv_screen_size *= rd_radius;
        }
        if (primary) {
          // bdpt.hlsl:246-253: the first-hit lookup (emission, albedo, visibility normal) works on a copy of sd,
          // next_vertex() then loads the material again into the real one (path.hlsli:1068): same values
          uint32_t n_copy = sd.packed_shading_normal, t_copy = sd.packed_tangent;
          m.load_textured(p.scene, in.material_address(), sd.u, sd.v, sd.uv_screen_size, n_copy, t_copy, p.sampling_flags);
          first_hit_normal = n_copy;
        }
        m.load_textured(p.scene, in.material_address(), sd.u, sd.v, sd.uv_screen_size, sd.packed_shading_normal, sd.packed_tangent, p.sampling_flags);
      } else if (p.lds_material_bytes && in.material_address() + 72u <= p.lds_material_bytes) {
        m.load_lds((DisneyMaterial::LdsFloat*)shade_lds, in.material_address());
      } else {
        m.load(p.scene, in.material_address());
      }
      const f3 Le = m.Le();

      // eval_emission, path.hlsli:847-894 (emissive surface, uniform light choice, no BDPT)
      auto eval_emission = [&]() {
        if (all_le0(Le)) return;
        const f3 contrib = beta * Le;
        const float cos_theta_light = -dot3(gn, direction);
        if (cos_theta_light < 0) return;
        float light_pdfA = shape_pdf;  // point_on_light_pdf, light.hlsli:163-171
        light_pdfA /= (float)p.pc.gLightCount;
        if (EXT) {
          if (!has_emissives) light_pdfA = 0;
          if (has_env) light_pdfA *= 1 - p.pc.gEnvironmentSampleProbability;
          if (!shape_pdf_area_measure) light_pdfA = light_pdfA * G;  // pdfWtoA, path.hlsli:864
        }
        if (MEDIA && !flag(p, STHIP_eDeferShadowRays)) light_pdfA *= T_nee_pdf;  // path.hlsli:866
        float weight = 1;
        if (path_length > 2) {
          if (LT) {  // path.hlsli:870-880
            if (use_mis) {
              const float p_rev_k = cosine_hemisphere_pdfW(fabsf(cos_theta_light)) * (fabsf(prev_cos_out) / len_sqr(origin - sd.position));
              if (connect_paths)
                weight = 1 / (1 + connection_dVC(dVC, p_rev_k, bsdf_pdf * G, prev_specular) * pow2f(light_pdfA));
              else
                weight = prev_specular ? 0.0f : mis2(true, path_pdf, path_pdf_rev * p_rev_k * light_pdfA);
            } else {
              weight = path_weight(p, path_length, 0);
            }
          } else if (use_nee)
            weight = (EXT && flag(p, STHIP_eNEEReservoirs)) ? 0.5f : mis2(use_mis, bsdf_pdf * G, light_pdfA);  // path.hlsli:881-886
        }
        if (debug_is(STHIP_DEBUG_VIEW_TRACE_CONTRIBUTION)) debug_add(contrib);  // path.hlsli:890-891
        debug_path_length(contrib, path_length, 0);
        radiance = radiance + contrib * weight;
        radiance_dirty = true;
      };

      if (primary) {
        eval_emission();  // bdpt.hlsl:253
        if (DEBUG) {  // bdpt.hlsl:257-260, :294-295
          if (p.debug_mode == STHIP_DEBUG_ALBEDO) debug_set(m.albedo());
          else if (p.debug_mode == STHIP_DEBUG_SPECULAR) debug_set(F3s(m.is_specular() ? 1.0f : 0.0f));
          else if (p.debug_mode == STHIP_DEBUG_EMISSION) debug_set(Le);
          else if (p.debug_mode == STHIP_DEBUG_SHADING_NORMAL) debug_set(unpack_normal_octahedron(first_hit_normal) * .5f + F3s(.5f));
          else if (p.debug_mode == STHIP_DEBUG_PREV_UV) debug_prev_uv(inst_index, sd.position);
        }
        if (p.write_aov && seed_index == 0) {
          const int view_index = get_view_index(p, px, py);
          const sthip_ViewData& view = p.views[view_index];
          const Xf t = load_xf(p.view_xf, (uint32_t)view_index);
          if (p.out_albedo) {
            const f3 a = m.albedo();
            p.out_albedo[pixel] = make_float4(a.x, a.y, a.z, 1);
          }
          if (p.out_visibility) {
            sthip_VisibilityInfo vis;
            vis.instance_primitive_index = ip;
            vis.packed_normal = first_hit_normal;
            p.out_visibility[pixel] = vis;
          }
          if (p.out_depth || p.out_prev_uv) {
            const Xf prev_inv_view = load_xf(p.prev_inv_view_xf, (uint32_t)view_index);
            const Xf motion = load_xf(p.scene.motion_xf, inst_index);
            const f3 prev_cam_pos = xf_point(xf_mul(prev_inv_view, motion), sd.position);  // bdpt.hlsl:265
            if (p.out_depth) {
              sthip_DepthInfo dpt;
              dpt.z = length3(sd.position - origin);
              dpt.prev_z = length3(prev_cam_pos);
              const f3 dir_x = primary_dir(view, t, (float)(px + 1), (float)py, nullptr);
              dpt.dz_dxy[0] = ray_plane(origin - sd.position, dir_x, gn) - dpt.z;
              const f3 dir_y = primary_dir(view, t, (float)px, (float)(py + 1), nullptr);
              dpt.dz_dxy[1] = ray_plane(origin - sd.position, dir_y, gn) - dpt.z;
              p.out_depth[pixel] = dpt;
            }
            if (p.out_prev_uv) {
              float4 pc4 = project_point(p.prev_views[view_index].projection, prev_cam_pos);
              pc4.y = -pc4.y;
              pc4.x = pc4.x / pc4.w;
              pc4.y = pc4.y / pc4.w;
              p.out_prev_uv[pixel] = make_float2(pc4.x * .5f + .5f, pc4.y * .5f + .5f);
            }
          }
        }
      }
      // loop condition of bdpt.hlsl:298
      if (!(any_gt0(beta) && !any_nan(beta))) break;

      // ---- next_vertex(), path.hlsli:1048-1075 + :955-998 ----
      const Frame3 frame = make_frame(sd);
      const f3 local_dir_in = normalize3(frame.to_local(-direction));
      if (path_length > 2) eval_emission();
      if (!m.can_eval() || path_length >= p.pc.gMaxPathVertices) break;
      if (!m.is_specular()) {
        diffuse_vertices++;
        if (diffuse_vertices > p.pc.gMaxDiffuseVertices) break;
        if (path_length >= p.pc.gMinPathVertices) {
          // russian_roulette, path.hlsli:829-845
          float rr = luminance3(beta) / eta_scale * 0.95f;
          const bool coherent = !MEDIA && p.rr != nullptr;
          bool group_kills = false;
          if (coherent) {
            if (PROBE && p.probe_kind == 1) {
              Rng peek = rng;
              p.rr[slot] = make_float4(rr, peek.next_float(), 1.0f, 0.0f);
              break;
            }
            const float4 verdict = p.rr[slot];  // the group's: (WaveActiveMax(p), WaveReadLaneFirst(rnd > p))
            rr = verdict.x;
            group_kills = verdict.y != 0.0f;
          }
          if (!(rr >= 1)) {
            const float rnd = rng.next_float();  // every lane draws; with eCoherentRR only the first lane's comparison counts
            if (coherent ? group_kills : rnd > rr) break;
            beta = beta / rr;
            path_pdf *= rr;  // path.hlsli:842
          }
        }
        if (PROBE && p.probe_kind == 1) break;  // a path that passes the roulette's place without reaching it has nothing to report
        const uint32_t group_lane = (py & 3u) * 8u + (px & 7u);  // WaveGetLaneIndex() of the reference's 8x4 workgroup
        if (use_nee) {
          // connect_light, path.hlsli:311-366; sample_Le :141-164; sample_point_on_light, light.hlsli:37-152
          // one light candidate: DirectLightSample's two constructors (path.hlsli:179-201) in front of setup()
          const bool presampled = flag(p, STHIP_ePresampleLights);
          auto light_candidate = [&](uint32_t ti, f3& cLe, float& c_pdfA, f3& c_dir, float& c_dist, float& c_G, f3& c_pos, uint32_t& c_pgn) {
            if (presampled) {
              // path.hlsli:313-320: one of the tile's presampled points; DirectLightSample(_isect, PresampledLightPoint) :184-201
              uint32_t path_index;  // map_pixel_coord, bdpt_util.hlsli:76-83, with the 8x4 groups of bdpt.hlsl:11-12
              if (flag(p, STHIP_eRemapThreads))
                path_index = ((py >> 2) * ((p.pc.gOutputExtent[0] + 7u) >> 3) + (px >> 3)) * 32u + (py & 3u) * 8u + (px & 7u);
              else
                path_index = py * p.pc.gOutputExtent[0] + px;
              const uint32_t tile_size = p.pc.gLightPresampleTileSize;
              const uint32_t tile_offset = ((path_index / tile_size) % p.pc.gLightPresampleTileCount) * tile_size;
              const float4* lp = p.presampled + 2 * ((size_t)seed_index * tile_size * p.pc.gLightPresampleTileCount + tile_offset + ti % tile_size);
              const float4 l0 = lp[0], l1 = lp[1];
              cLe = xyz(l1);
              c_pdfA = l1.w;
              c_pos = xyz(l0);
              c_pgn = __float_as_uint(l0.w);
              c_dir = xyz(l0) - sd.position;
              const float dist2 = len_sqr(c_dir);
              c_dist = sqrtf(dist2);
              c_dir = c_dir / c_dist;
              c_G = fabsf(dot3(c_dir, unpack_normal_octahedron(__float_as_uint(l0.w)))) / dist2;
            } else {
              const float r0 = rng.next_float(), r1 = rng.next_float(), r2 = rng.next_float(), r3 = rng.next_float();
              LightSample ls;
              sample_point_on_light<TEXTURED, EXT>(p, has_env, has_emissives, r0, r1, r2, r3, sd.position, ls);
              cLe = ls.Le;
              c_dir = ls.to_light;
              c_dist = ls.dist;
              c_pdfA = ls.pdf;
              c_pos = ls.position;
              c_pgn = pack_normal_octahedron(ls.normal);
              if (ls.is_env) {  // sample_Le, path.hlsli:156-162
                c_G = 1;
              } else {
                c_G = fabsf(dot3(c_dir, ls.normal)) / pow2f(c_dist);
                if (!ls.area_measure) c_pdfA = c_pdfA * c_G;
              }
            }
          };
          f3 lLe, to_light;
          float pdfA, cG, ls_dist;
          float ris_W = 0;  // eNEEReservoirs: the reservoir's weight replaces 1 / pdfA
          const bool reservoirs = EXT && flag(p, STHIP_eNEEReservoirs);
          if (reservoirs) {
            // connect_light_reservoir, path.hlsli:368-486, without spatial reuse: resampled importance sampling over
            // gReservoirM candidates with target = luminance(Le) G |cos| (Reservoir, reservoir.h)
            float total_weight = 0, r_target_pdf = 0;
            uint32_t M = 0;
            lLe = to_light = F3s(0.0f);
            pdfA = cG = ls_dist = 0;
            uint32_t ti = rng.next_uint();
            const bool coherent_ti = presampled && p.cs_nee != nullptr;  // path.hlsli:379-380,385-387
            if (coherent_ti) {
              if (PROBE && p.probe_kind == 2) {
                p.cs_nee[slot] = make_uint2(1u, ti);
                break;
              }
              ti = p.cs_nee[slot].y + group_lane;
            }
            f3 y_pos = F3s(0.0f);  // the chosen candidate's PresampledLightPoint (what a reservoir stores)
            uint32_t y_pgn = 0;
            for (uint32_t k = 0; k < p.pc.gReservoirM; k++) {
              if (presampled && !coherent_ti) ti = rng.next_uint();
              const uint32_t ti_k = ti;
              if (coherent_ti) ti += 32u;  // WaveGetLaneCount()
              f3 cLe, c_dir, c_pos = F3s(0.0f);
              float c_pdfA, c_dist, c_G;
              uint32_t c_pgn = 0;
              light_candidate(ti_k, cLe, c_pdfA, c_dir, c_dist, c_G, c_pos, c_pgn);
              if (c_pdfA <= 0 || all_le0(cLe)) continue;
              const f3 c_local = normalize3(frame.to_local(c_dir));
              const float target_pdf = luminance3(cLe) * c_G * fabsf(c_local.z);
              const float w = target_pdf / c_pdfA;
              M++;
              total_weight += w;
              if (rng.next_float() * total_weight <= w) {
                r_target_pdf = target_pdf;
                lLe = cLe;
                to_light = c_dir;
                pdfA = c_pdfA;
                cG = c_G;
                ls_dist = c_dist;
                y_pos = c_pos;
                y_pgn = c_pgn;
              }
            }
            // spatial reuse through the previous seed's hash grid, path.hlsli:402-428, and this vertex's append, :434-439
            const bool reuse = p.hg_appends != nullptr;
            f3 hg_t = F3s(0.0f), hg_b = F3s(0.0f);
            float cell_size = 0;
            auto jittered = [&]() {  // the position a lookup / an append hashes: jittered in the tangent plane with eHashGridJitter
              const float phi = rng.next_float() * 2 * DET_PI;
              if (!flag(p, STHIP_eHashGridJitter)) return sd.position;
              const float radius = cell_size * rng.next_float();
              float sn, cs;
              det_sincosf(phi, &sn, &cs);
              return sd.position + (hg_t * cs + hg_b * sn) * radius;
            };
            if (reuse) {
              make_orthonormal(gn, hg_t, hg_b);
              const Xf vt = load_xf(p.view_xf, 0);
              cell_size = hashgrid_cell_size(p.pc, p.views[0], F3(vt.r0.w, vt.r1.w, vt.r2.w), sd.position);
              if (p.hg_prev && p.pc.gReservoirSpatialM > 0) {
                const f3 at = jittered();
                const uint32_t bucket = hashgrid_find(p.hg_checksums, p.pc.gHashGridBucketCount, at, cell_size);
                if (bucket != 0xFFFFFFFFu) {
                  const uint32_t bucket_start = p.hg_indices[bucket], bucket_size = p.hg_counters[bucket];
                  uint32_t Msum = M;
                  for (uint32_t k = 0; k < p.pc.gReservoirSpatialM; k++) {
                    const float4* pr = p.hg_data + 3 * (size_t)(bucket_start + rng.next_uint() % bucket_size);
                    const float4 q0 = pr[0], q1 = pr[1], q2 = pr[2];
                    // DirectLightSample(_isect, prev_reservoir.y), path.hlsli:184-201 (surface points only)
                    const f3 cLe = xyz(q2);
                    const float c_pdfA = q2.w;
                    f3 c_dir = xyz(q1) - sd.position;
                    const float dist2 = len_sqr(c_dir);
                    const float c_dist = sqrtf(dist2);
                    c_dir = c_dir / c_dist;
                    const float c_G = fabsf(dot3(c_dir, unpack_normal_octahedron(__float_as_uint(q1.w)))) / dist2;
                    if (c_pdfA <= 0 || all_le0(cLe)) continue;
                    const uint32_t prev_M = __float_as_uint(q0.y);
                    Msum += prev_M;
                    const f3 c_local = normalize3(frame.to_local(c_dir));
                    const float target_pdf = luminance3(cLe) * c_G * fabsf(c_local.z);
                    const float w = target_pdf * q0.w * (float)prev_M;
                    M++;
                    total_weight += w;
                    if (rng.next_float() * total_weight <= w) {
                      r_target_pdf = target_pdf;
                      lLe = cLe;
                      to_light = c_dir;
                      pdfA = c_pdfA;
                      cG = c_G;
                      ls_dist = c_dist;
                      y_pos = xyz(q1);
                      y_pgn = __float_as_uint(q1.w);
                    }
                  }
                  M = Msum;
                }
              }
            }
            ris_W = (r_target_pdf > 0 && M > 0) ? total_weight / ((float)M * r_target_pdf) : 0;
            if (ris_W <= 1e-6f || ris_W != ris_W) {
              lLe = F3s(0.0f), pdfA = 0;  // :440: nothing to connect (and nothing appended)
            } else if (reuse) {
              const f3 at = jittered();  // (draws: also in a probe)
              uint32_t path_index;  // map_pixel_coord, bdpt_util.hlsli:76-83
              if (flag(p, STHIP_eRemapThreads))
                path_index = ((py >> 2) * ((p.pc.gOutputExtent[0] + 7u) >> 3) + (px >> 3)) * 32u + (py & 3u) * 8u + (px & 7u);
              else
                path_index = py * p.pc.gOutputExtent[0] + px;
              if (!PROBE) {
                float4* a = p.hg_appends + 4 * ((size_t)path_index * p.pc.gMaxDiffuseVertices + (diffuse_vertices - 1));
                a[0] = make_float4(at.x, at.y, at.z, total_weight);
                a[1] = make_float4(__uint_as_float(min(M, p.pc.gReservoirMaxM)), __uint_as_float(sd.packed_geometry_normal), ris_W, pdfA);
                a[2] = make_float4(y_pos.x, y_pos.y, y_pos.z, cell_size);
                a[3] = make_float4(lLe.x, lLe.y, lLe.z, __uint_as_float(y_pgn));
              }
            }
          } else {
            f3 unused_pos;
            uint32_t unused_pgn;
            uint32_t ti = 0;
            if (presampled) {
              ti = rng.next_uint() % p.pc.gLightPresampleTileSize;  // path.hlsli:316
              if (p.cs_nee) {                                       // :317-318
                if (PROBE && p.probe_kind == 2) {
                  p.cs_nee[slot] = make_uint2(1u, ti);
                  break;
                }
                ti = p.cs_nee[slot].y + group_lane;  // (light_candidate takes it modulo the tile size)
              }
            }
            light_candidate(ti, lLe, pdfA, to_light, ls_dist, cG, unused_pos, unused_pgn);
          }
          if (PROBE && p.probe_kind == 2) break;  // passed the NEE index without drawing one
          // DirectLightSample::setup, path.hlsli:204-221
          const f3 local_to_light = normalize3(frame.to_local(to_light));
          const float ngdotout = dot3(gn, to_light);
          const f3 ray_origin = ray_offset(sd.position, ngdotout > 0 ? gn : -gn);
          const float ray_distance = ls_dist * 0.999f;
          if (!PROBE) do {  // (no draws from here to the end of connect_light: a probe for connect_lvc's index skips it)
            if (all_le0(lLe) && pdfA < 1e-6f) break;
            MaterialEvalRecord ev;
            m.eval(ev, local_dir_in, local_to_light, false);
            const float pdfA_fwd = ev.pdf_fwd * cG;
            if (!reservoirs && pdfA_fwd < 1e-6f) break;
            if (MEDIA == 2 && reservoirs) {
              // connect_light_reservoir's inline tail, path.hlsli:441-485: the walk attenuates the finished contribution
              const float wG = cG * shading_normal_correction(local_dir_in.z, local_to_light.z, ngdotin, ngdotout, dot3(gn, sd.shading_normal()), EXT && flag(p, STHIP_eShadingNormalShadowFix));
              f3 contrib = lLe * ev.f * wG * ris_W;
              if (all_le0(contrib) || pdfA < 1e-6f) break;
              float weight = sample_bsdfs ? 1 - 0.5f : 1.0f;
              if (LT) {  // path.hlsli:458-465
                const float emission_pdfA = cosine_hemisphere_pdfW(ngdotout) * (ngdotout / pow2f(ray_distance));
                const float dL = connection_dVC(ris_W, emission_pdfA, 1 / ris_W, false);
                const float G_rev = prev_cos_out / len_sqr(origin - sd.position);
                const float dE = connection_dVC(dVC, ev.pdf_rev * G_rev, bsdf_pdf * G, prev_specular);
                weight = 1 / (1 + dE * pow2f(emission_pdfA) + dL * pow2f(ev.pdf_fwd * wG));
              }
              float dir_pdf = 1, nee_pdf = 1;
              walk_segments += visibility_walk_media(p, rng, ray_origin, to_light, ray_distance, medium, contrib, dir_pdf, nee_pdf, p.shade_stack + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * p.bvh.stack_depth);
              if (nee_pdf <= 0) break;
              contrib = contrib / nee_pdf;
              if (all_le0(contrib)) break;
              if (debug_is(STHIP_DEBUG_RESERVOIR_WEIGHT)) debug_add(F3s(ris_W));
              debug_path_length(beta * contrib, path_length, 1);
              radiance = radiance + (beta * contrib) * weight;
              radiance_dirty = true;
              break;
            }
            if (MEDIA == 2) {
              // path.hlsli:329-365 in upstream's order: the walk first — it attenuates Le, scales both pdfs and advances THIS
              // path's stream — then the shading-normal term, the contribution and its weight, added at once
              f3 wLe = lLe;
              float w_fwd = pdfA_fwd, w_pdfA = pdfA;
              walk_segments += visibility_walk_media(p, rng, ray_origin, to_light, ray_distance, medium, wLe, w_fwd, w_pdfA, p.shade_stack + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * p.bvh.stack_depth);
              if (all_le0(wLe)) break;
              const float wG = cG * shading_normal_correction(local_dir_in.z, local_to_light.z, ngdotin, ngdotout, dot3(gn, sd.shading_normal()), EXT && flag(p, STHIP_eShadingNormalShadowFix));
              const f3 contrib = wLe * ev.f * wG / w_pdfA;
              if (all_le0(contrib)) break;
              float weight = 1;
              if (LT) {  // BDPT MIS, path.hlsli:341-351, with the pdfs as the walk left them
                if (use_mis) {
                  const float emission_pdfA = cosine_hemisphere_pdfW(ngdotout) * (ngdotout / pow2f(ray_distance));
                  const float dL = connection_dVC(1 / w_pdfA, emission_pdfA, w_pdfA, false);
                  const float G_rev = prev_cos_out / len_sqr(origin - sd.position);
                  const float dE = connection_dVC(dVC, ev.pdf_rev * G_rev, bsdf_pdf * G, prev_specular);
                  weight = 1 / (1 + dE * pow2f(emission_pdfA) + dL * pow2f(w_fwd));
                } else {
                  weight = path_weight(p, path_length, 1);
                }
              } else if (sample_bsdfs)
                weight = mis2(use_mis, w_pdfA, w_fwd);
              debug_path_length(beta * contrib, path_length, 1);  // accumulate_contribution, path.hlsli:302-303
              radiance = radiance + (beta * contrib) * weight;
              radiance_dirty = true;
              break;
            }
            cG *= shading_normal_correction(local_dir_in.z, local_to_light.z, ngdotin, ngdotout, dot3(gn, sd.shading_normal()), EXT && flag(p, STHIP_eShadingNormalShadowFix));
            const f3 contrib = reservoirs ? lLe * ev.f * cG * ris_W : lLe * ev.f * cG / pdfA;
            // Without eDeferShadowRays the reference has traced (and counted) the visibility ray by now (path.hlsli:329-332,
            // before the shading-normal term and this test): a contribution that turns out zero still costs its ray
            const bool inline_ray = !reservoirs && !flag(p, STHIP_eDeferShadowRays);
            const bool nothing = all_le0(contrib);
            if (nothing && !inline_ray) break;
            if (reservoirs && pdfA < 1e-6f) break;  // path.hlsli:456
            float weight = 1;
            if (nothing) {
              weight = 0;
            } else if (LT && reservoirs) {  // connect_light_reservoir's BDPT weight, path.hlsli:458-465 (no eMIS test upstream; c.G already carries the shading-normal term)
              const float emission_pdfA = cosine_hemisphere_pdfW(ngdotout) * (ngdotout / pow2f(ray_distance));
              const float dL = connection_dVC(ris_W, emission_pdfA, 1 / ris_W, false);
              const float G_rev = prev_cos_out / len_sqr(origin - sd.position);
              const float dE = connection_dVC(dVC, ev.pdf_rev * G_rev, bsdf_pdf * G, prev_specular);
              weight = 1 / (1 + dE * pow2f(emission_pdfA) + dL * pow2f(ev.pdf_fwd * cG));
            } else if (LT) {  // BDPT MIS, path.hlsli:341-351
              if (use_mis) {
                const float emission_pdfA = cosine_hemisphere_pdfW(ngdotout) * (ngdotout / pow2f(ray_distance));  // setup(), :219
                const float dL = connection_dVC(1 / pdfA, emission_pdfA, pdfA, false);
                const float G_rev = prev_cos_out / len_sqr(origin - sd.position);
                const float dE = connection_dVC(dVC, ev.pdf_rev * G_rev, bsdf_pdf * G, prev_specular);
                weight = 1 / (1 + dE * pow2f(emission_pdfA) + dL * pow2f(pdfA_fwd));
              } else {
                weight = path_weight(p, path_length, 1);
              }
            } else if (sample_bsdfs)
              weight = reservoirs ? 1 - 0.5f : mis2(use_mis, pdfA, pdfA_fwd);  // reservoir_bsdf_mis, :175-177
            const f3 c = nothing ? F3s(0.0f) : beta * contrib * weight;
            // what an unoccluded INLINE shadow ray adds to gDebugImage (path.hlsli:482-485, :365 -> :302-303); a deferred record adds nothing (bdpt.hlsl:311-325)
            f3 dc = F3s(0.0f);
            if (DEBUG && !flag(p, STHIP_eDeferShadowRays) && !nothing) {
              if (p.debug_mode == STHIP_DEBUG_RESERVOIR_WEIGHT && reservoirs) dc = F3s(ris_W);
              if (p.debug_mode == STHIP_DEBUG_PATH_LENGTH_CONTRIBUTION && p.pc.gDebugLightPathLength == 1u && path_length == p.pc.gDebugViewPathLength) dc = beta * contrib;
            }
            // deferred: a zero/negative contribution never adds light and trace_shadows traces no ray for it (bdpt.hlsl:313)
            if (all_le0(c) && !inline_ray) break;
            // eLVC with eDeferShadowRays: connect_lvc stores its record into the SAME gShadowRays slot right after this one
            // (path.hlsli:775-783 after :355-364), unconditionally, so upstream never traces this record: neither do we
            if (LT && flag(p, STHIP_eLVC) && flag(p, STHIP_eConnectToLightPaths) && flag(p, STHIP_eDeferShadowRays)) break;
            if (MEDIA) {  // the record of a walk through the media (k_shadow_media); its result lands in its own entry
              const uint32_t entry = slot * p.pc.gMaxDiffuseVertices + (diffuse_vertices - 1);
              if (!(ray_distance > 1e-6f)) {
                p.shadow_result[entry] = make_float4(c.x, c.y, c.z, 0.0f);
                break;
              }
              const uint32_t k = (uint32_t)atomicAdd(shadow_size, 1ull);
              shadow_out[3 * (size_t)k] = make_float4(ray_origin.x, ray_origin.y, ray_origin.z, ray_distance);
              shadow_out[3 * (size_t)k + 1] = make_float4(to_light.x, to_light.y, to_light.z, __uint_as_float(slot));
              shadow_out[3 * (size_t)k + 2] = make_float4(c.x, c.y, c.z, __uint_as_float(medium));
              p.shadow_ext[shadow_base + k] = make_float4(__uint_as_float(rng.counter), 1.0f, __uint_as_float(entry), 0.0f);
              break;
            }
            if (!(ray_distance > 1e-6f)) {
              // trace_visibility_ray's `while (t_max > 1e-6f)` (intersection.hlsli:195) never runs: unoccluded,
              // no ray. This stage sits between shadow stage depth-1 and depth, so adding here keeps the order.
              float4* target = flag(p, STHIP_eDeferShadowRays) ? p.shadow_sum : nullptr;
              if (target) {
                float4 acc = target[slot];
                target[slot] = make_float4(acc.x + c.x, acc.y + c.y, acc.z + c.z, 0.0f);
              } else {
                radiance = radiance + c;
                radiance_dirty = true;
                if (DEBUG) debug_add(dc);
              }
              break;
            }
            const uint32_t k = (uint32_t)atomicAdd(shadow_size, 1ull);
            shadow_out[3 * (size_t)k] = make_float4(ray_origin.x, ray_origin.y, ray_origin.z, ray_distance);
            shadow_out[3 * (size_t)k + 1] = make_float4(to_light.x, to_light.y, to_light.z, __uint_as_float(slot));
            shadow_out[3 * (size_t)k + 2] = make_float4(c.x, c.y, c.z, 0.0f);
            if (DEBUG && p.shadow_debug && !all_le0(dc)) {
              // the same ray with what it adds to the debug image, in a queue of its own (control lines of "kind 2"): traced once
              // more, into p.debug (api.hip), the way the ray above is traced into the radiance
              const uint32_t k2 = (uint32_t)atomicAdd(&queue_ctl(p.qctl, 2, depth, seg)[QCTL_SIZE], 1ull);
              float4* dbg_out = p.shadow_debug + 3 * shadow_base;
              dbg_out[3 * (size_t)k2] = make_float4(ray_origin.x, ray_origin.y, ray_origin.z, ray_distance);
              dbg_out[3 * (size_t)k2 + 1] = make_float4(to_light.x, to_light.y, to_light.z, __uint_as_float(slot));
              dbg_out[3 * (size_t)k2 + 2] = make_float4(dc.x, dc.y, dc.z, 0.0f);
            }
          } while (0);
        }
        if (PROBE && p.probe_kind == 2) break;  // no NEE at this vertex: nothing to report
        if (connect_paths) {
          // connect_light_subpath, path.hlsli:802-822: this vertex to the stored vertices of the light subpath with the
          // same path index. Each connection that survives queues a visibility ray whose contribution lands in this
          // path's entry i - 1 of `conn`; a slot beyond the buffer reads as a zero vertex (robust buffer access).
          uint32_t path_index;  // map_pixel_coord, bdpt_util.hlsli:76-83
          if (flag(p, STHIP_eRemapThreads))
            path_index = ((py >> 2) * ((p.pc.gOutputExtent[0] + 7u) >> 3) + (px >> 3)) * 32u + (py & 3u) * 8u + (px & 7u);
          else
            path_index = py * p.pc.gOutputExtent[0] + px;
          const size_t per_seed = (size_t)p.pc.gLightPathCount * p.pc.gMaxDiffuseVertices;
          const size_t level = (size_t)p.pc.gOutputExtent[0] * p.pc.gOutputExtent[1];
          // PathVertex fields of a stored vertex (bdpt.h:108-156): fits() = the three tests in front of every connection
          auto vertex_fits = [&](const float4* lvp) {
            const float4 v2 = lvp[2];
            const uint32_t pb0 = __float_as_uint(v2.z), pb1 = __float_as_uint(v2.w);
            const f3 lv_beta = F3(det_f16tof32(pb0 & 0xFFFFu), det_f16tof32(pb0 >> 16), det_f16tof32(pb1 & 0xFFFFu));
            const uint32_t lv_length = (pb1 >> 16) & 0x7Fu, lv_diffuse = (pb1 >> 23) & 0x1Fu;
            return !(lv_length + path_length > p.pc.gMaxPathVertices || lv_diffuse + diffuse_vertices > p.pc.gMaxDiffuseVertices || all_le0(lv_beta));
          };
          // connect_light_vertex, path.hlsli:618-680: the contribution (without beta) of connecting this vertex to the stored
          // light vertex, zero when the connection fails; weight and the visibility ray come out with it
          auto connect_light_vertex = [&](const float4* lvp, float& weight, f3& ray_origin, f3& ray_direction, float& ray_distance) -> f3 {
            const float4 v0 = lvp[0], v2 = lvp[2], v3 = lvp[3];
            const uint32_t pb0 = __float_as_uint(v2.z), pb1 = __float_as_uint(v2.w);
            const uint32_t lv_length = (pb1 >> 16) & 0x7Fu;
            f3 contrib = F3(det_f16tof32(pb0 & 0xFFFFu), det_f16tof32(pb0 >> 16), det_f16tof32(pb1 & 0xFFFFu));
            const f3 none = F3s(0.0f);
            if (all_le0(contrib) || any_nan(contrib)) return none;
            ray_origin = sd.position;
            ray_direction = xyz(v0) - sd.position;
            ray_distance = length3(ray_direction);
            const float rcp_dist = 1 / ray_distance;
            ray_direction = ray_direction * rcp_dist;
            const float rcp_dist2 = pow2f(rcp_dist);
            contrib = contrib * rcp_dist2;
            float connection_G_fwd = rcp_dist2;
            if (!(MEDIA && ((pb1 >> 28) & 4u))) ray_distance = ray_distance * 0.999f;  // visibility_distance_epsilon (not towards a vertex inside a medium, path.hlsli:634-635)
            MaterialEvalRecord lev;
            float cos_theta_light = 0;
            if (!eval_light_vertex(lvp, -ray_direction, lev, cos_theta_light)) return none;
            contrib = contrib * lev.f;
            connection_G_fwd *= fabsf(cos_theta_light);
            const float dL = connection_dVC(v3.x, lev.pdf_rev * v3.y, v3.z, ((pb1 >> 28) & 8u) != 0);
            float pdfA_rev = lev.pdf_fwd * rcp_dist2;
            if (all_le0(contrib) || any_nan(contrib)) return none;
            const f3 local_to_light = normalize3(frame.to_local(ray_direction));
            const float ngdotout = dot3(gn, ray_direction);
            ray_origin = ray_offset(sd.position, ngdotout > 0 ? gn : -gn);
            pdfA_rev *= fabsf(ngdotout);
            contrib = contrib * shading_normal_correction(local_dir_in.z, local_to_light.z, ngdotin, ngdotout, dot3(gn, sd.shading_normal()), flag(p, STHIP_eShadingNormalShadowFix), false);
            MaterialEvalRecord ev;
            m.eval(ev, local_dir_in, local_to_light, false);
            if (ev.pdf_fwd < 1e-6f) return none;
            contrib = contrib * ev.f;
            if (all_le0(contrib)) return none;
            if (use_mis) {
              const float G_rev = prev_cos_out / len_sqr(origin - sd.position);
              const float dE = connection_dVC(dVC, ev.pdf_rev * G_rev, bsdf_pdf * G, prev_specular);
              weight = 1 / (1 + dE * pow2f(pdfA_rev) + dL * pow2f(ev.pdf_fwd * connection_G_fwd));
            } else
              weight = path_weight(p, path_length, lv_length);
            return contrib;
          };
          // a connection that survived: its visibility ray; the contribution lands in entry `entry` of `conn`
          auto queue_connection = [&](f3 c, uint32_t entry, f3 ray_origin, f3 ray_direction, float ray_distance) {
            if (!(ray_distance > 1e-6f)) {  // trace_visibility_ray's loop never runs: visible, no ray
              p.conn[entry] = make_float4(c.x, c.y, c.z, 0.0f);
              return;
            }
            const uint32_t k = (uint32_t)atomicAdd(shadow_size, 1ull);
            shadow_out[3 * (size_t)k] = make_float4(ray_origin.x, ray_origin.y, ray_origin.z, ray_distance);
            shadow_out[3 * (size_t)k + 1] = make_float4(ray_direction.x, ray_direction.y, ray_direction.z, __uint_as_float(0x40000000u | entry));
            shadow_out[3 * (size_t)k + 2] = make_float4(c.x, c.y, c.z, 0.0f);
          };
          // accumulate_contribution's debug half for a connection (path.hlsli:302-303): the UNWEIGHTED contribution of the one
          // (view, light) length pair asked for, where the connection is visible — the same ray once more, in the debug queue
          auto debug_connection = [&](f3 dc, uint32_t light_length, f3 ray_origin, f3 ray_direction, float ray_distance) {
            if (!(debug_is(STHIP_DEBUG_PATH_LENGTH_CONTRIBUTION) && p.pc.gDebugLightPathLength == light_length && path_length == p.pc.gDebugViewPathLength)) return;
            if (!(ray_distance > 1e-6f)) {
              debug_add(dc);
            } else if (p.shadow_debug) {
              const uint32_t k2 = (uint32_t)atomicAdd(&queue_ctl(p.qctl, 2, depth, seg)[QCTL_SIZE], 1ull);
              float4* dbg_out = p.shadow_debug + 3 * shadow_base;
              dbg_out[3 * (size_t)k2] = make_float4(ray_origin.x, ray_origin.y, ray_origin.z, ray_distance);
              dbg_out[3 * (size_t)k2 + 1] = make_float4(ray_direction.x, ray_direction.y, ray_direction.z, __uint_as_float(slot));
              dbg_out[3 * (size_t)k2 + 2] = make_float4(dc.x, dc.y, dc.z, 0.0f);
            }
          };
          if (flag(p, STHIP_eLVC)) {
            // connect_lvc, path.hlsli:683-800 (without the reuse through the hash grid): ONE vertex of the cache, picked
            // uniformly or, with eLVCReservoirs, by resampled importance sampling over gReservoirM uniform picks. An empty
            // cache is a division by zero upstream; here the random numbers are drawn and nothing connects.
            const size_t per_seed_lvc = (size_t)p.pc.gLightPathCount * p.pc.gMaxDiffuseVertices;
            const float4* cache = p.light_vertices + 4 * (size_t)seed_index * per_seed_lvc;
            const uint32_t n = (uint32_t)min((unsigned long long)p.lvc_count[seed_index], (unsigned long long)per_seed_lvc);
            uint32_t li0 = rng.next_uint();
            const bool coherent_li = p.cs_lvc != nullptr;  // path.hlsli:688,703
            if (coherent_li) {
              if (PROBE && p.probe_kind == 3) {
                p.cs_lvc[slot] = make_uint2(1u, li0);
                break;
              }
              li0 = p.cs_lvc[slot].y + group_lane;
            }
            f3 contrib = F3s(0.0f), ray_origin = F3s(0.0f), ray_direction = F3s(0.0f);
            float weight = 1, ray_distance = 0;
            uint32_t lvc_length = n ? (__float_as_uint(cache[4 * (size_t)(li0 % n) + 2].w) >> 16) & 0x7Fu : 0u;  // subpath_length() of the vertex connected to (MEDIA: accumulate_contribution's debug half)
            if (flag(p, STHIP_eLVCReservoirs)) {
              float total_weight = 0, r_target_pdf = 0;  // Reservoir, reservoir.h:4-27
              uint32_t M = 0;
              const float4* chosen = n ? cache + 4 * (size_t)(li0 % n) : nullptr;  // `lv`: what the reuse appends (a zero vertex for an empty cache)
              auto resample = [&](const float4* lvp) {  // one candidate of the RIS pass / of the reuse loop
                f3 ro_i = F3s(0.0f), rd_i = F3s(0.0f);
                float dist_i = 0, weight_i = 0;
                const f3 contrib_i = connect_light_vertex(lvp, weight_i, ro_i, rd_i, dist_i);
                const float target_pdf_i = luminance3(contrib_i);
                const float w = target_pdf_i / lvp[3].w;  // / lv_i.path_pdf
                M++;
                total_weight += w;
                if (rng.next_float() * total_weight <= w) {
                  contrib = contrib_i;
                  weight = weight_i;
                  ray_origin = ro_i;
                  ray_direction = rd_i;
                  ray_distance = dist_i;
                  r_target_pdf = target_pdf_i;
                  chosen = lvp;
                  lvc_length = (__float_as_uint(lvp[2].w) >> 16) & 0x7Fu;
                }
              };
              for (uint32_t ri = 0; ri < p.pc.gReservoirM; ri++) {
                const uint32_t pick = coherent_li ? li0 + (1u + ri) * 32u : rng.next_uint();
                if (!n) continue;
                const float4* lvp = cache + 4 * (size_t)(pick % n);
                if (!vertex_fits(lvp)) continue;
                resample(lvp);
              }
              if (p.lg_appends) {  // eLVCReservoirReuse, path.hlsli:727-768
                f3 hg_t, hg_b;
                make_orthonormal(gn, hg_t, hg_b);
                const Xf vt = load_xf(p.view_xf, 0);
                const float cell_size = hashgrid_cell_size(p.pc, p.views[0], F3(vt.r0.w, vt.r1.w, vt.r2.w), sd.position);
                auto jittered = [&]() {
                  const float phi = rng.next_float() * 2 * DET_PI;
                  if (!flag(p, STHIP_eHashGridJitter)) return sd.position;
                  const float radius = cell_size * rng.next_float();
                  float sn, cs;
                  det_sincosf(phi, &sn, &cs);
                  return sd.position + (hg_t * cs + hg_b * sn) * radius;
                };
                if (p.hg_prev && p.pc.gReservoirSpatialM > 0) {
                  const f3 at = jittered();
                  const uint32_t bucket = hashgrid_find(p.lg_checksums, p.pc.gHashGridBucketCount, at, cell_size);
                  if (bucket != 0xFFFFFFFFu) {
                    const uint32_t bucket_start = p.lg_indices[bucket], bucket_size = p.lg_counters[bucket];
                    uint32_t Msum = M;
                    for (uint32_t k = 0; k < p.pc.gReservoirSpatialM; k++) {
                      const float4* pr = p.lg_data + 5 * (size_t)(bucket_start + rng.next_uint() % bucket_size);
                      if (!vertex_fits(pr + 1)) continue;
                      Msum += __float_as_uint(pr[0].y);
                      resample(pr + 1);
                    }
                    M = Msum;
                  }
                }
                const float W = (r_target_pdf > 0 && M > 0) ? total_weight / ((float)M * r_target_pdf) : 0.0f;
                const f3 at = jittered();
                M = min(M, p.pc.gReservoirMaxM);
                uint32_t path_index;  // map_pixel_coord, bdpt_util.hlsli:76-83
                if (flag(p, STHIP_eRemapThreads))
                  path_index = ((py >> 2) * ((p.pc.gOutputExtent[0] + 7u) >> 3) + (px >> 3)) * 32u + (py & 3u) * 8u + (px & 7u);
                else
                  path_index = py * p.pc.gOutputExtent[0] + px;
                float4* a = p.lg_appends + 6 * ((size_t)path_index * p.pc.gMaxDiffuseVertices + (diffuse_vertices - 1));
                a[0] = make_float4(at.x, at.y, at.z, cell_size);
                a[1] = make_float4(total_weight, __uint_as_float(M), __uint_as_float(sd.packed_geometry_normal), W);
                for (int q = 0; q < 4; q++) a[2 + q] = chosen ? chosen[q] : make_float4(0, 0, 0, 0);
              }
              contrib = contrib * ((r_target_pdf > 0 && M > 0) ? total_weight / ((float)M * r_target_pdf) : 0.0f);
            } else if (n) {
              const float4* lvp = cache + 4 * (size_t)(li0 % n);
              if (vertex_fits(lvp)) contrib = connect_light_vertex(lvp, weight, ray_origin, ray_direction, ray_distance);
            }
            contrib = contrib * (float)(p.pc.gMaxDiffuseVertices - 1);
            contrib = contrib * beta;
            const f3 c = contrib * weight;
            if (flag(p, STHIP_eDeferShadowRays)) {
              // the record takes this vertex's gShadowRays slot (the one connect_light wrote, which is why that record was
              // not queued above); trace_shadows skips a record whose contribution is <= 0 (bdpt.hlsl:313)
              if (MEDIA) {  // the record of a walk through the media (k_shadow_media), in this vertex's own entry — the one connect_light's record would have taken
                if (!all_le0(c)) {
                  const uint32_t entry = slot * p.pc.gMaxDiffuseVertices + (diffuse_vertices - 1);
                  if (!(ray_distance > 1e-6f)) {
                    p.shadow_result[entry] = make_float4(c.x, c.y, c.z, 0.0f);
                  } else {
                    const uint32_t k = (uint32_t)atomicAdd(shadow_size, 1ull);
                    shadow_out[3 * (size_t)k] = make_float4(ray_origin.x, ray_origin.y, ray_origin.z, ray_distance);
                    shadow_out[3 * (size_t)k + 1] = make_float4(ray_direction.x, ray_direction.y, ray_direction.z, __uint_as_float(slot));
                    shadow_out[3 * (size_t)k + 2] = make_float4(c.x, c.y, c.z, __uint_as_float(medium));
                    p.shadow_ext[shadow_base + k] = make_float4(__uint_as_float(rng.counter), 1.0f, __uint_as_float(entry), 0.0f);
                  }
                }
              } else if (!all_le0(c)) {
                if (!(ray_distance > 1e-6f)) {
                  float4 acc = p.shadow_sum[slot];
                  p.shadow_sum[slot] = make_float4(acc.x + c.x, acc.y + c.y, acc.z + c.z, 0.0f);
                } else {
                  const uint32_t k = (uint32_t)atomicAdd(shadow_size, 1ull);
                  shadow_out[3 * (size_t)k] = make_float4(ray_origin.x, ray_origin.y, ray_origin.z, ray_distance);
                  shadow_out[3 * (size_t)k + 1] = make_float4(ray_direction.x, ray_direction.y, ray_direction.z, __uint_as_float(slot));
                  shadow_out[3 * (size_t)k + 2] = make_float4(c.x, c.y, c.z, 0.0f);
                }
              }
            } else if (any_gt0(contrib) && weight > 0) {
              if (MEDIA) {  // path.hlsli:791-797: walked and added at once
                float dir_pdf = 1, nee_pdf = 1;
                walk_segments += visibility_walk_media(p, rng, ray_origin, ray_direction, ray_distance, medium, contrib, dir_pdf, nee_pdf, p.shade_stack + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * p.bvh.stack_depth);
                if (!(all_le0(contrib) || nee_pdf <= 0)) {
                  contrib = contrib / nee_pdf;
                  debug_path_length(contrib, path_length, lvc_length);
                  radiance = radiance + contrib * weight;
                  radiance_dirty = true;
                }
              } else {
                queue_connection(c, slot * (p.pc.gMaxDiffuseVertices - 1), ray_origin, ray_direction, ray_distance);
                if (DEBUG) debug_connection(contrib, lvc_length, ray_origin, ray_direction, ray_distance);
              }
            }
          } else
          for (uint32_t li = 1; li < p.pc.gMaxDiffuseVertices; li++) {
            const size_t idx = level * (li - 1) + path_index;
            if (idx >= per_seed) break;
            const float4* lvp = p.light_vertices + 4 * ((size_t)seed_index * per_seed + idx);
            if (!vertex_fits(lvp)) break;
            f3 ray_origin = F3s(0.0f), ray_direction = F3s(0.0f);
            float ray_distance = 0, weight = 0;
            f3 contrib = beta * connect_light_vertex(lvp, weight, ray_origin, ray_direction, ray_distance);
            if (all_le0(contrib) || weight <= 0) continue;
            if (MEDIA) {  // path.hlsli:814-820: the visibility ray walks through the media now, in this path's stream; added at once
              float dir_pdf = 1, nee_pdf = 1;
              walk_segments += visibility_walk_media(p, rng, ray_origin, ray_direction, ray_distance, medium, contrib, dir_pdf, nee_pdf, p.shade_stack + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * p.bvh.stack_depth);
              if (all_le0(contrib) || nee_pdf <= 0) continue;
              contrib = contrib / nee_pdf;
              debug_path_length(contrib, path_length, (__float_as_uint(lvp[2].w) >> 16) & 0x7Fu);
              radiance = radiance + contrib * weight;
              radiance_dirty = true;
              continue;
            }
            queue_connection(contrib * weight, slot * (p.pc.gMaxDiffuseVertices - 1) + (li - 1), ray_origin, ray_direction, ray_distance);
            if (DEBUG) debug_connection(contrib, (__float_as_uint(lvp[2].w) >> 16) & 0x7Fu, ray_origin, ray_direction, ray_distance);
          }
        }
      }
      if (!sample_bsdfs) break;
      // sample_direction, path.hlsli:898-952
      const float s0 = rng.next_float(), s1 = rng.next_float(), s2 = rng.next_float();
      MaterialSampleRecord ms;
      m.sample(ms, F3(s0, s1, s2), local_dir_in, beta, false);
      if (ms.pdf_fwd < 1e-6f) break;
      if (ms.eta != 0) eta_scale /= pow2f(ms.eta);
      if (TEXTURED && flag(p, STHIP_eRayCones)) {  // path.hlsli:911-916, RayDifferential::reflect / refract :232-243
        float spec_spread = rd_spread + 2 * sd.mean_curvature * rd_radius;
        if (ms.eta != 0) spec_spread = spec_spread / ms.eta;
        rd_spread = fmaxf(0.0f, lerp1(spec_spread, 0.2f, ms.roughness));
      }
      if (LT) {  // MIS quantities, path.hlsli:919-925
        const float G_rev = prev_cos_out / len_sqr(origin - sd.position);
        if (path_length > 2) path_pdf_rev *= ms.pdf_rev * G_rev;
        dVC = connection_dVC(dVC, ms.pdf_rev * G_rev, bsdf_pdf * G, m.is_specular());
        prev_specular = m.is_specular();
      }
      bsdf_pdf = ms.pdf_fwd;
      const float ndotout = ms.dir_out.z;
      const f3 dir_out = normalize3(frame.to_world(ms.dir_out));
      const float ngdotout = dot3(gn, dir_out);
      if (LT) prev_cos_out = ngdotout;
      new_origin = ray_offset(sd.position, ngdotout > 0 ? gn : -gn);
      beta = beta * shading_normal_correction(local_dir_in.z, ndotout, ngdotin, ngdotout, dot3(gn, sd.shading_normal()), EXT && flag(p, STHIP_eShadingNormalShadowFix));
      if (all_le0(beta)) break;
      new_direction = dir_out;
      if (debug_is(STHIP_DEBUG_DIR_OUT)) debug_set(dir_out * .5f + F3s(.5f));  // path.hlsli:950
      alive = true;
    } while (0);
    if (PROBE) continue;

    if (MEDIA && walk_segments) atomicAdd(&p.counters[CNT_RAYS_SHADOW], (unsigned long long)walk_segments);  // (inline walks: not in any queue)
    if (DEBUG && dbg_dirty) p.debug[slot] = dbg;
    if (radiance_dirty) p.radiance[slot] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
    if (!TEXTURED && !EXT && !LT && !MEDIA && p.emitter_count) {  // (the plain instantiation only: in the extended one this code cost 10 % of k_shade in spills)
      // the path's next vertex is its last one and its ray cannot reach an emitter: answered here (aims_at_emitter)
      const bool last_next = alive && (path_length + 1u >= p.pc.gMaxPathVertices || (p.no_specular && diffuse_vertices + 1u > p.pc.gMaxDiffuseVertices));
      bool answered = false;
      if (__any(last_next)) answered = last_next && !aims_at_emitter(p, new_origin, new_direction);
      if (answered) {  // counted where the queue's own size is (one of eight lines; a single counter would serialise the waves)
        alive = false;
        atomicAdd(queue_size + (QCTL_ANSWERED - QCTL_SIZE), 1ull);
      }
    }
    if (alive) {
      p.ray_o[slot] = make_float4(new_origin.x, new_origin.y, new_origin.z, bsdf_pdf);
      p.ray_d[slot] = make_float4(new_direction.x, new_direction.y, new_direction.z, eta_scale);
      p.beta[slot] = make_float4(beta.x, beta.y, beta.z, __uint_as_float(rng.counter));
      p.meta[slot] = path_length | (diffuse_vertices << 8) | ((LT && prev_specular) ? 1u << 16 : 0u);
      if (LT) p.bdpt[slot] = make_float4(path_pdf, path_pdf_rev, dVC, prev_cos_out);
      if (TEXTURED) p.cone[slot] = make_float2(rd_radius, rd_spread);
      if (MEDIA) {  // a new trace() starts at this vertex: T_dir_pdf = T_nee_pdf = 1, no segment walked yet
        p.media_state[2 * (size_t)slot] = make_float4(new_origin.x, new_origin.y, new_origin.z, 1.0f);
        p.media_state[2 * (size_t)slot + 1] = make_float4(1.0f, __uint_as_float(medium), __uint_as_float(0u), 0.0f);
      }
      const uint32_t k = (uint32_t)atomicAdd(queue_size, 1ull);
      queue_out[k] = slot;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// cull_terminal: in front of k_shade in a round where paths can reach their last vertex (the path or diffuse budget ends
// there, path.hlsli:964-966,1061): such a vertex can only still ADD the emission of what was hit (eval_emission,
// path.hlsli:847-894), so a path that ends on a surface without emission — nearly all of them — has nothing left to do:
// k_shade would fetch indices, vertices and transform, build the shading data, find Le = 0 and stop, writing back the
// radiance it read. This pass reads the hit and one byte about the instance's material only and keeps the paths that have something to do, packed
// (gathered per block in LDS, one atomic per block), so that k_shade's waves are full of them. Scenes without images only (Le would
// need the uv), no light subpaths, no media; with an environment a miss is kept (it adds the background).
// ---------------------------------------------------------------------------------------------
#define CULL_BLOCKS_PER_SEGMENT 64u  // a block gathers what it keeps in LDS and appends it with ONE atomic (a wave-level append would
                                     // put ~25 000 atomics on the eight control lines: 11 ns each, longer than the pass itself)
#ifndef STHIP_TEMPLATE_INSTANCES_ONLY
inline __global__ void __launch_bounds__(STHIP_BLOCK) k_cull_terminal(FrameParams p, uint32_t depth, uint32_t* kept, uint32_t capacity) {
  extern __shared__ uint32_t cull_lds[];  // [0] count, [1] base, [2 ..] up to `capacity` kept slots
  const uint32_t seg = blockIdx.x % QUEUE_SEGMENTS;
  const uint32_t seg_base = seg * p.seg_stride;
  unsigned long long* line = queue_ctl(p.qctl, 0, depth, seg);
  const uint32_t n = (uint32_t)line[QCTL_SIZE];
  const uint32_t first = (blockIdx.x / QUEUE_SEGMENTS) * blockDim.x + threadIdx.x;
  const uint32_t step = ((gridDim.x - seg + QUEUE_SEGMENTS - 1u) / QUEUE_SEGMENTS) * blockDim.x;
  const uint32_t* queue_in = p.queue[depth & 1u] + seg_base;
  if (threadIdx.x == 0) cull_lds[0] = 0;
  __syncthreads();
  for (uint32_t i = first; i < n; i += step) {
    const uint32_t slot = queue_in[i];
    const uint32_t meta = p.meta[slot];
    const uint32_t ip = __float_as_uint(p.hit[slot].w);
    if (meta >= 0xFFFFFFFEu) continue;
    bool keep = false;
    if (ip == 0xFFFFFFFFu) {
      keep = (p.scene_flags & STHIP_BDPT_FLAG_HAS_ENVIRONMENT) != 0;  // (a miss without an environment adds nothing: path.hlsli:1049-1058)
    } else {
      const uint32_t path_length = (meta & 0xFFu) + 1u, diffuse_vertices = (meta >> 8) & 0xFFu;
      const uint32_t f = p.inst_flags[ip & 0xFFFFu];
      const bool last = !(f & INST_FLAG_CAN_EVAL) || path_length >= p.pc.gMaxPathVertices || (!(f & INST_FLAG_SPECULAR) && diffuse_vertices + 1u > p.pc.gMaxDiffuseVertices);
      keep = (f & (INST_FLAG_KEEP | INST_FLAG_EMITS)) || !last;
    }
    if (keep) {
      const uint32_t k = atomicAdd(&cull_lds[0], 1u);
      if (k < capacity) cull_lds[2 + k] = slot;  // (capacity = the most entries a block can meet: always true)
    }
  }
  __syncthreads();
  const uint32_t count = min(cull_lds[0], capacity);
  if (threadIdx.x == 0 && count) cull_lds[1] = (uint32_t)atomicAdd(&line[QCTL_KEPT], (unsigned long long)count);
  __syncthreads();
  const uint32_t base = cull_lds[1];
  for (uint32_t k = threadIdx.x; k < count; k += blockDim.x) kept[seg_base + base + k] = cull_lds[2 + k];
}
#endif

// ---------------------------------------------------------------------------------------------
// shadow_media: one step of trace_visibility_ray with media (intersection.hlsli:192-239) for every shadow record of
// round `depth`, after k_trace stored the closest hit of its current segment: a surface ends the walk with nothing,
// a volume boundary is crossed (delta tracking that cannot scatter over the segment if it ran inside a medium) and the
// record is queued again for round depth + 1, the end of the ray (or a miss) finishes it: contribution / nee_pdf goes to
// the record's own entry of shadow_result, which k_resolve sums in the order of trace_shadows (bdpt.hlsl:311-325).
// ---------------------------------------------------------------------------------------------
#ifndef STHIP_TEMPLATE_INSTANCES_ONLY
inline __global__ void __launch_bounds__(STHIP_BLOCK) k_shadow_media(FrameParams p, uint32_t depth) {
  const uint32_t seg = blockIdx.x % QUEUE_SEGMENTS;
  const uint32_t n = (uint32_t)queue_ctl(p.qctl, 1, depth, seg)[QCTL_SIZE];
  const uint32_t first = (blockIdx.x / QUEUE_SEGMENTS) * blockDim.x + threadIdx.x;
  const uint32_t step = ((gridDim.x - seg + QUEUE_SEGMENTS - 1u) / QUEUE_SEGMENTS) * blockDim.x;
  const size_t in_base = (size_t)seg * p.shadow_stride + ((depth & 1u) ? p.shadow_alt : 0u);
  const size_t out_base = (size_t)seg * p.shadow_stride + (((depth + 1u) & 1u) ? p.shadow_alt : 0u);
  unsigned long long* out_size = &queue_ctl(p.qctl, 1, depth + 1, seg)[QCTL_SIZE];
  for (uint32_t i = first; i < n; i += step) {
    const size_t r = in_base + i;
    const float4 s0 = p.shadow_rays[3 * r], s1 = p.shadow_rays[3 * r + 1], s2 = p.shadow_rays[3 * r + 2], ext = p.shadow_ext[r], hh = p.shadow_hit[r];
    f3 o = xyz(s0);
    float t_max = s0.w;
    const f3 d = xyz(s1);
    const uint32_t slot = __float_as_uint(s1.w);
    f3 contribution = xyz(s2);
    uint32_t cur_medium = __float_as_uint(s2.w);
    float nee_pdf = ext.y;
    const uint32_t entry = __float_as_uint(ext.z);
    uint32_t px, py;
    slot_to_pixel(p, slot, px, py);
    Rng rng;  // rng_init(pixel_coord, rd.rng_offset), bdpt.hlsl:315
    rng.x = px;
    rng.y = py;
    rng.seed = p.seed + slot / p.paths_per_seed;
    rng.counter = __float_as_uint(ext.x);
    float dir_pdf = 1;  // (only the inline form reads it)
    const bool done = visibility_step_media(p, rng, o, d, t_max, cur_medium, contribution, dir_pdf, nee_pdf, hh.x, __float_as_uint(hh.w));
    if (done) {
      if (nee_pdf > 0) contribution = contribution / nee_pdf;
      p.shadow_result[entry] = make_float4(contribution.x, contribution.y, contribution.z, 0.0f);
    } else {
      const size_t k = out_base + (uint32_t)atomicAdd(out_size, 1ull);
      p.shadow_rays[3 * k] = make_float4(o.x, o.y, o.z, t_max);
      p.shadow_rays[3 * k + 1] = make_float4(d.x, d.y, d.z, __uint_as_float(slot));
      p.shadow_rays[3 * k + 2] = make_float4(contribution.x, contribution.y, contribution.z, __uint_as_float(cur_medium));
      p.shadow_ext[k] = make_float4(__uint_as_float(rng.counter), nee_pdf, __uint_as_float(entry), 0.0f);
    }
  }
}
#endif

// ---------------------------------------------------------------------------------------------
// resolve: gRadiance += c (bdpt.hlsl:325), then the running mean that defines N samples per pixel
// (temporal_accumulation.hlsl:102-131), and on the last seed the scatter to the output image
// ---------------------------------------------------------------------------------------------
#ifndef STHIP_TEMPLATE_INSTANCES_ONLY
inline __global__ void __launch_bounds__(STHIP_BLOCK) k_resolve(FrameParams p, uint32_t first_seed, uint32_t last_seed, uint32_t primary_rays) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // every queued path / shadow record was traced exactly once: ray counts are the queue sizes
    // (a last ray answered by k_shade is a trace_ray call too: QCTL_ANSWERED of the bounce it belongs to)
    unsigned long long closest = primary_rays, shadow = 0, answered = 0;
    for (uint32_t d = 0; d < p.rounds; d++)
      for (uint32_t s = 0; s < QUEUE_SEGMENTS; s++) {
        if (d) closest += queue_ctl(p.qctl, 0, d, s)[QCTL_SIZE];
        if (d) answered += queue_ctl(p.qctl, 0, d, s)[QCTL_ANSWERED];
        shadow += queue_ctl(p.qctl, 1, d, s)[QCTL_SIZE];
      }
    p.counters[CNT_RAYS_CLOSEST] += closest + answered;
    p.counters[CNT_RAYS_ANSWERED] += answered;
    p.counters[CNT_RAYS_SHADOW] += shadow;
  }
  for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < p.paths_per_seed; q += gridDim.x * blockDim.x) {
    uint32_t px, py;
    const bool inside = slot_to_pixel(p, q, px, py);
    if (!inside) continue;
    if (p.meta[q] == 0xFFFFFFFFu) {
      // no view covers the pixel: sample_visibility returned at once, but add_light_trace runs for every pixel of the image
      // (bdpt.hlsl:328-338) and, in two debug modes, writes what it loaded — nothing, no splat lands outside the views — there
      if ((p.light_trace || p.light_trace_empty) && (p.debug_mode == STHIP_DEBUG_LIGHT_TRACE_CONTRIBUTION || (p.debug_mode == STHIP_DEBUG_PATH_LENGTH_CONTRIBUTION && p.pc.gDebugViewPathLength == 1))) {
        f3 lc = F3s(0.0f);
        if (p.light_trace) {
          const uint4 v = *reinterpret_cast<const uint4*>(p.light_trace + 4 * ((size_t)(p.seeds_in_flight - 1u) * p.pc.gOutputExtent[0] * p.pc.gOutputExtent[1] + (size_t)py * p.pc.gOutputExtent[0] + px));
          const float qq = (float)p.light_trace_quantization;
          lc = F3((float)((v.w & 1u) ? 0xFFFFFFFFu : v.x), (float)((v.w & 2u) ? 0xFFFFFFFFu : v.y), (float)((v.w & 4u) ? 0xFFFFFFFFu : v.z)) / qq;
          if (lc.x < 0 || lc.y < 0 || lc.z < 0 || any_nan(lc)) lc = F3s(0.0f);
        }
        p.out_debug[(size_t)py * p.pc.gOutputExtent[0] + px] = make_float4(lc.x, lc.y, lc.z, 1);
      }
      continue;
    }
    float4 acc = first_seed ? make_float4(0, 0, 0, 0) : p.accum[q];
    // the seeds in flight are folded in seed order, so the result does not depend on how many were in flight
    for (uint32_t s = 0; s < p.seeds_in_flight; s++) {
      const uint32_t slot = s * p.paths_per_seed + q;
      float4 r = p.radiance[slot];
      float4 c = p.shadow_sum[slot];
      if (p.media) {  // trace_shadows, bdpt.hlsl:311-325: c += the result of the walk of diffuse vertex i, i = 1..gMaxDiffuseVertices
        c = make_float4(0, 0, 0, 0);
        for (uint32_t k = 0; k < p.pc.gMaxDiffuseVertices; k++) {
          const float4 e = p.shadow_result[(size_t)slot * p.pc.gMaxDiffuseVertices + k];
          c.x = c.x + e.x;
          c.y = c.y + e.y;
          c.z = c.z + e.z;
        }
      }
      if (p.conn) {  // the light-subpath connections of the path's last vertex (see k_shade)
        const float4* cn = p.conn + (size_t)slot * (p.pc.gMaxDiffuseVertices - 1);
        for (uint32_t k = 0; k + 1 < p.pc.gMaxDiffuseVertices; k++) {
          const float4 e = cn[k];
          if (e.x != 0 || e.y != 0 || e.z != 0) {
            r.x = r.x + e.x;
            r.y = r.y + e.y;
            r.z = r.z + e.z;
          }
        }
      }
      float4 cur = make_float4(r.x + c.x, r.y + c.y, r.z + c.z, 1.0f);
      if (p.light_trace && p.debug_mode != STHIP_DEBUG_VIEW_TRACE_CONTRIBUTION) {  // add_light_trace, bdpt.hlsl:328-338 with load_light_sample, path.hlsli:38-46
        const uint4 v = *reinterpret_cast<const uint4*>(p.light_trace + 4 * ((size_t)s * p.pc.gOutputExtent[0] * p.pc.gOutputExtent[1] + (size_t)py * p.pc.gOutputExtent[0] + px));
        const float q = (float)p.light_trace_quantization;
        f3 lc = F3((float)((v.w & 1u) ? 0xFFFFFFFFu : v.x), (float)((v.w & 2u) ? 0xFFFFFFFFu : v.y), (float)((v.w & 4u) ? 0xFFFFFFFFu : v.z)) / q;
        if (lc.x < 0 || lc.y < 0 || lc.z < 0 || any_nan(lc)) lc = F3s(0.0f);
        cur.x = cur.x + lc.x;
        cur.y = cur.y + lc.y;
        cur.z = cur.z + lc.z;
        if (p.debug_mode == STHIP_DEBUG_LIGHT_TRACE_CONTRIBUTION || (p.debug_mode == STHIP_DEBUG_PATH_LENGTH_CONTRIBUTION && p.pc.gDebugViewPathLength == 1)) p.debug[slot] = make_float4(lc.x, lc.y, lc.z, 1);  // bdpt.hlsl:335-336
      }
      if (p.light_trace_empty && p.debug_mode != STHIP_DEBUG_VIEW_TRACE_CONTRIBUTION &&
          (p.debug_mode == STHIP_DEBUG_LIGHT_TRACE_CONTRIBUTION || (p.debug_mode == STHIP_DEBUG_PATH_LENGTH_CONTRIBUTION && p.pc.gDebugViewPathLength == 1)))
        p.debug[slot] = make_float4(0, 0, 0, 1);
      if (p.debug_mode) p.out_debug[(size_t)py * p.pc.gOutputExtent[0] + px] = p.debug[slot];  // (one seed in flight: the seeds of a call are upstream's successive frames)
      if (isinf(cur.x) || isinf(cur.y) || isinf(cur.z) || cur.x != cur.x || cur.y != cur.y || cur.z != cur.z) cur = make_float4(0, 0, 0, 0);
      if (acc.w > 0) {
        const float nn = acc.w + cur.w;
        const float alpha = fminf(fmaxf(cur.w / nn, 0.0f), 1.0f);
        acc.x = lerp1(acc.x, cur.x, alpha);
        acc.y = lerp1(acc.y, cur.y, alpha);
        acc.z = lerp1(acc.z, cur.z, alpha);
        acc.w = nn;
      } else {
        acc = cur;
      }
    }
    p.accum[q] = acc;
    if (last_seed && p.out_radiance) p.out_radiance[p.out_packed ? (size_t)q : (size_t)py * p.pc.gOutputExtent[0] + px] = acc;
  }
}
#endif

// the zero fills a pass starts with, as one launch: up to three ranges of 64-bit words
#ifndef STHIP_TEMPLATE_INSTANCES_ONLY
inline __global__ void __launch_bounds__(STHIP_BLOCK) k_clear(unsigned long long* a, uint32_t na, unsigned long long* b, uint32_t nb, unsigned long long* c, uint32_t nc) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < na + nb + nc; i += gridDim.x * blockDim.x) {
    if (i < na)
      a[i] = 0ull;
    else if (i < na + nb)
      b[i - na] = 0ull;
    else
      c[i - na - nb] = 0ull;
  }
}
#endif

// gRayCount[0] = every trace_ray call, [1] = path (closest-hit) rays; intersection.hlsli:66, path.hlsli:1006
// eCoherentRR: the probes of a round (FrameParams::rr) -> one verdict per 8x4 group. A wave covers 64 consecutive slots = one
// 8x8 pixel block; its lanes 0-31 (rows 0-3) and 32-63 (rows 4-7) are the two reference workgroups in it, in the
// workgroup's own thread order (y * 8 + x).
#ifndef STHIP_TEMPLATE_INSTANCES_ONLY
inline __global__ void __launch_bounds__(STHIP_BLOCK) k_rr_reduce(FrameParams p) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u, half = lane >> 5;
  const float4 v = slot < p.path_count ? p.rr[slot] : make_float4(0, 0, 0, 0);
  const bool reached = v.z != 0.0f;
  float pm = reached ? v.x : -1.0f;
  for (int off = 16; off > 0; off >>= 1) pm = fmaxf(pm, __shfl_xor(pm, off, 64));  // WaveActiveMax over the 32 lanes of the group
  const unsigned long long all = __ballot(reached);
  const uint32_t mine = (uint32_t)(all >> (half * 32u));
  const int first = mine ? (int)(half * 32u) + (__ffs((int)mine) - 1) : (int)lane;
  const float rnd_first = __shfl(v.y, first, 64);  // WaveReadLaneFirst
  if (slot < p.path_count) p.rr[slot] = make_float4(pm, (mine && pm < 1.0f && rnd_first > pm) ? 1.0f : 0.0f, mine ? 1.0f : 0.0f, 0.0f);
}
#endif

// eCoherentSampling: WaveReadLaneFirst over the group's lanes that reached the site (entries (1, own draw) written by the probe)
#ifndef STHIP_TEMPLATE_INSTANCES_ONLY
inline __global__ void __launch_bounds__(STHIP_BLOCK) k_cs_reduce(uint2* values, uint32_t path_count) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u, half = lane >> 5;
  const uint2 v = slot < path_count ? values[slot] : make_uint2(0u, 0u);
  const unsigned long long all = __ballot(v.x != 0u);
  const uint32_t mine = (uint32_t)(all >> (half * 32u));
  const int first = mine ? (int)(half * 32u) + (__ffs((int)mine) - 1) : (int)lane;
  const uint32_t first_value = (uint32_t)__shfl((int)v.y, first, 64);
  if (slot < path_count) values[slot] = make_uint2(mine ? 1u : 0u, first_value);
}
#endif

// ---- hash grid build (hashgrid.h): keys of the compacted appends, and the scatter into the bucket ranges ----
#ifndef STHIP_TEMPLATE_INSTANCES_ONLY
inline __global__ void __launch_bounds__(STHIP_BLOCK) k_hg_keys(const float4* appends, const uint32_t* count, uint32_t bucket_count, uint2* keys) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= *count) return;
  const float4 a0 = appends[4 * (size_t)i], a2 = appends[4 * (size_t)i + 2];
  uint32_t checksum;
  const uint32_t home = hashgrid_bucket_index(xyz(a0), a2.w, bucket_count, checksum);
  keys[i] = make_uint2(home, checksum);
}
#endif
// dest[i] = index into gNEEHashGrid.mData of append i, or 0xFFFFFFFF when its 32 probes found no slot (dropped, hashgrid.hlsli:56-58)
#ifndef STHIP_TEMPLATE_INSTANCES_ONLY
inline __global__ void __launch_bounds__(STHIP_BLOCK) k_hg_scatter(const float4* appends, const uint32_t* count, const uint32_t* dest, float4* data) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= *count) return;
  const uint32_t d = dest[i];
  if (d == 0xFFFFFFFFu) return;
  const float4 a0 = appends[4 * (size_t)i], a1 = appends[4 * (size_t)i + 1], a2 = appends[4 * (size_t)i + 2], a3 = appends[4 * (size_t)i + 3];
  data[3 * (size_t)d] = make_float4(a0.w, a1.x, a1.y, a1.z);      // r.total_weight, bits(r.M), bits(packed_geometry_normal), W
  data[3 * (size_t)d + 1] = make_float4(a2.x, a2.y, a2.z, a3.w);  // y.position, bits(y.packed_geometry_normal)
  data[3 * (size_t)d + 2] = make_float4(a3.x, a3.y, a3.z, a1.w);  // y.Le, y.pdfA
}
#endif

#ifndef STHIP_TEMPLATE_INSTANCES_ONLY
inline __global__ void __launch_bounds__(STHIP_BLOCK) k_hg_keys_lvc(const float4* appends, const uint32_t* count, uint32_t bucket_count, uint2* keys) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= *count) return;
  const float4 a0 = appends[6 * (size_t)i];
  uint32_t checksum;
  const uint32_t home = hashgrid_bucket_index(xyz(a0), a0.w, bucket_count, checksum);
  keys[i] = make_uint2(home, checksum);
}
#endif
#ifndef STHIP_TEMPLATE_INSTANCES_ONLY
inline __global__ void __launch_bounds__(STHIP_BLOCK) k_hg_scatter_lvc(const float4* appends, const uint32_t* count, const uint32_t* dest, float4* data) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= *count) return;
  const uint32_t d = dest[i];
  if (d == 0xFFFFFFFFu) return;
  for (int q = 0; q < 5; q++) data[5 * (size_t)d + q] = appends[6 * (size_t)i + 1 + q];
}
#endif

#ifndef STHIP_TEMPLATE_INSTANCES_ONLY
inline __global__ void k_write_ray_count(const unsigned long long* counters, unsigned long long* out) {
  out[0] = counters[CNT_RAYS_CLOSEST] + counters[CNT_RAYS_SHADOW];
  out[1] = counters[CNT_RAYS_CLOSEST] - counters[CNT_CROSSINGS];  // one per trace() call (path.hlsli:1006), however many segments it walked
}
#endif

// plain ray batches: the traversal contract on its own (sthip_trace_rays)
template <bool ANY_HIT, bool COUNT>
__global__ void __launch_bounds__(STHIP_BLOCK) k_trace_batch(DeviceBvh bvh, const sthip_ray* rays, uint32_t n, sthip_hit* hits, unsigned long long* counters) {
  extern __shared__ uint32_t lds_stack[];
  TraverseCounters cnt;
  cnt.clear();
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float4* r = reinterpret_cast<const float4*>(rays + i);
    const float4 a = r[0], b = r[1];
    RayHit h;
    if (bvh.spill)  // bounded LDS stacks: the batch entry point walks with the full-height stack in global memory
      traverse<ANY_HIT ? TRAV_ANY : TRAV_CLOSEST, COUNT, 1, true>(bvh, xyz(a), xyz(b), a.w, b.w, bvh.spill + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * bvh.stack_depth, h, cnt);
    else
      traverse<ANY_HIT ? TRAV_ANY : TRAV_CLOSEST, COUNT, STHIP_BLOCK, true>(bvh, xyz(a), xyz(b), a.w, b.w, lds_stack + threadIdx.x, h, cnt);
    sthip_hit out;
    out.t = h.t;
    out.b1 = h.b1;
    out.b2 = h.b2;
    out.instance_primitive_index = h.ip;
    hits[i] = out;
  }
  if (COUNT && counters) {
    atomicAdd(&counters[CNT_NODES + (ANY_HIT ? 1 : 0)], (unsigned long long)cnt.nodes);
    atomicAdd(&counters[CNT_TRIS + (ANY_HIT ? 1 : 0)], (unsigned long long)cnt.tris);
  }
}


// Assembles the frame from the packed tiles of every shard (sthip_assemble_tiles, sthip_assemble_tiles_bytes): entry
// (rank, slot) of `packed` is the pixel slot_to_pixel() gives for that shard; one thread per packed entry, disjoint pixels.
// An entry is `words` 32-bit words (4: RGBA32F radiance / albedo / DepthInfo, 2: VisibilityInfo / prev-uv).
DEV bool shard_slot_pixel(uint32_t rank, uint32_t slot, uint32_t shard_count, uint32_t slots, uint32_t tile_w, uint32_t tile_h, uint32_t width, uint32_t height, uint32_t& px, uint32_t& py) {
  FrameParams p;  // only the fields slot_to_pixel reads
  p.paths_per_seed = slots;
  p.tile_w = tile_w;
  p.tile_h = tile_h;
  p.shard_count = shard_count;
  p.shard_rank = rank;
  p.tiles_x = (width + tile_w - 1) / tile_w;
  p.tiles_y = (height + tile_h - 1) / tile_h;
  p.pc.gOutputExtent[0] = width;
  p.pc.gOutputExtent[1] = height;
  return slot_to_pixel(p, slot, px, py);
}
#ifndef STHIP_TEMPLATE_INSTANCES_ONLY
inline __global__ void __launch_bounds__(STHIP_BLOCK) k_assemble_tiles(const uint32_t* packed, size_t rank_stride, uint32_t shard_count, uint32_t slots, uint32_t tile_w, uint32_t tile_h, uint32_t width,
                                                                 uint32_t height, uint32_t words, uint32_t* frame) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)shard_count * slots) return;
  const uint32_t rank = (uint32_t)(i / slots), slot = (uint32_t)(i % slots);
  uint32_t px, py;
  if (!shard_slot_pixel(rank, slot, shard_count, slots, tile_w, tile_h, width, height, px, py)) return;
  const uint32_t* src = packed + ((size_t)rank * rank_stride + slot) * words;
  uint32_t* dst = frame + ((size_t)py * width + px) * words;
  for (uint32_t k = 0; k < words; k++) dst[k] = src[k];
}
#endif
// The other direction (sthip_pack_tiles): a W x H image on one rank -> that rank's tiles in slot order (slots outside the
// image: zero). What the ranks exchange of the G-buffer outputs, which sthip_render writes as images.
// BDPTDebugMode eEnvironmentSampleTest / eEnvironmentSamplePDF (bdpt.hlsl:190-205): sample_visibility returns before it
// traces anything — eight environment samples drawn from the path's stream as spots around the view direction, or the
// environment's pdf of that direction. Runs behind k_generate (which has made the view ray and loaded / cleared the pixel of
// the debug image) in place of the rounds; the paths are marked dead, radiance stays 0.
#ifndef STHIP_TEMPLATE_INSTANCES_ONLY
inline __global__ void __launch_bounds__(STHIP_BLOCK) k_debug_environment(FrameParams p) {
  for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < p.path_count; slot += gridDim.x * blockDim.x) {
    const uint32_t meta = p.meta[slot];
    if (meta >= 0xFFFFFFFEu) continue;
    p.meta[slot] = 0xFFFFFFFEu;  // nothing is traced
    p.radiance[slot] = make_float4(0, 0, 0, 0);
    if (!(p.scene_flags & STHIP_BDPT_FLAG_HAS_ENVIRONMENT)) continue;  // (the record at gEnvironmentMaterialAddress is not an environment: not restated)
    uint32_t px, py;
    slot_to_pixel(p, slot, px, py);
    const f3 direction = xyz(p.ray_d[slot]);
    Environment env;
    env.load(p.scene, p.pc.gEnvironmentMaterialAddress);
    float4 dbg = p.debug[slot];
    if (p.debug_mode == STHIP_DEBUG_ENVIRONMENT_SAMPLE_TEST) {
      Rng rng;
      rng.x = px;
      rng.y = py;
      rng.seed = p.seed + slot / p.paths_per_seed;
      rng.counter = 0u;
      for (uint32_t i = 0; i < 8; i++) {
        float pdf;
        f3 dir;
        const float r0 = rng.next_float(), r1 = rng.next_float();
        (void)env.sample(p.scene, r0, r1, dir, pdf, flag(p, STHIP_eSampleEnvironmentMapDirectly));
        const float v = 1024 * det_powf(fmaxf(0.0f, dot3(dir, direction)), 1024.0f);
        dbg.x = dbg.x + v;
        dbg.y = dbg.y + v;
        dbg.z = dbg.z + v;
      }
    } else {
      const float pdf = env.eval_pdf(p.scene, direction, flag(p, STHIP_eSampleEnvironmentMapDirectly));
      dbg.x = dbg.y = dbg.z = pdf;
    }
    p.debug[slot] = dbg;
  }
}
#endif

// What shading needs of a leaf triangle's vertices apart from their positions (bvh.h: BvhTriShade), written once per scene
// upload beside the leaf triangles as they lie in HBM: the triangle says where its index triple is (BvhTri::src_indices /
// src_vertex), the vertices are read again — so the record holds the floats shading would have gathered (shading_data.hlsli:2-6)
#ifndef STHIP_TEMPLATE_INSTANCES_ONLY
inline __global__ void __launch_bounds__(STHIP_BLOCK) k_fill_tri_shade(const BvhTri* tris, uint32_t n, const uint8_t* is_tri, const sthip_PackedVertexData* vertices, uint32_t vertex_count, const uint8_t* indices,
                                                                uint64_t indices_bytes, BvhTriShade* out, BvhTriUv* out_uv) {
  for (uint32_t i = blockIdx.x * STHIP_BLOCK + threadIdx.x; i < n; i += gridDim.x * STHIP_BLOCK) {
    BvhTriShade r;
    memset(&r, 0, sizeof(r));
    if (!is_tri || is_tri[i]) {
      const BvhTri t = tris[i];
      const uint32_t stride = (t.src_vertex >> 31) ? 4u : 2u, first = t.src_vertex & 0x7FFFFFFFu;
      uint32_t idx[3] = {0, 0, 0};
      if ((uint64_t)t.src_indices + 3u * stride <= indices_bytes)
        for (int k = 0; k < 3; k++) {  // byte loads: an index buffer's byte offset need not be aligned to its stride
          const uint8_t* q = indices + (size_t)t.src_indices + (size_t)k * stride;
          idx[k] = stride == 2u ? ((uint32_t)q[0] | (uint32_t)q[1] << 8) : ((uint32_t)q[0] | (uint32_t)q[1] << 8 | (uint32_t)q[2] << 16 | (uint32_t)q[3] << 24);
          idx[k] += first;
          if (idx[k] >= vertex_count) idx[k] = 0;  // (the builders have refused such scenes already)
        }
      const sthip_PackedVertexData a = vertices[idx[0]], b = vertices[idx[1]], c = vertices[idx[2]];
      for (int k = 0; k < 3; k++) {
        r.n0[k] = a.normal[k];
        r.n1[k] = b.normal[k];
        r.n2[k] = c.normal[k];
      }
      r.v0 = a.v;
      r.v1 = b.v;
      r.v2 = c.v;
      r.u[0] = a.u;
      r.u[1] = b.u;
      r.u[2] = c.u;
    }
    out[i] = r;
    if (out_uv) {  // (scenes with alpha masks) what the traversal's alpha test interpolates
      BvhTriUv q;
      q.uv[0][0] = r.u[0];
      q.uv[0][1] = r.v0;
      q.uv[1][0] = r.u[1];
      q.uv[1][1] = r.v1;
      q.uv[2][0] = r.u[2];
      q.uv[2][1] = r.v2;
      out_uv[i] = q;
    }
  }
}
#endif

// The seed-split replica mode (sthip.h: sthip_radiance_to_sums): a call's output is (mean over its seeds, their number); what a
// sum-reduce over replicas can add up is (sum over its seeds, their number). Back: mean = sum / number, correctly rounded.
#ifndef STHIP_TEMPLATE_INSTANCES_ONLY
inline __global__ void __launch_bounds__(STHIP_BLOCK) k_radiance_sums(float4* image, size_t n, uint32_t to_sums) {
  for (size_t i = blockIdx.x * (size_t)STHIP_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * STHIP_BLOCK) {
    float4 v = image[i];
    if (to_sums) {
      v.x = v.x * v.w;
      v.y = v.y * v.w;
      v.z = v.z * v.w;
    } else if (v.w > 0.0f) {
      v.x = v.x / v.w;
      v.y = v.y / v.w;
      v.z = v.z / v.w;
    }
    image[i] = v;
  }
}
#endif

#ifndef STHIP_TEMPLATE_INSTANCES_ONLY
inline __global__ void __launch_bounds__(STHIP_BLOCK) k_pack_tiles(const uint32_t* image, uint32_t rank, uint32_t shard_count, uint32_t slots, uint32_t tile_w, uint32_t tile_h, uint32_t width, uint32_t height,
                                                             uint32_t words, uint32_t* packed) {
  const size_t slot = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= slots) return;
  uint32_t px, py;
  const bool inside = shard_slot_pixel(rank, (uint32_t)slot, shard_count, slots, tile_w, tile_h, width, height, px, py);
  uint32_t* dst = packed + slot * words;
  const uint32_t* src = image + ((size_t)py * width + px) * words;
  for (uint32_t k = 0; k < words; k++) dst[k] = inside ? src[k] : 0u;
}
#endif
