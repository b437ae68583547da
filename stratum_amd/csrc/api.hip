// api.hip — the C ABI of include/sthip.h: context, scene upload, the per-frame launch sequence.
//
// sthip_render is the replacement of the dispatch sequence BDPT::render records
// (src/Node/BDPT.cpp:607-720: fill gShadowRays, dispatch sample_visibility, barrier, dispatch
// trace_shadows) and of the running-mean accumulation the denoiser applies with default settings
// (src/Node/Denoiser.cpp:73,186-213). There is no CPU fallback: every entry point needs a HIP device.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

#include "../../include/sthip.h"
#include "bvh_build.h"
#include "kernel_instances.h"  // kernels.h + the instantiations that live in shade_*.hip / trace_kernels.hip
#include "post.h"
#include "ceilings.h"

STHIP_DECLARE_KERNEL_INSTANCES

namespace sthip {
hipError_t lvc_compact(const float4* staging, uint32_t slots_per_seed, uint32_t seeds, uint32_t vertices_per_seed, float4* cache, uint32_t* counts, uint32_t* flags, uint32_t* offsets,
                       void* tmp, size_t& tmp_bytes, hipStream_t stream, uint32_t rec = 4, uint32_t flag_at = 2);  // lvc.hip
hipError_t hashgrid_build_device(const uint2* keys, const uint32_t* count, uint32_t slots, uint32_t buckets, uint32_t* checksums, uint32_t* counters, uint32_t* indices, uint32_t* dest, uint32_t* owner,
                                 uint32_t* bucket_of, uint32_t* append_index, uint32_t* sorted_bucket, uint32_t* sorted_append, unsigned long long* key64, unsigned long long* sorted_key64, void* tmp,
                                 size_t& tmp_bytes, hipStream_t stream, bool force_serial = false);  // hashgrid.hip
}

namespace {
thread_local std::string g_create_error;

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }  // an early return (HIP_TRY) must not leak a local staging buffer
  hipError_t ensure(size_t count) {
    if (count <= n && p) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
    if (count == 0) return hipSuccess;
    hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
    if (e == hipSuccess) n = count;
    // tests: STHIP_POISON_ALLOC=<byte> fills every new device buffer with that byte, so that a read of something nothing has
    // written shows in a fresh process too (where new device memory is zero pages) and not only once the heap is recycled
    static const int poison = [] {
      const char* v = getenv("STHIP_POISON_ALLOC");
      return v && *v ? (int)(strtoul(v, nullptr, 0) & 0xFFu) : -1;
    }();
    if (e == hipSuccess && poison >= 0) e = hipMemset(p, poison, count * sizeof(T));
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
};
// The unpacked nodes the host keeps (treetop selection, top-level rebuilds): page-locked, so that the device-resident
// builder's one node copy runs at PCIe speed, and never zero-filled
struct PinnedNodes {
  BvhNode* p = nullptr;
  size_t n = 0, cap = 0;
  PinnedNodes() = default;
  PinnedNodes(const PinnedNodes&) = delete;
  PinnedNodes& operator=(const PinnedNodes&) = delete;
  ~PinnedNodes() {
    if (p) (void)hipHostFree(p);
  }
  hipError_t ensure(size_t count) {  // capacity only; contents are lost when it grows
    if (count <= cap && p) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = n = 0;
    hipError_t e = hipHostMalloc((void**)&p, std::max<size_t>(1, count) * sizeof(BvhNode), hipHostMallocDefault);
    if (e == hipSuccess) cap = std::max<size_t>(1, count);
    return e;
  }
  bool empty() const { return n == 0; }
  size_t size() const { return n; }
  BvhNode* data() { return p; }
};
}  // namespace

// The scene as it was uploaded, kept on the host ("keep_scene" = 1, the default): what a change that the resident tree cannot
// follow — an instance of the merged world-space mesh that moves — is rebuilt from, inside sthip_scene_update_transforms
// (the reference rebuilds whatever is dirty, Scene.cpp:345,435-459,614-629: it has the scene graph to rebuild from).
struct KeptScene {
  std::vector<sthip_PackedVertexData> vertices;
  std::vector<uint8_t> indices, materials;
  std::vector<sthip_InstanceData> instances;
  std::vector<sthip_TransformData> xf, inv, motion;
  std::vector<uint32_t> lights;
  std::vector<std::vector<float>> image_pixels, image1_pixels;
  std::vector<sthip_image_desc> images, images1;
  std::vector<float> distributions;
  std::vector<std::vector<uint8_t>> volume_bytes;
  std::vector<sthip_volume_desc> volumes;
  bool valid = false;
  void keep(const sthip_scene_desc& s) {
    vertices.assign(s.gVertices, s.gVertices + (s.gVertices ? s.vertex_count : 0));
    indices.assign((const uint8_t*)s.gIndices, (const uint8_t*)s.gIndices + (s.gIndices ? s.indices_bytes : 0));
    materials.assign((const uint8_t*)s.gMaterialData, (const uint8_t*)s.gMaterialData + (s.gMaterialData ? s.material_bytes : 0));
    instances.assign(s.gInstances, s.gInstances + s.instance_count);
    xf.assign(s.gInstanceTransforms, s.gInstanceTransforms + s.instance_count);
    inv.assign(s.gInstanceInverseTransforms, s.gInstanceInverseTransforms + s.instance_count);
    motion.clear();
    if (s.gInstanceMotionTransforms) motion.assign(s.gInstanceMotionTransforms, s.gInstanceMotionTransforms + s.instance_count);
    lights.assign(s.gLightInstances, s.gLightInstances + (s.gLightInstances ? s.light_count : 0));
    auto take = [](const sthip_image_desc* in, uint32_t n, size_t channels, std::vector<std::vector<float>>& px, std::vector<sthip_image_desc>& d) {
      px.assign(n, {});
      d.assign(n, sthip_image_desc{});
      for (uint32_t i = 0; i < n; i++) {
        px[i].assign(in[i].pixels, in[i].pixels + (size_t)in[i].width * in[i].height * channels);
        d[i] = sthip_image_desc{px[i].data(), in[i].width, in[i].height};
      }
    };
    take(s.gImages, s.gImages ? s.image_count : 0, 4, image_pixels, images);
    take(s.gImage1s, s.gImage1s ? s.image1_count : 0, 1, image1_pixels, images1);
    distributions.assign(s.gDistributions, s.gDistributions + (s.gDistributions ? s.distribution_count : 0));
    volume_bytes.assign(s.gVolumes ? s.volume_count : 0, {});
    volumes.assign(volume_bytes.size(), sthip_volume_desc{});
    for (size_t i = 0; i < volume_bytes.size(); i++) {
      volume_bytes[i].assign((const uint8_t*)s.gVolumes[i].data, (const uint8_t*)s.gVolumes[i].data + s.gVolumes[i].bytes);
      volumes[i] = sthip_volume_desc{volume_bytes[i].data(), s.gVolumes[i].bytes};
    }
    valid = true;
  }
  sthip_scene_desc desc() const {
    sthip_scene_desc d{};
    d.gVertices = vertices.data();
    d.vertex_count = (uint32_t)vertices.size();
    d.gIndices = indices.data();
    d.indices_bytes = (uint32_t)indices.size();
    d.gInstances = instances.data();
    d.instance_count = (uint32_t)instances.size();
    d.gInstanceTransforms = xf.data();
    d.gInstanceInverseTransforms = inv.data();
    d.gInstanceMotionTransforms = motion.empty() ? nullptr : motion.data();
    d.gMaterialData = materials.data();
    d.material_bytes = (uint32_t)materials.size();
    d.gLightInstances = lights.data();
    d.light_count = (uint32_t)lights.size();
    d.gImages = images.empty() ? nullptr : images.data();
    d.image_count = (uint32_t)images.size();
    d.gDistributions = distributions.empty() ? nullptr : distributions.data();
    d.distribution_count = (uint32_t)distributions.size();
    d.gImage1s = images1.empty() ? nullptr : images1.data();
    d.image1_count = (uint32_t)images1.size();
    d.gVolumes = volumes.empty() ? nullptr : volumes.data();
    d.volume_count = (uint32_t)volumes.size();
    return d;
  }
};

struct sthip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string error;
  int cu_count = 256;
  // scene
  bool has_scene = false;
  bool textured = false;      // some material binds an image: k_shade<true> (ray cones, image values, normal maps)
  bool has_alpha = false;     // some triangle material has an alpha mask (gImage1s)
  bool has_spheres = false;   // some instance is a sphere: k_shade<., true>
  bool has_volumes = false;   // some instance is a volume (a Medium): the media instantiations
  uint32_t volume_count = 0, volume_instances = 0;
  std::vector<uint8_t> instance_is_volume;
  sthip::TopLevelState top;  // what sthip_scene_update_transforms rebuilds the top level from
  KeptScene kept;            // ... and what it rebuilds everything from when the top level alone cannot follow the change
  bool keep_scene = true;    // "keep_scene"
  size_t nodes_capacity = 0;
  DevBuf<uint32_t> volume_words;
  DevBuf<DeviceVolume> volumes;
  std::vector<uint8_t> materials_host;                    // gMaterialData as uploaded (validation of the environment record)
  std::vector<std::pair<uint32_t, uint32_t>> image_dims;  // (width, height) of gImages
  uint32_t distribution_count = 0;
  bool has_specular = false;  // some material satisfies DisneyMaterial::is_specular (disney_material.hlsli:125)
  DevBuf<sthip_PackedVertexData> vertices;
  DevBuf<uint8_t> indices;
  DevBuf<sthip_InstanceData> instances;
  DevBuf<sthip_TransformData> xf, inv_xf, motion_xf;
  DevBuf<uint8_t> materials;
  DevBuf<uint32_t> lights;
  DevBuf<DeviceImage> images;
  DevBuf<float4> image_texels;
  uint32_t image_count = 0;
  DevBuf<float2> cone;
  uint32_t instance_count = 0, light_count = 0;
  DevBuf<BvhNodeSlot> nodes;
  DevBuf<BvhTri> tris;
  DevBuf<TlasEntry> entries;
  DevBuf<WideNode> wide_nodes;
  DevBuf<TlasEntry> wide_entries;
  DevBuf<Wide8Node> wide8_nodes;
  DevBuf<TlasEntry> wide8_entries;
  std::vector<Wide8Node> wide8_host;  // the 8-wide nodes as uploaded: a transforms-only update makes the top level's again behind the bottom levels'
  size_t wide8_node_count = 0;
  uint32_t tri_min_lanes = 1;  // "tri_min_lanes" (DeviceBvh::wide8_tri_min)
  DeviceBvh bvh{};
  uint64_t bvh_nodes = 0, bvh_tris = 0;
  // treetop (bvh_build.h): rebuilt whenever the top level changes; needs the nodes on the host
  PinnedNodes nodes_host;
  DevBuf<BvhNode> raw_nodes;  // device-resident build only: the unpacked nodes before their one copy to nodes_host
  DevBuf<BvhNodePacked> top_nodes;
  DevBuf<TlasEntry> top_entries;
  bool use_treetop = false;  // "treetop": measured +1.5 % before k_trace's loop lost its other exec-mask regions, -0.5 % after
  // host builds: leaf triangles in the node array behind their parent (bvh_build.h: BuiltBvh::embedded). Measured on the
  // bench scene: k_trace 2.30 ms either way (the leaf fetch is not what a step waits for), so off: two arrays are simpler
  bool embed_leaves = false;
  // "wide_bvh": k_trace walks the 4-wide form of the tree (takes effect at the next sthip_scene_upload). 1 (default) = always:
  // host-built trees are collapsed on the host, GPU-built ones and the trees of a transforms-only update on the device
  // (wide.hip). 0 = never (the binary walk every other kernel uses), 2 = only when the binary nodes do not fit one XCD's L2
  // (4 MiB). Bench scene: k_trace -20 % against the binary walk; instanced forest (a 2.6 MB tree): equal.
  // 3 = the 8-wide compressed form (bvh.h: Wide8Node) for host-built trees (collapsed on the host, at upload and at every
  // transforms-only update); trees it cannot take (GPU-built ones, embedded leaves) get the 4-wide form as with 1.
  int use_wide = 1;
  size_t wide_node_count = 0;
  sthip::DeviceWideScratch* wide_scratch = nullptr;  // buffers of the device-side collapse (wide.hip), kept between calls
  bool want_wide = false;                            // the current scene is walked in its 4-wide form (decided at upload)
  bool lds_materials = true;  // k_shade stages gMaterialData in LDS when it fits 32 KB
  // frame
  DevBuf<uint8_t> views;  // gViews | gViewTransforms | gPrevViews | gPrevInverseViewTransforms
  DevBuf<float4> ray_o, ray_d, hit, beta, radiance, shadow_sum, accum, shadow_rays, light_vertices, conn, media_state, shadow_hit, shadow_ext, shadow_result;
  DevBuf<uint32_t> view_medium;
  DevBuf<uint32_t> meta, queue0, queue1, queue_kept;
  // "cull_terminal": 1 (default) = in a round where paths can reach their last vertex, k_cull_terminal hands k_shade only the
  // paths that still have something to do (kernels.h); 0 = k_shade sees the whole queue. Same frames either way.
  int cull_terminal = 1;
  DevBuf<unsigned long long> counters;
  DevBuf<float> distributions;  // gDistributions
  DevBuf<float4> presampled;    // gPresampledLights
  DevBuf<float4> bdpt;          // BDPT quantities per path (eConnectToViews)
  DevBuf<float4> rr;  // eCoherentRR: probes / group verdicts of the round in flight (FrameParams::rr)
  DevBuf<uint2> cs_nee, cs_lvc;  // eCoherentSampling: probes / group values of the round in flight (FrameParams::cs_nee, cs_lvc)
  DevBuf<float4> lvc_staging, path_contrib;  // eLVC: staged light vertices, the light paths' path_contrib (eLVCReservoirs)
  DevBuf<uint32_t> lvc_count, lvc_flags, lvc_offsets;
  DevBuf<uint8_t> lvc_tmp;
  // eNEEReservoirReuse: the append stage and its compaction, the keys / destinations of the host-side probing, the grid
  DevBuf<float4> hg_appends, hg_compact, hg_data;
  DevBuf<uint32_t> hg_count, hg_flags, hg_offsets, hg_dest, hg_checksums, hg_counters, hg_indices;
  DevBuf<uint32_t> hg_owner, hg_bucket_of, hg_append, hg_sorted_bucket, hg_sorted_append;  // scratch of the device-side grid build (hashgrid.hip)
  DevBuf<unsigned long long> hg_key64, hg_sorted_key64;
  DevBuf<uint2> hg_keys;
  DevBuf<uint8_t> hg_tmp;
  DevBuf<float4> lg_appends, lg_compact, lg_data;  // eLVCReservoirReuse: the same for gLVCHashGrid (keys / flags / tmp are shared)
  DevBuf<uint32_t> lg_checksums, lg_counters, lg_indices;
  DevBuf<uint32_t> light_trace; // gLightTraceSamples
  DevBuf<DeviceImage1> images1;  // gImage1s (alpha masks)
  DevBuf<float> image1_texels;
  DevBuf<BvhTriUv> tri_uvs;
  DevBuf<BvhTriShade> tri_shade;  // beside the leaf triangles: their vertices' normals and uvs (k_fill_tri_shade)
  DevBuf<uint32_t> hit_leaf;      // per path: the leaf triangle of its hit
  DevBuf<uint32_t> shade_stack;   // media without eDeferShadowRays: a traversal stack column per k_shade thread (visibility_walk_media)
  DevBuf<float4> debug, out_debug, shadow_debug;  // BDPTDebugMode: per-path pixel of gDebugImage, the image's staging (host pointers), the debug halves of inline shadow rays
  DevBuf<uint32_t> inst_alpha;
  DevBuf<uint8_t> inst_flags;  // per instance: INST_FLAG_* of its (untextured) material, for k_cull_terminal
  std::vector<uint8_t> inst_flags_host;
  // "answer_last_rays": 1 (default) = the last ray of a path is only queued if it can reach the bounds of an emissive instance
  // (kernels.h: aims_at_emitter); 0 = every ray is queued and traced. Same frames and ray counts either way.
  int answer_last_rays = 1;
  DevBuf<EmitterBounds> emitters;
  uint32_t emitter_count = 0;  // 0: not applicable to this scene (no or too many emissive triangle instances)
  DevBuf<unsigned long long> qctl;  // queue control lines (queue_ctl)
  DevBuf<uint32_t> post_scratch;  // maxima / metric accumulator of post.h
  DevBuf<sthip_ray> ray_staging;  // sthip_trace_rays with host pointers
  DevBuf<sthip_hit> hit_staging;
  DevBuf<float4> out_radiance, out_albedo;
  DevBuf<sthip_VisibilityInfo> out_visibility;
  DevBuf<sthip_DepthInfo> out_depth;
  DevBuf<float2> out_prev_uv;
  uint32_t shard_rank = 0, shard_count = 1, tile_w = 64, tile_h = 32;
  // options / stats
  bool count_traversal = false, time_kernels = false;
  bool hashgrid_serial = false;  // "hashgrid_serial": build the reuse grids with the one-thread probe sequence (hashgrid.hip's rare-case path; tests)
  // "reuse_grids_persist": the grids the LAST seed of a call leaves are what the FIRST seed of the next call looks into — upstream's
  // previous frame for a host that renders one frame per call (BDPT.cpp:482-483,621-627). The key says what they were built for.
  bool reuse_persist = false, reuse_grids_valid = false;
  uint64_t reuse_key[3] = {0, 0, 0};
  uint32_t refill_idle = 16, inner_min_lanes = 24, trace_blocks_per_cu = 0;
  uint64_t max_paths_in_flight = 1ull << 22;  // (sthip_create sizes it to the device: 2^26 on a 288 GB MI355X — launches large enough that their
                                              // tails stop mattering: atrium x 8 seeds +7 %, forest at 4K x 16 +38 % over 2^22; tools/in_flight_sweep.py)
  bool packet_primary = true;  // the first bounce is traced as wave packets (k_trace_primary)
  bool fuse_trace = true;  // closest-hit rays of a bounce and the shadow rays of the previous one in one launch
  int bvh_builder = 0;  // sthip::BvhBuilderKind
  uint32_t sah_top_size = 64;  // "sah_top" (measured 32 .. 16384: 64 traces fastest): the GPU builder's subtrees of at most this many triangles get a host-built SAH top (0: off)
  int lbvh_algorithm = 1, ploc_radius = 4;  // of the GPU builder (bvh_build.h: DeviceBuildTarget); radius measured: 4 traces fastest (2 .. 32 tried)
  // levels of the per-lane LDS traversal stack at most; a higher tree runs the BOUNDED instantiations (traverse.h) with the
  // full stack of an overflowing ray in global memory (spill)
  // measured (atrium, PLOC tree 40 high): the bounded instantiation costs ~2.5 % per step, a fourth resident block is worth
  // ~13 %, a treetop ~3 %: so the whole stack stays in LDS as long as four blocks of it fit (40 levels = 160 KB per CU)
  uint32_t lds_stack_threshold = 40, lds_stack_cap = 32;
  DevBuf<uint32_t> spill;
  DevBuf<float4> deep_rays;  // rays that overflowed a bounded LDS stack (k_trace_deep), and their count
  DevBuf<uint32_t> deep_count;
  sthip_stats stats{};
  hipError_t last_hip_error = hipSuccess;  // of the last failed HIP_TRY (sthip_render halves its batch after an out-of-memory)
  bool render_launched = false;            // the render call in progress has enqueued work (no second attempt from here on)
  bool stats_pending = false;  // ray / traversal counters of the last render still live on the device
  hipEvent_t ev[2] = {nullptr, nullptr};
};

// The traversal stack is stack_depth x 1 KB of dynamic LDS per 256-thread block. Up to 64 KB needs nothing; beyond it
// (deep LBVH trees of clustered scenes) the kernels must be told (hipFuncAttributeMaxDynamicSharedMemorySize), and the
// CU holds fewer blocks. 152 KB leaves room for the runtime's own LDS use.
#define STHIP_MAX_STACK_LDS ((size_t)152 * 1024)
// Trees higher than the LDS cap (lds_stack_levels) keep only that many levels in LDS; the rare ray that needs more is traced
// again with a global-memory stack of the tree's full height — bounded here so that the spill buffer stays small
#define STHIP_MAX_STACK_DEPTH 512u

// k_trace's instantiations by (count_traversal, alpha / volumes, bounded stack, treetop)
static const void* trace_kernel(bool count, bool alpha, bool bounded, bool top) {
  static const void* const table[16] = {
      (const void*)&k_trace<false, false, false, false>, (const void*)&k_trace<true, false, false, false>, (const void*)&k_trace<false, true, false, false>, (const void*)&k_trace<true, true, false, false>,
      (const void*)&k_trace<false, false, true, false>,  (const void*)&k_trace<true, false, true, false>,  (const void*)&k_trace<false, true, true, false>,  (const void*)&k_trace<true, true, true, false>,
      (const void*)&k_trace<false, false, false, true>,  (const void*)&k_trace<true, false, false, true>,  (const void*)&k_trace<false, true, false, true>,  (const void*)&k_trace<true, true, false, true>,
      (const void*)&k_trace<false, false, true, true>,   (const void*)&k_trace<true, false, true, true>,   (const void*)&k_trace<false, true, true, true>,   (const void*)&k_trace<true, true, true, true>};
  return table[(count ? 1 : 0) | (alpha ? 2 : 0) | (bounded ? 4 : 0) | (top ? 8 : 0)];
}
// ... over the 8-wide compressed tree ("wide_bvh" = 3)
static const void* trace_kernel_wide8(bool count, bool alpha, bool bounded) {
  static const void* const table[8] = {
      (const void*)&k_trace<false, false, false, false, 2>, (const void*)&k_trace<true, false, false, false, 2>, (const void*)&k_trace<false, true, false, false, 2>, (const void*)&k_trace<true, true, false, false, 2>,
      (const void*)&k_trace<false, false, true, false, 2>,  (const void*)&k_trace<true, false, true, false, 2>,  (const void*)&k_trace<false, true, true, false, 2>,  (const void*)&k_trace<true, true, true, false, 2>};
  return table[(count ? 1 : 0) | (alpha ? 2 : 0) | (bounded ? 4 : 0)];
}
// ... and over the 4-wide tree ("wide_bvh"; never with the treetop)
static const void* trace_kernel_wide(bool count, bool alpha, bool bounded) {
  static const void* const table[8] = {
      (const void*)&k_trace<false, false, false, false, 1>, (const void*)&k_trace<true, false, false, false, 1>, (const void*)&k_trace<false, true, false, false, 1>, (const void*)&k_trace<true, true, false, false, 1>,
      (const void*)&k_trace<false, false, true, false, 1>,  (const void*)&k_trace<true, false, true, false, 1>,  (const void*)&k_trace<false, true, true, false, 1>,  (const void*)&k_trace<true, true, true, false, 1>};
  return table[(count ? 1 : 0) | (alpha ? 2 : 0) | (bounded ? 4 : 0)];
}

#define HIP_TRY(ctx, expr)                                                                            \
  do {                                                                                                \
    hipError_t _e = (expr);                                                                           \
    if (_e != hipSuccess) {                                                                           \
      (ctx)->error = std::string(#expr) + ": " + hipGetErrorString(_e);                               \
      (ctx)->last_hip_error = _e;                                                                     \
      return STHIP_ERR_HIP;                                                                           \
    }                                                                                                 \
  } while (0)

static void fill_counter_stats(sthip_ctx* ctx, const unsigned long long* c) {
  ctx->stats.rays_total = c[CNT_RAYS_CLOSEST] + c[CNT_RAYS_SHADOW];
  ctx->stats.rays_path = c[CNT_RAYS_CLOSEST] - c[CNT_CROSSINGS];
  ctx->stats.rays_shadow = c[CNT_RAYS_SHADOW];
  ctx->stats.nodes_visited = c[CNT_NODES];
  ctx->stats.tris_tested = c[CNT_TRIS];
  ctx->stats.nodes_visited_shadow = c[CNT_NODES + 1];
  ctx->stats.tris_tested_shadow = c[CNT_TRIS + 1];
  for (int k = 0; k < 2; k++) {
    ctx->stats.inner_slots[k] = c[CNT_INNER_SLOTS + k];
    ctx->stats.tri_slots[k] = c[CNT_TRI_SLOTS + k];
    ctx->stats.round_slots[k] = c[CNT_ROUND_SLOTS + k];
    ctx->stats.busy_rounds[k] = c[CNT_BUSY_ROUNDS + k];
  }
  for (int k = 0; k < 8; k++) ctx->stats.lane_states[k] = c[CNT_LANE_STATES + k];
  ctx->stats.rays_answered = c[CNT_RAYS_ANSWERED];
  ctx->stats.nodes_visited_primary = c[CNT_NODES_PRIMARY];
  ctx->stats.tris_tested_primary = c[CNT_TRIS_PRIMARY];
}

static uint32_t grid_for_early(const sthip_ctx* ctx, size_t n) {  // (grid_for, for code above its definition)
  const size_t blocks = (n + STHIP_BLOCK - 1) / STHIP_BLOCK;
  return (uint32_t)std::max<size_t>(1, std::min(blocks, (size_t)ctx->cu_count * 32));
}

static int fail(sthip_ctx* ctx, int code, const std::string& msg) {
  ctx->error = msg;
  return code;
}

extern "C" {

int sthip_abi_version(void) { return STHIP_ABI_VERSION; }

int sthip_create(int device, sthip_ctx** out_ctx) {
  if (!out_ctx) {
    g_create_error = "out_ctx is NULL";
    return STHIP_ERR_INVALID_ARGUMENT;
  }
  *out_ctx = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0) {
    g_create_error = std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0") +
                     " (libstratum_hip has no CPU backend)";
    return STHIP_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= count) {
    g_create_error = "device index out of range";
    return STHIP_ERR_INVALID_ARGUMENT;
  }
  e = hipSetDevice(device);
  if (e != hipSuccess) {
    g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
    return STHIP_ERR_HIP;
  }
  sthip_ctx* ctx = new sthip_ctx();
  ctx->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
    ctx->cu_count = prop.multiProcessorCount;
    // paths in flight by default: the largest power of two that leaves 4 KB of device memory per path (a path costs ~330 B of
    // state, queues and shadow records at the default flags, more with light subpaths): 2^26 on a 288 GB MI355X
    // — of the memory that is FREE now: a host application (or other contexts on the device) may hold most of it; should the
    // batch still not fit when a render allocates it, sthip_render halves it and tries again
    size_t free_bytes = 0, total_bytes = 0;
    if (hipMemGetInfo(&free_bytes, &total_bytes) != hipSuccess) free_bytes = prop.totalGlobalMem;
    uint64_t paths = 1ull << 18;
    while (paths * 2 * 4096 <= (uint64_t)free_bytes && paths < (1ull << 27)) paths *= 2;
    ctx->max_paths_in_flight = paths;
  }
  (void)hipEventCreate(&ctx->ev[0]);
  (void)hipEventCreate(&ctx->ev[1]);
  {  // dynamic LDS beyond the 64 KB default for the kernels that carry the traversal stack
    const int lds_max = 160 * 1024;
    for (int k = 0; k < 16; k++) (void)hipFuncSetAttribute(trace_kernel((k & 1) != 0, (k & 2) != 0, (k & 4) != 0, (k & 8) != 0), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    for (int k = 0; k < 8; k++) (void)hipFuncSetAttribute(trace_kernel_wide((k & 1) != 0, (k & 2) != 0, (k & 4) != 0), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    for (int k = 0; k < 8; k++) (void)hipFuncSetAttribute(trace_kernel_wide8((k & 1) != 0, (k & 2) != 0, (k & 4) != 0), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trace_batch<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trace_batch<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trace_batch<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trace_batch<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
  }
  ctx->stats.bvh_node_bytes = sizeof(BvhNodePacked);
  ctx->stats.bvh_tri_bytes = sizeof(BvhTri);
  *out_ctx = ctx;
  return STHIP_OK;
}

void sthip_destroy(sthip_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  sthip::device_wide_scratch_destroy(ctx->wide_scratch);
  ctx->wide_scratch = nullptr;
  ctx->vertices.release();
  ctx->volume_words.release();
  ctx->volumes.release();
  ctx->indices.release();
  ctx->instances.release();
  ctx->xf.release();
  ctx->inv_xf.release();
  ctx->motion_xf.release();
  ctx->materials.release();
  ctx->lights.release();
  ctx->images.release();
  ctx->image_texels.release();
  ctx->cone.release();
  ctx->nodes.release();
  ctx->tris.release();
  ctx->entries.release();
  ctx->views.release();
  ctx->ray_o.release();
  ctx->ray_d.release();
  ctx->hit.release();
  ctx->hit_leaf.release();
  ctx->debug.release();
  ctx->shadow_debug.release();
  ctx->beta.release();
  ctx->radiance.release();
  ctx->shadow_sum.release();
  ctx->accum.release();
  ctx->shadow_rays.release();
  ctx->light_vertices.release();
  ctx->media_state.release();
  ctx->shade_stack.release();
  ctx->shadow_hit.release();
  ctx->shadow_ext.release();
  ctx->shadow_result.release();
  ctx->view_medium.release();
  ctx->conn.release();
  ctx->meta.release();
  ctx->queue0.release();
  ctx->queue1.release();
  ctx->counters.release();
  ctx->distributions.release();
  ctx->presampled.release();
  ctx->bdpt.release();
  ctx->light_trace.release();
  ctx->images1.release();
  ctx->image1_texels.release();
  ctx->tri_uvs.release();
  ctx->tri_shade.release();
  ctx->hit_leaf.release();
  ctx->inst_alpha.release();
  ctx->inst_flags.release();
  ctx->qctl.release();
  ctx->post_scratch.release();
  ctx->out_radiance.release();
  ctx->out_albedo.release();
  ctx->out_visibility.release();
  ctx->out_depth.release();
  ctx->out_prev_uv.release();
  if (ctx->ev[0]) (void)hipEventDestroy(ctx->ev[0]);
  if (ctx->ev[1]) (void)hipEventDestroy(ctx->ev[1]);
  delete ctx;
}

const char* sthip_last_error(const sthip_ctx* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int sthip_set_stream(sthip_ctx* ctx, void* hip_stream) {
  if (!ctx) return STHIP_ERR_INVALID_ARGUMENT;
  ctx->stream = (hipStream_t)hip_stream;
  return STHIP_OK;
}

int sthip_set_shard(sthip_ctx* ctx, uint32_t shard_rank, uint32_t shard_count, uint32_t tile_w, uint32_t tile_h) {
  if (!ctx) return STHIP_ERR_INVALID_ARGUMENT;
  if (shard_count == 0 || shard_rank >= shard_count || tile_w == 0 || tile_h == 0 || (tile_w & 7u) || (tile_h & 7u))
    return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "shard: need rank < count and tile sizes that are multiples of 8");
  ctx->shard_rank = shard_rank;
  ctx->shard_count = shard_count;
  ctx->tile_w = tile_w;
  ctx->tile_h = tile_h;
  return STHIP_OK;
}

int sthip_set_option(sthip_ctx* ctx, const char* name, int64_t value) {
  if (!ctx || !name) return STHIP_ERR_INVALID_ARGUMENT;
  if (!strcmp(name, "count_traversal"))
    ctx->count_traversal = value != 0;
  else if (!strcmp(name, "time_kernels"))
    ctx->time_kernels = value != 0;
  else if (!strcmp(name, "fuse_trace"))
    ctx->fuse_trace = value != 0;
  else if (!strcmp(name, "packet_primary"))
    ctx->packet_primary = value != 0;
  else if (!strcmp(name, "answer_last_rays"))
    ctx->answer_last_rays = value != 0;
  else if (!strcmp(name, "cull_terminal"))
    ctx->cull_terminal = value != 0;
  else if (!strcmp(name, "refill_idle"))
    ctx->refill_idle = (uint32_t)std::min<int64_t>(64, std::max<int64_t>(1, value));
  else if (!strcmp(name, "bvh_builder"))
    ctx->bvh_builder = value == 1 ? 1 : 0;
  else if (!strcmp(name, "lds_materials"))
    ctx->lds_materials = value != 0;
  else if (!strcmp(name, "embed_leaves"))  // takes effect at the next sthip_scene_upload
    ctx->embed_leaves = value != 0;
  else if (!strcmp(name, "keep_scene")) {  // takes effect at the next sthip_scene_upload
    ctx->keep_scene = value != 0;
    if (!ctx->keep_scene) ctx->kept = KeptScene();
  } else if (!strcmp(name, "wide_bvh"))  // takes effect at the next sthip_scene_upload (host-built trees only)
    ctx->use_wide = (int)std::min<int64_t>(std::max<int64_t>(value, 0), 3);
  else if (!strcmp(name, "poison_deep_queue")) {  // tests: leaves the deep queue's control words as a call cut short between k_trace and k_trace_deep would (the next render must not care)
    if (value && ctx->deep_count.p) {
      const uint32_t junk[2] = {5u, 3u};
      (void)hipSetDevice(ctx->device);
      (void)hipStreamSynchronize(ctx->stream);
      if (hipMemcpy(ctx->deep_count.p, junk, 8, hipMemcpyHostToDevice) != hipSuccess) return fail(ctx, STHIP_ERR_HIP, "poison_deep_queue: copy failed");
    }
  } else if (!strcmp(name, "tri_min_lanes"))  // the 8-wide walk's leaf phase goes on while at least this many lanes hold a triangle
    ctx->tri_min_lanes = ctx->bvh.wide8_tri_min = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 1), 64);
  else if (!strcmp(name, "lbvh_algorithm"))  // 0: Karras radix tree, 1: PLOC (default)
    ctx->lbvh_algorithm = value == 0 ? 0 : 1;
  else if (!strcmp(name, "lds_stack_levels")) {  // takes effect at the next sthip_scene_upload / sthip_scene_update_transforms
    ctx->lds_stack_cap = ctx->lds_stack_threshold = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 4), 150);
  } else if (!strcmp(name, "ploc_radius"))
    ctx->ploc_radius = (int)std::min<int64_t>(std::max<int64_t>(value, 1), 32);
  else if (!strcmp(name, "sah_top"))
    ctx->sah_top_size = value <= 0 ? 0u : (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 16), 1 << 20);  // (the reserved host-node room covers frontiers down to 16)
  else if (!strcmp(name, "max_paths_in_flight"))
    ctx->max_paths_in_flight = (uint64_t)std::max<int64_t>(1, value);
  else if (!strcmp(name, "trace_blocks_per_cu"))
    ctx->trace_blocks_per_cu = (uint32_t)std::min<int64_t>(16, std::max<int64_t>(0, value));
  else if (!strcmp(name, "treetop"))  // takes effect at the next sthip_scene_upload / sthip_scene_update_transforms
    ctx->use_treetop = value != 0;
  else if (!strcmp(name, "hashgrid_serial"))
    ctx->hashgrid_serial = value != 0;
  else if (!strcmp(name, "reuse_grids_persist")) {  // setting it (to either value) also drops the grids kept so far
    ctx->reuse_persist = value != 0;
    ctx->reuse_grids_valid = false;
  }
  else if (!strcmp(name, "inner_min_lanes"))
    ctx->inner_min_lanes = (uint32_t)std::min<int64_t>(64, std::max<int64_t>(1, value));
  else
    return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, std::string("unknown option ") + name);
  return STHIP_OK;
}

int sthip_get_stats(sthip_ctx* ctx, sthip_stats* out) {
  if (!ctx || !out) return STHIP_ERR_INVALID_ARGUMENT;
  if (ctx->stats_pending) {
    unsigned long long c[CNT_TOTAL];
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(c, ctx->counters.p, sizeof(c), hipMemcpyDeviceToHost));
    fill_counter_stats(ctx, c);
    ctx->stats_pending = false;
  }
  *out = ctx->stats;
  out->bvh_node_bytes = ctx->bvh.wide8_nodes ? (uint32_t)sizeof(Wide8Node) : ctx->bvh.wide_nodes ? (uint32_t)sizeof(WideNode) : (uint32_t)sizeof(BvhNodePacked);  // of the nodes k_trace walks (its visits are what the counters count)
  out->bvh_tri_bytes = sizeof(BvhTri);
  out->bvh_nodes = ctx->bvh_nodes;
  out->bvh_tris = ctx->bvh_tris;
  return STHIP_OK;
}

static size_t stack_bytes(const sthip_ctx* ctx);
static size_t trace_lds_bytes(const sthip_ctx* ctx);
static int trace_occupancy(const sthip_ctx* ctx, size_t lds_bytes);
static int configure_stack(sthip_ctx* ctx);
static int refresh_treetop(sthip_ctx* ctx);
// packs `count` nodes and writes them into the node array from slot `first` on
static hipError_t upload_nodes(sthip_ctx* ctx, size_t first, const BvhNode* nodes, size_t count) {
  std::vector<BvhNodePacked> packed;
  sthip::pack_nodes(nodes, count, packed);
  std::vector<BvhNodeSlot> slots(count);
  memset(slots.data(), 0, count * sizeof(BvhNodeSlot));
  for (size_t i = 0; i < count; i++) slots[i].n = packed[i];
  return hipMemcpy(ctx->nodes.p + first, slots.data(), count * sizeof(BvhNodeSlot), hipMemcpyHostToDevice);
}

// The 4-wide form of the tree that is resident now (ctx->nodes / ctx->entries / ctx->bvh.root_ref), made on the device
// (wide.hip): after a GPU build and after a transforms-only update. On failure the binary walk stays (STHIP_OK): a tree whose
// boxes do not fit the wide nodes' grid is still a tree.
static int collapse_resident_tree(sthip_ctx* ctx) {
  ctx->bvh.wide_nodes = nullptr;
  ctx->bvh.wide_entries = nullptr;
  ctx->bvh.wide_root_ref = BVH_INVALID_REF;
  ctx->bvh.wide_stack_depth = 0;
  ctx->wide_node_count = 0;
  const size_t count = (size_t)ctx->bvh_nodes;
  if (count == 0 || ctx->bvh.root_ref == BVH_INVALID_REF || (ctx->bvh.root_ref & BVH_LEAF_BIT) || count * sizeof(WideNode) > 0xFFFFFFFFull) return STHIP_OK;
  if (!ctx->wide_scratch) ctx->wide_scratch = sthip::device_wide_scratch_create();
  HIP_TRY(ctx, ctx->wide_nodes.ensure(count));
  HIP_TRY(ctx, ctx->wide_entries.ensure(std::max<size_t>(1, ctx->top.entries.size())));
  sthip::DeviceWideResult res;
  std::string err;
  if (!sthip::collapse_wide_device(ctx->wide_scratch, reinterpret_cast<const BvhNodeSlot*>(ctx->nodes.p), (uint32_t)count, ctx->entries.p, (uint32_t)ctx->top.entries.size(), ctx->bvh.root_ref,
                                   ctx->bvh.top_is_world_blas != 0, ctx->bvh.stack_depth, ctx->wide_nodes.p, ctx->wide_entries.p, ctx->stream, res, err)) {
    if (getenv("STHIP_VERBOSE")) fprintf(stderr, "[sthip] no wide tree: %s\n", err.c_str());
    return STHIP_OK;
  }
  ctx->bvh.wide_nodes = reinterpret_cast<const uint4*>(ctx->wide_nodes.p);
  ctx->bvh.wide_entries = ctx->wide_entries.p;
  ctx->bvh.wide_root_ref = res.root_ref;
  ctx->bvh.wide_stack_depth = res.stack_depth;
  ctx->wide_node_count = res.node_count;
  ctx->stats.bvh_build_gpu_ms += res.gpu_ms;
  return STHIP_OK;
}

int sthip_scene_upload(sthip_ctx* ctx, const sthip_scene_desc* s) {
  if (!ctx) return STHIP_ERR_INVALID_ARGUMENT;
  if (!s || !s->gInstances || !s->gInstanceTransforms || !s->gInstanceInverseTransforms || !s->gMaterialData || s->instance_count == 0)
    return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "scene: a required array is NULL or there are no instances");
  if ((s->vertex_count && !s->gVertices) || (s->indices_bytes && !s->gIndices))  // a scene of sphere instances alone has neither
    return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "scene: vertex_count / indices_bytes > 0 but the array is NULL");
  if (s->instance_count > 0xFFFF) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "scene: more than 65535 instances (16-bit instance index, scene.h:23)");
  if (s->light_count && !s->gLightInstances) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "scene: light_count > 0 but gLightInstances is NULL");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // frames of the previous scene may still be in flight on the caller's stream (device output pointers: sthip_render only
  // enqueues), and the copies below go through the null stream, which a non-blocking stream does not wait for
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->has_scene = false;
  for (uint32_t i = 0; i < s->light_count; i++)
    if (s->gLightInstances[i] >= s->instance_count) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "scene: gLightInstances entry out of range");
  bool any_specular = false, any_image = false, any_alpha = false;
  if (s->image_count && !s->gImages) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "scene: image_count > 0 but gImages is NULL");
  // Per-channel extremes of an image (level 0; every mip level is an average of it, and bilinear / trilinear taps are
  // convex combinations, so a sampled value lies between them). Scanned on first use: the device multiplies the
  // constant by the texel (image_value.h:194-198), so whether a material can be specular depends on the texels.
  std::vector<int> scanned(s->image_count, 0);
  std::vector<float> tex_min((size_t)s->image_count * 4, 0.0f), tex_max((size_t)s->image_count * 4, 0.0f);
  auto image_range = [&](uint32_t index, int channel, float& lo, float& hi) {
    if (!scanned[index]) {
      const float* px = s->gImages[index].pixels;
      const size_t count = (size_t)s->gImages[index].width * s->gImages[index].height;
      for (int c = 0; c < 4; c++) {
        float a = __builtin_inff(), b = -__builtin_inff();
        bool nan = false;
        for (size_t k = 0; px && k < count; k++) {
          const float t = px[4 * k + c];
          if (t != t) nan = true;
          a = std::min(a, t);
          b = std::max(b, t);
        }
        if (nan || !px || !count) a = -__builtin_inff(), b = __builtin_inff();  // unknown: everything is possible
        tex_min[(size_t)index * 4 + c] = a;
        tex_max[(size_t)index * 4 + c] = b;
      }
      scanned[index] = 1;
    }
    lo = tex_min[(size_t)index * 4 + channel];
    hi = tex_max[(size_t)index * 4 + channel];
  };
  // bounds of one component of an image value: constant, or constant * texel (zero when no component of the constant is positive)
  auto value_range = [&](const sthip_MaterialRecord& rec, int k, int channel, float& lo, float& hi) {
    const float c = rec.values[k].value[channel];
    lo = hi = c;
    const uint32_t index = rec.values[k].image_index;
    if (index >= STHIP_IMAGE_COUNT || index >= s->image_count) return;
    const float* v = rec.values[k].value;
    if (!(v[0] > 0 || v[1] > 0 || v[2] > 0 || v[3] > 0)) {
      lo = hi = 0.0f;
      return;
    }
    float a, b;
    image_range(index, channel, a, b);
    lo = std::min(c * a, c * b);
    hi = std::max(c * a, c * b);
    if (lo != lo || hi != hi) lo = -__builtin_inff(), hi = __builtin_inff();
  };
  // materials: constant values or image values over gImages (image_value.h:183-207)
  std::vector<uint8_t> inst_flags(std::max<uint32_t>(1, s->instance_count), (uint8_t)INST_FLAG_KEEP);  // k_cull_terminal's table (kernels.h)
  for (uint32_t i = 0; i < s->instance_count; i++) {
    const uint32_t addr = s->gInstances[i].packed[0] >> 4;
    if ((s->gInstances[i].packed[0] & 0xF) == STHIP_INSTANCE_TYPE_VOLUME) {  // a Medium record (Material.hpp:80-87), 40 bytes
      if ((size_t)addr + 40 > s->material_bytes || (addr & 3)) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "scene: medium material_address out of range");
      uint32_t vol[2];
      memcpy(vol, (const uint8_t*)s->gMaterialData + addr + 32, 8);
      if (vol[0] >= s->volume_count || (vol[1] != 0xFFFFFFFFu && vol[1] >= s->volume_count)) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "scene: a medium refers to a volume that is not in gVolumes");
      float anisotropy;
      memcpy(&anisotropy, (const uint8_t*)s->gMaterialData + addr + 12, 4);
      if (!(fabsf(anisotropy) <= 0.999f)) any_specular = true;  // Medium::is_specular (medium.hlsli:22): its vertices are not diffuse vertices
      continue;
    }
    if ((size_t)addr + sizeof(sthip_MaterialRecord) > s->material_bytes) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "scene: material_address out of range");
    sthip_MaterialRecord rec;
    memcpy(&rec, (const uint8_t*)s->gMaterialData + addr, sizeof(rec));
    for (int k = 0; k < 3; k++)
      if (rec.values[k].image_index < STHIP_IMAGE_COUNT) {
        if (rec.values[k].image_index >= s->image_count) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "scene: a material refers to an image that is not in gImages");
        any_image = true;
      }
    if (rec.bump_index < STHIP_IMAGE_COUNT) {
      if (rec.bump_index >= s->image_count) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "scene: a bump map refers to an image that is not in gImages");
      any_image = true;
    }
    if (rec.alpha_mask_index < STHIP_IMAGE_COUNT && (s->gInstances[i].packed[0] & 0xF) == STHIP_INSTANCE_TYPE_TRIANGLES) {
      if (rec.alpha_mask_index >= s->image1_count) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "scene: a material refers to an alpha mask that is not in gImage1s");
      any_alpha = true;
    }
    // DisneyMaterial::is_specular (disney_material.hlsli:125) is evaluated per hit on value * texel, so the host test is
    // over what the product can reach: conservative (a "maybe" only costs bounce rounds that find empty queues)
    float lo, metallic_hi, roughness_lo, transmission_hi;
    value_range(rec, 1, 0, lo, metallic_hi);
    value_range(rec, 1, 1, roughness_lo, lo);
    value_range(rec, 2, 2, lo, transmission_hi);
    if ((metallic_hi > 0.999f || transmission_hi > 0.999f) && roughness_lo <= 1e-2f) any_specular = true;
    if ((s->gInstances[i].packed[0] & 0xF) == STHIP_INSTANCE_TYPE_TRIANGLES || (s->gInstances[i].packed[0] & 0xF) == STHIP_INSTANCE_TYPE_SPHERE) {
      // what DisneyMaterial::load reads of an untextured record (shading.h), with the device's arithmetic: Le = base_color *
      // emission, can_eval, is_specular (only the untextured k_shade instantiations, i.e. a scene without images, consult this)
      float f[14];
      memcpy(f, (const uint8_t*)s->gMaterialData + addr, sizeof(f));
      const float le[3] = {f[0] * f[3], f[1] * f[3], f[2] * f[3]};
      const bool emits = le[0] > 0 || le[1] > 0 || le[2] > 0;
      const bool can_eval = f[3] <= 0 && (f[0] > 0 || f[1] > 0 || f[2] > 0);
      const bool specular = (f[5] > 0.999f || f[12] > 0.999f) && f[6] <= 1e-2f;
      inst_flags[i] = (uint8_t)((emits ? INST_FLAG_EMITS : 0) | (can_eval ? INST_FLAG_CAN_EVAL : 0) | (specular ? INST_FLAG_SPECULAR : 0));
    }
  }
  ctx->inst_flags_host = inst_flags;
  HIP_TRY(ctx, ctx->inst_flags.ensure(inst_flags.size()));
  HIP_TRY(ctx, hipMemcpy(ctx->inst_flags.p, inst_flags.data(), inst_flags.size(), hipMemcpyHostToDevice));
  ctx->has_specular = any_specular;
  ctx->textured = any_image;
  if (s->image_count && !s->gImages) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "scene: image_count > 0 but gImages is NULL");
  if (s->image1_count && !s->gImage1s) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "scene: image1_count > 0 but gImage1s is NULL");
  if (s->volume_count && !s->gVolumes) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "scene: volume_count > 0 but gVolumes is NULL");
  sthip::BuiltBvh built;
  std::string err;
  const uint32_t n = s->instance_count;
  // The GPU builder works on the device-resident scene: vertices and indices go up BEFORE the build, and the bottom levels
  // are written in place (lbvh.hip: lbvh_build_device). From here on the previous scene is gone, also when the call fails.
  const bool device_build = ctx->bvh_builder == sthip::BVH_BUILDER_LBVH_GPU;
  sthip::DeviceBuildTarget target;
  if (device_build) {
    ctx->has_scene = false;
    HIP_TRY(ctx, ctx->vertices.ensure(std::max(1u, s->vertex_count)));
    HIP_TRY(ctx, ctx->indices.ensure((size_t)s->indices_bytes + 8));
    if (s->vertex_count) HIP_TRY(ctx, hipMemcpy(ctx->vertices.p, s->gVertices, (size_t)s->vertex_count * sizeof(sthip_PackedVertexData), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemset(ctx->indices.p + s->indices_bytes, 0, 8));
    if (s->indices_bytes) HIP_TRY(ctx, hipMemcpy(ctx->indices.p, s->gIndices, s->indices_bytes, hipMemcpyHostToDevice));
    target.vertices = ctx->vertices.p;
    target.vertex_count = s->vertex_count;
    target.indices = ctx->indices.p;
    target.stream = ctx->stream;
    target.algorithm = ctx->lbvh_algorithm;
    target.ploc_radius = ctx->ploc_radius;
    target.sah_top_size = ctx->lbvh_algorithm == 1 ? ctx->sah_top_size : 0;
    target.user = ctx;
    target.reserve = [](void* user, size_t node_capacity, size_t tri_capacity, sthip::DeviceBuildTarget& self) {
      sthip_ctx* c = (sthip_ctx*)user;
      if (c->nodes.ensure(node_capacity) != hipSuccess || c->raw_nodes.ensure(node_capacity) != hipSuccess || c->tris.ensure(tri_capacity) != hipSuccess) return false;
      self.nodes = c->nodes.p;
      self.raw_nodes = c->raw_nodes.p;
      self.tris = c->tris.p;
      return true;
    };
  }
  const auto t_build0 = std::chrono::steady_clock::now();
  if (!sthip::build_scene_bvh(*s, built, err, ctx->bvh_builder, device_build ? &target : nullptr, ctx->embed_leaves && STHIP_NODE_STRIDE == 48))
    return fail(ctx, err.find("only triangle") != std::string::npos ? STHIP_ERR_UNSUPPORTED : STHIP_ERR_INVALID_ARGUMENT, "scene: " + err);

  if (((size_t)built.dev_nodes + built.nodes.size() + 2 * built.entries.size() + 2) * sizeof(BvhNodeSlot) > 0xFFFFFFFFull || ((size_t)built.dev_tris + built.tris.size()) * sizeof(BvhTri) > 0xFFFFFFFFull)
    return fail(ctx, STHIP_ERR_UNSUPPORTED, "scene: the acceleration structure exceeds 4 GiB (the traversal addresses nodes and triangles with 32-bit byte offsets)");
  if (built.stack_depth > STHIP_MAX_STACK_DEPTH) return fail(ctx, STHIP_ERR_UNSUPPORTED, "scene: the acceleration structure is too deep for the traversal stack (use the SAH builder)");
  ctx->stats.bvh_build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_build0).count();
  ctx->stats.bvh_build_gpu_ms = built.gpu_build_ms;
  {
    // The bounds of the emissive triangle instances (kernels.h: EmitterBounds), from the validated scene arrays: the box of the
    // vertices an instance's triangles refer to, widened by 2^-15 of its coordinates' magnitude as the packed nodes of the
    // tree are, in world space for an instance with identity transforms (that is where its triangles are tested, whether the
    // builder merged it or not: the identity's fmaf chain returns the world-space ray) and in object space otherwise.
    std::vector<EmitterBounds> bounds;
    bool usable = true;
    for (uint32_t i = 0; i < s->instance_count && usable; i++) {
      // (sphere lights, environments: scenes of the extended k_shade instantiation, which does not answer last rays)
      if ((s->gInstances[i].packed[0] & 0xF) != STHIP_INSTANCE_TYPE_TRIANGLES || !(ctx->inst_flags_host[i] & INST_FLAG_EMITS)) continue;
      if (bounds.size() == STHIP_MAX_EMITTER_BOUNDS) {
        usable = false;
        break;
      }
      const uint32_t prims = (s->gInstances[i].packed[1] >> 12) & 0xFFFFu, stride = s->gInstances[i].packed[1] >> 28;
      const uint32_t first_vertex = s->gInstances[i].packed[2];
      const uint8_t* ib = (const uint8_t*)s->gIndices + s->gInstances[i].packed[3];
      EmitterBounds b{};
      for (int a = 0; a < 3; a++) b.lo[a] = __builtin_inff(), b.hi[a] = -__builtin_inff();
      for (uint32_t k = 0; k < 3 * prims; k++) {
        uint32_t index;
        if (stride == 2) {
          uint16_t w;
          memcpy(&w, ib + 2 * (size_t)k, 2);
          index = w;
        } else {
          memcpy(&index, ib + 4 * (size_t)k, 4);
        }
        if ((size_t)first_vertex + index >= s->vertex_count) {  // (the builders have refused such a scene already)
          usable = false;
          break;
        }
        const float* pos = s->gVertices[first_vertex + index].position;
        for (int a = 0; a < 3; a++) {
          b.lo[a] = std::min(b.lo[a], pos[a]);
          b.hi[a] = std::max(b.hi[a], pos[a]);
        }
      }
      if (!prims || !(b.lo[0] <= b.hi[0])) continue;  // (no triangle: nothing to hit)
      double diag = 0;
      for (int a = 0; a < 3; a++) {
        const float mag = std::max(fabsf(b.lo[a]), fabsf(b.hi[a])) * (1.0f / 32768.0f) + 1e-30f;
        b.lo[a] -= mag;
        b.hi[a] += mag;
        b.sphere[a] = 0.5f * b.lo[a] + 0.5f * b.hi[a];
        diag += ((double)b.hi[a] - b.lo[a]) * ((double)b.hi[a] - b.lo[a]);
      }
      b.sphere[3] = (float)sqrt(diag);  // (twice the box's own radius: the padding only has to be large enough)
      if (!std::isfinite(b.sphere[3])) usable = false;
      b.instance = i;
      static const float ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
      b.identity = (!memcmp(&s->gInstanceTransforms[i], ident, 48) && !memcmp(&s->gInstanceInverseTransforms[i], ident, 48)) ? 1u : 0u;
      bounds.push_back(b);
    }
    ctx->emitter_count = usable ? (uint32_t)bounds.size() : 0u;
    if (ctx->emitter_count) {
      HIP_TRY(ctx, ctx->emitters.ensure(bounds.size()));
      HIP_TRY(ctx, hipMemcpy(ctx->emitters.p, bounds.data(), bounds.size() * sizeof(EmitterBounds), hipMemcpyHostToDevice));
    }
  }
  if (!device_build) {
    HIP_TRY(ctx, ctx->vertices.ensure(std::max(1u, s->vertex_count)));
    HIP_TRY(ctx, ctx->indices.ensure((size_t)s->indices_bytes + 8));
  }
  HIP_TRY(ctx, ctx->instances.ensure(n));
  HIP_TRY(ctx, ctx->xf.ensure(n));
  HIP_TRY(ctx, ctx->inv_xf.ensure(n));
  HIP_TRY(ctx, ctx->motion_xf.ensure(n));
  HIP_TRY(ctx, ctx->materials.ensure(s->material_bytes));
  HIP_TRY(ctx, ctx->lights.ensure(std::max(1u, s->light_count)));
  if (!device_build) {
    if (s->vertex_count) HIP_TRY(ctx, hipMemcpy(ctx->vertices.p, s->gVertices, (size_t)s->vertex_count * sizeof(sthip_PackedVertexData), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemset(ctx->indices.p, 0, (size_t)s->indices_bytes + 8));
    if (s->indices_bytes) HIP_TRY(ctx, hipMemcpy(ctx->indices.p, s->gIndices, s->indices_bytes, hipMemcpyHostToDevice));
  }
  HIP_TRY(ctx, hipMemcpy(ctx->instances.p, s->gInstances, (size_t)n * 16, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(ctx->xf.p, s->gInstanceTransforms, (size_t)n * 48, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(ctx->inv_xf.p, s->gInstanceInverseTransforms, (size_t)n * 48, hipMemcpyHostToDevice));
  if (s->gInstanceMotionTransforms) {
    HIP_TRY(ctx, hipMemcpy(ctx->motion_xf.p, s->gInstanceMotionTransforms, (size_t)n * 48, hipMemcpyHostToDevice));
  } else {
    std::vector<sthip_TransformData> I(n);
    memset(I.data(), 0, (size_t)n * 48);
    for (auto& t : I) t.m[0][0] = t.m[1][1] = t.m[2][2] = 1;
    HIP_TRY(ctx, hipMemcpy(ctx->motion_xf.p, I.data(), (size_t)n * 48, hipMemcpyHostToDevice));
  }
  HIP_TRY(ctx, hipMemcpy(ctx->materials.p, s->gMaterialData, s->material_bytes, hipMemcpyHostToDevice));
  if (s->light_count) HIP_TRY(ctx, hipMemcpy(ctx->lights.p, s->gLightInstances, (size_t)s->light_count * 4, hipMemcpyHostToDevice));
  ctx->instance_count = n;
  ctx->light_count = s->light_count;
  ctx->materials_host.assign((const uint8_t*)s->gMaterialData, (const uint8_t*)s->gMaterialData + s->material_bytes);
  ctx->has_spheres = false;
  for (uint32_t i = 0; i < n; i++)
    if ((s->gInstances[i].packed[0] & 0xF) == STHIP_INSTANCE_TYPE_SPHERE) ctx->has_spheres = true;
  if (s->distribution_count && !s->gDistributions) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "scene: distribution_count > 0 but gDistributions is NULL");
  HIP_TRY(ctx, ctx->distributions.ensure(std::max<size_t>(1, s->distribution_count)));
  if (s->distribution_count) HIP_TRY(ctx, hipMemcpy(ctx->distributions.p, s->gDistributions, (size_t)s->distribution_count * 4, hipMemcpyHostToDevice));
  ctx->distribution_count = s->distribution_count;
  ctx->image_dims.clear();
  for (uint32_t i = 0; i < s->image_count; i++) ctx->image_dims.emplace_back(s->gImages[i].width, s->gImages[i].height);
  // images: mip chain by 2x2 box filter, level k+1 = max(1, floor(dim / 2)), ((a + b) + (c + d)) * 0.25
  {
    std::vector<DeviceImage> table(s->image_count);
    std::vector<float> texels;
    for (uint32_t i = 0; i < s->image_count; i++) {
      uint32_t w = s->gImages[i].width, h = s->gImages[i].height;
      if (!s->gImages[i].pixels || w == 0 || h == 0 || w > 0xFFFF || h > 0xFFFF) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "scene: bad image");
      DeviceImage& im = table[i];
      memset(&im, 0, sizeof(im));
      size_t level_start = texels.size();
      texels.insert(texels.end(), s->gImages[i].pixels, s->gImages[i].pixels + (size_t)w * h * 4);
      for (uint32_t level = 0;; level++) {
        im.offset[level] = (uint32_t)(level_start / 4);
        im.w[level] = (uint16_t)w;
        im.h[level] = (uint16_t)h;
        im.levels = level + 1;
        if ((w == 1 && h == 1) || level + 1 == STHIP_MAX_MIPS) break;
        const uint32_t nw = std::max(1u, w / 2), nh = std::max(1u, h / 2);
        const size_t next_start = texels.size();
        texels.resize(next_start + (size_t)nw * nh * 4);
        const float* prev = texels.data() + level_start;
        float* next = texels.data() + next_start;
        for (uint32_t y = 0; y < nh; y++)
          for (uint32_t x = 0; x < nw; x++) {
            const uint32_t x0 = std::min(2 * x, w - 1), x1 = std::min(2 * x + 1, w - 1), y0 = std::min(2 * y, h - 1), y1 = std::min(2 * y + 1, h - 1);
            for (int k = 0; k < 4; k++) {
              const float a = prev[4 * ((size_t)y0 * w + x0) + k], b = prev[4 * ((size_t)y0 * w + x1) + k];
              const float c = prev[4 * ((size_t)y1 * w + x0) + k], e = prev[4 * ((size_t)y1 * w + x1) + k];
              next[4 * ((size_t)y * nw + x) + k] = ((a + b) + (c + e)) * 0.25f;
            }
          }
        level_start = next_start;
        w = nw;
        h = nh;
      }
    }
    HIP_TRY(ctx, ctx->images.ensure(std::max<size_t>(1, table.size())));
    HIP_TRY(ctx, ctx->image_texels.ensure(std::max<size_t>(1, texels.size() / 4)));
    if (!table.empty()) HIP_TRY(ctx, hipMemcpy(ctx->images.p, table.data(), table.size() * sizeof(DeviceImage), hipMemcpyHostToDevice));
    if (!texels.empty()) HIP_TRY(ctx, hipMemcpy(ctx->image_texels.p, texels.data(), texels.size() * sizeof(float), hipMemcpyHostToDevice));
    ctx->image_count = s->image_count;
  }

  // the 8-wide form permutes the leaf triangles (bvh_build.h): made before anything of the tree is uploaded
  if (ctx->use_wide == 3 && !ctx->use_treetop && !built.embedded && built.dev_nodes == 0) sthip::build_wide8_bvh(built);
  // headroom: a transforms-only update may build a top level with more inner nodes than this one (at most 2 per entry)
  const size_t nodes_total = (size_t)built.dev_nodes + built.nodes.size(), tris_total = (size_t)built.dev_tris + built.tris.size();
  const size_t nodes_needed = std::max<size_t>(1, built.top.blas_nodes + 2 * built.entries.size() + 2);
  if (device_build && built.dev_nodes) {  // the device region is already in place: growing the arrays now would lose it
    if (ctx->nodes.n < std::max(nodes_needed, nodes_total) || ctx->tris.n < std::max<size_t>(1, tris_total)) return fail(ctx, STHIP_ERR_HIP, "scene: the reserved device arrays are too small");
  } else {
    HIP_TRY(ctx, ctx->nodes.ensure(std::max(nodes_needed, nodes_total)));
    HIP_TRY(ctx, ctx->tris.ensure(built.embedded ? 1 : std::max<size_t>(1, tris_total)));
  }
  HIP_TRY(ctx, ctx->entries.ensure(std::max<size_t>(1, built.entries.size())));
  if (built.embedded) {  // one array: packed nodes, and the leaf triangles in the units behind their parents
    std::vector<BvhNodePacked> packed;
    sthip::pack_nodes(built.nodes.data(), built.nodes.size(), packed);
    static_assert(sizeof(BvhNodePacked) == sizeof(BvhTri), "a leaf triangle takes one node unit");
    for (size_t u = 0; u < built.unit_tri.size(); u++)
      if (built.unit_tri[u] != 0xFFFFFFFFu) memcpy(&packed[u], &built.tris[built.unit_tri[u]], sizeof(BvhTri));
    if (!packed.empty()) HIP_TRY(ctx, hipMemcpy(ctx->nodes.p, packed.data(), packed.size() * sizeof(BvhNodePacked), hipMemcpyHostToDevice));
  } else {
    if (!built.nodes.empty()) HIP_TRY(ctx, upload_nodes(ctx, built.dev_nodes, built.nodes.data(), built.nodes.size()));
    if (!built.tris.empty()) HIP_TRY(ctx, hipMemcpy(ctx->tris.p + built.dev_tris, built.tris.data(), built.tris.size() * sizeof(BvhTri), hipMemcpyHostToDevice));
  }
  if (!built.entries.empty()) HIP_TRY(ctx, hipMemcpy(ctx->entries.p, built.entries.data(), built.entries.size() * sizeof(TlasEntry), hipMemcpyHostToDevice));
  ctx->bvh.wide_nodes = nullptr;
  ctx->bvh.wide_entries = nullptr;
  ctx->bvh.wide_root_ref = BVH_INVALID_REF;
  ctx->bvh.wide_stack_depth = 0;
  ctx->wide_node_count = 0;
  ctx->bvh.wide8_nodes = nullptr;
  ctx->bvh.wide8_entries = nullptr;
  ctx->bvh.wide8_root = BVH_INVALID_REF;
  ctx->bvh.wide8_stack_depth = 0;
  ctx->bvh.wide8_tri_min = ctx->tri_min_lanes;
  ctx->wide8_node_count = 0;
  ctx->wide8_host.clear();
  if (!built.wide8_nodes.empty()) {
    // (room for a rebuilt top level: at most one node per entry and per two entries above them, and a copy of the merged mesh's root)
    HIP_TRY(ctx, ctx->wide8_nodes.ensure(built.top.wide8_blas_nodes + 2 * built.entries.size() + 4));
    HIP_TRY(ctx, ctx->wide8_entries.ensure(std::max<size_t>(1, built.entries.size())));
    HIP_TRY(ctx, hipMemcpy(ctx->wide8_nodes.p, built.wide8_nodes.data(), built.wide8_nodes.size() * sizeof(Wide8Node), hipMemcpyHostToDevice));
    if (!built.wide8_entries.empty()) HIP_TRY(ctx, hipMemcpy(ctx->wide8_entries.p, built.wide8_entries.data(), built.wide8_entries.size() * sizeof(TlasEntry), hipMemcpyHostToDevice));
    ctx->bvh.wide8_nodes = reinterpret_cast<const uint4*>(ctx->wide8_nodes.p);
    ctx->bvh.wide8_entries = ctx->wide8_entries.p;
    ctx->bvh.wide8_root = built.wide8_root;
    ctx->bvh.wide8_stack_depth = built.wide8_stack_depth;
    ctx->wide8_node_count = built.wide8_nodes.size();
    ctx->wide8_host = std::move(built.wide8_nodes);
  }
  ctx->want_wide = (ctx->use_wide == 1 || (ctx->use_wide == 3 && !ctx->bvh.wide8_nodes) || (ctx->use_wide == 2 && nodes_total * sizeof(BvhNodePacked) > ((size_t)4 << 20))) && !ctx->use_treetop && !built.embedded;
  if (ctx->want_wide && built.dev_nodes == 0) {  // a host-built tree: collapsed on the host from its exact boxes (a device build: below, from the nodes in HBM)
    sthip::build_wide_bvh(built);
    if (!built.wide_nodes.empty() && built.wide_nodes.size() * sizeof(WideNode) <= 0xFFFFFFFFull) {
      HIP_TRY(ctx, ctx->wide_nodes.ensure(built.wide_nodes.size()));
      HIP_TRY(ctx, ctx->wide_entries.ensure(std::max<size_t>(1, built.wide_entries.size())));
      HIP_TRY(ctx, hipMemcpy(ctx->wide_nodes.p, built.wide_nodes.data(), built.wide_nodes.size() * sizeof(WideNode), hipMemcpyHostToDevice));
      if (!built.wide_entries.empty()) HIP_TRY(ctx, hipMemcpy(ctx->wide_entries.p, built.wide_entries.data(), built.wide_entries.size() * sizeof(TlasEntry), hipMemcpyHostToDevice));
      ctx->bvh.wide_nodes = reinterpret_cast<const uint4*>(ctx->wide_nodes.p);
      ctx->bvh.wide_entries = ctx->wide_entries.p;
      ctx->bvh.wide_root_ref = built.wide_root_ref;
      ctx->bvh.wide_stack_depth = built.wide_stack_depth;
      ctx->wide_node_count = built.wide_nodes.size();
    }
  }
  {  // alpha masks: one-channel images and the per-triangle uvs the traversal interpolates
    std::vector<DeviceImage1> table(s->image1_count);
    std::vector<float> texels;
    for (uint32_t i = 0; i < s->image1_count; i++) {
      const sthip_image_desc& im = s->gImage1s[i];
      if (!im.pixels || im.width == 0 || im.height == 0 || im.width > 0xFFFF || im.height > 0xFFFF) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "scene: bad alpha-mask image");
      table[i].offset = (uint32_t)texels.size();
      table[i].w = im.width;
      table[i].h = im.height;
      table[i].pad = 0;
      texels.insert(texels.end(), im.pixels, im.pixels + (size_t)im.width * im.height);
    }
    HIP_TRY(ctx, ctx->images1.ensure(std::max<size_t>(1, table.size())));
    HIP_TRY(ctx, ctx->image1_texels.ensure(std::max<size_t>(1, texels.size())));
    HIP_TRY(ctx, ctx->tri_uvs.ensure(1));  // (sized and filled with the shading records further down, once the triangles are resident)
    if (!table.empty()) HIP_TRY(ctx, hipMemcpy(ctx->images1.p, table.data(), table.size() * sizeof(DeviceImage1), hipMemcpyHostToDevice));
    if (!texels.empty()) HIP_TRY(ctx, hipMemcpy(ctx->image1_texels.p, texels.data(), texels.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(ctx, ctx->inst_alpha.ensure(std::max<size_t>(1, built.inst_alpha.size())));
    if (!built.inst_alpha.empty()) HIP_TRY(ctx, hipMemcpy(ctx->inst_alpha.p, built.inst_alpha.data(), built.inst_alpha.size() * 4, hipMemcpyHostToDevice));
    ctx->bvh.inst_alpha = ctx->inst_alpha.p;
    ctx->has_alpha = any_alpha && built.any_alpha;
    ctx->bvh.tri_uv = reinterpret_cast<const float2*>(ctx->tri_uvs.p);
    ctx->bvh.images1 = ctx->images1.p;
    ctx->bvh.image1_texels = ctx->image1_texels.p;
    ctx->bvh.alpha_test = 0;
    ctx->bvh.flip_uvs = 0;
  }
  {  // gVolumes: the grids back to back as 32-bit words, and their parsed headers
    size_t words = 0;
    for (uint32_t i = 0; i < s->volume_count; i++) {
      built.volumes[i].first_word = (uint32_t)words;
      words += (size_t)(s->gVolumes[i].bytes / 4);
    }
    if (words > 0xFFFFFFFFull) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "scene: gVolumes exceed 16 GiB");
    HIP_TRY(ctx, ctx->volume_words.ensure(std::max<size_t>(1, words)));
    HIP_TRY(ctx, ctx->volumes.ensure(std::max<size_t>(1, built.volumes.size())));
    for (uint32_t i = 0; i < s->volume_count; i++)
      HIP_TRY(ctx, hipMemcpy(ctx->volume_words.p + built.volumes[i].first_word, s->gVolumes[i].data, (size_t)s->gVolumes[i].bytes, hipMemcpyHostToDevice));
    if (!built.volumes.empty()) HIP_TRY(ctx, hipMemcpy(ctx->volumes.p, built.volumes.data(), built.volumes.size() * sizeof(DeviceVolume), hipMemcpyHostToDevice));
    ctx->volume_count = s->volume_count;
    ctx->has_volumes = false;
    ctx->volume_instances = 0;
    ctx->instance_is_volume.assign(n, 0);
    for (uint32_t i = 0; i < n; i++)
      if ((s->gInstances[i].packed[0] & 0xF) == STHIP_INSTANCE_TYPE_VOLUME) {
        ctx->has_volumes = true;
        ctx->volume_instances++;
        ctx->instance_is_volume[i] = 1;
      }
    ctx->bvh.volumes = ctx->volumes.p;
  }
  ctx->bvh.nodes = reinterpret_cast<const float4*>(ctx->nodes.p);
  ctx->bvh.tris = built.embedded ? reinterpret_cast<const float4*>(ctx->nodes.p) : reinterpret_cast<const float4*>(ctx->tris.p);
  ctx->bvh.entries = ctx->entries.p;
  ctx->bvh.root_ref = built.root_ref;
  ctx->bvh.top_is_world_blas = built.top_is_world_blas;
  ctx->bvh.stack_depth = built.stack_depth;
  ctx->bvh.scene_cx = built.scene_center[0];
  ctx->bvh.scene_cy = built.scene_center[1];
  ctx->bvh.scene_cz = built.scene_center[2];
  ctx->bvh.scene_radius = built.scene_radius;
  ctx->bvh_nodes = nodes_total;
  ctx->bvh_tris = tris_total;
  ctx->top = std::move(built.top);
  if (ctx->want_wide && built.dev_nodes != 0) {  // the GPU builder's tree: its wide form is made where the nodes are
    const int rc = collapse_resident_tree(ctx);
    if (rc != STHIP_OK) return rc;
  }
  // the host's copy of the unpacked nodes, with room for a rebuilt top level
  // (only the treetop selection reads it: without the treetop a device-resident build copies no node to the host at all)
  ctx->nodes_host.n = 0;
  if (ctx->use_treetop) {
    HIP_TRY(ctx, ctx->nodes_host.ensure(std::max<size_t>(nodes_total, ctx->nodes.n)));
    if (built.dev_nodes) HIP_TRY(ctx, hipMemcpy(ctx->nodes_host.p, ctx->raw_nodes.p, (size_t)built.dev_nodes * sizeof(BvhNode), hipMemcpyDeviceToHost));
    if (!built.nodes.empty()) memcpy(ctx->nodes_host.p + built.dev_nodes, built.nodes.data(), built.nodes.size() * sizeof(BvhNode));
    ctx->nodes_host.n = nodes_total;
  }
  {  // the shading records beside the leaf triangles (bvh.h: BvhTriShade), from the triangles as they lie in HBM now
    const size_t units = built.embedded ? nodes_total : tris_total;  // (embedded leaves: a triangle is a unit of the node array; the units that are nodes get a record nobody reads)
    HIP_TRY(ctx, ctx->tri_shade.ensure(std::max<size_t>(1, units)));
    if (ctx->has_alpha) {  // the uvs the traversal's alpha test interpolates (gAlphaTest, intersection.hlsli:117-131), in the same order
      HIP_TRY(ctx, ctx->tri_uvs.ensure(std::max<size_t>(1, units)));
      ctx->bvh.tri_uv = reinterpret_cast<const float2*>(ctx->tri_uvs.p);
    }
    if (units && s->vertex_count) {
      DevBuf<uint8_t> is_tri;  // embedded leaves: which units are triangles
      if (built.embedded) {
        std::vector<uint8_t> flags(units, 0);
        for (size_t u = 0; u < built.unit_tri.size() && u < units; u++) flags[u] = built.unit_tri[u] != 0xFFFFFFFFu;
        HIP_TRY(ctx, is_tri.ensure(units));
        HIP_TRY(ctx, hipMemcpy(is_tri.p, flags.data(), units, hipMemcpyHostToDevice));
      }
      hipLaunchKernelGGL(k_fill_tri_shade, dim3(grid_for_early(ctx, units)), dim3(STHIP_BLOCK), 0, ctx->stream, reinterpret_cast<const BvhTri*>(ctx->bvh.tris), (uint32_t)units, is_tri.p, ctx->vertices.p, s->vertex_count,
                         ctx->indices.p, (uint64_t)s->indices_bytes, ctx->tri_shade.p, ctx->has_alpha ? ctx->tri_uvs.p : nullptr);
      HIP_TRY(ctx, hipGetLastError());
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // (`is_tri` goes out of scope)
    }
  }
  {
    const int rc = configure_stack(ctx);
    if (rc != STHIP_OK) return rc;
  }
  ctx->has_scene = true;
  ctx->reuse_grids_valid = false;  // (stored samples name materials and lights of the scene they were taken in)
  if (ctx->keep_scene) {
    if (s->gVertices != ctx->kept.vertices.data() || !ctx->kept.valid) ctx->kept.keep(*s);  // (a rebuild from the kept copy itself keeps nothing anew)
  } else {
    ctx->kept = KeptScene();
  }
  if (getenv("STHIP_VERBOSE")) {
    int per_cu = 0;
    per_cu = trace_occupancy(ctx, trace_lds_bytes(ctx));
    fprintf(stderr, "[sthip] bvh (%s): %zu nodes, %zu tris, %zu top-level entries, stack depth %u (%zu B LDS / block%s, treetop %u nodes), %d trace blocks / CU, build %.1f ms (GPU kernels %.2f ms)\n",
            ctx->bvh_builder ? "lbvh/gpu" : "sah/host", (size_t)ctx->bvh_nodes, (size_t)ctx->bvh_tris, built.entries.size(), built.stack_depth, stack_bytes(ctx), ctx->bvh.spill ? ", bounded" : "", ctx->bvh.top_count, per_cu, ctx->stats.bvh_build_ms,
            ctx->stats.bvh_build_gpu_ms);
  }
  return STHIP_OK;
}

static unsigned long long* queue_ctl_host(unsigned long long* qctl, uint32_t kind, uint32_t depth) {
  return qctl + (size_t)((kind * 64u + depth) * QUEUE_SEGMENTS) * QCTL_STRIDE;
}
static uint32_t grid_for(const sthip_ctx* ctx, size_t n) {
  const size_t blocks = (n + STHIP_BLOCK - 1) / STHIP_BLOCK;
  const size_t cap = (size_t)ctx->cu_count * 32;  // grid-stride beyond this
  return (uint32_t)std::max<size_t>(1, std::min(blocks, cap));
}
// Persistent trace kernels: as many blocks as are resident at once (LDS stack and VGPRs bound it).
static int trace_occupancy(const sthip_ctx* ctx, size_t lds_bytes) {  // resident k_trace blocks per CU with that much dynamic LDS
  int per_cu = 0;
  hipError_t e;
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu,
                                                   ctx->bvh.wide8_nodes  ? trace_kernel_wide8(false, false, ctx->bvh.spill != nullptr)
                                                   : ctx->bvh.wide_nodes ? trace_kernel_wide(false, false, ctx->bvh.spill != nullptr)
                                                                         : trace_kernel(false, false, ctx->bvh.spill != nullptr, ctx->use_treetop),
                                                   STHIP_BLOCK, lds_bytes);
  return e == hipSuccess ? per_cu : 0;
}
static uint32_t trace_grid(sthip_ctx* ctx, size_t lds_bytes) {
  int per_cu = trace_occupancy(ctx, lds_bytes);
  if (per_cu < 1) per_cu = 2;
  if (ctx->trace_blocks_per_cu) per_cu = (int)ctx->trace_blocks_per_cu;
  return (uint32_t)(ctx->cu_count * per_cu);
}
static size_t stack_bytes(const sthip_ctx* ctx) { return (size_t)ctx->bvh.lds_levels * STHIP_BLOCK * (ctx->bvh.wide8_nodes ? sizeof(uint2) : sizeof(uint32_t)); }  // (the 8-wide walk's entries are 64-bit groups)

// LDS of one k_trace block: the per-lane stacks and, behind them, the treetop
static size_t trace_lds_bytes(const sthip_ctx* ctx) { return stack_bytes(ctx) + (size_t)ctx->bvh.top_count * sizeof(BvhNodePacked); }

// Decides how the traversal stack of the current tree is held: all of it in LDS, or lds_stack_cap levels there and the
// full height in global memory for the rays that overflow. Then the treetop takes the LDS that is left.
static int configure_stack(sthip_ctx* ctx) {
  // (the wide walk pushes up to three children per level, and every step writes the three levels from `top` on)
  const uint32_t need = ctx->bvh.wide8_nodes ? ctx->bvh.wide8_stack_depth : ctx->bvh.wide_nodes ? ctx->bvh.wide_stack_depth + 3u : ctx->bvh.stack_depth;
  // (a 64-bit entry counts as two levels of the limits, which are about LDS bytes)
  const bool bounded = ctx->bvh.wide8_nodes ? 2u * need > ctx->lds_stack_threshold : need > ctx->lds_stack_threshold;
  ctx->bvh.lds_levels = bounded ? (ctx->bvh.wide8_nodes ? std::max(4u, ctx->lds_stack_cap / 2u) : ctx->lds_stack_cap) : need;
  ctx->bvh.spill = nullptr;
  if (bounded) {
    // one column per lane of the largest grid a trace launch can have (persistent: resident blocks; ray batches use it too)
    HIP_TRY(ctx, ctx->spill.ensure((size_t)ctx->cu_count * 8 * STHIP_BLOCK * ctx->bvh.stack_depth));
    ctx->bvh.spill = ctx->spill.p;
  }
  return refresh_treetop(ctx);
}

// (Re)builds the treetop for the current top level: as many nodes as fit into the LDS the stacks leave free at the
// occupancy the kernel's registers allow anyway (the treetop must not cost a resident block).
static int refresh_treetop(sthip_ctx* ctx) {
  ctx->bvh.top_nodes = nullptr;
  ctx->bvh.top_entries = ctx->bvh.entries;
  ctx->bvh.top_root_ref = ctx->bvh.root_ref;
  ctx->bvh.top_count = 0;
  if (!ctx->use_treetop || ctx->nodes_host.empty()) return STHIP_OK;
  int per_cu = 0;
  const size_t stack = stack_bytes(ctx);
  per_cu = trace_occupancy(ctx, stack);
  if (per_cu < 1) return STHIP_OK;
  if (ctx->trace_blocks_per_cu) per_cu = (int)ctx->trace_blocks_per_cu;
  const size_t lds_per_cu = 160 * 1024, per_block = lds_per_cu / (size_t)per_cu;
  if (per_block < stack + 2048) return STHIP_OK;
  uint32_t capacity = (uint32_t)std::min<size_t>((per_block - stack - 1024) / sizeof(BvhNodePacked), 2048);
  while (capacity >= 16) {  // the allocation granularity is the runtime's: ask it
    int got = 0;
    got = trace_occupancy(ctx, stack + (size_t)capacity * sizeof(BvhNodePacked));
    if (got >= per_cu) break;
    capacity -= 16;
  }
  if (capacity < 16) return STHIP_OK;
  sthip::Treetop tt;
  sthip::build_treetop(ctx->nodes_host.data(), (size_t)ctx->bvh_nodes, ctx->top.entries, ctx->bvh.root_ref, capacity, tt);
  if (tt.nodes.empty()) return STHIP_OK;
  HIP_TRY(ctx, ctx->top_nodes.ensure(tt.nodes.size()));
  HIP_TRY(ctx, ctx->top_entries.ensure(std::max<size_t>(1, tt.entries.size())));
  {
    std::vector<BvhNodePacked> packed;
    sthip::pack_nodes(tt.nodes.data(), tt.nodes.size(), packed);
    HIP_TRY(ctx, hipMemcpy(ctx->top_nodes.p, packed.data(), packed.size() * sizeof(BvhNodePacked), hipMemcpyHostToDevice));
  }
  if (!tt.entries.empty()) HIP_TRY(ctx, hipMemcpy(ctx->top_entries.p, tt.entries.data(), tt.entries.size() * sizeof(TlasEntry), hipMemcpyHostToDevice));
  ctx->bvh.top_nodes = reinterpret_cast<const float4*>(ctx->top_nodes.p);
  ctx->bvh.top_entries = ctx->top_entries.p;
  ctx->bvh.top_root_ref = tt.root_ref;
  ctx->bvh.top_count = (uint32_t)tt.nodes.size();
  return STHIP_OK;
}

int sthip_trace_rays(sthip_ctx* ctx, const sthip_ray* rays, uint32_t ray_count, sthip_hit* hits, uint32_t any_hit, uint32_t device_ptrs) {
  if (!ctx || !rays || !hits) return STHIP_ERR_INVALID_ARGUMENT;
  if (!ctx->has_scene) return fail(ctx, STHIP_ERR_NO_SCENE, "no scene uploaded");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ray_count == 0) return STHIP_OK;
  const sthip_ray* d_rays = rays;
  sthip_hit* d_hits = hits;
  DevBuf<sthip_ray>& rb = ctx->ray_staging;  // kept in the context: no hipMalloc / hipFree per call
  DevBuf<sthip_hit>& hb = ctx->hit_staging;
  if (!device_ptrs) {
    HIP_TRY(ctx, rb.ensure(ray_count));
    HIP_TRY(ctx, hb.ensure(ray_count));
    HIP_TRY(ctx, hipMemcpyAsync(rb.p, rays, (size_t)ray_count * sizeof(sthip_ray), hipMemcpyHostToDevice, ctx->stream));
    d_rays = rb.p;
    d_hits = hb.p;
  }
  HIP_TRY(ctx, ctx->counters.ensure(CNT_TOTAL));
  HIP_TRY(ctx, hipMemsetAsync(ctx->counters.p, 0, CNT_TOTAL * sizeof(unsigned long long), ctx->stream));
  // bounded stacks: the batch kernel walks with full-height stacks in global memory, one column per lane of a grid the spill buffer covers
  const uint32_t grid = ctx->bvh.spill ? std::min<uint32_t>(grid_for(ctx, ray_count), (uint32_t)ctx->cu_count * 8u) : grid_for(ctx, ray_count);
  const size_t lds = ctx->bvh.spill ? 0 : stack_bytes(ctx);
  DeviceBvh bvh = ctx->bvh;
  bvh.alpha_test = (ctx->has_alpha && (any_hit & 2u)) ? 1u : 0u;
  bvh.flip_uvs = (any_hit & 4u) ? 1u : 0u;
  any_hit &= 1u;
  if (any_hit) {
    if (ctx->count_traversal)
      hipLaunchKernelGGL((k_trace_batch<true, true>), dim3(grid), dim3(STHIP_BLOCK), lds, ctx->stream, bvh, d_rays, ray_count, d_hits, ctx->counters.p);
    else
      hipLaunchKernelGGL((k_trace_batch<true, false>), dim3(grid), dim3(STHIP_BLOCK), lds, ctx->stream, bvh, d_rays, ray_count, d_hits, ctx->counters.p);
  } else {
    if (ctx->count_traversal)
      hipLaunchKernelGGL((k_trace_batch<false, true>), dim3(grid), dim3(STHIP_BLOCK), lds, ctx->stream, bvh, d_rays, ray_count, d_hits, ctx->counters.p);
    else
      hipLaunchKernelGGL((k_trace_batch<false, false>), dim3(grid), dim3(STHIP_BLOCK), lds, ctx->stream, bvh, d_rays, ray_count, d_hits, ctx->counters.p);
  }
  HIP_TRY(ctx, hipGetLastError());
  if (!device_ptrs) {
    HIP_TRY(ctx, hipMemcpyAsync(hits, hb.p, (size_t)ray_count * sizeof(sthip_hit), hipMemcpyDeviceToHost, ctx->stream));
    std::vector<unsigned long long> c(CNT_TOTAL);
    HIP_TRY(ctx, hipMemcpyAsync(c.data(), ctx->counters.p, CNT_TOTAL * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stats.nodes_visited = c[CNT_NODES];
    ctx->stats.tris_tested = c[CNT_TRIS];
    ctx->stats.nodes_visited_shadow = c[CNT_NODES + 1];
    ctx->stats.tris_tested_shadow = c[CNT_TRIS + 1];
  }
  return STHIP_OK;
}

int sthip_scene_update_transforms(sthip_ctx* ctx, const sthip_TransformData* xf, const sthip_TransformData* inv, const sthip_TransformData* motion, uint32_t instance_count) {
  if (!ctx) return STHIP_ERR_INVALID_ARGUMENT;
  if (!ctx->has_scene) return fail(ctx, STHIP_ERR_NO_SCENE, "sthip_scene_update_transforms before sthip_scene_upload");
  if (!xf || !inv) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "sthip_scene_update_transforms: transforms and inverse transforms are required");
  if (instance_count != ctx->instance_count) return fail(ctx, STHIP_ERR_UNSUPPORTED, "sthip_scene_update_transforms: the instance count changed: upload the scene again");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  sthip::TopLevelState next = ctx->top;  // nothing changes unless everything succeeds
  std::vector<BvhNode> tlas;
  uint32_t root_ref = 0, top_is_world = 1, stack_depth = 4;
  float center[3] = {ctx->bvh.scene_cx, ctx->bvh.scene_cy, ctx->bvh.scene_cz}, radius = ctx->bvh.scene_radius;
  std::string err;
  if (!sthip::rebuild_top_level(next, xf, inv, instance_count, tlas, root_ref, top_is_world, stack_depth, center, radius, err)) {
    // An instance of the merged world-space mesh (identity transform at upload) moved: the tree that is resident cannot follow,
    // the scene is built again from the copy kept at upload, with the new transforms — the configured builder ("bvh_builder" = 1:
    // ~10 ms per million triangles on the device), everything else as uploaded. Without the copy ("keep_scene" = 0) it is refused.
    if (ctx->kept.valid && ctx->kept.instances.size() == instance_count) {
      ctx->kept.xf.assign(xf, xf + instance_count);
      ctx->kept.inv.assign(inv, inv + instance_count);
      if (motion)
        ctx->kept.motion.assign(motion, motion + instance_count);
      else
        ctx->kept.motion.clear();
      const sthip_scene_desc d = ctx->kept.desc();
      ctx->stats.full_rebuilds++;
      return sthip_scene_upload(ctx, &d);
    }
    return fail(ctx, STHIP_ERR_UNSUPPORTED, "sthip_scene_update_transforms: " + err);
  }
  if (stack_depth > STHIP_MAX_STACK_DEPTH) return fail(ctx, STHIP_ERR_UNSUPPORTED, "sthip_scene_update_transforms: the new top level is too deep for the traversal stack");
  if ((size_t)next.blas_nodes + tlas.size() > ctx->nodes.n) return fail(ctx, STHIP_ERR_UNSUPPORTED, "sthip_scene_update_transforms: the new top level does not fit: upload the scene again");
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // frames in flight still read the old top level
  const uint32_t n = instance_count;
  HIP_TRY(ctx, hipMemcpy(ctx->xf.p, xf, (size_t)n * 48, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(ctx->inv_xf.p, inv, (size_t)n * 48, hipMemcpyHostToDevice));
  if (motion) {
    HIP_TRY(ctx, hipMemcpy(ctx->motion_xf.p, motion, (size_t)n * 48, hipMemcpyHostToDevice));
  } else {
    std::vector<sthip_TransformData> I(n);
    memset(I.data(), 0, (size_t)n * 48);
    for (auto& t : I) t.m[0][0] = t.m[1][1] = t.m[2][2] = 1;
    HIP_TRY(ctx, hipMemcpy(ctx->motion_xf.p, I.data(), (size_t)n * 48, hipMemcpyHostToDevice));
  }
  if (!next.entries.empty()) HIP_TRY(ctx, hipMemcpy(ctx->entries.p, next.entries.data(), next.entries.size() * sizeof(TlasEntry), hipMemcpyHostToDevice));
  if (!tlas.empty()) HIP_TRY(ctx, upload_nodes(ctx, next.blas_nodes, tlas.data(), tlas.size()));
  ctx->bvh.root_ref = root_ref;
  ctx->bvh.top_is_world_blas = top_is_world;
  ctx->bvh.stack_depth = stack_depth;
  ctx->bvh.scene_cx = center[0];
  ctx->bvh.scene_cy = center[1];
  ctx->bvh.scene_cz = center[2];
  ctx->bvh.scene_radius = radius;
  ctx->bvh_nodes = next.blas_nodes + tlas.size();
  if (ctx->nodes_host.n && ctx->nodes_host.cap >= (size_t)next.blas_nodes + tlas.size()) {
    if (!tlas.empty()) memcpy(ctx->nodes_host.p + next.blas_nodes, tlas.data(), tlas.size() * sizeof(BvhNode));
    ctx->nodes_host.n = std::max(ctx->nodes_host.n, (size_t)next.blas_nodes + tlas.size());
  }
  ctx->top = std::move(next);
  // the wide form was made from the old top level: made again from the nodes in HBM (the bottom levels come out as before,
  // the top level new), so a moved scene keeps the walk it was uploaded with
  ctx->bvh.wide_nodes = nullptr;
  ctx->bvh.wide_entries = nullptr;
  if (ctx->bvh.wide8_nodes) {  // the 8-wide form: its top level again, on the host (a node per few entries), behind the bottom levels' nodes
    std::vector<TlasEntry> entries8;
    uint32_t root8 = BVH_INVALID_REF, depth8 = 0;
    ctx->bvh.wide8_nodes = nullptr;
    if (sthip::build_wide8_top(ctx->top, tlas.data(), ctx->top.blas_nodes, root_ref, top_is_world != 0, ctx->wide8_host, entries8, root8, depth8) && ctx->wide8_host.size() <= ctx->wide8_nodes.n &&
        entries8.size() <= ctx->wide8_entries.n) {
      const size_t first = ctx->top.wide8_blas_nodes;
      if (ctx->wide8_host.size() > first)
        HIP_TRY(ctx, hipMemcpy(ctx->wide8_nodes.p + first, ctx->wide8_host.data() + first, (ctx->wide8_host.size() - first) * sizeof(Wide8Node), hipMemcpyHostToDevice));
      if (!entries8.empty()) HIP_TRY(ctx, hipMemcpy(ctx->wide8_entries.p, entries8.data(), entries8.size() * sizeof(TlasEntry), hipMemcpyHostToDevice));
      ctx->bvh.wide8_nodes = reinterpret_cast<const uint4*>(ctx->wide8_nodes.p);
      ctx->bvh.wide8_entries = ctx->wide8_entries.p;
      ctx->bvh.wide8_root = root8;
      ctx->bvh.wide8_stack_depth = depth8;
      ctx->wide8_node_count = ctx->wide8_host.size();
    } else {  // (a top level that cannot take the form: the 4-wide one from here on)
      ctx->want_wide = !ctx->use_treetop;
    }
  }
  if (ctx->want_wide && !ctx->bvh.wide8_nodes) {
    const int rc = collapse_resident_tree(ctx);
    if (rc != STHIP_OK) {  // (an allocation of the collapse failed: the new binary tree is resident and walked; its stack must be configured for it)
      const std::string why = ctx->error;
      (void)configure_stack(ctx);
      ctx->error = why;
      return rc;
    }
  }
  return configure_stack(ctx);
}

// One hash grid from one seed's staged appends (hashgrid.h): compact the stage in (path, vertex) order, hash the keys, build
// the table the serial probe sequence of find_or_insert would build (hashgrid.hlsli:43-58; in parallel: hashgrid.hip), prefix
// the bucket counters (compute_indices, :72-79) and scatter the records into their bucket ranges (swizzle, :81-88) — all of it
// enqueued on the stream, nothing visits the host.
static int build_hash_grid(sthip_ctx* ctx, hipStream_t st, const float4* appends, float4* compact, float4* data, size_t slots, uint32_t rec, uint32_t flag_at, bool lvc_records,
                           uint32_t bucket_count, DevBuf<uint32_t>& d_checksums, DevBuf<uint32_t>& d_counters, DevBuf<uint32_t>& d_indices) {
  const uint32_t buckets = bucket_count + 32u;  // probing does not wrap
  size_t tmp_bytes = ctx->hg_tmp.n;
  HIP_TRY(ctx, sthip::lvc_compact(appends, (uint32_t)slots, 1, (uint32_t)slots, compact, ctx->hg_count.p, ctx->hg_flags.p, ctx->hg_offsets.p, ctx->hg_tmp.p, tmp_bytes, st, rec, flag_at));
  const unsigned kgrid = (unsigned)((slots + STHIP_BLOCK - 1) / STHIP_BLOCK);
  if (lvc_records)
    hipLaunchKernelGGL(k_hg_keys_lvc, dim3(kgrid), dim3(STHIP_BLOCK), 0, st, compact, ctx->hg_count.p, bucket_count, ctx->hg_keys.p);
  else
    hipLaunchKernelGGL(k_hg_keys, dim3(kgrid), dim3(STHIP_BLOCK), 0, st, compact, ctx->hg_count.p, bucket_count, ctx->hg_keys.p);
  // the probe sequence, the bucket ranges and every record's place, on the device (hashgrid.hip): no host hop
  size_t build_bytes = ctx->hg_tmp.n;
  HIP_TRY(ctx, sthip::hashgrid_build_device(ctx->hg_keys.p, ctx->hg_count.p, (uint32_t)slots, buckets, d_checksums.p, d_counters.p, d_indices.p, ctx->hg_dest.p, ctx->hg_owner.p, ctx->hg_bucket_of.p,
                                            ctx->hg_append.p, ctx->hg_sorted_bucket.p, ctx->hg_sorted_append.p, ctx->hg_key64.p, ctx->hg_sorted_key64.p, ctx->hg_tmp.p, build_bytes, st, ctx->hashgrid_serial));
  if (lvc_records)
    hipLaunchKernelGGL(k_hg_scatter_lvc, dim3(kgrid), dim3(STHIP_BLOCK), 0, st, compact, ctx->hg_count.p, ctx->hg_dest.p, data);
  else
    hipLaunchKernelGGL(k_hg_scatter, dim3(kgrid), dim3(STHIP_BLOCK), 0, st, compact, ctx->hg_count.p, ctx->hg_dest.p, data);
  HIP_TRY(ctx, hipGetLastError());
  return STHIP_OK;
}

// The buffers sthip_render sizes by the number of paths in flight (released before a second attempt with half the batch)
static void release_path_state(sthip_ctx* ctx) {
  (void)hipStreamSynchronize(ctx->stream);  // an earlier call's kernels may still read them
  ctx->ray_o.release();
  ctx->ray_d.release();
  ctx->hit.release();
  ctx->beta.release();
  ctx->radiance.release();
  ctx->shadow_sum.release();
  ctx->shadow_rays.release();
  ctx->light_vertices.release();
  ctx->conn.release();
  ctx->media_state.release();
  ctx->shade_stack.release();
  ctx->shadow_hit.release();
  ctx->shadow_ext.release();
  ctx->shadow_result.release();
  ctx->cone.release();
  ctx->meta.release();
  ctx->queue0.release();
  ctx->queue1.release();
  ctx->queue_kept.release();
  ctx->bdpt.release();
  ctx->rr.release();
  ctx->cs_nee.release();
  ctx->cs_lvc.release();
  ctx->lvc_staging.release();
  ctx->lvc_flags.release();
  ctx->lvc_offsets.release();
  ctx->lvc_tmp.release();
  ctx->path_contrib.release();
  ctx->light_trace.release();
  ctx->presampled.release();
  ctx->deep_rays.release();
}

static int render_once(sthip_ctx* ctx, const sthip_BDPTPushConstants* pc, uint32_t sampling_flags, uint32_t scene_flags, const sthip_frame_desc* frame, uint32_t seed_begin, uint32_t seed_count,
                       const sthip_outputs* out);

// A render allocates its path state (~330 B per path in flight at the default flags) before it enqueues anything. Should the
// device not have that much left — a host application that holds memory of its own, several contexts on one device — the
// batch is halved (fewer seeds traced together: the same frame, a little slower) and the call tried again, down to one seed;
// the smaller batch stays for the calls that follow (stats: max_paths_in_flight, batch_halvings).
int sthip_render(sthip_ctx* ctx, const sthip_BDPTPushConstants* pc, uint32_t sampling_flags, uint32_t scene_flags, const sthip_frame_desc* frame, uint32_t seed_begin,
                 uint32_t seed_count, const sthip_outputs* out) {
  if (!ctx) return STHIP_ERR_INVALID_ARGUMENT;
  for (;;) {
    ctx->last_hip_error = hipSuccess;
    ctx->render_launched = false;
    const int rc = render_once(ctx, pc, sampling_flags, scene_flags, frame, seed_begin, seed_count, out);
    ctx->stats.max_paths_in_flight = ctx->max_paths_in_flight;
    if (rc != STHIP_ERR_HIP || ctx->last_hip_error != hipErrorOutOfMemory || ctx->render_launched) return rc;
    (void)hipGetLastError();  // (the allocation's error is not sticky, but it is the "last error" until read)
    const uint64_t per_seed = std::max<uint64_t>(1, ctx->stats.paths_per_seed);
    if (ctx->max_paths_in_flight <= per_seed || ctx->max_paths_in_flight <= 1) return rc;  // one seed in flight already: it does not fit
    release_path_state(ctx);
    ctx->max_paths_in_flight = std::max<uint64_t>(per_seed, ctx->max_paths_in_flight / 2);
    ctx->stats.batch_halvings++;
    if (getenv("STHIP_VERBOSE")) fprintf(stderr, "[sthip] out of device memory (%s): max_paths_in_flight -> %llu\n", ctx->error.c_str(), (unsigned long long)ctx->max_paths_in_flight);
  }
}

static int render_once(sthip_ctx* ctx, const sthip_BDPTPushConstants* pc, uint32_t sampling_flags, uint32_t scene_flags, const sthip_frame_desc* frame, uint32_t seed_begin, uint32_t seed_count,
                       const sthip_outputs* out) {
  if (!pc || !frame || !out || !out->gRadiance || !frame->gViews || !frame->gViewTransforms || frame->view_count == 0 || seed_count == 0)
    return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "render: a required argument is NULL/zero");
  if (!ctx->has_scene) return fail(ctx, STHIP_ERR_NO_SCENE, "no scene uploaded");
  if (pc->gViewCount != frame->view_count) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "render: gViewCount != frame.view_count");
  if ((out->gDepth || out->gPrevUVs) && !frame->gInverseViewTransforms && !frame->gPrevInverseViewTransforms)
    return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "render: depth / prev-uv outputs need gInverseViewTransforms");
  // BDPT_FLAG_TRACE_LIGHT is a per-kernel specialisation of the reference (sample_photons), never a caller's choice
  if (scene_flags & STHIP_BDPT_FLAG_TRACE_LIGHT) return fail(ctx, STHIP_ERR_UNSUPPORTED, "render: BDPT_FLAG_TRACE_LIGHT is not a scene flag a caller sets");
  const uint32_t unsupported = (1u << STHIP_eSampleLightPower);
  if (sampling_flags & unsupported) return fail(ctx, STHIP_ERR_UNSUPPORTED, "render: a sampling flag outside the built hot path is set");
  if (pc->gMaxPathVertices > 60 || pc->gMaxDiffuseVertices > 60) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "render: path length limits above 60");
  if (pc->gLightCount > ctx->light_count) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "render: gLightCount exceeds the uploaded light list");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;

  // flag resolution of BDPT::render (BDPT.cpp:486-523), idempotent if the host has done it already
  sthip_BDPTPushConstants pcn = *pc;
  if (pcn.gLightCount == 0) scene_flags &= ~STHIP_BDPT_FLAG_HAS_EMISSIVES;
  const bool has_env = (scene_flags & STHIP_BDPT_FLAG_HAS_ENVIRONMENT) != 0, has_emissives = (scene_flags & STHIP_BDPT_FLAG_HAS_EMISSIVES) != 0;
  if (!has_env) pcn.gEnvironmentSampleProbability = 0;
  if (!has_emissives) pcn.gEnvironmentSampleProbability = 1;
  if (!has_emissives && !has_env) sampling_flags &= ~((1u << STHIP_eNEE) | (1u << STHIP_eConnectToViews) | (1u << STHIP_eConnectToLightPaths));
  if (!(sampling_flags & (1u << STHIP_eNEE))) sampling_flags &= ~((1u << STHIP_ePresampleLights) | (1u << STHIP_eNEEReservoirs) | (1u << STHIP_eNEEReservoirReuse));
  if (!(sampling_flags & (1u << STHIP_eNEEReservoirs))) sampling_flags &= ~(1u << STHIP_eNEEReservoirReuse);  // only connect_light_reservoir touches the grid
  if (!(sampling_flags & (1u << STHIP_eLVC))) sampling_flags &= ~((1u << STHIP_eLVCReservoirs) | (1u << STHIP_eLVCReservoirReuse));  // BDPT.cpp:517-520
  if (!(sampling_flags & (1u << STHIP_eConnectToLightPaths))) sampling_flags &= ~((1u << STHIP_eLVC) | (1u << STHIP_eLVCReservoirs) | (1u << STHIP_eLVCReservoirReuse));  // only connect_lvc reads the cache
  if (!(sampling_flags & (1u << STHIP_eLVCReservoirs))) sampling_flags &= ~(1u << STHIP_eLVCReservoirReuse);  // the reuse sits inside connect_lvc's reservoir branch
  if (!(sampling_flags & ((1u << STHIP_eNEE) | (1u << STHIP_eLVC)))) sampling_flags &= ~(1u << STHIP_eDeferShadowRays);  // BDPT.cpp:522-523
  // eCoherentSampling only touches the index of a presampled light (path.hlsli:317,379) and connect_lvc's (:688,703)
  if (!(sampling_flags & ((1u << STHIP_ePresampleLights) | (1u << STHIP_eLVC)))) sampling_flags &= ~(1u << STHIP_eCoherentSampling);
  pc = &pcn;
  if (has_env) {  // the Environment record (environment.h:17-22): ImageValue3, then 4 offsets into gDistributions when an image is bound
    const size_t addr = pcn.gEnvironmentMaterialAddress;
    if (addr + 16 > ctx->materials_host.size() || (addr & 3)) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "render: gEnvironmentMaterialAddress is outside gMaterialData");
    uint32_t rec[8] = {0};
    memcpy(rec, ctx->materials_host.data() + addr, 16);
    if (rec[3] < STHIP_IMAGE_COUNT) {
      if (rec[3] >= ctx->image_count || addr + 32 > ctx->materials_host.size()) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "render: the environment refers to an image that is not in gImages");
      memcpy(rec, ctx->materials_host.data() + addr, 32);
      const size_t w = ctx->image_dims[rec[3]].first, h = ctx->image_dims[rec[3]].second;
      const size_t need[4] = {h, w * h, h + 1, (w + 1) * h};  // marginal_pdf, row_pdf, marginal_cdf, row_cdf (dist2.h)
      for (int k = 0; k < 4; k++)
        if ((size_t)rec[4 + k] + need[k] > ctx->distribution_count) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "render: an environment distribution table lies outside gDistributions");
    }
  }

  // participating media (BDPT_FLAG_HAS_MEDIA, BDPT.cpp:497-500)
  if (ctx->has_volumes && !(scene_flags & STHIP_BDPT_FLAG_HAS_MEDIA)) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "render: the scene has volume instances but BDPT_FLAG_HAS_MEDIA is not set");
  const bool media = ctx->has_volumes;
  if (media && (sampling_flags & (1u << STHIP_eCoherentSampling))) return fail(ctx, STHIP_ERR_UNSUPPORTED, "render: eCoherentSampling with media (walks through volumes break the lockstep of a workgroup)");
  // With media every visibility ray draws random numbers. A deferred NEE ray carries its own offset (k_shadow_media walks it);
  // everything else draws from the path's own stream in the middle of a vertex — NEE without eDeferShadowRays, light tracing's
  // connect_view, the connections to stored light vertices or to the light vertex cache — and k_shade / k_shade_light walk those
  // themselves (visibility_walk_media).
  if (!media) pcn.gMaxNullCollisions = 0;
  const uint32_t W = pc->gOutputExtent[0], H = pc->gOutputExtent[1];
  if (W == 0 || H == 0) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "render: empty output extent");
  const size_t pixels = (size_t)W * H;
  FrameParams p;
  memset(&p, 0, sizeof(p));
  p.pc = *pc;
  p.sampling_flags = sampling_flags;
  p.scene_flags = scene_flags;
  p.shard_rank = ctx->shard_rank;
  p.shard_count = ctx->shard_count;
  p.tile_w = ctx->tile_w;
  p.tile_h = ctx->tile_h;
  p.tiles_x = (W + p.tile_w - 1) / p.tile_w;
  p.tiles_y = (H + p.tile_h - 1) / p.tile_h;
  const uint32_t tiles = p.tiles_x * p.tiles_y;
  const uint32_t owned = tiles > p.shard_rank ? (tiles - p.shard_rank + p.shard_count - 1) / p.shard_count : 0;
  p.paths_per_seed = owned * p.tile_w * p.tile_h;
  // Seeds traced together in one pass. A shard of a frame is small (1/8 of 1080p = 259 K paths does not fill
  // 256 CUs of persistent waves), so several seeds of the owned pixels share the launches, up to ~4 M paths.
  const uint32_t max_in_flight = (uint32_t)std::max<uint64_t>(1, (ctx->max_paths_in_flight) / std::max(1u, p.paths_per_seed));
  // Reservoir reuse couples the seeds of a call: seed s looks into the hash grid seed s - 1 built (the reference's frame
  // and previous frame), so they are traced one at a time and the grid is built between them.
  const bool nee_reuse = (sampling_flags & (1u << STHIP_eNEEReservoirReuse)) != 0;
  const bool lvc_reuse = (sampling_flags & (1u << STHIP_eLVCReservoirReuse)) != 0;
  // BDPTDebugMode: upstream's gDebugImage persists from frame to frame and most modes add to it or overwrite it: the seeds of a
  // call are traced one after the other, as its frames are
  const uint32_t debug_mode = out->gDebugImage ? out->debug_mode : 0u;
  if (debug_mode >= STHIP_DEBUG_MODE_COUNT) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "render: debug_mode is not a BDPTDebugMode");
  if (debug_mode && out->radiance_layout == STHIP_LAYOUT_SHARD_TILES && false) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "render: debug image with packed tiles");
  const uint32_t batch = (nee_reuse || lvc_reuse || debug_mode) ? 1u : std::min(seed_count, max_in_flight);
  p.path_count = batch * p.paths_per_seed;
  ctx->stats.paths_per_seed = p.paths_per_seed;
  ctx->stats.seeds_in_flight = batch;
  // light tracing (eConnectToViews, BDPT.cpp:653-667): sample_photons' padded dispatch, dispatch_over(W, ceil(gLightPathCount / W))
  const bool connect_views = (sampling_flags & (1u << STHIP_eConnectToViews)) != 0;
  const bool connect_paths = (sampling_flags & (1u << STHIP_eConnectToLightPaths)) != 0;  // light-subpath connections, no light vertex cache
  const bool bdpt = connect_views || connect_paths;
  const bool light_tracing = bdpt && pc->gMaxPathVertices > 2;
  if (bdpt) {
    if (has_env) return fail(ctx, STHIP_ERR_UNSUPPORTED, "render: light subpaths with an environment (upstream starts environment light paths from an unset position, bdpt.hlsl:109-113)");
    if (!frame->gInverseViewTransforms) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "render: eConnectToViews / eConnectToLightPaths need gInverseViewTransforms");
    if (connect_paths && !(sampling_flags & (1u << STHIP_eRemapThreads)) && (W & 7u))
      return fail(ctx, STHIP_ERR_UNSUPPORTED, "render: eConnectToLightPaths without eRemapThreads needs a width that is a multiple of 8 (upstream's padding threads race on the vertex slots of the next row)");
    if (connect_paths && pc->gMaxDiffuseVertices < 1) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "render: eConnectToLightPaths needs gMaxDiffuseVertices >= 1");
  }
  const bool lvc = connect_paths && (sampling_flags & (1u << STHIP_eLVC));
  const bool lvc_reservoirs = lvc && (sampling_flags & (1u << STHIP_eLVCReservoirs));
  if (lvc) {
    if (pc->gMaxDiffuseVertices < 2) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "render: eLVC needs gMaxDiffuseVertices >= 2 (a light path stores vertices 1 .. gMaxDiffuseVertices - 1)");
    if (pc->gLightPathCount == 0 || (uint64_t)pc->gLightPathCount * pc->gMaxDiffuseVertices > (1ull << 28)) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "render: eLVC needs 0 < gLightPathCount * gMaxDiffuseVertices <= 2^28");
  }
  const uint32_t light_rows = (pc->gLightPathCount + W - 1) / W;
  const uint32_t light_threads = light_tracing ? ((W + 7) / 8) * 8 * ((light_rows + 3) / 4) * 4 : 0;
  if ((uint64_t)light_threads * batch > 0x7FFFFFFFull || (light_tracing && (uint64_t)batch * W * H > 0x7FFFFFFFull)) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "render: too many light paths in flight");
  const size_t P = std::max<size_t>(std::max<size_t>(1, p.path_count), (size_t)light_threads * batch);
  const size_t P0 = std::max<size_t>(1, p.paths_per_seed);
  const size_t vertices_per_seed = connect_paths ? (size_t)pc->gLightPathCount * pc->gMaxDiffuseVertices : 0;
  const size_t conn_per_path = connect_paths ? pc->gMaxDiffuseVertices - 1 : 0;
  if (bdpt) {
    HIP_TRY(ctx, ctx->bdpt.ensure(P));
    if (light_tracing && connect_views) HIP_TRY(ctx, ctx->light_trace.ensure((size_t)batch * W * H * 4));
    if (connect_paths) {
      if ((uint64_t)P * std::max<size_t>(1, conn_per_path) >= 0x40000000ull) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "render: too many connection entries in flight");
      HIP_TRY(ctx, ctx->light_vertices.ensure(4 * std::max<size_t>(1, vertices_per_seed * batch)));
      HIP_TRY(ctx, ctx->conn.ensure(std::max<size_t>(1, (size_t)p.path_count * conn_per_path)));
    }
    if (lvc) {
      const size_t slots = (size_t)pc->gLightPathCount * (pc->gMaxDiffuseVertices - 1) * batch;
      if (slots > 0x7FFFFFFFull) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "render: too many light-vertex-cache slots in flight");
      HIP_TRY(ctx, ctx->lvc_staging.ensure(4 * slots));
      HIP_TRY(ctx, ctx->lvc_flags.ensure(slots));
      HIP_TRY(ctx, ctx->lvc_offsets.ensure(slots));
      HIP_TRY(ctx, ctx->lvc_count.ensure(batch));
      size_t tmp_bytes = 0;
      HIP_TRY(ctx, sthip::lvc_compact(nullptr, (uint32_t)(slots / batch), batch, 0, nullptr, nullptr, ctx->lvc_flags.p, ctx->lvc_offsets.p, nullptr, tmp_bytes, ctx->stream));
      HIP_TRY(ctx, ctx->lvc_tmp.ensure(std::max<size_t>(16, tmp_bytes)));
      if (lvc_reservoirs) HIP_TRY(ctx, ctx->path_contrib.ensure(P));
    }
  }

  size_t hg_slots = 0;
  const uint32_t hg_buckets = (nee_reuse || lvc_reuse) ? pc->gHashGridBucketCount + 32u : 0u;  // probing does not wrap (hashgrid.h)
  if (nee_reuse || lvc_reuse) {
    if (has_env) return fail(ctx, STHIP_ERR_UNSUPPORTED, "render: eNEEReservoirReuse with an environment (a stored environment sample is read back as a surface point upstream: sample_Le leaves its pdfA positive)");
    if (ctx->shard_count > 1) return fail(ctx, STHIP_ERR_UNSUPPORTED, "render: eNEEReservoirReuse on a pixel-tile shard (the grid is a whole-frame structure: render replicas and reduce)");
    if (pc->gHashGridBucketCount == 0 || pc->gHashGridBucketCount > (1u << 28) || !(pc->gHashGridMinBucketRadius > 0)) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "render: eNEEReservoirReuse needs 0 < gHashGridBucketCount <= 2^28 and gHashGridMinBucketRadius > 0");
    hg_slots = (size_t)((W + 7) / 8) * ((H + 3) / 4) * 32 * std::max(1u, pc->gMaxDiffuseVertices);  // covers both map_pixel_coord forms
    if (hg_slots > 0x7FFFFFFFull) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "render: too many hash-grid append slots");
    if (nee_reuse) {
      HIP_TRY(ctx, ctx->hg_appends.ensure(4 * hg_slots));
      HIP_TRY(ctx, ctx->hg_compact.ensure(4 * hg_slots));
      HIP_TRY(ctx, ctx->hg_data.ensure(3 * hg_slots));
    }
    if (lvc_reuse) {
      HIP_TRY(ctx, ctx->lg_appends.ensure(6 * hg_slots));
      HIP_TRY(ctx, ctx->lg_compact.ensure(6 * hg_slots));
      HIP_TRY(ctx, ctx->lg_data.ensure(5 * hg_slots));
      HIP_TRY(ctx, ctx->lg_checksums.ensure(hg_buckets));
      HIP_TRY(ctx, ctx->lg_counters.ensure(hg_buckets));
      HIP_TRY(ctx, ctx->lg_indices.ensure(hg_buckets));
    }
    HIP_TRY(ctx, ctx->hg_flags.ensure(hg_slots));
    HIP_TRY(ctx, ctx->hg_offsets.ensure(hg_slots));
    HIP_TRY(ctx, ctx->hg_dest.ensure(hg_slots));
    HIP_TRY(ctx, ctx->hg_keys.ensure(hg_slots));
    HIP_TRY(ctx, ctx->hg_count.ensure(1));
    HIP_TRY(ctx, ctx->hg_checksums.ensure(hg_buckets));
    HIP_TRY(ctx, ctx->hg_counters.ensure(hg_buckets));
    HIP_TRY(ctx, ctx->hg_indices.ensure(hg_buckets));
    HIP_TRY(ctx, ctx->hg_owner.ensure(hg_buckets + 2 + 1024));  // (+ hashgrid.hip's control words and special-cell list)
    HIP_TRY(ctx, ctx->hg_key64.ensure(hg_slots));
    HIP_TRY(ctx, ctx->hg_sorted_key64.ensure(hg_slots));
    HIP_TRY(ctx, ctx->hg_bucket_of.ensure(hg_slots));
    HIP_TRY(ctx, ctx->hg_append.ensure(hg_slots));
    HIP_TRY(ctx, ctx->hg_sorted_bucket.ensure(hg_slots));
    HIP_TRY(ctx, ctx->hg_sorted_append.ensure(hg_slots));
    size_t tmp_bytes = 0, build_bytes = 0;
    HIP_TRY(ctx, sthip::lvc_compact(nullptr, (uint32_t)hg_slots, 1, 0, nullptr, nullptr, ctx->hg_flags.p, ctx->hg_offsets.p, nullptr, tmp_bytes, ctx->stream));
    HIP_TRY(ctx, sthip::hashgrid_build_device(nullptr, nullptr, (uint32_t)hg_slots, hg_buckets, nullptr, ctx->hg_counters.p, ctx->hg_indices.p, nullptr, nullptr, ctx->hg_bucket_of.p, ctx->hg_append.p,
                                              ctx->hg_sorted_bucket.p, ctx->hg_sorted_append.p, ctx->hg_key64.p, ctx->hg_sorted_key64.p, nullptr, build_bytes, ctx->stream));
    HIP_TRY(ctx, ctx->hg_tmp.ensure(std::max<size_t>(16, std::max(tmp_bytes, build_bytes))));
  }

  HIP_TRY(ctx, ctx->ray_o.ensure(P));
  HIP_TRY(ctx, ctx->ray_d.ensure(P));
  HIP_TRY(ctx, ctx->hit.ensure(P));
  HIP_TRY(ctx, ctx->hit_leaf.ensure(P));
  HIP_TRY(ctx, ctx->beta.ensure(P));
  HIP_TRY(ctx, ctx->radiance.ensure(P));
  HIP_TRY(ctx, ctx->shadow_sum.ensure(P));
  HIP_TRY(ctx, ctx->accum.ensure(P0));
  if (ctx->textured || debug_mode) HIP_TRY(ctx, ctx->cone.ensure(P));  // (a debug mode runs the general instantiation of k_shade)
  if (debug_mode) HIP_TRY(ctx, ctx->debug.ensure(P));
  // The queues are cut into QUEUE_SEGMENTS segments (traverse.h). A segment starts as a contiguous eighth of the
  // slots and only shrinks from bounce to bounce, which bounds it and the distance between segments.
  const uint32_t shade_grid = std::max<uint32_t>(grid_for(ctx, P), QUEUE_SEGMENTS);
  const size_t seg_stride = (((P + QUEUE_SEGMENTS - 1) / QUEUE_SEGMENTS) + 63) & ~(size_t)63;
  if (seg_stride * QUEUE_SEGMENTS > 0xFFFFFFFFull) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "render: too many paths in flight");
  // a vertex queues at most one NEE ray, plus one visibility ray per stored light vertex it connects to
  // (media: the walks of earlier vertices are still in the queue when a vertex adds its own — at most one per diffuse vertex and path)
  const size_t shadow_stride = seg_stride * (media ? std::max<size_t>(1, pc->gMaxDiffuseVertices) : 1 + conn_per_path);
  if (shadow_stride * QUEUE_SEGMENTS > 0xFFFFFFFFull) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "render: too many shadow rays in flight");
  const size_t shadow_entries = shadow_stride * QUEUE_SEGMENTS;  // per round; media ping-pong between two such regions
  HIP_TRY(ctx, ctx->shadow_rays.ensure(3 * shadow_entries * (media ? 2 : 1)));
  if (ctx->bvh.spill) {  // bounded LDS stacks: room for every ray of a trace launch to overflow (4 x float4 each)
    HIP_TRY(ctx, ctx->deep_rays.ensure(4 * (P + shadow_entries)));
    if (ctx->deep_count.n < 2) {  // [0] rays in the deep queue, [1] k_trace_deep blocks that are through: both zero between launches (k_trace_deep resets them)
      HIP_TRY(ctx, ctx->deep_count.ensure(2));
      HIP_TRY(ctx, hipMemsetAsync(ctx->deep_count.p, 0, 8, ctx->stream));
    }
  }
  if (media) {
    if (2 * shadow_entries > 0xFFFFFFFFull) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "render: too many shadow rays in flight");
    HIP_TRY(ctx, ctx->media_state.ensure(2 * P));
    HIP_TRY(ctx, ctx->shadow_hit.ensure(2 * shadow_entries));
    HIP_TRY(ctx, ctx->shadow_ext.ensure(2 * shadow_entries));
    HIP_TRY(ctx, ctx->shadow_result.ensure(P * std::max(1u, pc->gMaxDiffuseVertices)));
    HIP_TRY(ctx, ctx->view_medium.ensure(std::max(1u, frame->view_count)));
  }
  HIP_TRY(ctx, ctx->meta.ensure(P));
  HIP_TRY(ctx, ctx->queue0.ensure(seg_stride * QUEUE_SEGMENTS));
  HIP_TRY(ctx, ctx->queue1.ensure(seg_stride * QUEUE_SEGMENTS));
  HIP_TRY(ctx, ctx->queue_kept.ensure(seg_stride * QUEUE_SEGMENTS));
  HIP_TRY(ctx, ctx->counters.ensure(CNT_TOTAL));
  HIP_TRY(ctx, ctx->qctl.ensure((size_t)3 * 64 * QUEUE_SEGMENTS * QCTL_STRIDE));  // path queues, shadow queues, and (BDPTDebugMode) the shadow rays' debug halves
  // ePresampleLights (BDPT.cpp:644-651): gLightPresampleTileSize x TileCount light points per seed in flight
  const bool presample = (sampling_flags & (1u << STHIP_ePresampleLights)) && pc->gMaxPathVertices > 2;
  const size_t presample_n = (size_t)pc->gLightPresampleTileSize * pc->gLightPresampleTileCount;
  if (sampling_flags & (1u << STHIP_ePresampleLights)) {
    if (has_env) return fail(ctx, STHIP_ERR_UNSUPPORTED, "render: ePresampleLights with an environment (upstream leaves the presampled environment direction unset, bdpt.hlsl:93)");
    if (presample_n == 0 || presample_n > (1u << 24)) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "render: gLightPresampleTileSize * gLightPresampleTileCount must be in 1 .. 2^24");
    HIP_TRY(ctx, ctx->presampled.ensure(2 * presample_n * batch));
  }

  // views
  const uint32_t nv = frame->view_count;
  const size_t vbytes = (size_t)nv * 48;
  HIP_TRY(ctx, ctx->views.ensure(5 * vbytes));
  {
    std::vector<uint8_t> host(5 * vbytes);
    if (frame->gInverseViewTransforms)
      memcpy(host.data() + 4 * vbytes, frame->gInverseViewTransforms, vbytes);
    else
      memset(host.data() + 4 * vbytes, 0, vbytes);
    memcpy(host.data(), frame->gViews, vbytes);
    memcpy(host.data() + vbytes, frame->gViewTransforms, vbytes);
    memcpy(host.data() + 2 * vbytes, frame->gPrevViews ? frame->gPrevViews : frame->gViews, vbytes);
    const sthip_TransformData* piv = frame->gPrevInverseViewTransforms ? frame->gPrevInverseViewTransforms : frame->gInverseViewTransforms;
    if (piv)
      memcpy(host.data() + 3 * vbytes, piv, vbytes);
    else
      memset(host.data() + 3 * vbytes, 0, vbytes);
    HIP_TRY(ctx, hipMemcpyAsync(ctx->views.p, host.data(), 5 * vbytes, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));  // `host` goes out of scope
  }
  p.views = reinterpret_cast<const sthip_ViewData*>(ctx->views.p);
  p.view_xf = reinterpret_cast<const sthip_TransformData*>(ctx->views.p + vbytes);
  p.prev_views = reinterpret_cast<const sthip_ViewData*>(ctx->views.p + 2 * vbytes);
  p.prev_inv_view_xf = reinterpret_cast<const sthip_TransformData*>(ctx->views.p + 3 * vbytes);
  p.inv_view_xf = reinterpret_cast<const sthip_TransformData*>(ctx->views.p + 4 * vbytes);
  p.bdpt = bdpt ? ctx->bdpt.p : nullptr;
  p.light_trace = light_tracing && connect_views ? ctx->light_trace.p : nullptr;
  p.light_trace_empty = connect_views && !light_tracing ? 1u : 0u;
  p.light_vertices = connect_paths ? ctx->light_vertices.p : nullptr;
  p.conn = connect_paths && conn_per_path ? ctx->conn.p : nullptr;
  // eCoherentRR takes effect in rounds in which a path can reach the roulette (see run below); never with media
  const bool coherent_rr = (sampling_flags & (1u << STHIP_eCoherentRR)) && !media;
  if (coherent_rr) HIP_TRY(ctx, ctx->rr.ensure(P));
  p.rr = nullptr;
  // eCoherentSampling: one probe per site and round (FrameParams::cs_nee / cs_lvc)
  const bool coherent_nee = (sampling_flags & (1u << STHIP_eCoherentSampling)) && (sampling_flags & (1u << STHIP_ePresampleLights)) && (sampling_flags & (1u << STHIP_eNEE));
  const bool coherent_lvc = (sampling_flags & (1u << STHIP_eCoherentSampling)) && lvc;
  if (coherent_nee) HIP_TRY(ctx, ctx->cs_nee.ensure(P));
  if (coherent_lvc) HIP_TRY(ctx, ctx->cs_lvc.ensure(P));
  p.cs_nee = nullptr;
  p.cs_lvc = nullptr;
  p.probe_kind = 0;
  p.lds_material_bytes = (ctx->lds_materials && !ctx->textured && ctx->materials_host.size() <= 32768) ? (uint32_t)(ctx->materials_host.size() & ~(size_t)3) : 0u;
  const size_t shade_lds = p.lds_material_bytes;
  p.hg_checksums = ctx->hg_checksums.p;
  p.hg_counters = ctx->hg_counters.p;
  p.hg_indices = ctx->hg_indices.p;
  p.hg_data = ctx->hg_data.p;
  // the first seed of the call looks into the grids the previous call left, if it left them for the same estimator and table
  const uint64_t reuse_key[3] = {(uint64_t)(nee_reuse ? 1u : 0u) | (lvc_reuse ? 2u : 0u), hg_buckets, hg_slots};
  p.hg_prev = ((nee_reuse || lvc_reuse) && ctx->reuse_persist && ctx->reuse_grids_valid && !memcmp(reuse_key, ctx->reuse_key, sizeof reuse_key)) ? 1u : 0u;
  if (nee_reuse || lvc_reuse) ctx->reuse_grids_valid = false;  // (until this call has left its own)
  p.hg_appends = nee_reuse ? ctx->hg_appends.p : nullptr;
  p.lg_checksums = ctx->lg_checksums.p;
  p.lg_counters = ctx->lg_counters.p;
  p.lg_indices = ctx->lg_indices.p;
  p.lg_data = ctx->lg_data.p;
  p.lg_appends = lvc_reuse ? ctx->lg_appends.p : nullptr;
  p.lvc_staging = lvc ? ctx->lvc_staging.p : nullptr;
  p.lvc_count = lvc ? ctx->lvc_count.p : nullptr;
  p.path_contrib = lvc_reservoirs ? ctx->path_contrib.p : nullptr;
  p.light_threads = light_threads;
  p.light_trace_quantization = 65536;  // BDPT.hpp:55 mLightTraceQuantization

  p.bvh = ctx->bvh;
  p.bvh.alpha_test = (ctx->has_alpha && (sampling_flags & (1u << STHIP_eAlphaTest))) ? 1u : 0u;  // intersection.hlsli:118
  p.bvh.flip_uvs = (sampling_flags & (1u << STHIP_eFlipTriangleUVs)) ? 1u : 0u;
  p.scene.vertices = ctx->vertices.p;
  p.scene.indices = ctx->indices.p;
  p.scene.instances = ctx->instances.p;
  p.scene.xf = ctx->xf.p;
  p.scene.inv_xf = ctx->inv_xf.p;
  p.scene.motion_xf = ctx->motion_xf.p;
  p.scene.materials = ctx->materials.p;
  p.scene.lights = ctx->lights.p;
  p.scene.instance_count = ctx->instance_count;
  p.scene.light_count = ctx->light_count;
  p.scene.images = ctx->images.p;
  p.scene.image_texels = ctx->image_texels.p;
  p.scene.image_count = ctx->image_count;
  p.scene.distributions = ctx->distributions.p;
  p.scene.distribution_count = ctx->distribution_count;
  p.scene.volume_words = ctx->volume_words.p;
  p.scene.volumes = ctx->volumes.p;
  p.scene.volume_count = ctx->volume_count;
  p.scene.leaf_tris = ctx->bvh.tris;
  p.scene.leaf_shade = reinterpret_cast<const float4*>(ctx->tri_shade.p);
  p.ray_o = ctx->ray_o.p;
  p.ray_d = ctx->ray_d.p;
  p.hit = ctx->hit.p;
  p.hit_leaf = ctx->hit_leaf.p;
  p.beta = ctx->beta.p;
  p.meta = ctx->meta.p;
  p.radiance = ctx->radiance.p;
  p.shadow_sum = ctx->shadow_sum.p;
  p.accum = ctx->accum.p;
  p.cone = (ctx->textured || debug_mode) ? ctx->cone.p : nullptr;
  p.queue[0] = ctx->queue0.p;
  p.queue[1] = ctx->queue1.p;
  p.shadow_rays = ctx->shadow_rays.p;
  p.deep_rays = ctx->deep_rays.p;
  p.deep_count = ctx->deep_count.p;
  p.presampled = ctx->presampled.p;
  p.counters = ctx->counters.p;
  p.qctl = ctx->qctl.p;
  p.seg_stride = (uint32_t)seg_stride;
  p.shadow_stride = (uint32_t)shadow_stride;
  p.media = media ? 1u : 0u;
  const bool inline_media = media && (sampling_flags & (1u << STHIP_eNEE)) && !(sampling_flags & (1u << STHIP_eDeferShadowRays));
  p.inline_media = inline_media ? 1u : 0u;
  if (inline_media || (media && bdpt)) {  // (light tracing's connect_view and the light-subpath connections walk inline whatever eDeferShadowRays says)
    HIP_TRY(ctx, ctx->shade_stack.ensure((size_t)shade_grid * STHIP_BLOCK * std::max(1u, ctx->bvh.stack_depth)));
    p.shade_stack = ctx->shade_stack.p;
  }
  if (media) {
    p.shadow_alt = (uint32_t)shadow_entries;
    p.media_state = ctx->media_state.p;
    p.shadow_hit = ctx->shadow_hit.p;
    p.shadow_ext = ctx->shadow_ext.p;
    p.shadow_result = ctx->shadow_result.p;
    p.view_medium = nullptr;
    if (frame->gViewMediumInstances) {
      for (uint32_t v = 0; v < frame->view_count; v++) {
        const uint32_t mi = frame->gViewMediumInstances[v];
        if (mi != 0xFFFFu && (mi >= ctx->instance_count || !ctx->instance_is_volume[mi]))
          return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "render: gViewMediumInstances entry is not a volume instance");
      }
      HIP_TRY(ctx, hipMemcpyAsync(ctx->view_medium.p, frame->gViewMediumInstances, (size_t)frame->view_count * 4, hipMemcpyHostToDevice, st));
      HIP_TRY(ctx, hipStreamSynchronize(st));
      p.view_medium = ctx->view_medium.p;
    }
  }
  p.count_traversal = ctx->count_traversal ? 1u : 0u;
  p.refill_idle = ctx->refill_idle;
  p.culled = 0;
  p.inst_flags = ctx->inst_flags.p;
  p.emitters = ctx->emitters.p;
  p.emitter_count = 0;  // (set where the view pass starts: only the plain pipeline answers last rays)
  p.no_specular = ctx->has_specular ? 0u : 1u;
  p.inner_min_lanes = ctx->inner_min_lanes;

  // outputs: device pointers are written in place, host pointers go through staging buffers
  const bool dev = out->device_ptrs != 0;
  p.debug_mode = debug_mode;
  p.debug = debug_mode ? ctx->debug.p : nullptr;
  p.out_debug = nullptr;
  p.shadow_debug = nullptr;
  if (debug_mode) {
    if (dev) {
      p.out_debug = reinterpret_cast<float4*>(out->gDebugImage);
    } else {  // in / out: what the caller's image holds goes up first
      HIP_TRY(ctx, ctx->out_debug.ensure(pixels));
      HIP_TRY(ctx, hipMemcpyAsync(ctx->out_debug.p, out->gDebugImage, pixels * 16, hipMemcpyHostToDevice, st));
      p.out_debug = ctx->out_debug.p;
    }
    // inline shadow rays add to the debug image only where they are unoccluded: they are traced once more, with what they add
    const bool inline_adds = (debug_mode == STHIP_DEBUG_RESERVOIR_WEIGHT || (debug_mode == STHIP_DEBUG_PATH_LENGTH_CONTRIBUTION && pc->gDebugLightPathLength == 1)) &&
                             (sampling_flags & (1u << STHIP_eNEE)) && !(sampling_flags & (1u << STHIP_eDeferShadowRays)) && !media;
    // so do light-subpath connections (accumulate_contribution with the light vertex's length, path.hlsli:797,820); connect_lvc's
    // deferred record (:781-789) adds to the radiance only
    const bool connection_adds = debug_mode == STHIP_DEBUG_PATH_LENGTH_CONTRIBUTION && pc->gDebugLightPathLength >= 2 && connect_paths && !media &&
                                 !((sampling_flags & (1u << STHIP_eLVC)) && (sampling_flags & (1u << STHIP_eDeferShadowRays)));
    if (inline_adds || connection_adds) {
      HIP_TRY(ctx, ctx->shadow_debug.ensure(ctx->shadow_rays.n));
      p.shadow_debug = ctx->shadow_debug.p;
    }
  }
  p.out_packed = out->radiance_layout == STHIP_LAYOUT_SHARD_TILES ? 1u : 0u;
  if (out->radiance_layout > STHIP_LAYOUT_SHARD_TILES) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "render: unknown radiance_layout");
  const size_t radiance_entries = p.out_packed ? std::max<size_t>(1, p.paths_per_seed) : pixels;
  if (dev) {
    p.out_radiance = reinterpret_cast<float4*>(out->gRadiance);
    p.out_albedo = reinterpret_cast<float4*>(out->gAlbedo);
    p.out_visibility = out->gVisibility;
    p.out_depth = out->gDepth;
    p.out_prev_uv = reinterpret_cast<float2*>(out->gPrevUVs);
  } else {
    HIP_TRY(ctx, ctx->out_radiance.ensure(radiance_entries));
    p.out_radiance = ctx->out_radiance.p;
    if (out->gAlbedo) {
      HIP_TRY(ctx, ctx->out_albedo.ensure(pixels));
      p.out_albedo = ctx->out_albedo.p;
    }
    if (out->gVisibility) {
      HIP_TRY(ctx, ctx->out_visibility.ensure(pixels));
      p.out_visibility = ctx->out_visibility.p;
    }
    if (out->gDepth) {
      HIP_TRY(ctx, ctx->out_depth.ensure(pixels));
      p.out_depth = ctx->out_depth.p;
    }
    if (out->gPrevUVs) {
      HIP_TRY(ctx, ctx->out_prev_uv.ensure(pixels));
      p.out_prev_uv = ctx->out_prev_uv.p;
    }
  }
  // primary rays = owned pixels that lie inside the image and inside a view (known without asking the GPU)
  uint32_t primary_rays = 0;
  if (pc->gMaxPathVertices >= 2) {
    for (uint32_t t = p.shard_rank; t < tiles; t += p.shard_count) {
      const uint32_t ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
      const int x0 = (int)(tx * p.tile_w), y0 = (int)(ty * p.tile_h);
      const int x1 = (int)std::min(W, (tx + 1) * p.tile_w), y1 = (int)std::min(H, (ty + 1) * p.tile_h);
      if (nv == 1) {
        const int ax0 = std::max(x0, frame->gViews[0].image_min[0]), ay0 = std::max(y0, frame->gViews[0].image_min[1]);
        const int ax1 = std::min(x1, frame->gViews[0].image_max[0]), ay1 = std::min(y1, frame->gViews[0].image_max[1]);
        if (ax1 > ax0 && ay1 > ay0) primary_rays += (uint32_t)(ax1 - ax0) * (uint32_t)(ay1 - ay0);
      } else {
        for (int y = y0; y < y1; y++)
          for (int x = x0; x < x1; x++)
            for (uint32_t v = 0; v < nv; v++) {
              const sthip_ViewData& vw = frame->gViews[v];
              if (x >= vw.image_min[0] && y >= vw.image_min[1] && x < vw.image_max[0] && y < vw.image_max[1]) {
                primary_rays++;
                break;
              }
            }
      }
    }
  }
  // Pixels this shard does not own and pixels outside every view are zero (a sum-reduce over shards assembles the frame).
  // When the shard is the whole frame and every pixel lies in a view, every output entry is written by the pass itself —
  // the first vertex's G-buffer stores (hit or miss) and k_resolve — so the five fills (131 MB at 1080p) are left out.
  const bool every_entry_written = p.shard_count == 1 && !p.out_packed && !media && pc->gMaxPathVertices >= 2 && (size_t)primary_rays == pixels;
  if (!every_entry_written) {
    HIP_TRY(ctx, hipMemsetAsync(p.out_radiance, 0, radiance_entries * 16, st));
    if (p.out_albedo) HIP_TRY(ctx, hipMemsetAsync(p.out_albedo, 0, pixels * 16, st));
    if (p.out_visibility) HIP_TRY(ctx, hipMemsetAsync(p.out_visibility, 0, pixels * 8, st));
    if (p.out_depth) HIP_TRY(ctx, hipMemsetAsync(p.out_depth, 0, pixels * 16, st));
    if (p.out_prev_uv) HIP_TRY(ctx, hipMemsetAsync(p.out_prev_uv, 0, pixels * 8, st));
  }
  const uint32_t grid = shade_grid;
  const size_t lds = trace_lds_bytes(ctx);
  uint32_t tgrid = std::min<uint32_t>(trace_grid(ctx, lds), (uint32_t)((P + STHIP_BLOCK - 1) / STHIP_BLOCK));
  if (ctx->bvh.spill) tgrid = std::min<uint32_t>(tgrid, (uint32_t)ctx->cu_count * 8u);  // what the spill buffer has columns for (configure_stack)
  // closest-hit rays per path <= gMaxPathVertices - 1 (path.hlsli:960); without specular materials every scattering
  // vertex counts as a diffuse vertex, so the path also ends after gMaxDiffuseVertices + 1 rays (path.hlsli:964-966):
  // rounds beyond that would only be empty launches
  uint32_t max_bounce_rounds = pc->gMaxPathVertices >= 2 ? pc->gMaxPathVertices - 1 : 0;
  if (!ctx->has_specular) max_bounce_rounds = std::min(max_bounce_rounds, pc->gMaxDiffuseVertices + 1);
  // media: a trace() call walks from volume boundary to volume boundary, one k_trace round per segment (up to 2 per
  // volume instance and ray); a shadow ray likewise, so its last segments need rounds of their own after the last bounce.
  // The queue control words exist for 64 rounds; paths / shadow rays still walking after that are dropped.
  uint32_t drain_rounds = 0;
  if (media) {
    drain_rounds = std::min(8u, 2 * ctx->volume_instances + 1);
    max_bounce_rounds = std::min<uint64_t>(62 - drain_rounds, (uint64_t)max_bounce_rounds * (1 + 2 * ctx->volume_instances));
  }
  p.rounds = max_bounce_rounds + drain_rounds;
  // The deepest round whose vertices can queue a visibility ray. Without specular materials the vertex shaded in round d is
  // diffuse vertex d + 1, and the diffuse budget ends a path before NEE (path.hlsli:964-966 precede :978): rounds beyond
  // gMaxDiffuseVertices - 1 leave their shadow queue empty, and the launch that would trace it is left out.
  uint32_t max_shadow_round = 0xFFFFFFFFu;
  if (!ctx->has_specular && !media && !bdpt) max_shadow_round = pc->gMaxDiffuseVertices ? pc->gMaxDiffuseVertices - 1 : 0u;
  const bool timing = ctx->time_kernels;
  float ms_trace = 0, ms_primary = 0, ms_shade = 0, ms_other = 0;
  uint32_t launches_trace = 0, launches_primary = 0;
  uint64_t rays_primary = 0;
  auto timed = [&](float& acc, auto&& launch) -> int {
    if (timing) HIP_TRY(ctx, hipEventRecord(ctx->ev[0], st));
    launch();
    HIP_TRY(ctx, hipGetLastError());
    if (timing) {
      HIP_TRY(ctx, hipEventRecord(ctx->ev[1], st));
      HIP_TRY(ctx, hipEventSynchronize(ctx->ev[1]));
      float ms = 0;
      HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
      acc += ms;
    }
    return STHIP_OK;
  };

  ctx->render_launched = true;  // everything the call needs is allocated: from here on work is enqueued
  // the deep queue's control words: k_trace_deep leaves them at zero, but a call that was cut short between k_trace and
  // k_trace_deep (a failed launch, an error return) would not have: one 8-byte fill per call keeps every call self-contained
  if (ctx->bvh.spill && ctx->deep_count.p) HIP_TRY(ctx, hipMemsetAsync(ctx->deep_count.p, 0, 8, st));
  bool counters_cleared = false;  // (the first pass's k_clear takes the counters along with its queue control words)
  const bool nee = (sampling_flags & (1u << STHIP_eNEE)) != 0;
  const bool ext = ctx->has_spheres || has_env || bdpt || (sampling_flags & ((1u << STHIP_eNEEReservoirs) | (1u << STHIP_eShadingNormalShadowFix)));
  for (uint32_t s = 0; s < seed_count; s += batch) {
    const uint32_t in_flight = std::min(batch, seed_count - s);
    p.seed = seed_begin + s;
    p.seeds_in_flight = in_flight;
    p.write_aov = s == 0 ? 1u : 0u;
    int rc = STHIP_OK;
    // queue sizes and heads are per pass; the ray / traversal counters run over the whole call
    auto reset_queues = [&]() -> int {
      const uint32_t per_depth = QUEUE_SEGMENTS * QCTL_STRIDE;  // 64-bit words
      const uint32_t words = (max_bounce_rounds + drain_rounds + 1) * per_depth;  // the last round's shade appends to depth + 1; k_resolve sums p.rounds of them
      hipLaunchKernelGGL(k_clear, dim3(std::max(1u, std::min(64u, (2 * words + CNT_TOTAL + STHIP_BLOCK - 1) / STHIP_BLOCK))), dim3(STHIP_BLOCK), 0, st, queue_ctl_host(ctx->qctl.p, 0, 0), words,
                         queue_ctl_host(ctx->qctl.p, 1, 0), words, ctx->counters.p, counters_cleared ? 0u : (uint32_t)CNT_TOTAL);
      counters_cleared = true;
      if (p.shadow_debug) HIP_TRY(ctx, hipMemsetAsync(queue_ctl_host(ctx->qctl.p, 2, 0), 0, (size_t)words * 8, st));  // the debug halves' queues (BDPTDebugMode)
      HIP_TRY(ctx, hipGetLastError());
      return STHIP_OK;
    };
    auto trace = [&](uint32_t dc, uint32_t ds) -> int {
      if (dc == TRACE_NONE && ds == TRACE_NONE) return STHIP_OK;
      launches_trace++;
      return timed(ms_trace, [&]() {
        const bool alpha = p.bvh.alpha_test || ctx->has_volumes;  // alpha masks under eAlphaTest, volume instances: the instantiation that carries them
        void* kargs[3] = {(void*)&p, (void*)&dc, (void*)&ds};
        (void)hipLaunchKernel(p.bvh.wide8_nodes  ? trace_kernel_wide8(ctx->count_traversal, alpha, p.bvh.spill != nullptr)
                              : p.bvh.wide_nodes ? trace_kernel_wide(ctx->count_traversal, alpha, p.bvh.spill != nullptr)
                                                 : trace_kernel(ctx->count_traversal, alpha, p.bvh.spill != nullptr, p.bvh.top_count != 0),
                              dim3(tgrid), dim3(STHIP_BLOCK), kargs, lds, st);
        if (p.bvh.spill) {  // a tree higher than the LDS stack ran the bounded instantiation: now the rays that overflowed
          const uint32_t dgrid = (uint32_t)ctx->cu_count * 8u;  // one spill column per thread (configure_stack)
          if (alpha) {
            if (ctx->count_traversal)
              hipLaunchKernelGGL((k_trace_deep<true, true>), dim3(dgrid), dim3(STHIP_BLOCK), 0, st, p);
            else
              hipLaunchKernelGGL((k_trace_deep<false, true>), dim3(dgrid), dim3(STHIP_BLOCK), 0, st, p);
          } else if (ctx->count_traversal)
            hipLaunchKernelGGL((k_trace_deep<true, false>), dim3(dgrid), dim3(STHIP_BLOCK), 0, st, p);
          else
            hipLaunchKernelGGL((k_trace_deep<false, false>), dim3(dgrid), dim3(STHIP_BLOCK), 0, st, p);
        }
        if (p.shadow_debug && ds != TRACE_NONE) {
          // BDPTDebugMode: the same shadow rays once more, carrying what each adds to the debug image, accumulated into the paths'
          // debug pixels the way the first pass accumulated their contributions into the radiance (finish_ray)
          FrameParams q = p;
          q.shadow_rays = p.shadow_debug;
          q.radiance = p.debug;
          q.shadow_sum = p.debug;  // (a connection's debug half exists while NEE's rays are deferred: finish_ray's target either way)
          q.qctl = p.qctl + (size_t)64 * QUEUE_SEGMENTS * QCTL_STRIDE;  // its "shadow queues" (kind 1) are the debug queues (kind 2) k_shade filled
          const uint32_t none = TRACE_NONE;
          void* qargs[3] = {(void*)&q, (void*)&none, (void*)&ds};
          (void)hipLaunchKernel(p.bvh.wide8_nodes  ? trace_kernel_wide8(false, alpha, p.bvh.spill != nullptr)
                                : p.bvh.wide_nodes ? trace_kernel_wide(false, alpha, p.bvh.spill != nullptr)
                                                   : trace_kernel(false, alpha, p.bvh.spill != nullptr, p.bvh.top_count != 0),
                                dim3(tgrid), dim3(STHIP_BLOCK), qargs, lds, st);
          if (p.bvh.spill) {
            const uint32_t dgrid = (uint32_t)ctx->cu_count * 8u;
            if (alpha)
              hipLaunchKernelGGL((k_trace_deep<false, true>), dim3(dgrid), dim3(STHIP_BLOCK), 0, st, q);
            else
              hipLaunchKernelGGL((k_trace_deep<false, false>), dim3(dgrid), dim3(STHIP_BLOCK), 0, st, q);
          }
        }
      });
    };
    // the first bounce as wave packets (k_trace_primary): one 8x8 pixel block per wave
    auto trace_primary = [&]() -> int {
      launches_primary++;
      rays_primary += (uint64_t)primary_rays * in_flight;
      const uint32_t packets = (p.path_count + 63) / 64;
      const unsigned pgrid = std::max(1u, std::min((packets + 3) / 4, (uint32_t)ctx->cu_count * 64u));
      const size_t plds = ((size_t)ctx->bvh.stack_depth * (STHIP_BLOCK / 64) + 13 * STHIP_BLOCK) * sizeof(uint32_t);  // the per-wave stacks + every lane's saved world-space ray constants
      return timed(ms_primary, [&]() {
        if (p.bvh.alpha_test) {
          if (ctx->count_traversal)
            hipLaunchKernelGGL((k_trace_primary<true, true>), dim3(pgrid), dim3(STHIP_BLOCK), plds, st, p);
          else
            hipLaunchKernelGGL((k_trace_primary<false, true>), dim3(pgrid), dim3(STHIP_BLOCK), plds, st, p);
        } else if (ctx->count_traversal)
          hipLaunchKernelGGL((k_trace_primary<true, false>), dim3(pgrid), dim3(STHIP_BLOCK), plds, st, p);
        else
          hipLaunchKernelGGL((k_trace_primary<false, false>), dim3(pgrid), dim3(STHIP_BLOCK), plds, st, p);
      });
    };
    // Round r traces the paths entering bounce r together with the shadow rays bounce r - 1 produced (one launch,
    // k_trace), then shades bounce r; a last launch traces the shadow rays of the last bounce. `shade(depth)` is the
    // pass's shading kernel; the light pass (sample_photons) has visibility rays to the camera in place of NEE rays.
    auto run_rounds = [&](bool light, bool shadow_rays, auto&& shade) -> int {
      for (uint32_t depth = 0; depth <= max_bounce_rounds; depth++) {
        const uint32_t dc = depth < max_bounce_rounds ? depth : TRACE_NONE;
        const uint32_t ds = depth >= 1 && shadow_rays && depth - 1 <= max_shadow_round ? depth - 1 : TRACE_NONE;
        int r;
        if (!light && depth == 0 && dc == 0 && ctx->packet_primary && !ctx->has_volumes) {
          r = trace_primary();
        } else if (ctx->fuse_trace) {
          r = trace(dc, ds);
        } else {  // analysis: the two ray kinds in launches of their own
          r = trace(TRACE_NONE, ds);
          if (!r) r = trace(dc, TRACE_NONE);
        }
        if (r) return r;
        if (media && ds != TRACE_NONE) {
          r = timed(ms_shade, [&]() { hipLaunchKernelGGL(k_shadow_media, dim3(grid), dim3(STHIP_BLOCK), 0, st, p, ds); });
          if (r) return r;
        }
        if (dc == TRACE_NONE) break;
        r = timed(ms_shade, [&]() { shade(depth); });
        if (r) return r;
      }
      if (media && shadow_rays)  // the shadow rays still walking after the last bounce
        for (uint32_t ds = max_bounce_rounds; ds < max_bounce_rounds + drain_rounds; ds++) {
          int r = trace(TRACE_NONE, ds);
          if (!r) r = timed(ms_shade, [&]() { hipLaunchKernelGGL(k_shadow_media, dim3(grid), dim3(STHIP_BLOCK), 0, st, p, ds); });
          if (r) return r;
        }
      return STHIP_OK;
    };

    if (connect_paths) {  // BDPT.cpp:655-659; `conn` holds no pending entries when a pass starts
      HIP_TRY(ctx, hipMemsetAsync(ctx->light_vertices.p, 0, std::max<size_t>(1, vertices_per_seed * in_flight) * 64, st));
      if (lvc) HIP_TRY(ctx, hipMemsetAsync(ctx->lvc_staging.p, 0, (size_t)in_flight * pc->gLightPathCount * (pc->gMaxDiffuseVertices - 1) * 64, st));
      if (p.conn) HIP_TRY(ctx, hipMemsetAsync(ctx->conn.p, 0, (size_t)in_flight * p.paths_per_seed * conn_per_path * 16, st));
    }
    if (light_tracing) {  // sample_photons before the view paths, BDPT.cpp:653-667
      if (connect_views) HIP_TRY(ctx, hipMemsetAsync(ctx->light_trace.p, 0, (size_t)in_flight * W * H * 16, st));
      p.light_pass = 1;
      p.path_count = in_flight * light_threads;
      rc = reset_queues();
      if (rc) return rc;
      const uint32_t lgrid = grid_for(ctx, p.path_count);
      rc = timed(ms_other, [&]() {
        if (ctx->textured)
          hipLaunchKernelGGL((k_generate_light<true, true>), dim3(lgrid), dim3(STHIP_BLOCK), 0, st, p);
        else
          hipLaunchKernelGGL((k_generate_light<false, true>), dim3(lgrid), dim3(STHIP_BLOCK), 0, st, p);
      });
      if (rc) return rc;
      rc = run_rounds(true, !media, [&](uint32_t depth) {  // (media: connect_view walks its ray itself: nothing is queued)
        if (media && ctx->textured)
          hipLaunchKernelGGL((k_shade_light<true, true, true>), dim3(grid), dim3(STHIP_BLOCK), 0, st, p, depth);
        else if (media)
          hipLaunchKernelGGL((k_shade_light<false, true, true>), dim3(grid), dim3(STHIP_BLOCK), 0, st, p, depth);
        else if (ctx->textured)
          hipLaunchKernelGGL((k_shade_light<true, true>), dim3(grid), dim3(STHIP_BLOCK), 0, st, p, depth);
        else
          hipLaunchKernelGGL((k_shade_light<false, true>), dim3(grid), dim3(STHIP_BLOCK), 0, st, p, depth);
      });
      if (rc) return rc;
      rc = timed(ms_other, [&]() { hipLaunchKernelGGL(k_count_rays, dim3(1), dim3(1), 0, st, p); });
      if (rc) return rc;
      p.light_pass = 0;
    }
    if (lvc) {  // the cache in its defined order: compact the staged vertices of every seed in flight (lvc.hip)
      const uint32_t slots_per_seed = pc->gLightPathCount * (pc->gMaxDiffuseVertices - 1);
      size_t tmp_bytes = ctx->lvc_tmp.n;
      if (light_tracing)
        HIP_TRY(ctx, sthip::lvc_compact(ctx->lvc_staging.p, slots_per_seed, in_flight, (uint32_t)vertices_per_seed, ctx->light_vertices.p, ctx->lvc_count.p, ctx->lvc_flags.p, ctx->lvc_offsets.p,
                                        ctx->lvc_tmp.p, tmp_bytes, st));
      else
        HIP_TRY(ctx, hipMemsetAsync(ctx->lvc_count.p, 0, (size_t)in_flight * 4, st));
    }

    if (nee_reuse) HIP_TRY(ctx, hipMemsetAsync(ctx->hg_appends.p, 0, hg_slots * 64, st));
    if (lvc_reuse) HIP_TRY(ctx, hipMemsetAsync(ctx->lg_appends.p, 0, hg_slots * 96, st));
    p.path_count = in_flight * p.paths_per_seed;
    rc = reset_queues();
    if (rc) return rc;
    rc = timed(ms_other, [&]() { hipLaunchKernelGGL(k_generate, dim3(grid), dim3(STHIP_BLOCK), 0, st, p); });
    if (rc) return rc;
    if (debug_mode == STHIP_DEBUG_ENVIRONMENT_SAMPLE_TEST || debug_mode == STHIP_DEBUG_ENVIRONMENT_SAMPLE_PDF) {
      // bdpt.hlsl:190-205: sample_visibility returns before it traces anything; the frame stays (0, 0, 0, 1), no ray is counted
      // (the G-buffer outputs are not written upstream either: they are left zero here)
      if (s == 0) {
        if (p.out_albedo) HIP_TRY(ctx, hipMemsetAsync(p.out_albedo, 0, pixels * 16, st));
        if (p.out_visibility) HIP_TRY(ctx, hipMemsetAsync(p.out_visibility, 0, pixels * 8, st));
        if (p.out_depth) HIP_TRY(ctx, hipMemsetAsync(p.out_depth, 0, pixels * 16, st));
        if (p.out_prev_uv) HIP_TRY(ctx, hipMemsetAsync(p.out_prev_uv, 0, pixels * 8, st));
      }
      rc = timed(ms_other, [&]() {
        hipLaunchKernelGGL(k_debug_environment, dim3(grid), dim3(STHIP_BLOCK), 0, st, p);
        hipLaunchKernelGGL(k_resolve, dim3(grid), dim3(STHIP_BLOCK), 0, st, p, s == 0 ? 1u : 0u, s + in_flight == seed_count ? 1u : 0u, 0u);
      });
      if (rc) return rc;
      rays_primary = 0;
      continue;
    }
    if (presample) {
      const unsigned pgrid = (unsigned)((presample_n * in_flight + STHIP_BLOCK - 1) / STHIP_BLOCK);
      const bool pext = ctx->has_spheres || has_env;
      rc = timed(ms_other, [&]() {
        if (ctx->textured && pext)
          hipLaunchKernelGGL((k_presample_lights<true, true>), dim3(pgrid), dim3(STHIP_BLOCK), 0, st, p);
        else if (ctx->textured)
          hipLaunchKernelGGL((k_presample_lights<true, false>), dim3(pgrid), dim3(STHIP_BLOCK), 0, st, p);
        else if (pext)
          hipLaunchKernelGGL((k_presample_lights<false, true>), dim3(pgrid), dim3(STHIP_BLOCK), 0, st, p);
        else
          hipLaunchKernelGGL((k_presample_lights<false, false>), dim3(pgrid), dim3(STHIP_BLOCK), 0, st, p);
      });
      if (rc) return rc;
    }
    p.emitter_count = ctx->answer_last_rays ? ctx->emitter_count : 0u;  // (only the plain k_shade instantiation looks at it)
    rc = run_rounds(false, (nee || connect_paths) && !inline_media, [&](uint32_t depth) {  // (inline walks through media: nothing is queued)
      // eCoherentRR: a vertex shaded in round `depth` has path_length depth + 2; the roulette runs for
      // gMinPathVertices <= path_length < gMaxPathVertices at a non-specular vertex that is within the diffuse budget —
      // without specular materials that is vertex number depth + 1 of at most gMaxDiffuseVertices. In such a round the
      // paths first report their p (k_shade<PROBE>), the 8x4 groups agree (k_rr_reduce), then the round proper runs.
      // A probe: the round's k_shade without any output, up to the statement `kind` names (FrameParams::probe_kind)
      auto launch_probe = [&](uint32_t kind) {
        FrameParams probe = p;
        probe.probe_kind = kind;
        probe.out_albedo = nullptr;
        probe.out_visibility = nullptr;
        probe.out_depth = nullptr;
        probe.out_prev_uv = nullptr;
        if (bdpt) {
          if (ctx->textured)
            hipLaunchKernelGGL((k_shade<true, true, true, false, true>), dim3(grid), dim3(STHIP_BLOCK), 0, st, probe, depth);
          else
            hipLaunchKernelGGL((k_shade<false, true, true, false, true>), dim3(grid), dim3(STHIP_BLOCK), shade_lds, st, probe, depth);
        } else if (ctx->textured)
          hipLaunchKernelGGL((k_shade<true, true, false, false, true>), dim3(grid), dim3(STHIP_BLOCK), 0, st, probe, depth);
        else
          hipLaunchKernelGGL((k_shade<false, true, false, false, true>), dim3(grid), dim3(STHIP_BLOCK), shade_lds, st, probe, depth);
      };
      const unsigned reduce_grid = (unsigned)((p.path_count + STHIP_BLOCK - 1) / STHIP_BLOCK);
      p.rr = nullptr;
      if (coherent_rr && depth + 2 >= pc->gMinPathVertices && depth + 2 < pc->gMaxPathVertices && (ctx->has_specular || depth + 1 <= pc->gMaxDiffuseVertices)) {
        p.rr = ctx->rr.p;
        (void)hipMemsetAsync(ctx->rr.p, 0, (size_t)p.path_count * 16, st);
        launch_probe(1);
        hipLaunchKernelGGL(k_rr_reduce, dim3(reduce_grid), dim3(STHIP_BLOCK), 0, st, p);
      }
      // eCoherentSampling: the NEE index first (with the roulette's verdict known), then connect_lvc's (with the NEE index
      // known: how many numbers a path draws in between depends on the candidates it looked at)
      p.cs_nee = nullptr;
      p.cs_lvc = nullptr;
      if (coherent_nee) {
        p.cs_nee = ctx->cs_nee.p;
        (void)hipMemsetAsync(ctx->cs_nee.p, 0, (size_t)p.path_count * 8, st);
        launch_probe(2);
        hipLaunchKernelGGL(k_cs_reduce, dim3(reduce_grid), dim3(STHIP_BLOCK), 0, st, p.cs_nee, p.path_count);
      }
      if (coherent_lvc) {
        p.cs_lvc = ctx->cs_lvc.p;
        (void)hipMemsetAsync(ctx->cs_lvc.p, 0, (size_t)p.path_count * 8, st);
        launch_probe(3);
        hipLaunchKernelGGL(k_cs_reduce, dim3(reduce_grid), dim3(STHIP_BLOCK), 0, st, p.cs_lvc, p.path_count);
      }
      if (debug_mode) {  // BDPTDebugMode: the general instantiation with the statements that feed gDebugImage
        if (media && bdpt && inline_media)
          hipLaunchKernelGGL((k_shade<true, true, true, 2, false, true>), dim3(grid), dim3(STHIP_BLOCK), 0, st, p, depth);
        else if (media && bdpt)
          hipLaunchKernelGGL((k_shade<true, true, true, 1, false, true>), dim3(grid), dim3(STHIP_BLOCK), 0, st, p, depth);
        else if (media && inline_media)
          hipLaunchKernelGGL((k_shade<true, true, false, 2, false, true>), dim3(grid), dim3(STHIP_BLOCK), 0, st, p, depth);
        else if (media)
          hipLaunchKernelGGL((k_shade<true, true, false, 1, false, true>), dim3(grid), dim3(STHIP_BLOCK), 0, st, p, depth);
        else if (bdpt)
          hipLaunchKernelGGL((k_shade<true, true, true, false, false, true>), dim3(grid), dim3(STHIP_BLOCK), 0, st, p, depth);
        else
          hipLaunchKernelGGL((k_shade<true, true, false, false, false, true>), dim3(grid), dim3(STHIP_BLOCK), 0, st, p, depth);
      } else if (media && bdpt) {  // light tracing through media: the view paths carry the BDPT quantities
        if (ctx->textured && inline_media)
          hipLaunchKernelGGL((k_shade<true, true, true, 2>), dim3(grid), dim3(STHIP_BLOCK), 0, st, p, depth);
        else if (ctx->textured)
          hipLaunchKernelGGL((k_shade<true, true, true, 1>), dim3(grid), dim3(STHIP_BLOCK), 0, st, p, depth);
        else if (inline_media)
          hipLaunchKernelGGL((k_shade<false, true, true, 2>), dim3(grid), dim3(STHIP_BLOCK), shade_lds, st, p, depth);
        else
          hipLaunchKernelGGL((k_shade<false, true, true, 1>), dim3(grid), dim3(STHIP_BLOCK), shade_lds, st, p, depth);
      } else if (media && inline_media) {
        if (ctx->textured)
          hipLaunchKernelGGL((k_shade<true, true, false, 2>), dim3(grid), dim3(STHIP_BLOCK), 0, st, p, depth);
        else
          hipLaunchKernelGGL((k_shade<false, true, false, 2>), dim3(grid), dim3(STHIP_BLOCK), shade_lds, st, p, depth);
      } else if (media) {
        if (ctx->textured)
          hipLaunchKernelGGL((k_shade<true, true, false, 1>), dim3(grid), dim3(STHIP_BLOCK), 0, st, p, depth);
        else
          hipLaunchKernelGGL((k_shade<false, true, false, 1>), dim3(grid), dim3(STHIP_BLOCK), shade_lds, st, p, depth);
      } else if (bdpt) {
        if (ctx->textured)
          hipLaunchKernelGGL((k_shade<true, true, true>), dim3(grid), dim3(STHIP_BLOCK), 0, st, p, depth);
        else
          hipLaunchKernelGGL((k_shade<false, true, true>), dim3(grid), dim3(STHIP_BLOCK), shade_lds, st, p, depth);
      } else if (ctx->textured && ext)
        hipLaunchKernelGGL((k_shade<true, true>), dim3(grid), dim3(STHIP_BLOCK), 0, st, p, depth);
      else if (ctx->textured)
        hipLaunchKernelGGL((k_shade<true, false>), dim3(grid), dim3(STHIP_BLOCK), 0, st, p, depth);
      else {
        // untextured scenes, no light subpaths, no media. Where the path or diffuse budget can end at this round's vertex, only
        // the paths that still have something to do reach k_shade (k_cull_terminal).
        FrameParams pk = p;
        if (ctx->cull_terminal && depth >= 1 && !p.rr && !p.cs_nee && !p.cs_lvc && (depth + 2 >= pc->gMaxPathVertices || depth + 1 > pc->gMaxDiffuseVertices)) {
          // a block keeps what it meets in LDS: as many blocks per segment as it takes for a segment's share to fit (16 KB at 1080p)
          uint32_t per_segment = CULL_BLOCKS_PER_SEGMENT, per_block;
          for (;; per_segment *= 2) {
            per_block = (uint32_t)((((size_t)p.seg_stride + (size_t)per_segment * STHIP_BLOCK - 1) / ((size_t)per_segment * STHIP_BLOCK)) * STHIP_BLOCK);  // entries a block can meet
            if ((size_t)(per_block + 2) * 4 <= 48 * 1024) break;
          }
          hipLaunchKernelGGL(k_cull_terminal, dim3(QUEUE_SEGMENTS * per_segment), dim3(STHIP_BLOCK), (size_t)(per_block + 2) * 4, st, p, depth, ctx->queue_kept.p, per_block);
          pk.queue[depth & 1u] = ctx->queue_kept.p;
          pk.culled = 1;
        }
        if (ext)
          hipLaunchKernelGGL((k_shade<false, true>), dim3(grid), dim3(STHIP_BLOCK), shade_lds, st, pk, depth);
        else
          hipLaunchKernelGGL((k_shade<false, false>), dim3(grid), dim3(STHIP_BLOCK), shade_lds, st, pk, depth);
      }
    });
    if (rc) return rc;
    rc = timed(ms_other, [&]() { hipLaunchKernelGGL(k_resolve, dim3(grid), dim3(STHIP_BLOCK), 0, st, p, s == 0 ? 1u : 0u, s + in_flight == seed_count ? 1u : 0u, primary_rays * in_flight); });
    if (rc) return rc;
    if ((nee_reuse || lvc_reuse) && (s + in_flight < seed_count || ctx->reuse_persist)) {
      // This seed's appends become the grids the next seed looks up (hashgrid.h)
      if (nee_reuse) {
        const int rc2 = build_hash_grid(ctx, st, ctx->hg_appends.p, ctx->hg_compact.p, ctx->hg_data.p, hg_slots, 4, 2, false, pc->gHashGridBucketCount, ctx->hg_checksums, ctx->hg_counters, ctx->hg_indices);
        if (rc2) return rc2;
      }
      if (lvc_reuse) {
        const int rc2 = build_hash_grid(ctx, st, ctx->lg_appends.p, ctx->lg_compact.p, ctx->lg_data.p, hg_slots, 6, 0, true, pc->gHashGridBucketCount, ctx->lg_checksums, ctx->lg_counters, ctx->lg_indices);
        if (rc2) return rc2;
      }
      p.hg_prev = 1;
    }
  }
  if ((nee_reuse || lvc_reuse) && ctx->reuse_persist) {  // what the next call's first seed may look into
    memcpy(ctx->reuse_key, reuse_key, sizeof reuse_key);
    ctx->reuse_grids_valid = true;
  }

  if (!dev) {
    HIP_TRY(ctx, hipMemcpyAsync(out->gRadiance, p.out_radiance, radiance_entries * 16, hipMemcpyDeviceToHost, st));
    if (out->gAlbedo) HIP_TRY(ctx, hipMemcpyAsync(out->gAlbedo, p.out_albedo, pixels * 16, hipMemcpyDeviceToHost, st));
    if (out->gVisibility) HIP_TRY(ctx, hipMemcpyAsync(out->gVisibility, p.out_visibility, pixels * 8, hipMemcpyDeviceToHost, st));
    if (out->gDepth) HIP_TRY(ctx, hipMemcpyAsync(out->gDepth, p.out_depth, pixels * 16, hipMemcpyDeviceToHost, st));
    if (out->gPrevUVs) HIP_TRY(ctx, hipMemcpyAsync(out->gPrevUVs, p.out_prev_uv, pixels * 8, hipMemcpyDeviceToHost, st));
    if (debug_mode) HIP_TRY(ctx, hipMemcpyAsync(out->gDebugImage, p.out_debug, pixels * 16, hipMemcpyDeviceToHost, st));
    unsigned long long c[CNT_TOTAL];
    HIP_TRY(ctx, hipMemcpyAsync(c, ctx->counters.p, sizeof(c), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (out->gRayCount) {
      out->gRayCount[0] = c[CNT_RAYS_CLOSEST] + c[CNT_RAYS_SHADOW];
      out->gRayCount[1] = c[CNT_RAYS_CLOSEST] - c[CNT_CROSSINGS];
    }
    fill_counter_stats(ctx, c);
    ctx->stats_pending = false;
  } else {
    if (out->gRayCount) hipLaunchKernelGGL(k_write_ray_count, dim3(1), dim3(1), 0, st, ctx->counters.p, reinterpret_cast<unsigned long long*>(out->gRayCount));
    ctx->stats_pending = true;
  }
  if (timing) {
    ctx->stats.ms_trace = ms_trace;
    ctx->stats.ms_shade = ms_shade;
    ctx->stats.ms_total = ms_trace + ms_primary + ms_shade + ms_other;
    ctx->stats.launches_trace = launches_trace;
    ctx->stats.ms_trace_primary = ms_primary;
    ctx->stats.launches_primary = launches_primary;
  }
  ctx->stats.rays_primary_packets = rays_primary;
  return STHIP_OK;
}

// ---- multi-GPU assembly: the packed tiles of every shard -> the frame ----

uint32_t sthip_shard_slot_count(uint32_t width, uint32_t height, uint32_t shard_rank, uint32_t shard_count, uint32_t tile_w, uint32_t tile_h) {
  if (!width || !height || !shard_count || !tile_w || !tile_h || shard_rank >= shard_count) return 0;
  const uint32_t tiles = ((width + tile_w - 1) / tile_w) * ((height + tile_h - 1) / tile_h);
  const uint32_t owned = tiles > shard_rank ? (tiles - shard_rank + shard_count - 1) / shard_count : 0;
  return owned * tile_w * tile_h;
}

int sthip_assemble_tiles_bytes(sthip_ctx* ctx, const void* packed, uint64_t rank_stride, uint32_t shard_count, uint32_t tile_w, uint32_t tile_h, uint32_t width, uint32_t height,
                               uint32_t entry_bytes, void* frame) {
  if (!ctx) return STHIP_ERR_INVALID_ARGUMENT;
  if (!packed || !frame || !shard_count || !width || !height) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "sthip_assemble_tiles: a required argument is NULL/zero");
  if (tile_w == 0 || tile_h == 0 || (tile_w & 7) || (tile_h & 7)) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "sthip_assemble_tiles: tile size must be a multiple of 8");
  if (entry_bytes == 0 || entry_bytes > 64 || (entry_bytes & 3)) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "sthip_assemble_tiles: entry size must be a multiple of 4 bytes, at most 64");
  const uint32_t slots = sthip_shard_slot_count(width, height, 0, shard_count, tile_w, tile_h);  // rank 0 owns the most tiles
  if (rank_stride < slots) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "sthip_assemble_tiles: rank_stride is smaller than a shard's slot count");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t n = (size_t)shard_count * slots;
  hipLaunchKernelGGL(k_assemble_tiles, dim3((unsigned)((n + STHIP_BLOCK - 1) / STHIP_BLOCK)), dim3(STHIP_BLOCK), 0, ctx->stream, reinterpret_cast<const uint32_t*>(packed), (size_t)rank_stride, shard_count,
                     slots, tile_w, tile_h, width, height, entry_bytes / 4, reinterpret_cast<uint32_t*>(frame));
  HIP_TRY(ctx, hipGetLastError());
  return STHIP_OK;
}
int sthip_assemble_tiles(sthip_ctx* ctx, const float* packed, uint64_t rank_stride, uint32_t shard_count, uint32_t tile_w, uint32_t tile_h, uint32_t width, uint32_t height,
                         float* frame) {
  return sthip_assemble_tiles_bytes(ctx, packed, rank_stride, shard_count, tile_w, tile_h, width, height, 16, frame);
}
int sthip_radiance_to_sums(sthip_ctx* ctx, float* image, uint64_t entries, uint32_t back) {
  if (!ctx) return STHIP_ERR_INVALID_ARGUMENT;
  if (!image) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "sthip_radiance_to_sums: image is NULL");
  if (entries == 0) return STHIP_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_radiance_sums, dim3(grid_for(ctx, (size_t)entries)), dim3(STHIP_BLOCK), 0, ctx->stream, reinterpret_cast<float4*>(image), (size_t)entries, back ? 0u : 1u);
  HIP_TRY(ctx, hipGetLastError());
  return STHIP_OK;
}

int sthip_pack_tiles(sthip_ctx* ctx, const void* image, uint32_t width, uint32_t height, uint32_t entry_bytes, void* packed) {
  if (!ctx) return STHIP_ERR_INVALID_ARGUMENT;
  if (!image || !packed || !width || !height) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "sthip_pack_tiles: a required argument is NULL/zero");
  if (entry_bytes == 0 || entry_bytes > 64 || (entry_bytes & 3)) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "sthip_pack_tiles: entry size must be a multiple of 4 bytes, at most 64");
  const uint32_t slots = sthip_shard_slot_count(width, height, ctx->shard_rank, ctx->shard_count, ctx->tile_w, ctx->tile_h);
  if (slots == 0) return STHIP_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_pack_tiles, dim3((unsigned)((slots + STHIP_BLOCK - 1) / STHIP_BLOCK)), dim3(STHIP_BLOCK), 0, ctx->stream, reinterpret_cast<const uint32_t*>(image), ctx->shard_rank, ctx->shard_count,
                     slots, ctx->tile_w, ctx->tile_h, width, height, entry_bytes / 4, reinterpret_cast<uint32_t*>(packed));
  HIP_TRY(ctx, hipGetLastError());
  return STHIP_OK;
}

// ---- after the path: temporal accumulation, tonemap and image metric (post.h) ----

int sthip_accumulate(sthip_ctx* ctx, const sthip_accumulate_desc* d) {
  if (!ctx || !d) return STHIP_ERR_INVALID_ARGUMENT;
  if (!d->gViews || !d->view_count || !d->gRadiance || !d->gPrevAccumColor || !d->gPrevAccumMoments || !d->gAccumColor || !d->gAccumMoments || !d->width || !d->height)
    return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "sthip_accumulate: gViews, gRadiance, gPrevAccum*, gAccum* and a non-empty extent are required");
  if (d->demodulate_albedo && !d->gAlbedo) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "sthip_accumulate: gDemodulateAlbedo needs gAlbedo");
  if (d->reprojection && (!d->gVisibility || !d->gDepth || !d->gPrevUVs || !d->gPrevVisibility || !d->gPrevDepth))
    return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "sthip_accumulate: gReprojection needs gVisibility, gDepth, gPrevUVs, gPrevVisibility, gPrevDepth");
  if ((uint64_t)d->width * d->height > 0x7FFFFFFFull) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "sthip_accumulate: extent too large");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t n = (size_t)d->width * d->height;
  std::vector<void*> staged;  // device copies of host inputs, freed on return
  auto release = [&]() {
    for (void* q : staged) (void)hipFree(q);
  };
  bool bad = false;
  auto in = [&](const void* host, size_t bytes) -> const void* {
    if (!host || d->device_ptrs) return host;
    void* q = nullptr;
    if (hipMalloc(&q, bytes) != hipSuccess || hipMemcpyAsync(q, host, bytes, hipMemcpyHostToDevice, st) != hipSuccess) {
      bad = true;
      return nullptr;
    }
    staged.push_back(q);
    return q;
  };
  AccumulateParams p{};
  p.width = d->width;
  p.height = d->height;
  p.view_count = d->view_count;
  p.reprojection = d->reprojection;
  p.demodulate_albedo = d->demodulate_albedo;
  p.history_limit = d->history_limit;
  p.instance_count = d->instance_count;
  if (d->view_count <= ACCUMULATE_INLINE_VIEWS) {  // gViews is a host array in either mode: a few views ride in the kernel's arguments
    p.views = nullptr;
    memcpy(p.inline_views, d->gViews, (size_t)d->view_count * sizeof(sthip_ViewData));
  } else {
    void* q = nullptr;
    if (hipMalloc(&q, (size_t)d->view_count * sizeof(sthip_ViewData)) != hipSuccess ||
        hipMemcpyAsync(q, d->gViews, (size_t)d->view_count * sizeof(sthip_ViewData), hipMemcpyHostToDevice, st) != hipSuccess)
      bad = true;
    else
      staged.push_back(q);
    p.views = (const sthip_ViewData*)q;
  }
  p.radiance = (const float4*)in(d->gRadiance, n * 16);
  p.albedo = (const float4*)in(d->gAlbedo, n * 16);
  p.visibility = (const sthip_VisibilityInfo*)in(d->gVisibility, n * 8);
  p.depth = (const sthip_DepthInfo*)in(d->gDepth, n * 16);
  p.prev_uvs = (const float2*)in(d->gPrevUVs, n * 8);
  p.prev_visibility = (const sthip_VisibilityInfo*)in(d->gPrevVisibility, n * 8);
  p.prev_depth = (const sthip_DepthInfo*)in(d->gPrevDepth, n * 16);
  p.prev_accum_color = (const float4*)in(d->gPrevAccumColor, n * 16);
  p.prev_accum_moments = (const float2*)in(d->gPrevAccumMoments, n * 8);
  p.instance_index_map = (const uint32_t*)in(d->gInstanceIndexMap, (size_t)d->instance_count * 4);
  float4* out_c = reinterpret_cast<float4*>(d->gAccumColor);
  float2* out_m = reinterpret_cast<float2*>(d->gAccumMoments);
  if (!d->device_ptrs) {
    void *qc = nullptr, *qm = nullptr;
    if (hipMalloc(&qc, n * 16) != hipSuccess || hipMalloc(&qm, n * 8) != hipSuccess) bad = true;
    if (qc) staged.push_back(qc);
    if (qm) staged.push_back(qm);
    out_c = (float4*)qc;
    out_m = (float2*)qm;
    // pixels outside every view keep what the caller's buffers hold
    if (!bad && (hipMemcpyAsync(qc, d->gAccumColor, n * 16, hipMemcpyHostToDevice, st) != hipSuccess || hipMemcpyAsync(qm, d->gAccumMoments, n * 8, hipMemcpyHostToDevice, st) != hipSuccess)) bad = true;
  }
  if (bad) {
    (void)hipStreamSynchronize(st);
    release();
    return fail(ctx, STHIP_ERR_HIP, "sthip_accumulate: staging the host images on the device failed");
  }
  p.accum_color = out_c;
  p.accum_moments = out_m;
  hipLaunchKernelGGL(k_accumulate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && !d->device_ptrs) {
    e = hipMemcpyAsync(d->gAccumColor, out_c, n * 16, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d->gAccumMoments, out_m, n * 8, hipMemcpyDeviceToHost, st);
  }
  // staged copies must outlive the kernel; with device pointers and inline views nothing is staged: the call is only enqueued
  const hipError_t es = staged.empty() ? hipSuccess : hipStreamSynchronize(st);
  release();
  if (e != hipSuccess || es != hipSuccess) return fail(ctx, STHIP_ERR_HIP, std::string("sthip_accumulate: ") + hipGetErrorString(e != hipSuccess ? e : es));
  return STHIP_OK;
}

int sthip_tonemap(sthip_ctx* ctx, const sthip_tonemap_desc* d) {
  if (!ctx || !d) return STHIP_ERR_INVALID_ARGUMENT;
  if (!d->gInput || !d->gOutput || d->width == 0 || d->height == 0) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "sthip_tonemap: gInput, gOutput and a non-empty extent are required");
  if (d->mode >= eTonemapModeCount) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "sthip_tonemap: unknown mode " + std::to_string(d->mode));
  if (d->modulate_albedo && !d->gAlbedo) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "sthip_tonemap: gModulateAlbedo needs gAlbedo");
  if ((uint64_t)d->width * d->height > 0xFFFFFFFFull) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "sthip_tonemap: extent too large");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const uint32_t n = d->width * d->height;
  const float4 *in = reinterpret_cast<const float4*>(d->gInput), *alb = reinterpret_cast<const float4*>(d->gAlbedo);
  float4* out = reinterpret_cast<float4*>(d->gOutput);
  DevBuf<float4> bin, balb, bout;
  if (!d->device_ptrs) {
    HIP_TRY(ctx, bin.ensure(n));
    HIP_TRY(ctx, bout.ensure(n));
    HIP_TRY(ctx, hipMemcpyAsync(bin.p, d->gInput, (size_t)n * 16, hipMemcpyHostToDevice, st));
    in = bin.p;
    out = bout.p;
    if (d->modulate_albedo) {
      HIP_TRY(ctx, balb.ensure(n));
      HIP_TRY(ctx, hipMemcpyAsync(balb.p, d->gAlbedo, (size_t)n * 16, hipMemcpyHostToDevice, st));
      alb = balb.p;
    }
  }
  HIP_TRY(ctx, ctx->post_scratch.ensure(16));  // 4 quantised maxima, then the 6 floats of the exposure state
  HIP_TRY(ctx, hipMemsetAsync(ctx->post_scratch.p, 0, 16, st));
  TonemapState prev;
  for (int k = 0; k < 6; k++) prev.v[k] = d->exposure_state ? d->exposure_state[k] : 0.0f;
  float* state_out = d->exposure_state ? reinterpret_cast<float*>(ctx->post_scratch.p + 4) : nullptr;
  const uint32_t grid = (uint32_t)std::min<size_t>(((size_t)n + 255) / 256, (size_t)ctx->cu_count * 16);
  hipLaunchKernelGGL(k_tonemap_reduce_max, dim3(grid), dim3(256), 0, st, in, alb, n, d->modulate_albedo, ctx->post_scratch.p);
  hipLaunchKernelGGL(k_tonemap, dim3(grid), dim3(256), 0, st, in, alb, out, n, d->mode, d->modulate_albedo, d->gamma_correction, d->exposure, ctx->post_scratch.p,
                     d->exposure_state ? d->exposure_alpha : 0.0f, prev, state_out);
  HIP_TRY(ctx, hipGetLastError());
  if (!d->device_ptrs) HIP_TRY(ctx, hipMemcpyAsync(d->gOutput, bout.p, (size_t)n * 16, hipMemcpyDeviceToHost, st));
  if (d->out_max) {
    uint32_t m[4];
    HIP_TRY(ctx, hipMemcpyAsync(m, ctx->post_scratch.p, 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    for (int k = 0; k < 4; k++) d->out_max[k] = (float)m[k] / TONEMAP_MAX_QUANTIZATION;
  } else if (!d->device_ptrs) {
    HIP_TRY(ctx, hipStreamSynchronize(st));
  }
  if (d->exposure_state) {  // the state goes back to the caller: this form of the call synchronises
    HIP_TRY(ctx, hipMemcpyAsync(d->exposure_state, state_out, 24, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
  }
  return STHIP_OK;
}

int sthip_image_compare(sthip_ctx* ctx, const float* image1, const float* image2, uint32_t width, uint32_t height, uint32_t metric, uint32_t quantization, uint32_t device_ptrs,
                        uint32_t* sum_out, uint32_t* overflow_out) {
  if (!ctx) return STHIP_ERR_INVALID_ARGUMENT;
  if (!image1 || !image2 || !sum_out || width == 0 || height == 0) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "sthip_image_compare: two images, sum_out and a non-empty extent are required");
  if (metric > 2) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "sthip_image_compare: unknown metric " + std::to_string(metric));
  if ((uint64_t)width * height * 3 > 0xFFFFFFFFull) return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "sthip_image_compare: extent too large");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const uint32_t n = width * height;
  const float4 *a = reinterpret_cast<const float4*>(image1), *b = reinterpret_cast<const float4*>(image2);
  DevBuf<float4> ba, bb;
  if (!device_ptrs) {
    HIP_TRY(ctx, ba.ensure(n));
    HIP_TRY(ctx, bb.ensure(n));
    HIP_TRY(ctx, hipMemcpyAsync(ba.p, image1, (size_t)n * 16, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(bb.p, image2, (size_t)n * 16, hipMemcpyHostToDevice, st));
    a = ba.p;
    b = bb.p;
  }
  HIP_TRY(ctx, ctx->post_scratch.ensure(4));
  HIP_TRY(ctx, hipMemsetAsync(ctx->post_scratch.p, 0, 16, st));
  hipLaunchKernelGGL(k_image_compare, dim3((n + 63) / 64), dim3(64), 0, st, a, b, n, metric, quantization, ctx->post_scratch.p);
  HIP_TRY(ctx, hipGetLastError());
  uint32_t r[2];
  HIP_TRY(ctx, hipMemcpyAsync(r, ctx->post_scratch.p, 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  *sum_out = r[0];
  if (overflow_out) *overflow_out = r[1];
  return STHIP_OK;
}

// ---- measured ceilings for the roofline (ceilings.h) ----
int sthip_measure_ceiling(sthip_ctx* ctx, uint32_t kind, double* gbytes_per_s) {
  if (!ctx || !gbytes_per_s) return STHIP_ERR_INVALID_ARGUMENT;
  *gbytes_per_s = 0.0;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const uint32_t blocks = (uint32_t)ctx->cu_count * 8u;  // 2048 threads per CU: every wave slot
  float best_ms = 0.0f;
  double bytes = 0.0;
  if (kind == STHIP_CEILING_TRIAD) {
    const size_t n = (size_t)32 << 20;  // 3 arrays of 512 MiB: far beyond the 256 MiB Infinity Cache
    DevBuf<float4> a, b, c;
    HIP_TRY(ctx, a.ensure(n));
    HIP_TRY(ctx, b.ensure(n));
    HIP_TRY(ctx, c.ensure(n));
    HIP_TRY(ctx, hipMemsetAsync(a.p, 0, n * 16, st));
    HIP_TRY(ctx, hipMemsetAsync(b.p, 0, n * 16, st));
    bytes = 3.0 * 16.0 * (double)n;
    for (int rep = 0; rep < 4; rep++) {
      HIP_TRY(ctx, hipEventRecord(ctx->ev[0], st));
      hipLaunchKernelGGL(k_ceiling_triad, dim3(blocks), dim3(256), 0, st, a.p, b.p, c.p, 0.5f, n);
      HIP_TRY(ctx, hipEventRecord(ctx->ev[1], st));
      HIP_TRY(ctx, hipEventSynchronize(ctx->ev[1]));
      float ms = 0;
      HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
      if (rep > 0 && (best_ms == 0.0f || ms < best_ms)) best_ms = ms;
    }
  } else if (kind == STHIP_CEILING_NODE_GATHER_TABLE || kind == STHIP_CEILING_NODE_GATHER_L2 || kind == STHIP_CEILING_NODE_GATHER_L1) {
    if (!ctx->has_scene || !ctx->bvh_nodes) return fail(ctx, STHIP_ERR_NO_SCENE, "sthip_measure_ceiling: the node-gather ceilings read the resident acceleration structure: upload a scene first");
    // the nodes k_trace walks: the 48-byte binary ones (3 loads), the 64-byte 4-wide ones (4 loads) or the 80-byte 8-wide ones (5 loads)
    const bool wide8 = ctx->bvh.wide8_nodes != nullptr, wide = !wide8 && ctx->bvh.wide_nodes != nullptr;
    const uint32_t nb = wide8 ? (uint32_t)sizeof(Wide8Node) : wide ? (uint32_t)sizeof(WideNode) : BVH_NODE_BYTES;
    uint32_t count = (uint32_t)std::min<uint64_t>(wide8 ? ctx->wide8_node_count : wide ? ctx->wide_node_count : ctx->bvh_nodes, 0xFFFFFFFFull);
    if (kind == STHIP_CEILING_NODE_GATHER_L2) count = std::min<uint32_t>(count, (2u << 20) / nb);
    if (kind == STHIP_CEILING_NODE_GATHER_L1) count = std::min<uint32_t>(count, (16u << 10) / nb);
    const uint32_t iterations = 64;
    DevBuf<float> sink;
    HIP_TRY(ctx, sink.ensure((size_t)blocks * 256));
    bytes = (double)(wide8 ? sizeof(Wide8Node) : wide ? sizeof(WideNode) : sizeof(BvhNodePacked)) * (double)blocks * 256.0 * iterations * CEIL_UNROLL;
    for (int rep = 0; rep < 4; rep++) {
      HIP_TRY(ctx, hipEventRecord(ctx->ev[0], st));
      if (wide8)
        hipLaunchKernelGGL(k_ceiling_node_gather<5>, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const float4*>(ctx->wide8_nodes.p), count, nb, iterations, sink.p);
      else if (wide)
        hipLaunchKernelGGL(k_ceiling_node_gather<4>, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const float4*>(ctx->wide_nodes.p), count, nb, iterations, sink.p);
      else
        hipLaunchKernelGGL(k_ceiling_node_gather<3>, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const float4*>(ctx->nodes.p), count, nb, iterations, sink.p);
      HIP_TRY(ctx, hipEventRecord(ctx->ev[1], st));
      HIP_TRY(ctx, hipEventSynchronize(ctx->ev[1]));
      float ms = 0;
      HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
      if (rep > 0 && (best_ms == 0.0f || ms < best_ms)) best_ms = ms;
    }
  } else {
    return fail(ctx, STHIP_ERR_INVALID_ARGUMENT, "sthip_measure_ceiling: unknown kind");
  }
  HIP_TRY(ctx, hipGetLastError());
  if (best_ms > 0.0f) *gbytes_per_s = bytes / ((double)best_ms * 1e-3) / 1e9;
  return STHIP_OK;
}

}  // extern "C"
