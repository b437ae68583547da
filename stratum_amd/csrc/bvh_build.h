// bvh_build.h — host-side acceleration-structure build (see bvh.h for the layout)
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/sthip.h"
#include "bvh.h"

namespace sthip {

struct BuiltBvh {
  std::vector<BvhNode> nodes;
  std::vector<BvhTri> tris;
  std::vector<BvhTriUv> tri_uvs;  // parallel to tris when some material has an alpha mask, else empty
  std::vector<uint32_t> inst_alpha;  // per instance: gImage1s index of its material's alpha mask or BVH_NO_ALPHA
  std::vector<TlasEntry> entries;
  std::vector<DeviceVolume> volumes;  // parsed headers of gVolumes (first_word is filled by the uploader)
  uint32_t root_ref = BVH_INVALID_REF;  // inner-node index traversal starts at; BVH_INVALID_REF: empty scene
  uint32_t top_is_world_blas = 1;       // 1: root_ref is the merged world-space mesh, no top level
  uint32_t stack_depth = 4;             // upper bound of the traversal stack height
  float scene_center[3] = {0, 0, 0};
  float scene_radius = 0;
  float gpu_build_ms = 0;  // device time of the LBVH kernels (0 for the host builder)
};

enum BvhBuilderKind { BVH_BUILDER_SAH_HOST = 0, BVH_BUILDER_LBVH_GPU = 1 };

// Validates the scene arrays and builds. Returns false and sets `err` on malformed input.
// `builder`: binned SAH on the host (default; best traversal) or LBVH on the GPU (fastest build; lbvh.hip).
bool build_scene_bvh(const sthip_scene_desc& scene, BuiltBvh& out, std::string& err, int builder = BVH_BUILDER_SAH_HOST);

// GPU LBVH of one mesh (lbvh.hip): appends nodes and leaf-ordered triangles to the outputs.
bool lbvh_build_gpu(const std::vector<BvhTri>& tris_in, std::vector<BvhNode>& nodes_out, std::vector<BvhTri>& tris_out, uint32_t& root_ref, uint32_t& stack_need, float& gpu_ms,
                    std::string& err);

}  // namespace sthip
