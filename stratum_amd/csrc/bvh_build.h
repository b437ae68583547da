// bvh_build.h — host-side acceleration-structure build (see bvh.h for the layout)
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/sthip.h"
#include "bvh.h"

namespace sthip {

// What a transforms-only update needs from the last full build (kept by the uploader): the bottom levels stay as they
// are in HBM, the top level is rebuilt over new world boxes. The analogue of the reference's BLAS cache (Scene.cpp:435-459).
struct TopLevelState {
  std::vector<TlasEntry> entries;
  std::vector<float> obj_box;       // 6 per entry: object-space bounds of a transformed mesh (unused for the other kinds)
  std::vector<uint8_t> merged;      // per instance: part of the merged world-space mesh — its transform must stay the identity
  std::vector<DeviceVolume> volumes;
  uint32_t instance_count = 0;
  uint32_t blas_nodes = 0;          // nodes [0, blas_nodes) are bottom-level nodes; the top level follows
  uint32_t blas_depth = 0;
  float merged_box[6] = {0, 0, 0, 0, 0, 0};  // world box of the merged mesh's entry
  bool has_merged = false;
};

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
  TopLevelState top;
};

enum BvhBuilderKind { BVH_BUILDER_SAH_HOST = 0, BVH_BUILDER_LBVH_GPU = 1 };

// Validates the scene arrays and builds. Returns false and sets `err` on malformed input.
// `builder`: binned SAH on the host (default; best traversal) or LBVH on the GPU (fastest build; lbvh.hip).
bool build_scene_bvh(const sthip_scene_desc& scene, BuiltBvh& out, std::string& err, int builder = BVH_BUILDER_SAH_HOST);

// Transforms-only update: new entry matrices and world boxes, a new top level. `tlas_nodes` come out with their child
// references already offset by st.blas_nodes (they go to nodes[st.blas_nodes ...]). Fails (false) when an instance of the
// merged world-space mesh no longer has the identity transform: that needs a full build.
bool rebuild_top_level(TopLevelState& st, const sthip_TransformData* xf, const sthip_TransformData* inv, uint32_t instance_count, std::vector<BvhNode>& tlas_nodes, uint32_t& root_ref,
                       uint32_t& top_is_world_blas, uint32_t& stack_depth, float scene_center[3], float& scene_radius, std::string& err);

// GPU LBVH of one mesh (lbvh.hip): appends nodes and leaf-ordered triangles to the outputs.
bool lbvh_build_gpu(const std::vector<BvhTri>& tris_in, std::vector<BvhNode>& nodes_out, std::vector<BvhTri>& tris_out, uint32_t& root_ref, uint32_t& stack_need, float& gpu_ms,
                    std::string& err);

}  // namespace sthip
