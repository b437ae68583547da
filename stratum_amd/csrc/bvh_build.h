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
  std::vector<TlasEntry> entries;
  uint32_t root_ref = BVH_INVALID_REF;  // inner-node index traversal starts at; BVH_INVALID_REF: empty scene
  uint32_t top_is_world_blas = 1;       // 1: root_ref is the merged world-space mesh, no top level
  uint32_t stack_depth = 4;             // upper bound of the traversal stack height
  float scene_center[3] = {0, 0, 0};
  float scene_radius = 0;
};

// Validates the scene arrays and builds. Returns false and sets `err` on malformed input.
bool build_scene_bvh(const sthip_scene_desc& scene, BuiltBvh& out, std::string& err);

}  // namespace sthip
