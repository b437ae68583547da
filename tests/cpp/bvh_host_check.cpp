// bvh_host_check.cpp — the HOST side of the acceleration-structure build (stratum_amd/csrc/bvh_build.cpp: validation,
// threaded binned-SAH bottom levels, embedded leaves, top level, transforms-only rebuild, treetop selection, node packing)
// as a plain CPU program, so that it can run under AddressSanitizer / UndefinedBehaviorSanitizer / ThreadSanitizer (GPU
// sanitizers are not available on the pool). Reads the raw scene arrays tests/test_host_cpp.py dumps:
//   bvh_host_check <dir>   with <dir>/{vertices,indices,instances,xf,inv_xf,materials}.bin
// Checks structural invariants of what comes out and prints one line. The GPU builder's entry points are never reached
// with the SAH builder; they are defined here only to satisfy the linker.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "../../stratum_amd/csrc/bvh_build.h"

namespace sthip {
bool lbvh_build_gpu(const std::vector<BvhTri>&, std::vector<BvhNode>&, std::vector<BvhTri>&, uint32_t&, uint32_t&, float&, std::string& err) {
  err = "no device in this harness";
  return false;
}
bool lbvh_build_device(const DeviceBuildTarget&, const std::vector<MeshPiece>&, uint32_t, uint32_t, uint32_t&, uint32_t&, float*, float&, std::string& err, std::vector<FrontierEntry>*, uint32_t) {
  err = "no device in this harness";
  return false;
}
}  // namespace sthip

template <typename T>
static std::vector<T> slurp(const std::string& path) {
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) {
    std::fprintf(stderr, "cannot open %s\n", path.c_str());
    std::exit(2);
  }
  const size_t n = (size_t)f.tellg();
  std::vector<T> v(n / sizeof(T));
  f.seekg(0);
  f.read((char*)v.data(), v.size() * sizeof(T));
  return v;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string d = argv[1];
  auto vertices = slurp<sthip_PackedVertexData>(d + "/vertices.bin");
  auto indices = slurp<uint8_t>(d + "/indices.bin");
  auto instances = slurp<sthip_InstanceData>(d + "/instances.bin");
  auto xf = slurp<sthip_TransformData>(d + "/xf.bin");
  auto inv = slurp<sthip_TransformData>(d + "/inv_xf.bin");
  auto materials = slurp<uint8_t>(d + "/materials.bin");
  indices.resize(indices.size() + 8, 0);
  sthip_scene_desc s{};
  s.gVertices = vertices.data();
  s.vertex_count = (uint32_t)vertices.size();
  s.gIndices = indices.data();
  s.indices_bytes = indices.size() - 8;
  s.gInstances = instances.data();
  s.gInstanceTransforms = xf.data();
  s.gInstanceInverseTransforms = inv.data();
  s.instance_count = (uint32_t)instances.size();
  s.gMaterialData = materials.data();
  s.material_bytes = materials.size();
  size_t total_nodes = 0;
  for (int embed = 0; embed < 2; embed++) {
    sthip::BuiltBvh b;
    std::string err;
    if (!sthip::build_scene_bvh(s, b, err, sthip::BVH_BUILDER_SAH_HOST, nullptr, embed != 0)) {
      std::printf("BUILD FAILED: %s\n", err.c_str());
      return 1;
    }
    // every reference stays inside its array; every triangle is referenced exactly once
    std::vector<uint32_t> seen(b.tris.size(), 0);
    for (size_t i = 0; i < b.nodes.size(); i++) {
      if (b.embedded && b.unit_tri[i] != 0xFFFFFFFFu) continue;  // a triangle's unit
      for (int c = 0; c < 2; c++) {
        const uint32_t r = b.nodes[i].ref[c];
        if (r == BVH_INVALID_REF) continue;
        if (!(r & BVH_LEAF_BIT)) {
          if (r >= b.nodes.size()) return std::printf("FAIL: inner reference out of range\n"), 1;
        } else if (r & BVH_INST_BIT) {
          if ((r & 0xFFFFu) >= b.entries.size()) return std::printf("FAIL: entry reference out of range\n"), 1;
        } else {
          const uint32_t first = (r & 0x3FFFFFFFu) >> 2, count = (r & 3u) + 1;
          for (uint32_t k = 0; k < count; k++) {
            uint32_t t = first + k;
            if (b.embedded) {
              if (t >= b.unit_tri.size() || b.unit_tri[t] == 0xFFFFFFFFu) return std::printf("FAIL: leaf unit is not a triangle\n"), 1;
              t = b.unit_tri[t];
            }
            if (t >= b.tris.size()) return std::printf("FAIL: triangle reference out of range\n"), 1;
            seen[t]++;
          }
        }
      }
    }
    size_t wrapped = 0;
    for (uint32_t c : seen) {
      if (c == 0) return std::printf("FAIL: a triangle is not referenced\n"), 1;
      if (c > 1) wrapped++;  // a single-leaf mesh is wrapped with its leaf in both child slots
    }
    // the 4-wide form (build_wide_bvh): every decoded child box contains the box the binary tree holds for the same subtree
    // (checked through the leaves: a wide node's child must enclose every binary child box met on the way down to the
    // same leaf references), every reference stays in range, and the wide tree reaches exactly the leaf references the
    // binary tree reaches, each as often
    {
      sthip::build_wide_bvh(b);
      if (b.wide_nodes.empty() && !b.nodes.empty() && b.root_ref != BVH_INVALID_REF) return std::printf("FAIL: no wide nodes\n"), 1;
      std::vector<uint32_t> bin_leaves, wide_leaves;
      std::vector<uint32_t> todo;
      // (in the wide tree the top-level leaf of the merged world-space mesh is that mesh's root — build_wide_bvh splices it in —
      // so there the mesh is reached from the top level and its entry is not a root of its own, nor a leaf reference)
      auto roots_of = [&](const std::vector<TlasEntry>& entries, uint32_t root_ref, bool top_is_world, bool spliced) {
        std::vector<uint32_t> r;
        if (root_ref != BVH_INVALID_REF && !top_is_world) r.push_back(root_ref);
        for (const TlasEntry& e : entries)
          if ((e.identity == TLAS_ENTRY_IDENTITY && !(spliced && !top_is_world)) || e.identity == TLAS_ENTRY_TRANSFORMED) r.push_back(e.root);
        std::sort(r.begin(), r.end());
        r.erase(std::unique(r.begin(), r.end()), r.end());
        return r;
      };
      for (uint32_t root : roots_of(b.entries, b.root_ref, b.top_is_world_blas != 0, false)) {
        todo.assign(1, root);
        while (!todo.empty()) {
          const uint32_t i = todo.back();
          todo.pop_back();
          const uint32_t r0 = b.nodes[i].ref[0], r1 = b.nodes[i].ref[1];
          for (int c = 0; c < 2; c++) {
            const uint32_t r = c ? r1 : r0;
            if (r == BVH_INVALID_REF || (c == 1 && r1 == r0 && (r & BVH_LEAF_BIT))) continue;
            const bool merged_entry = (r & (BVH_LEAF_BIT | BVH_INST_BIT)) == (BVH_LEAF_BIT | BVH_INST_BIT) && b.entries[r & 0xFFFFu].identity == TLAS_ENTRY_IDENTITY;
            if (merged_entry) continue;  // spliced in the wide tree: no leaf reference there
            if (r & BVH_LEAF_BIT) bin_leaves.push_back(r);
            else todo.push_back(r);
          }
        }
      }
      size_t loose = 0;
      for (uint32_t root : roots_of(b.wide_entries, b.wide_root_ref, b.top_is_world_blas != 0, true)) {
        todo.assign(1, root);
        while (!todo.empty()) {
          const uint32_t i = todo.back();
          todo.pop_back();
          if (i >= b.wide_nodes.size()) return std::printf("FAIL: wide reference out of range\n"), 1;
          const WideNode& w = b.wide_nodes[i];
          for (int k = 0; k < (int)w.exp[3]; k++) {
            const uint32_t r = w.ref[k];
            for (int a = 0; a < 3; a++) {
              if ((int8_t)w.exp[a] < -126) return std::printf("FAIL: plane step is not a normal power of two\n"), 1;
              if (w.q[2 * a][k] > w.q[2 * a + 1][k]) loose++;
            }
            if (r & BVH_LEAF_BIT) wide_leaves.push_back(r);
            else todo.push_back(r);
          }
        }
      }
      if (loose) return std::printf("FAIL: %zu inverted child boxes on used slots\n", loose), 1;
      std::sort(bin_leaves.begin(), bin_leaves.end());
      std::sort(wide_leaves.begin(), wide_leaves.end());
      if (bin_leaves != wide_leaves) return std::printf("FAIL: the wide tree reaches %zu leaf references, the binary tree %zu (or other ones)\n", wide_leaves.size(), bin_leaves.size()), 1;
      // containment: the decoded box of a leaf child against the binary tree's box of the same leaf reference (a leaf
      // reference occurs once per tree, apart from wrapped lone leaves whose two boxes are equal)
      std::vector<std::pair<uint32_t, std::array<float, 6>>> bin_box;
      for (size_t i = 0; i < b.nodes.size(); i++) {
        if (b.embedded && b.unit_tri[i] != 0xFFFFFFFFu) continue;
        for (int c = 0; c < 2; c++) {
          const uint32_t r = b.nodes[i].ref[c];
          if (r == BVH_INVALID_REF || !(r & BVH_LEAF_BIT)) continue;
          const float* xy = c ? b.nodes[i].n1xy : b.nodes[i].n0xy;
          bin_box.push_back({r, {xy[0], xy[1], xy[2], xy[3], b.nodes[i].nz[2 * c], b.nodes[i].nz[2 * c + 1]}});
        }
      }
      std::sort(bin_box.begin(), bin_box.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
      for (const WideNode& w : b.wide_nodes) {
        if (w.exp[3] < 1 || w.exp[3] > 4) return std::printf("FAIL: child count\n"), 1;
        for (int k = (int)w.exp[3]; k < 4; k++)
          for (int a = 0; a < 3; a++)
            if (w.q[2 * a][k] != 255 || w.q[2 * a + 1][k] != 0 || w.ref[k] != w.ref[0]) return std::printf("FAIL: unused slot\n"), 1;
        for (int k = 0; k < (int)w.exp[3]; k++) {
          const uint32_t r = w.ref[k];
          if (!(r & BVH_LEAF_BIT)) continue;
          auto it = std::lower_bound(bin_box.begin(), bin_box.end(), r, [](const auto& x, uint32_t v) { return x.first < v; });
          if (it == bin_box.end() || it->first != r) return std::printf("FAIL: a wide leaf reference the binary tree does not hold\n"), 1;
          bool inside_one = false;
          for (; it != bin_box.end() && it->first == r && !inside_one; ++it) {
            bool inside = true;
            for (int a = 0; a < 3; a++) {
              const double step = std::ldexp(1.0, (int)(int8_t)w.exp[a]);
              const double lo = (double)w.origin[a] + w.q[2 * a][k] * step, hi = (double)w.origin[a] + w.q[2 * a + 1][k] * step;
              if (lo > (double)it->second[2 * a] || hi < (double)it->second[2 * a + 1]) inside = false;
            }
            inside_one = inside;
          }
          if (!inside_one) return std::printf("FAIL: a decoded child box does not contain the box it stands for\n"), 1;
        }
      }
      if (b.wide_stack_depth < 4) return std::printf("FAIL: wide stack bound\n"), 1;
    }
    // the treetop and the packed nodes
    sthip::Treetop tt;
    sthip::build_treetop(b.nodes.data(), b.nodes.size(), b.entries, b.root_ref, 170, tt);
    std::vector<BvhNodePacked> packed;
    sthip::pack_nodes(b.nodes.data(), b.nodes.size(), packed);
    sthip::pack_nodes(tt.nodes.data(), tt.nodes.size(), packed);
    // a transforms-only rebuild of the top level with the same transforms
    std::vector<BvhNode> tlas;
    uint32_t root = 0, world = 0, depth = 0;
    float center[3], radius = 0;
    if (!sthip::rebuild_top_level(b.top, xf.data(), inv.data(), s.instance_count, tlas, root, world, depth, center, radius, err)) return std::printf("REBUILD FAILED: %s\n", err.c_str()), 1;
    if (depth != b.stack_depth && !b.entries.empty()) return std::printf("FAIL: rebuilt top level has stack depth %u, build had %u\n", depth, b.stack_depth), 1;
    total_nodes += b.nodes.size();
    std::printf("%s: %zu units, %zu triangles (%zu in wrapped leaves), %zu entries, stack depth %u, treetop %zu\n", embed ? "embedded" : "separate", b.nodes.size(), b.tris.size(), wrapped, b.entries.size(),
                b.stack_depth, tt.nodes.size());
  }
  // vertices no 8-bit grid can hold (an infinite one, and a pair 6e38 apart): the build either refuses the scene or keeps a
  // binary tree, and build_wide_bvh then reports "no wide tree" by leaving EVERY wide output empty (api.hip walks the binary
  // tree for such a scene) — never a half-filled one
  for (int kind = 0; kind < 2 && !vertices.empty(); kind++) {
    std::vector<sthip_PackedVertexData> bad = vertices;
    const size_t at = bad.size() / 2;
    if (kind == 0) bad[at].position[1] = INFINITY;
    else bad[at].position[0] = 3.0e38f, bad[(at + 1) % bad.size()].position[0] = -3.0e38f;
    sthip_scene_desc sb = s;
    sb.gVertices = bad.data();
    sthip::BuiltBvh b;
    std::string err;
    if (!sthip::build_scene_bvh(sb, b, err, sthip::BVH_BUILDER_SAH_HOST, nullptr, false)) {
      std::printf("non-finite case %d: refused (%s)\n", kind, err.c_str());
      continue;
    }
    sthip::build_wide_bvh(b);
    if (b.wide_nodes.empty()) {
      if (!b.wide_entries.empty() || b.wide_root_ref != BVH_INVALID_REF || b.wide_stack_depth != 0) return std::printf("FAIL: a failed wide build left outputs behind\n"), 1;
      std::printf("non-finite case %d: no wide tree\n", kind);
    } else {
      for (const WideNode& w : b.wide_nodes)
        for (int a = 0; a < 3; a++)
          if (!std::isfinite(w.origin[a]) || (int8_t)w.exp[a] < -126) return std::printf("FAIL: a wide node with a non-finite grid\n"), 1;
      std::printf("non-finite case %d: wide tree kept (%zu nodes)\n", kind, b.wide_nodes.size());
    }
  }
  std::printf("BVH HOST OK %zu\n", total_nodes);
  return 0;
}
