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


// ---- the 8-wide compressed form (build_wide8_bvh): a CPU walk with the kernel's group logic (traverse.h: Traversal8) against
// a walk of the binary tree, both with exact-in-double conservative box tests and the same triangle test, so the closest
// hits must agree whatever the trees look like ----
namespace w8 {
struct Ray {
  double o[3], d[3];
};
struct Hit {
  double t = 1e300;
  uint32_t tri = 0xFFFFFFFFu, inst = 0xFFFFFFFFu;
  bool operator==(const Hit& h) const { return t == h.t && tri == h.tri && inst == h.inst; }
};
static bool slab(const double lo[3], const double hi[3], const Ray& r, double tbest) {
  double tn = 0, tf = tbest;
  for (int a = 0; a < 3; a++) {
    const double inv = 1.0 / (std::fabs(r.d[a]) < 1e-300 ? 1e-300 : r.d[a]);
    double t0 = (lo[a] - 1e-9 - r.o[a]) * inv, t1 = (hi[a] + 1e-9 - r.o[a]) * inv;
    if (t0 > t1) std::swap(t0, t1);
    tn = std::max(tn, t0);
    tf = std::min(tf, t1);
  }
  return tn <= tf;
}
static void tri_hit(const BvhTri& t, uint32_t index, uint32_t inst, const Ray& r, Hit& h) {  // Moeller-Trumbore in double
  const double e1[3] = {(double)t.v1[0] - t.v0[0], (double)t.v1[1] - t.v0[1], (double)t.v1[2] - t.v0[2]}, e2[3] = {(double)t.v2[0] - t.v0[0], (double)t.v2[1] - t.v0[1], (double)t.v2[2] - t.v0[2]};
  const double p[3] = {r.d[1] * e2[2] - r.d[2] * e2[1], r.d[2] * e2[0] - r.d[0] * e2[2], r.d[0] * e2[1] - r.d[1] * e2[0]};
  const double det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
  if (det == 0) return;
  const double s[3] = {r.o[0] - t.v0[0], r.o[1] - t.v0[1], r.o[2] - t.v0[2]};
  const double u = (s[0] * p[0] + s[1] * p[1] + s[2] * p[2]) / det;
  const double q[3] = {s[1] * e1[2] - s[2] * e1[1], s[2] * e1[0] - s[0] * e1[2], s[0] * e1[1] - s[1] * e1[0]};
  const double v = (r.d[0] * q[0] + r.d[1] * q[1] + r.d[2] * q[2]) / det;
  const double tt = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) / det;
  if (u < 0 || v < 0 || u + v > 1 || !(tt > 0)) return;
  if (tt < h.t || (tt == h.t && (inst < h.inst || (inst == h.inst && index < h.tri)))) {
    h.t = tt;
    h.tri = index;
    h.inst = inst;
  }
}
static Ray to_object(const TlasEntry& e, const Ray& r) {
  Ray q;
  for (int a = 0; a < 3; a++) {
    q.o[a] = e.inv[4 * a] * r.o[0] + e.inv[4 * a + 1] * r.o[1] + e.inv[4 * a + 2] * r.o[2] + e.inv[4 * a + 3];
    q.d[a] = e.inv[4 * a] * r.d[0] + e.inv[4 * a + 1] * r.d[1] + e.inv[4 * a + 2] * r.d[2];
  }
  return q;
}
// the binary tree (triangle ids: the triangle's own id word, so that the permutation of the array does not matter)
static void walk2(const sthip::BuiltBvh& b, uint32_t root, const Ray& r, uint32_t inst, Hit& h, size_t& visits) {
  std::vector<uint32_t> todo(1, root);
  while (!todo.empty()) {
    const uint32_t i = todo.back();
    todo.pop_back();
    visits++;
    const BvhNode& n = b.nodes[i];
    for (int c = 0; c < 2; c++) {
      const uint32_t ref = n.ref[c];
      if (ref == BVH_INVALID_REF || (c == 1 && ref == n.ref[0] && (ref & BVH_LEAF_BIT))) continue;
      const float* xy = c ? n.n1xy : n.n0xy;
      const double lo[3] = {xy[0], xy[2], n.nz[2 * c]}, hi[3] = {xy[1], xy[3], n.nz[2 * c + 1]};
      if (!slab(lo, hi, r, h.t)) continue;
      if (!(ref & BVH_LEAF_BIT)) {
        todo.push_back(ref);
      } else if (ref & BVH_INST_BIT) {
        const TlasEntry& e = b.entries[ref & 0xFFFFu];
        if (e.identity == TLAS_ENTRY_IDENTITY) walk2(b, e.root, r, inst, h, visits);
        else if (e.identity == TLAS_ENTRY_TRANSFORMED) walk2(b, e.root, to_object(e, r), e.id_bits, h, visits);
      } else {
        const uint32_t first = (ref & 0x3FFFFFFFu) >> 2, count = (ref & 3u) + 1;
        for (uint32_t k = 0; k < count; k++) tri_hit(b.tris[first + k], b.tris[first + k].id, inst, r, h);
      }
    }
  }
}
// the 8-wide form, with the kernel's states and stack discipline; returns false on a malformed tree
static bool walk8(const sthip::BuiltBvh& b, const Ray& world, Hit& h, size_t& visits, size_t& max_stack) {
  struct Group {
    uint32_t x, y;
  };
  const uint32_t DONE = 0xFFFFFFFFu, EXIT = 0xFFFFFFFEu;
  std::vector<Group> st;
  st.push_back({DONE, 0});
  Ray r = world;
  uint32_t inst = 0xFFFFFFFFu;
  uint32_t gx = b.wide8_root, gy = 0x80000000u, tx = 0, ty = 0;
  auto take = [&](Group P) {
    gx = P.x;
    gy = P.y;
    tx = P.x;
    ty = P.y <= 0x00FFFFFFu ? P.y : 0u;
  };
  for (size_t guard = 0; guard < 10000000; guard++) {
    max_stack = std::max(max_stack, st.size());
    if (gy > 0x00FFFFFFu && ty == 0) {  // walking
      const uint32_t octinv = (r.d[0] < 0 ? 0u : 1u) | (r.d[1] < 0 ? 0u : 2u) | (r.d[2] < 0 ? 0u : 4u);
      const uint32_t bit = 31u - (uint32_t)__builtin_clz(gy);
      const uint32_t rest = gy & ~(1u << bit);
      const uint32_t slot = (bit - 24u) ^ octinv;
      const uint32_t index = gx + (uint32_t)__builtin_popcount(gy & ((1u << slot) - 1u));
      if (rest > 0x00FFFFFFu) st.push_back({gx, rest});
      if (index >= b.wide8_nodes.size()) return std::printf("FAIL: wide8 node index out of range\n"), false;
      const Wide8Node& n = b.wide8_nodes[index];
      visits++;
      uint32_t hitmask = 0;
      for (int s = 0; s < 8; s++) {
        const uint32_t m = n.meta[s];
        if (!m) continue;
        double lo[3], hi[3];
        for (int a = 0; a < 3; a++) {
          const double step = std::ldexp(1.0, (int)(int8_t)n.exp[a]);
          lo[a] = (double)n.origin[a] + n.q[2 * a][s] * step;
          hi[a] = (double)n.origin[a] + n.q[2 * a + 1][s] * step;
        }
        if (!slab(lo, hi, r, h.t)) continue;
        const bool inner = (m & 0x18u) == 0x18u;
        const uint32_t at = inner ? ((m ^ octinv) & 31u) : (m & 31u);
        hitmask |= (m >> 5) << at;
      }
      if (hitmask == 0) {
        take(st.back());
        st.pop_back();
      } else {
        gx = n.child_base;
        gy = (hitmask & 0xFF000000u) | n.imask;
        tx = n.leaf_base;
        ty = hitmask & 0x00FFFFFFu;
      }
    } else if (ty != 0) {
      const uint32_t bit = (uint32_t)__builtin_ctz(ty);
      ty &= ty - 1;
      if (tx & WIDE8_ENTRY_BIT) {
        const uint32_t k = (tx & ~WIDE8_ENTRY_BIT) + bit;
        if (k >= b.wide8_entries.size()) return std::printf("FAIL: wide8 entry index out of range\n"), false;
        const TlasEntry& e = b.wide8_entries[k];
        if (e.identity == TLAS_ENTRY_TRANSFORMED || e.identity == TLAS_ENTRY_IDENTITY) {
          if (gy > 0x00FFFFFFu) st.push_back({gx, gy});
          if (ty) st.push_back({tx, ty});
          if (e.identity == TLAS_ENTRY_TRANSFORMED) {
            st.push_back({EXIT, 0});
            r = to_object(e, world);
            inst = e.id_bits;
          }
          gx = e.root;
          gy = 0x80000000u;
          ty = 0;
          continue;
        }
        // (spheres / volumes: not part of this comparison)
      } else {
        const uint32_t k = tx + bit;
        if (k >= b.tris.size()) return std::printf("FAIL: wide8 triangle index out of range\n"), false;
        tri_hit(b.tris[k], b.tris[k].id, inst, r, h);
      }
      if (ty == 0 && gy <= 0x00FFFFFFu) {
        take(st.back());
        st.pop_back();
      }
    } else if (gx == EXIT) {
      r = world;
      inst = 0xFFFFFFFFu;
      take(st.back());
      st.pop_back();
    } else if (gx == DONE) {
      return st.empty() ? true : (std::printf("FAIL: wide8 walk ended with %zu stack entries\n", st.size()), false);
    } else {
      return std::printf("FAIL: wide8 walk in an undefined state\n"), false;
    }
  }
  return std::printf("FAIL: wide8 walk does not end\n"), false;
}
static uint32_t rnd_state = 12345u;
static double rnd() {
  rnd_state = rnd_state * 1664525u + 1013904223u;
  return (rnd_state >> 8) * (1.0 / 16777216.0);
}
// structure: indices in range, every triangle an item of exactly one node, items and inner children consistent with meta / imask
static bool check_structure(const sthip::BuiltBvh& b) {
  std::vector<uint32_t> seen(b.tris.size(), 0);
  std::vector<uint8_t> is_top(b.wide8_nodes.size(), 0);
  for (size_t i = b.top.wide8_blas_nodes; i < b.wide8_nodes.size(); i++) is_top[i] = 1;
  for (size_t i = 0; i < b.wide8_nodes.size(); i++) {
    const Wide8Node& n = b.wide8_nodes[i];
    uint32_t inner = 0, items = 0;
    for (int s = 0; s < 8; s++) {
      const uint32_t m = n.meta[s];
      const bool in = (n.imask >> s) & 1;
      if (in != ((m & 0x18u) == 0x18u && m != 0)) return std::printf("FAIL: wide8 imask / meta disagree\n"), false;
      if (!m) {
        for (int a = 0; a < 3; a++)
          if (n.q[2 * a][s] != 255 || n.q[2 * a + 1][s] != 0) return std::printf("FAIL: wide8 unused slot has a box\n"), false;
        continue;
      }
      for (int a = 0; a < 3; a++)
        if (n.q[2 * a][s] > n.q[2 * a + 1][s]) return std::printf("FAIL: wide8 inverted box on a used slot\n"), false;
      if (in) {
        if (m != (0x20u | (24u + (uint32_t)s))) return std::printf("FAIL: wide8 inner meta\n"), false;
        inner++;
        continue;
      }
      const uint32_t unary = m >> 5, first = m & 31u, count = unary == 1 ? 1 : unary == 3 ? 2 : unary == 7 ? 3 : 0;
      if (!count || first != items) return std::printf("FAIL: wide8 leaf meta (items not consecutive)\n"), false;
      items += count;
      if (items > WIDE8_MAX_ITEMS) return std::printf("FAIL: wide8 too many items\n"), false;
      if (n.leaf_base & WIDE8_ENTRY_BIT) {
        if (!is_top[i] || count != 1 || (n.leaf_base & ~WIDE8_ENTRY_BIT) + first >= b.wide8_entries.size()) return std::printf("FAIL: wide8 entry item\n"), false;
      } else {
        for (uint32_t k = 0; k < count; k++) {
          if (n.leaf_base + first + k >= b.tris.size()) return std::printf("FAIL: wide8 triangle item out of range\n"), false;
          if (!is_top[i]) seen[n.leaf_base + first + k]++;  // (a copy of the merged mesh's root in the top level repeats that root's items)
          // the decoded box holds the triangle
          const BvhTri& t = b.tris[n.leaf_base + first + k];
          for (int a = 0; a < 3; a++) {
            const double step = std::ldexp(1.0, (int)(int8_t)n.exp[a]);
            const double lo = (double)n.origin[a] + n.q[2 * a][s] * step, hi = (double)n.origin[a] + n.q[2 * a + 1][s] * step;
            for (const float* v : {t.v0, t.v1, t.v2})
              if ((double)v[a] < lo || (double)v[a] > hi) return std::printf("FAIL: a wide8 leaf box does not hold its triangle\n"), false;
          }
        }
      }
    }
    if (inner && (size_t)n.child_base + inner > b.wide8_nodes.size()) return std::printf("FAIL: wide8 children out of range\n"), false;
  }
  for (uint32_t c : seen)
    if (c != 1) return std::printf("FAIL: a triangle is an item of %u wide8 nodes\n", c), false;
  return true;
}
}  // namespace w8

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
    // the 8-wide compressed form: made from a build of its own (it permutes the triangles), walked on the CPU against the
    // binary tree of the same build: same closest hits for rays through the scene
    if (!embed) {
      sthip::BuiltBvh b8;
      if (!sthip::build_scene_bvh(s, b8, err, sthip::BVH_BUILDER_SAH_HOST, nullptr, false)) return std::printf("BUILD FAILED: %s\n", err.c_str()), 1;
      sthip::build_wide8_bvh(b8);
      bool has_tri_entry = false;
      for (const TlasEntry& e : b8.entries) has_tri_entry |= e.identity == TLAS_ENTRY_IDENTITY || e.identity == TLAS_ENTRY_TRANSFORMED;
      if (b8.wide8_nodes.empty() && has_tri_entry) return std::printf("FAIL: no wide8 nodes\n"), 1;
      if (!b8.wide8_nodes.empty()) {
        if (b8.tris.size() != b.tris.size()) return std::printf("FAIL: wide8 changed the triangle count\n"), 1;
        if (!w8::check_structure(b8)) return 1;
        // the binary leaves still name every triangle once
        std::vector<uint32_t> seen8(b8.tris.size(), 0);
        for (const BvhNode& nd : b8.nodes)
          for (int c = 0; c < 2; c++) {
            const uint32_t r = nd.ref[c];
            if ((r & (BVH_LEAF_BIT | BVH_INST_BIT)) != BVH_LEAF_BIT || (c == 1 && r == nd.ref[0])) continue;
            for (uint32_t k = 0; k <= (r & 3u); k++) {
              if (((r & 0x3FFFFFFFu) >> 2) + k >= b8.tris.size()) return std::printf("FAIL: remapped leaf out of range\n"), 1;
              seen8[((r & 0x3FFFFFFFu) >> 2) + k]++;
            }
          }
        for (uint32_t c : seen8)
          if (c != 1) return std::printf("FAIL: after the wide8 permutation a triangle is in %u binary leaves\n", c), 1;
        size_t v2 = 0, v8 = 0, max_stack = 0, hits = 0;
        const int RAYS = 3000;
        for (int k = 0; k < RAYS; k++) {
          w8::Ray r;
          for (int a = 0; a < 3; a++) {
            r.o[a] = b8.scene_center[a] + (w8::rnd() * 2 - 1) * 0.6 * b8.scene_radius;
            r.d[a] = w8::rnd() * 2 - 1;
          }
          w8::Hit h2, h8;
          if (b8.top_is_world_blas) w8::walk2(b8, b8.root_ref, r, 0xFFFFFFFFu, h2, v2);
          else w8::walk2(b8, b8.root_ref, r, 0xFFFFFFFFu, h2, v2);
          if (!w8::walk8(b8, r, h8, v8, max_stack)) return 1;
          if (!(h2 == h8)) return std::printf("FAIL: ray %d: binary walk t=%.9g tri=%u inst=%u, wide8 walk t=%.9g tri=%u inst=%u\n", k, h2.t, h2.tri, h2.inst, h8.t, h8.tri, h8.inst), 1;
          hits += h2.tri != 0xFFFFFFFFu;
        }
        if (max_stack > b8.wide8_stack_depth) return std::printf("FAIL: the wide8 walk needed %zu stack entries, the bound says %u\n", max_stack, b8.wide8_stack_depth), 1;
        // a transforms-only rebuild with the same transforms gives a top level that walks the same
        {
          std::vector<BvhNode> tlas8;
          uint32_t root8 = 0, world8 = 0, depth8 = 0;
          float c8[3], r8 = 0;
          sthip::TopLevelState st8 = b8.top;
          if (!sthip::rebuild_top_level(st8, xf.data(), inv.data(), s.instance_count, tlas8, root8, world8, depth8, c8, r8, err)) return std::printf("REBUILD FAILED: %s\n", err.c_str()), 1;
          std::vector<Wide8Node> nodes8 = b8.wide8_nodes;
          std::vector<TlasEntry> entries8;
          uint32_t wroot = 0, wdepth = 0;
          if (!sthip::build_wide8_top(st8, tlas8.data(), st8.blas_nodes, root8, world8 != 0, nodes8, entries8, wroot, wdepth)) return std::printf("FAIL: build_wide8_top\n"), 1;
          sthip::BuiltBvh moved = b8;
          moved.wide8_nodes = nodes8;
          moved.wide8_entries = entries8;
          moved.wide8_root = wroot;
          moved.wide8_stack_depth = wdepth;
          moved.top = st8;
          if (!w8::check_structure(moved)) return 1;
          size_t dummy = 0, ms = 0;
          for (int k = 0; k < 300; k++) {
            w8::Ray r;
            for (int a = 0; a < 3; a++) {
              r.o[a] = b8.scene_center[a] + (w8::rnd() * 2 - 1) * 0.6 * b8.scene_radius;
              r.d[a] = w8::rnd() * 2 - 1;
            }
            w8::Hit h2, h8;
            w8::walk2(b8, b8.root_ref, r, 0xFFFFFFFFu, h2, dummy);
            if (!w8::walk8(moved, r, h8, dummy, ms)) return 1;
            if (!(h2 == h8)) return std::printf("FAIL: rebuilt wide8 top level: ray %d differs\n", k), 1;
          }
          if (ms > wdepth) return std::printf("FAIL: rebuilt wide8 stack bound\n"), 1;
        }
        std::printf("wide8: %zu nodes (%u bottom-level), %zu entries, stack %u (walks used %zu); %d rays, %zu hit: %.2f binary / %.2f wide8 node visits per ray\n", b8.wide8_nodes.size(),
                    b8.top.wide8_blas_nodes, b8.wide8_entries.size(), b8.wide8_stack_depth, max_stack, RAYS, hits, (double)v2 / RAYS, (double)v8 / RAYS);
      }
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
