// bvh_build.cpp — host-side construction of the acceleration structure described in bvh.h.
//
// The reference hands BLAS/TLAS construction to the Vulkan driver with ePreferFastTrace
// (src/Core/AccelerationStructure.cpp:6,25); scenes are static, so the build is a load-time step
// (src/Node/Scene.cpp:345 rebuilds only when dirty). Here: a binned-SAH BVH2 per unique mesh with a
// hard depth cap (the traversal stack lives in LDS and is sized from it), every identity-transform
// instance flattened into one world-space mesh, and a small top level over what remains.
#include "bvh_build.h"

#include <math.h>
#include <cmath>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <limits>
#include <map>
#include <queue>
#include <thread>
#include <unordered_map>
#include <tuple>

namespace sthip {
namespace {

struct Box {
  float lo[3], hi[3];
  void reset() {
    for (int a = 0; a < 3; a++) {
      lo[a] = INFINITY;
      hi[a] = -INFINITY;
    }
  }
  void grow(const float* p) {
    for (int a = 0; a < 3; a++) {
      lo[a] = std::min(lo[a], p[a]);
      hi[a] = std::max(hi[a], p[a]);
    }
  }
  void grow(const Box& b) {
    for (int a = 0; a < 3; a++) {
      lo[a] = std::min(lo[a], b.lo[a]);
      hi[a] = std::max(hi[a], b.hi[a]);
    }
  }
  float half_area() const {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
  }
};

struct TmpNode {
  Box box;
  int32_t left, right;  // -1: leaf
  uint32_t first, count;
};

// Binary binned-SAH build over `boxes`; leaves hold at most `leaf_max` primitives; depth <= depth_cap.
struct Builder {
  const std::vector<Box>& boxes;
  std::vector<float> cen;
  std::vector<uint32_t> order;
  std::vector<TmpNode> nodes;
  uint32_t leaf_max;
  uint32_t depth_cap;
  uint32_t max_depth = 0;

  // parallel_min / task_min: below parallel_min primitives the build stays on the calling thread; a task holds at least task_min
  Builder(const std::vector<Box>& b, uint32_t leaf, uint32_t cap, size_t parallel_min = 50000, size_t task_min = 4096) : boxes(b), leaf_max(leaf), depth_cap(cap) {
    const size_t n = boxes.size();
    cen.resize(3 * n);
    order.resize(n);
    for (size_t i = 0; i < n; i++) {
      order[i] = (uint32_t)i;
      for (int a = 0; a < 3; a++) cen[3 * i + a] = 0.5f * (boxes[i].lo[a] + boxes[i].hi[a]);
    }
    nodes.reserve(2 * n / std::max(1u, leaf) + 16);
    if (!n) return;
    // Large builds run on several host threads: the top of the tree is built here, and every subtree small enough
    // (<= n / (8 * threads) primitives) becomes a task that builds into a node vector of its own over its own
    // (disjoint) range of `order`; the pieces are then appended with their indices shifted. The tree is the one the
    // single-threaded build makes (same splits, same order), only the numbering of TmpNode differs, and flatten()
    // renumbers depth-first anyway.
    unsigned threads = std::thread::hardware_concurrency();
    if (const char* e = getenv("STHIP_BUILD_THREADS")) threads = (unsigned)std::max(1, atoi(e));
    threads = std::min(threads, 32u);
    if (n < parallel_min || threads <= 1) {
      build(nodes, max_depth, nullptr, 0, (uint32_t)n, 0);
      return;
    }
    std::vector<Task> tasks;
    task_size = (uint32_t)std::max<size_t>(task_min, n / (8 * (size_t)threads));
    build(nodes, max_depth, &tasks, 0, (uint32_t)n, 0);
    std::vector<std::vector<TmpNode>> parts(tasks.size());
    std::vector<uint32_t> depths(tasks.size(), 0);
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; t++)
      pool.emplace_back([&]() {
        for (;;) {
          const size_t k = next.fetch_add(1);
          if (k >= tasks.size()) break;
          parts[k].reserve(2 * (tasks[k].hi - tasks[k].lo) / std::max(1u, leaf_max) + 16);
          build(parts[k], depths[k], nullptr, tasks[k].lo, tasks[k].hi, tasks[k].depth);
        }
      });
    for (auto& th : pool) th.join();
    for (size_t k = 0; k < tasks.size(); k++) {
      const int32_t base = (int32_t)nodes.size();
      for (TmpNode tn : parts[k]) {
        if (tn.left >= 0) {
          tn.left += base;
          tn.right += base;
        }
        nodes.push_back(tn);
      }
      (tasks[k].right_child ? nodes[tasks[k].parent].right : nodes[tasks[k].parent].left) = base;  // a task's root is its node 0
      max_depth = std::max(max_depth, depths[k]);
    }
  }
  struct Task {
    uint32_t lo, hi, depth;
    int32_t parent;
    bool right_child;
  };
  uint32_t task_size = 0;

  // largest primitive count a subtree rooted at `depth` may hold
  uint64_t capacity(uint32_t depth) const {
    const uint32_t rem = depth_cap > depth ? depth_cap - depth : 0;
    return (uint64_t)leaf_max << std::min(rem, 40u);
  }

  // builds the subtree over order[lo, hi) into `nodes` (indices relative to that vector). With `tasks`, subtrees of at
  // most task_size primitives below the root are not built but recorded (their parent's link is patched later).
  int32_t build(std::vector<TmpNode>& nodes, uint32_t& max_depth, std::vector<Task>* tasks, uint32_t lo, uint32_t hi, uint32_t depth) {
    const int32_t idx = (int32_t)nodes.size();
    nodes.push_back(TmpNode());
    max_depth = std::max(max_depth, depth);
    Box box, cbox;
    box.reset();
    cbox.reset();
    for (uint32_t i = lo; i < hi; i++) {
      box.grow(boxes[order[i]]);
      cbox.grow(&cen[3 * (size_t)order[i]]);
    }
    nodes[idx].box = box;
    nodes[idx].left = nodes[idx].right = -1;
    nodes[idx].first = lo;
    nodes[idx].count = hi - lo;
    const uint32_t n = hi - lo;
    if (n <= leaf_max) return idx;

#ifndef SAH_BINS
#define SAH_BINS 32
#endif
    const int NB = SAH_BINS;
    float best_cost = INFINITY;
    int best_axis = -1, best_bin = -1;
    for (int axis = 0; axis < 3; axis++) {
      const float ext = cbox.hi[axis] - cbox.lo[axis];
      if (!(ext > 0)) continue;
      Box bb[NB];
      uint32_t bc[NB];
      for (int b = 0; b < NB; b++) {
        bb[b].reset();
        bc[b] = 0;
      }
      const float k = NB / ext;
      for (uint32_t i = lo; i < hi; i++) {
        const uint32_t p = order[i];
        int b = (int)((cen[3 * (size_t)p + axis] - cbox.lo[axis]) * k);
        b = std::min(std::max(b, 0), NB - 1);
        bb[b].grow(boxes[p]);
        bc[b]++;
      }
      float ra[NB];
      uint32_t rc[NB];
      Box r;
      r.reset();
      uint32_t c = 0;
      for (int b = NB - 1; b > 0; b--) {
        r.grow(bb[b]);
        c += bc[b];
        ra[b] = r.half_area();
        rc[b] = c;
      }
      Box l;
      l.reset();
      c = 0;
      for (int b = 0; b < NB - 1; b++) {
        l.grow(bb[b]);
        c += bc[b];
        if (c == 0 || rc[b + 1] == 0) continue;
        const float cost = l.half_area() * (float)c + ra[b + 1] * (float)rc[b + 1];
        if (cost < best_cost) {
          best_cost = cost;
          best_axis = axis;
          best_bin = b;
        }
      }
    }
    uint32_t mid = lo;
    if (best_axis >= 0) {
      const float ext = cbox.hi[best_axis] - cbox.lo[best_axis];
      const float k = NB / ext;
      const float clo = cbox.lo[best_axis];
      uint32_t* m = std::partition(order.data() + lo, order.data() + hi, [&](uint32_t p) {
        int b = (int)((cen[3 * (size_t)p + best_axis] - clo) * k);
        b = std::min(std::max(b, 0), NB - 1);
        return b <= best_bin;
      });
      mid = (uint32_t)(m - order.data());
    }
    const uint64_t cap = capacity(depth + 1);
    const bool sah_ok = mid > lo && mid < hi && (uint64_t)(mid - lo) <= cap && (uint64_t)(hi - mid) <= cap;
    if (!sah_ok) {
      // balanced split along the widest centroid axis (also the degenerate all-same-centroid case)
      int axis = 0;
      for (int a = 1; a < 3; a++)
        if (cbox.hi[a] - cbox.lo[a] > cbox.hi[axis] - cbox.lo[axis]) axis = a;
      mid = lo + n / 2;
      std::nth_element(order.begin() + lo, order.begin() + mid, order.begin() + hi,
                       [&](uint32_t a, uint32_t b) { return cen[3 * (size_t)a + axis] < cen[3 * (size_t)b + axis]; });
    }
    auto child = [&](uint32_t a, uint32_t b, bool right) -> int32_t {
      if (tasks && b - a <= task_size && b - a > leaf_max) {
        tasks->push_back(Task{a, b, depth + 1, idx, right});
        return 0;  // patched when the task's nodes are appended
      }
      return build(nodes, max_depth, tasks, a, b, depth + 1);
    };
    const int32_t l = child(lo, mid, false);
    const int32_t r = child(mid, hi, true);
    nodes[idx].left = l;
    nodes[idx].right = r;
    nodes[idx].count = 0;
    return idx;
  }
};

inline void set_child(BvhNode& n, int c, const Box* b, uint32_t ref) {
  float* xy = c == 0 ? n.n0xy : n.n1xy;
  if (b) {
    xy[0] = b->lo[0];
    xy[1] = b->hi[0];
    xy[2] = b->lo[1];
    xy[3] = b->hi[1];
    n.nz[2 * c] = b->lo[2];
    n.nz[2 * c + 1] = b->hi[2];
  } else {  // empty: inverted box, never hit
    xy[0] = xy[2] = INFINITY;
    xy[1] = xy[3] = -INFINITY;
    n.nz[2 * c] = INFINITY;
    n.nz[2 * c + 1] = -INFINITY;
  }
  n.ref[c] = ref;
}

// Flattens a Builder tree into out.nodes (depth-first). `leaf_ref(first,count)` makes the leaf reference.
// Returns the reference of the root (an inner-node index, or a leaf reference for a single-leaf tree).
template <typename LeafRef>
uint32_t flatten(const Builder& b, int32_t t, std::vector<BvhNode>& out, LeafRef leaf_ref) {
  const TmpNode& tn = b.nodes[t];
  if (tn.left < 0) return leaf_ref(tn.first, tn.count);
  const uint32_t my = (uint32_t)out.size();
  out.push_back(BvhNode());
  memset(&out[my], 0, sizeof(BvhNode));
  // leaf children first: with embedded leaves their triangles take the units right behind this node
  const bool lleaf = b.nodes[tn.left].left < 0, rleaf = b.nodes[tn.right].left < 0;
  uint32_t l = 0, r = 0;
  if (lleaf) l = flatten(b, tn.left, out, leaf_ref);
  if (rleaf) r = flatten(b, tn.right, out, leaf_ref);
  if (!lleaf) l = flatten(b, tn.left, out, leaf_ref);
  if (!rleaf) r = flatten(b, tn.right, out, leaf_ref);
  set_child(out[my], 0, &b.nodes[tn.left].box, l);
  set_child(out[my], 1, &b.nodes[tn.right].box, r);
  return my;
}

inline uint32_t wrap_leaf(std::vector<BvhNode>& nodes, const Box& box, uint32_t leaf) {
  BvhNode n;
  memset(&n, 0, sizeof(n));
  // both slots hold the leaf: the traversal then needs no "is this child there" test in its inner loop; testing the
  // leaf's triangles twice changes neither the closest hit nor an occlusion answer
  set_child(n, 0, &box, leaf);
  set_child(n, 1, &box, leaf);
  nodes.push_back(n);
  return (uint32_t)nodes.size() - 1;
}

inline bool is_identity(const sthip_TransformData& t) {
  static const float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  return memcmp(&t, I, sizeof(I)) == 0;
}

struct InstView {
  uint32_t type, material_address, prim_count, stride, first_vertex, indices_byte_offset;
};
inline InstView view(const sthip_InstanceData& d) {
  InstView v;
  v.type = d.packed[0] & 0xF;
  v.material_address = d.packed[0] >> 4;
  v.prim_count = (d.packed[1] >> 12) & 0xFFFF;
  v.stride = d.packed[1] >> 28;
  v.first_vertex = d.packed[2];
  v.indices_byte_offset = d.packed[3];
  return v;
}
// scene.h:139-161
inline void load_tri(const sthip_scene_desc& s, const InstView& in, uint32_t prim, uint32_t tri[3]) {
  const uint8_t* ib = (const uint8_t*)s.gIndices;
  const size_t off = (size_t)in.indices_byte_offset + (size_t)prim * 3 * in.stride;
  if (in.stride == 2) {
    uint16_t w[3];
    memcpy(w, ib + off, 6);
    tri[0] = w[0];
    tri[1] = w[1];
    tri[2] = w[2];
  } else {
    memcpy(tri, ib + off, 12);
  }
  for (int k = 0; k < 3; k++) tri[k] += in.first_vertex;
}

}  // namespace

bool build_scene_bvh(const sthip_scene_desc& s, BuiltBvh& out, std::string& err, int builder, DeviceBuildTarget* device, bool embed_leaves) {
  out = BuiltBvh();
  // device mode: bottom levels of >= DEVICE_MIN_PRIMS triangles are built in place on the GPU from the uploaded scene arrays
  const bool dev_mode = device != nullptr && builder == BVH_BUILDER_LBVH_GPU;
  const uint32_t DEVICE_MIN_PRIMS = 64;
  // embedded leaves need every bottom level in one array the host lays out: not with the device builder, not through lbvh_build_gpu
  out.embedded = embed_leaves && builder == BVH_BUILDER_SAH_HOST;
  const uint32_t BLAS_DEPTH_CAP = 22, TLAS_DEPTH_CAP = 18, LBVH_MAX_HEIGHT = 120;
  // (a leaf triangle remembers where its index triple lies and its mesh's first vertex in 32 and 31 bits: bvh.h: BvhTri)
  if (s.indices_bytes > 0xFFFFFFFFull || s.vertex_count >= 0x80000000u) {
    err = "gIndices beyond 4 GiB or more than 2^31 vertices";
    return false;
  }
  // ---- validate + classify ----
  std::vector<uint32_t> merged, separate, spheres, volumes;
  // gVolumes: NanoVDB float grids, format 32.x (PNanoVDB.h:761-777,904-916,964-968, FLOAT row of the type constants)
  for (uint32_t i = 0; i < s.volume_count; i++) {
    const uint8_t* b = (const uint8_t*)s.gVolumes[i].data;
    const uint64_t n = s.gVolumes[i].bytes;
    auto rd32 = [&](uint64_t off) { uint32_t v = 0; if (off + 4 <= n) memcpy(&v, b + off, 4); return v; };
    auto rd64 = [&](uint64_t off) { uint64_t v = 0; if (off + 8 <= n) memcpy(&v, b + off, 8); return v; };
    auto rdf = [&](uint64_t off) { float v = 0; if (off + 4 <= n) memcpy(&v, b + off, 4); return v; };
    if (!b || n < 672 + 64 + 64 || n > 0xFFFFFFF0ull || (n & 3) || rd64(0) != 0x304244566f6e614eull || rd32(636) != 1u) {
      err = "gVolumes entry is not a NanoVDB float grid (magic / grid type / size)";
      return false;
    }
    DeviceVolume v;
    memset(&v, 0, sizeof(v));
    v.bytes = (uint32_t)n;
    const uint64_t root = 672 + rd64(672 + 24);
    if (root + 64 > n) {
      err = "gVolumes entry: the root node lies outside the buffer";
      return false;
    }
    v.root = (uint32_t)root;
    const uint64_t tiles = rd32(root + 24);  // the device reader walks the root's tile table linearly
    if (tiles > 65536 || root + 64 + 32 * tiles > n) {
      err = "gVolumes entry: the root tile table lies outside the buffer";
      return false;
    }
    for (int k = 0; k < 3; k++) {
      v.bbox_min[k] = (int32_t)rd32(root + 4 * k);
      v.bbox_max[k] = (int32_t)rd32(root + 12 + 4 * k);
      v.vecf[k] = rdf(296 + 72 + 4 * k);
    }
    v.root_max = rdf(root + 36);
    for (int k = 0; k < 9; k++) {
      v.matf[k] = rdf(296 + 4 * k);
      v.invmatf[k] = rdf(296 + 36 + 4 * k);
    }
    out.volumes.push_back(v);
  }
  for (uint32_t i = 0; i < s.instance_count; i++) {
    const InstView in = view(s.gInstances[i]);
    if (in.type == STHIP_INSTANCE_TYPE_SPHERE) {
      float r;
      memcpy(&r, &s.gInstances[i].packed[2], 4);
      if (!(fabsf(r) > 0.0f) || !std::isfinite(r)) {
        err = "sphere instance with a zero or non-finite radius";
        return false;
      }
      if ((size_t)in.material_address + sizeof(sthip_MaterialRecord) > s.material_bytes) {
        err = "instance material_address exceeds gMaterialData";
        return false;
      }
      spheres.push_back(i);
      continue;
    }
    if (in.type == STHIP_INSTANCE_TYPE_VOLUME) {
      if (s.gInstances[i].packed[2] >= s.volume_count) {
        err = "volume instance refers to a volume that is not in gVolumes";
        return false;
      }
      if ((size_t)in.material_address + 40 > s.material_bytes) {
        err = "instance material_address exceeds gMaterialData";
        return false;
      }
      volumes.push_back(i);
      continue;
    }
    if (in.type != STHIP_INSTANCE_TYPE_TRIANGLES) {
      err = "unknown instance type";
      return false;
    }
    if (in.stride != 2 && in.stride != 4) {
      err = "index stride must be 2 or 4";
      return false;
    }
    const size_t end = (size_t)in.indices_byte_offset + (size_t)in.prim_count * 3 * in.stride;
    if (end > s.indices_bytes) {
      err = "instance index range exceeds gIndices";
      return false;
    }
    if ((size_t)in.material_address + sizeof(sthip_MaterialRecord) > s.material_bytes) {
      err = "instance material_address exceeds gMaterialData";
      return false;
    }
    if (!dev_mode)  // (device mode: the fetch kernel checks the meshes it builds, add_blas the ones the host builds)
      for (uint32_t p = 0; p < in.prim_count; p++) {
        uint32_t tri[3];
        load_tri(s, in, p, tri);
        for (int k = 0; k < 3; k++)
          if (tri[k] >= s.vertex_count) {
            err = "vertex index exceeds gVertices";
            return false;
          }
      }
    if (is_identity(s.gInstanceInverseTransforms[i]) && is_identity(s.gInstanceTransforms[i]))
      merged.push_back(i);
    else
      separate.push_back(i);
  }

  // alpha masks (Material.hpp:35): MaterialRecord.alpha_mask_index of the instance's material, if it names an image
  std::vector<uint32_t> inst_alpha(s.instance_count, BVH_NO_ALPHA);
  bool any_alpha = false;
  for (uint32_t i = 0; i < s.instance_count; i++) {
    const InstView in = view(s.gInstances[i]);
    if (in.type != STHIP_INSTANCE_TYPE_TRIANGLES) continue;
    sthip_MaterialRecord rec;
    memcpy(&rec, (const uint8_t*)s.gMaterialData + in.material_address, sizeof(rec));
    if (rec.alpha_mask_index < STHIP_IMAGE_COUNT) {
      if (rec.alpha_mask_index >= s.image1_count) {
        err = "a material refers to an alpha mask that is not in gImage1s";
        return false;
      }
      inst_alpha[i] = rec.alpha_mask_index;
      any_alpha = true;
    }
  }
  // (the uvs the alpha test interpolates lie beside the leaf triangles and are filled on the device from the resident
  // triangles, whoever built the tree: api.hip, k_fill_tri_shade)

  std::string gpu_err;
  // ---- device mode: the region the GPU builder fills is sized before anything is built ----
  // A bottom level = the triangles of a list of instances (all merged ones, or the one that stands for a shared mesh).
  auto prim_total = [&](const std::vector<uint32_t>& insts) {
    size_t n = 0;
    for (uint32_t i : insts) n += view(s.gInstances[i]).prim_count;
    return n;
  };
  uint32_t dev_node_cursor = 0, dev_tri_cursor = 0;
  if (dev_mode) {
    size_t dev_nodes = 0, dev_tris = 0, all_prims = 0, meshes_n = 0;
    auto account = [&](size_t n) {
      if (n == 0) return;
      all_prims += n;
      meshes_n++;
      if (n >= DEVICE_MIN_PRIMS) {
        dev_nodes += n - 1;
        dev_tris += n;
      }
    };
    account(prim_total(merged));
    std::map<std::tuple<uint32_t, uint32_t, uint32_t, uint32_t>, int> seen;
    for (uint32_t i : separate) {
      const InstView in = view(s.gInstances[i]);
      if (seen.emplace(std::make_tuple(in.first_vertex, in.indices_byte_offset, in.prim_count, in.stride), 0).second) account(in.prim_count);
    }
    if (all_prims >= (1u << 28)) {
      err = "too many triangles";
      return false;
    }
    out.dev_nodes = (uint32_t)dev_nodes;
    out.dev_tris = (uint32_t)dev_tris;
    // upper bounds of what the host adds behind the device region: a mesh of n triangles has at most max(n - 1, 1)
    // nodes, the top level at most 2 per entry; + the headroom a transforms-only update may need
    const size_t entries_upper = (merged.empty() ? 0 : 1) + separate.size() + spheres.size() + volumes.size();
    // (+ all_prims / 4: the host-built SAH tops over the device-built subtrees: at most one node per frontier entry, and a
    // frontier of size s has about 2 n / s entries; s >= 16)
    if (!device->reserve || !device->reserve(device->user, all_prims + all_prims / 4 + 64 * meshes_n + 4 * entries_upper + 8, all_prims + 1, *device)) {
      err = "lbvh: could not reserve the device arrays";
      return false;
    }
  }
  auto add_blas = [&](const std::vector<uint32_t>& insts, bool with_instance_bits, Box& bounds, uint32_t& depth, bool& on_device) -> uint32_t {
    on_device = false;
    const size_t total = prim_total(insts);
    if (dev_mode && total >= DEVICE_MIN_PRIMS) {
      std::vector<MeshPiece> pieces;
      uint32_t begin = 0;
      for (uint32_t i : insts) {
        const InstView in = view(s.gInstances[i]);
        if (in.prim_count == 0) continue;
        pieces.push_back(MeshPiece{with_instance_bits ? i : 0u, in.first_vertex, in.indices_byte_offset, in.stride, in.prim_count, begin});
        begin += in.prim_count;
      }
      uint32_t root = 0;
      float ms = 0, bb[6];
      std::vector<FrontierEntry> frontier;
      if (!lbvh_build_device(*device, pieces, dev_node_cursor, dev_tri_cursor, root, depth, bb, ms, gpu_err, device->sah_top_size ? &frontier : nullptr, device->sah_top_size)) return BVH_INVALID_REF;
      out.gpu_build_ms += ms;
      if (depth <= LBVH_MAX_HEIGHT) {  // (else: the SAH builder below; the reserved stretch of the device region stays unused)
        memcpy(bounds.lo, bb, 12);
        memcpy(bounds.hi, bb + 3, 12);
        dev_node_cursor += (uint32_t)total - 1;
        dev_tri_cursor += (uint32_t)total;
        if (frontier.size() >= 2) {
          // A binned-SAH top over the device-built subtrees: host nodes (behind the device region, like the top level) whose
          // leaves are the subtrees' own references — final already, so they are marked for the offset pass at the end.
          std::vector<Box> fboxes(frontier.size());
          uint32_t below = 0;
          for (size_t k = 0; k < frontier.size(); k++) {
            memcpy(fboxes[k].lo, frontier[k].lo, 12);
            memcpy(fboxes[k].hi, frontier[k].hi, 12);
            below = std::max(below, frontier[k].height);
          }
          Builder tb(fboxes, 1, BLAS_DEPTH_CAP, 4000, 512);  // ten thousand boxes or so: worth a few threads, it sits on the rebuild's critical path
          const size_t first_node = out.nodes.size();
          const uint32_t TOKEN = BVH_LEAF_BIT | BVH_INST_BIT;  // never a bottom-level reference: stands for "frontier entry i" until the pass below
          const uint32_t top_root = flatten(tb, 0, out.nodes, [&](uint32_t first, uint32_t) { return TOKEN | tb.order[first]; });
          out.absolute_ref.resize(2 * out.nodes.size(), 0);
          for (size_t i = first_node; i < out.nodes.size(); i++)
            for (int c = 0; c < 2; c++)
              if ((out.nodes[i].ref[c] & TOKEN) == TOKEN && out.nodes[i].ref[c] != BVH_INVALID_REF) {
                out.nodes[i].ref[c] = frontier[out.nodes[i].ref[c] & ~TOKEN].ref;
                out.absolute_ref[2 * i + c] = 1;
              }
          depth = tb.max_depth + 1 + below;
          on_device = false;  // the root is a host node: offset with the other host-built roots
          return top_root;
        }
        on_device = true;
        return root;
      }
    }
    std::vector<std::pair<uint32_t, uint32_t>> prims;  // (instance, prim)
    prims.reserve(total);
    for (uint32_t i : insts) {
      const InstView in = view(s.gInstances[i]);
      for (uint32_t p = 0; p < in.prim_count; p++) prims.emplace_back(i, p);
    }
    if (dev_mode)  // the index check the validation loop left to the builders
      for (const auto& pr : prims) {
        uint32_t tri[3];
        load_tri(s, view(s.gInstances[pr.first]), pr.second, tri);
        for (int k = 0; k < 3; k++)
          if (tri[k] >= s.vertex_count) {
            gpu_err = "vertex index exceeds gVertices";
            return BVH_INVALID_REF;
          }
      }
    if (!dev_mode && builder == BVH_BUILDER_LBVH_GPU && prims.size() >= 64) {
      // GPU path through host vectors: hand the triangles over unsorted; the device sorts them along the Morton curve
      std::vector<BvhTri> tin(prims.size());
      bounds.reset();
      for (size_t k = 0; k < prims.size(); k++) {
        const InstView in = view(s.gInstances[prims[k].first]);
        uint32_t tri[3];
        load_tri(s, in, prims[k].second, tri);
        BvhTri& t = tin[k];
        memcpy(t.v0, s.gVertices[tri[0]].position, 12);
        memcpy(t.v1, s.gVertices[tri[1]].position, 12);
        memcpy(t.v2, s.gVertices[tri[2]].position, 12);
        t.id = (prims[k].second << 16) | (with_instance_bits ? prims[k].first : 0u);
        t.src_indices = in.indices_byte_offset + prims[k].second * 3u * in.stride;
        t.src_vertex = in.first_vertex | (in.stride == 4u ? 0x80000000u : 0u);
        bounds.grow(t.v0);
        bounds.grow(t.v1);
        bounds.grow(t.v2);
      }
      uint32_t root = 0;
      float ms = 0;
      const size_t nodes_before = out.nodes.size(), tris_before = out.tris.size();
      if (!lbvh_build_gpu(tin, out.nodes, out.tris, root, depth, ms, gpu_err)) return BVH_INVALID_REF;
      out.gpu_build_ms += ms;
      // The height of a radix tree is not bounded by log2(n): a run of equal Morton codes below a long common prefix
      // can reach ~96 levels. The traversal stack takes 1 KB of LDS per level and block; beyond what the LDS holds the
      // mesh is built with the (depth-capped) SAH builder instead.
      if (depth <= LBVH_MAX_HEIGHT) return root;
      out.nodes.resize(nodes_before);
      out.tris.resize(tris_before);
    }
    std::vector<Box> boxes(prims.size());
    bounds.reset();
    for (size_t k = 0; k < prims.size(); k++) {
      const InstView in = view(s.gInstances[prims[k].first]);
      uint32_t tri[3];
      load_tri(s, in, prims[k].second, tri);
      boxes[k].reset();
      for (int v = 0; v < 3; v++) boxes[k].grow(s.gVertices[tri[v]].position);
      bounds.grow(boxes[k]);
    }
    uint32_t leaf_tris = BVH_MAX_LEAF_TRIS;
    if (const char* e = getenv("STHIP_LEAF_TRIS")) leaf_tris = (uint32_t)std::min(4, std::max(1, atoi(e)));  // tuning experiments
    Builder b(boxes, leaf_tris, BLAS_DEPTH_CAP);
    depth = b.max_depth;
    const uint32_t tri_base = (uint32_t)out.tris.size();
    out.tris.resize(tri_base + prims.size());
    for (size_t k = 0; k < prims.size(); k++) {
      const auto& pr = prims[b.order[k]];
      const InstView in = view(s.gInstances[pr.first]);
      uint32_t tri[3];
      load_tri(s, in, pr.second, tri);
      BvhTri& t = out.tris[tri_base + k];
      memcpy(t.v0, s.gVertices[tri[0]].position, 12);
      memcpy(t.v1, s.gVertices[tri[1]].position, 12);
      memcpy(t.v2, s.gVertices[tri[2]].position, 12);
      t.id = (pr.second << 16) | (with_instance_bits ? pr.first : 0u);
      t.src_indices = in.indices_byte_offset + pr.second * 3u * in.stride;
      t.src_vertex = in.first_vertex | (in.stride == 4u ? 0x80000000u : 0u);
    }
    const uint32_t root = flatten(b, 0, out.nodes, [&](uint32_t first, uint32_t count) {
      if (!out.embedded) return BVH_LEAF_BIT | ((tri_base + first) << 2) | (count - 1);
      out.unit_tri.resize(out.nodes.size(), 0xFFFFFFFFu);
      const uint32_t unit = (uint32_t)out.nodes.size();
      for (uint32_t k = 0; k < count; k++) {
        out.nodes.push_back(BvhNode());
        memset(&out.nodes.back(), 0, sizeof(BvhNode));
        out.unit_tri.push_back(tri_base + first + k);
      }
      return BVH_LEAF_BIT | (unit << 2) | (count - 1);
    });
    if (root & BVH_LEAF_BIT) {  // a mesh of <= 4 triangles: give it a one-child root so kernels always start at an inner node
      depth = 1;
      return wrap_leaf(out.nodes, bounds, root);
    }
    return root;
  };

  Box scene_box;
  scene_box.reset();
  uint32_t blas_depth = 0;
  std::vector<Box> entry_boxes;

  std::vector<uint8_t> entry_root_local;  // device mode: the entry's root is an index into out.nodes (host-built), offset at the end
  if (!merged.empty()) {
    const size_t total = prim_total(merged);
    if (total > 0) {
      if (total >= (1u << 28)) {
        err = "too many triangles";
        return false;
      }
      Box bounds;
      uint32_t depth = 0;
      bool on_device = false;
      TlasEntry e;
      memset(&e, 0, sizeof(e));
      e.inv[0] = e.inv[5] = e.inv[10] = 1.0f;
      e.root = add_blas(merged, true, bounds, depth, on_device);
      if (e.root == BVH_INVALID_REF) {
        err = gpu_err;
        return false;
      }
      entry_root_local.push_back(on_device ? 0 : 1);
      e.id_bits = 0;
      e.identity = TLAS_ENTRY_IDENTITY;
      for (int a = 0; a < 3; a++) e.center[a] = 0.5f * (bounds.lo[a] + bounds.hi[a]);
      e.radius = 0.5f * sqrtf((bounds.hi[0] - bounds.lo[0]) * (bounds.hi[0] - bounds.lo[0]) + (bounds.hi[1] - bounds.lo[1]) * (bounds.hi[1] - bounds.lo[1]) +
                              (bounds.hi[2] - bounds.lo[2]) * (bounds.hi[2] - bounds.lo[2]));
      blas_depth = std::max(blas_depth, depth);
      out.entries.push_back(e);
      entry_boxes.push_back(bounds);
      scene_box.grow(bounds);
    }
  }
  struct MeshInfo {
    uint32_t root;
    Box bounds;
    bool on_device;
  };
  std::map<std::tuple<uint32_t, uint32_t, uint32_t, uint32_t>, MeshInfo> meshes;
  for (uint32_t i : separate) {
    const InstView in = view(s.gInstances[i]);
    if (in.prim_count == 0) continue;
    auto key = std::make_tuple(in.first_vertex, in.indices_byte_offset, in.prim_count, in.stride);
    auto it = meshes.find(key);
    if (it == meshes.end()) {
      MeshInfo mi;
      uint32_t depth = 0;
      mi.root = add_blas(std::vector<uint32_t>{i}, false, mi.bounds, depth, mi.on_device);
      if (mi.root == BVH_INVALID_REF) {
        err = gpu_err;
        return false;
      }
      blas_depth = std::max(blas_depth, depth);
      it = meshes.emplace(key, mi).first;
    }
    TlasEntry e;
    memset(&e, 0, sizeof(e));
    memcpy(e.inv, &s.gInstanceInverseTransforms[i], 48);
    e.root = it->second.root;
    e.id_bits = i;
    e.identity = TLAS_ENTRY_TRANSFORMED;
    const Box& ob = it->second.bounds;
    for (int a = 0; a < 3; a++) e.center[a] = 0.5f * (ob.lo[a] + ob.hi[a]);
    e.radius = 0.5f * sqrtf((ob.hi[0] - ob.lo[0]) * (ob.hi[0] - ob.lo[0]) + (ob.hi[1] - ob.lo[1]) * (ob.hi[1] - ob.lo[1]) + (ob.hi[2] - ob.lo[2]) * (ob.hi[2] - ob.lo[2]));
    // world box: the transformed vertices of the mesh, padded because the traversal ray reaches
    // object space through Minv while this box is placed through M (M*Minv is only ~identity)
    Box wb;
    wb.reset();
    const sthip_TransformData& M = s.gInstanceTransforms[i];
    if (it->second.on_device) {  // the triangles never came to the host: the corners of the object box (as rebuild_top_level does)
      for (int c = 0; c < 8; c++) {
        float q[3], w[3];
        for (int a = 0; a < 3; a++) q[a] = ((c >> a) & 1) ? ob.hi[a] : ob.lo[a];
        for (int r = 0; r < 3; r++) w[r] = M.m[r][0] * q[0] + M.m[r][1] * q[1] + M.m[r][2] * q[2] + M.m[r][3];
        wb.grow(w);
      }
    } else {
      for (uint32_t p = 0; p < in.prim_count; p++) {
        uint32_t tri[3];
        load_tri(s, in, p, tri);
        for (int v = 0; v < 3; v++) {
          const float* q = s.gVertices[tri[v]].position;
          float w[3];
          for (int r = 0; r < 3; r++) w[r] = M.m[r][0] * q[0] + M.m[r][1] * q[1] + M.m[r][2] * q[2] + M.m[r][3];
          wb.grow(w);
        }
      }
    }
    for (int a = 0; a < 3; a++) {
      const float m = std::max(std::max(fabsf(wb.lo[a]), fabsf(wb.hi[a])), wb.hi[a] - wb.lo[a]);
      wb.lo[a] -= 2e-5f * m;
      wb.hi[a] += 2e-5f * m;
    }
    out.entries.push_back(e);
    entry_root_local.push_back(it->second.on_device ? 0 : 1);
    entry_boxes.push_back(wb);
    scene_box.grow(wb);
  }

  // sphere instances (Scene.cpp:511-553): one top-level entry each, tested in place by the traversal. The box is the
  // sphere's world box, generously padded: sphere_test alone decides what is hit.
  for (uint32_t i : spheres) {
    TlasEntry e;
    memset(&e, 0, sizeof(e));
    memcpy(e.inv, &s.gInstanceInverseTransforms[i], 48);
    e.root = BVH_INVALID_REF;
    e.id_bits = i;
    e.identity = TLAS_ENTRY_SPHERE;
    memcpy(&e.radius, &s.gInstances[i].packed[2], 4);
    Box wb;
    const float r = fabsf(e.radius);
    for (int a = 0; a < 3; a++) {
      const float c = s.gInstanceTransforms[i].m[a][3];
      const float pad = 1e-3f * r + 1e-5f * fabsf(c);
      wb.lo[a] = c - r - pad;
      wb.hi[a] = c + r + pad;
    }
    out.entries.push_back(e);
    entry_boxes.push_back(wb);
    scene_box.grow(wb);
  }

  // volume instances (Scene.cpp:556-590): one top-level entry each, tested in place (volume_test, media.h). The box is
  // the world box of the grid's root bounding box, padded: volume_test alone decides what is hit.
  for (uint32_t i : volumes) {
    TlasEntry e;
    memset(&e, 0, sizeof(e));
    memcpy(e.inv, &s.gInstanceInverseTransforms[i], 48);
    e.root = s.gInstances[i].packed[2];
    e.id_bits = i;
    e.identity = TLAS_ENTRY_VOLUME;
    const DeviceVolume& v = out.volumes[e.root];
    const sthip_TransformData& xf = s.gInstanceTransforms[i];
    Box wb;
    wb.reset();
    for (int c = 0; c < 8; c++) {
      const float ip[3] = {(float)((c & 1) ? v.bbox_max[0] + 1 : v.bbox_min[0]), (float)((c & 2) ? v.bbox_max[1] + 1 : v.bbox_min[1]), (float)((c & 4) ? v.bbox_max[2] + 1 : v.bbox_min[2])};
      float gw[3], wp[3];
      for (int a = 0; a < 3; a++) gw[a] = ip[0] * v.matf[3 * a] + ip[1] * v.matf[3 * a + 1] + ip[2] * v.matf[3 * a + 2] + v.vecf[a];
      for (int a = 0; a < 3; a++) wp[a] = xf.m[a][0] * gw[0] + xf.m[a][1] * gw[1] + xf.m[a][2] * gw[2] + xf.m[a][3];
      wb.grow(wp);
    }
    for (int a = 0; a < 3; a++) {
      const float pad = 1e-3f * (wb.hi[a] - wb.lo[a]) + 1e-5f * std::max(fabsf(wb.lo[a]), fabsf(wb.hi[a]));
      wb.lo[a] -= pad;
      wb.hi[a] += pad;
    }
    out.entries.push_back(e);
    entry_boxes.push_back(wb);
    scene_box.grow(wb);
  }

  if (dev_mode)  // host-built bottom levels lie behind the device region
    for (size_t k = 0; k < entry_root_local.size(); k++)
      if (entry_root_local[k]) out.entries[k].root += out.dev_nodes;
  // ---- what a transforms-only update needs later ----
  out.top.entries = out.entries;
  out.top.volumes = out.volumes;
  out.top.instance_count = s.instance_count;
  out.top.merged.assign(s.instance_count, 0);
  for (uint32_t i : merged) out.top.merged[i] = 1;
  out.top.blas_nodes = out.dev_nodes + (uint32_t)out.nodes.size();
  out.top.blas_depth = blas_depth;
  out.top.obj_box.assign(6 * out.entries.size(), 0.f);
  for (size_t k = 0; k < out.entries.size(); k++) {
    const TlasEntry& e = out.entries[k];
    if (e.identity == TLAS_ENTRY_IDENTITY) {
      out.top.has_merged = true;
      memcpy(out.top.merged_box, entry_boxes[k].lo, 12);
      memcpy(out.top.merged_box + 3, entry_boxes[k].hi, 12);
    } else if (e.identity == TLAS_ENTRY_TRANSFORMED) {
      const InstView in = view(s.gInstances[e.id_bits]);
      const Box& ob = meshes.at(std::make_tuple(in.first_vertex, in.indices_byte_offset, in.prim_count, in.stride)).bounds;
      memcpy(&out.top.obj_box[6 * k], ob.lo, 12);
      memcpy(&out.top.obj_box[6 * k + 3], ob.hi, 12);
    }
  }

  // ---- top level ----
  uint32_t tlas_depth = 0;
  if (out.entries.empty()) {
    out.root_ref = BVH_INVALID_REF;
    out.top_is_world_blas = 1;
  } else if (out.entries.size() == 1 && out.entries[0].identity == TLAS_ENTRY_IDENTITY) {
    out.root_ref = out.entries[0].root;
    out.top_is_world_blas = 1;
  } else {
    if (out.entries.size() >= 0xFFFE) {  // 0xFFFE / 0xFFFF are the traversal's stack sentinels
      err = "too many top-level entries";
      return false;
    }
    Builder b(entry_boxes, 1, TLAS_DEPTH_CAP);
    tlas_depth = b.max_depth + 1;
    out.root_ref = flatten(b, 0, out.nodes, [&](uint32_t first, uint32_t) { return BVH_LEAF_BIT | BVH_INST_BIT | b.order[first]; });
    if (out.root_ref & BVH_LEAF_BIT) {
      out.root_ref = wrap_leaf(out.nodes, entry_boxes[0], out.root_ref);
      tlas_depth += 1;
    }
    out.top_is_world_blas = 0;
  }
  out.stack_depth = tlas_depth + blas_depth + 3;  // + the sentinels, + one spare level for the traversal's speculative push
  if (scene_box.lo[0] <= scene_box.hi[0]) {
    for (int a = 0; a < 3; a++) out.scene_center[a] = 0.5f * (scene_box.lo[a] + scene_box.hi[a]);
    out.scene_radius = 0.5f * sqrtf((scene_box.hi[0] - scene_box.lo[0]) * (scene_box.hi[0] - scene_box.lo[0]) + (scene_box.hi[1] - scene_box.lo[1]) * (scene_box.hi[1] - scene_box.lo[1]) +
                                    (scene_box.hi[2] - scene_box.lo[2]) * (scene_box.hi[2] - scene_box.lo[2]));
  }
  if (dev_mode) {  // out.nodes / out.tris start behind the device region: make their references final
    const uint32_t node_off = out.dev_nodes, tri_off = out.dev_tris << 2;
    out.absolute_ref.resize(2 * out.nodes.size(), 0);
    for (size_t i = 0; i < out.nodes.size(); i++)
      for (int c = 0; c < 2; c++) {
        BvhNode& nd = out.nodes[i];
        if (out.absolute_ref[2 * i + c]) continue;  // a reference into the device region (a SAH top's leaf)
        if (!(nd.ref[c] & BVH_LEAF_BIT))
          nd.ref[c] += node_off;
        else if (!(nd.ref[c] & BVH_INST_BIT))
          nd.ref[c] += tri_off;
      }
    if (!out.top_is_world_blas && out.root_ref != BVH_INVALID_REF) out.root_ref += node_off;  // the top level is host-built
  }
  out.any_alpha = any_alpha;
  if (out.embedded) {
    out.unit_tri.resize(out.nodes.size(), 0xFFFFFFFFu);
    if (out.nodes.size() >= (1u << 28)) {
      err = "too many triangles";
      return false;
    }
  }
  out.inst_alpha = inst_alpha;
  return true;
}

namespace {
struct WideChild {
  Box box;
  uint32_t ref;
};
inline Box child_box(const BvhNode& n, int c) {
  Box b;
  const float* xy = c == 0 ? n.n0xy : n.n1xy;
  b.lo[0] = xy[0];
  b.hi[0] = xy[1];
  b.lo[1] = xy[2];
  b.hi[1] = xy[3];
  b.lo[2] = n.nz[2 * c];
  b.hi[2] = n.nz[2 * c + 1];
  return b;
}
struct WideBuilder {
  const std::vector<BvhNode>& nodes;
  std::vector<WideNode>& out;
  std::vector<uint32_t> wide_of;   // binary inner node -> wide node made from it, or 0xFFFFFFFF
  std::vector<uint32_t> height_of; // of a wide node: levels of wide nodes below and including it
  bool ok = true;                  // false: some node's boxes do not fit the 8-bit grid (non-finite or astronomically large extents)
  const std::vector<TlasEntry>* entries = nullptr;  // set: a top-level leaf of the merged world-space mesh is replaced by that mesh's root
  WideBuilder(const std::vector<BvhNode>& n, std::vector<WideNode>& o) : nodes(n), out(o), wide_of(n.size(), 0xFFFFFFFFu) {}

  // The entry of the merged world-space mesh needs no change of space and no id bits (TLAS_ENTRY_IDENTITY: traverse.h), so in
  // the wide tree its top-level leaf IS the mesh's root: an inner reference the collapse may open like any other. A ray then
  // walks from the top level into the mesh without the stop at a leaf phase an instance entry costs.
  uint32_t resolve(uint32_t r) const {
    if (entries && (r & (BVH_LEAF_BIT | BVH_INST_BIT)) == (BVH_LEAF_BIT | BVH_INST_BIT) && r < 0xFFFFFFFEu) {
      const TlasEntry& e = (*entries)[r & 0xFFFFu];
      if (e.identity == TLAS_ENTRY_IDENTITY && !(e.root & BVH_LEAF_BIT)) return e.root;
    }
    return r;
  }

  // Bottom levels: which subtrees become the (up to four) children of a wide node is planned for the whole tree at once — the
  // cut that minimises the summed surface area of the wide nodes (Ylitie et al. 2017, section 3.1, for width 4): c(n, i) = the
  // least cost of the subtree of n as at most i children. (The largest-box-first rule below serves the top level and
  // STHIP_WIDE_GREEDY.)
  std::vector<float> cost;     // [3 * node + (i - 1)], i = 1..3
  std::vector<uint8_t> split;  // [4 * node + (j - 1)], j = 2..4: slots of the left child; 0 = a lone child
  static float area_of(const Box& b) {
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return (dx >= 0 && dy >= 0 && dz >= 0) ? dx * dy + dy * dz + dz * dx : 0.0f;
  }
  float ref_cost(uint32_t r, int i) const { return (r & BVH_LEAF_BIT) ? 0.0f : cost[3 * (size_t)r + (size_t)(i - 1)]; }  // (a leaf costs the same wherever it hangs)
  void plan(uint32_t root) {
    if (cost.empty()) {
      cost.assign(3 * nodes.size(), INFINITY);
      split.assign(4 * nodes.size(), 0);
    }
    std::vector<uint32_t> order, todo(1, root);
    while (!todo.empty()) {
      const uint32_t i = todo.back();
      todo.pop_back();
      order.push_back(i);
      for (int c = 0; c < 2; c++) {
        const uint32_t r = nodes[i].ref[c];
        if (r != BVH_INVALID_REF && !(r & BVH_LEAF_BIT) && !(c == 1 && r == nodes[i].ref[0])) todo.push_back(r);
      }
    }
    for (size_t k = order.size(); k-- > 0;) {
      const uint32_t i = order[k];
      const BvhNode& n = nodes[i];
      float* c = &cost[3 * (size_t)i];
      const bool lone = n.ref[1] == n.ref[0] || n.ref[1] == BVH_INVALID_REF || n.ref[0] == BVH_INVALID_REF;
      if (lone) {
        const uint32_t r = n.ref[0] == BVH_INVALID_REF ? n.ref[1] : n.ref[0];
        for (int j = 1; j <= 3; j++) c[j - 1] = ref_cost(r, j);
        continue;  // (split stays 0)
      }
      Box own = child_box(n, 0);
      own.grow(child_box(n, 1));
      float dist[5];
      for (int j = 2; j <= 4; j++) {
        float best = INFINITY;
        int bk = 1;
        for (int kk = 1; kk < j; kk++) {
          const float v = ref_cost(n.ref[0], std::min(kk, 3)) + ref_cost(n.ref[1], std::min(j - kk, 3));
          if (v < best) {
            best = v;
            bk = kk;
          }
        }
        dist[j] = best;
        split[4 * (size_t)i + (size_t)(j - 1)] = (uint8_t)bk;
      }
      c[0] = dist[4] + area_of(own);
      c[1] = std::min(dist[2], c[0]);
      c[2] = std::min(dist[3], c[1]);
    }
  }
  void place(const Box& box, uint32_t r, int slots, WideChild* ch, int& n) const {
    if (r & BVH_LEAF_BIT) {
      ch[n++] = WideChild{box, r};
      return;
    }
    slots = std::min(slots, 3);
    while (slots > 1 && cost[3 * (size_t)r + (size_t)(slots - 1)] == cost[3 * (size_t)r + (size_t)(slots - 2)]) slots--;
    if (slots == 1) {
      ch[n++] = WideChild{box, r};
      return;
    }
    distribute(r, slots, ch, n);
  }
  void distribute(uint32_t i, int j, WideChild* ch, int& n) const {
    const BvhNode& nd = nodes[i];
    const uint8_t k = split[4 * (size_t)i + (size_t)(j - 1)];
    if (k == 0) {
      const int c = nd.ref[0] == BVH_INVALID_REF ? 1 : 0;
      place(child_box(nd, c), nd.ref[c], j, ch, n);
      return;
    }
    place(child_box(nd, 0), nd.ref[0], k, ch, n);
    place(child_box(nd, 1), nd.ref[1], j - k, ch, n);
  }

  // the wide node that stands for binary inner node `i` (meshes shared by several entries are converted once)
  uint32_t convert(uint32_t i) {
    if (wide_of[i] != 0xFFFFFFFFu) return wide_of[i];
    WideChild ch[4];
    int n = 0;
    if (!entries && !cost.empty() && std::isfinite(cost[3 * (size_t)i])) {  // a planned bottom level
      distribute(i, 4, ch, n);
      return emit(i, ch, n);
    }
    for (int c = 0; c < 2; c++) {
      if (nodes[i].ref[c] == BVH_INVALID_REF) continue;
      const uint32_t r = resolve(nodes[i].ref[c]);
      if (n == 1 && r == ch[0].ref) continue;  // a wrapped lone leaf fills both slots: once is enough here
      ch[n].box = child_box(nodes[i], c);
      ch[n].ref = r;
      n++;
    }
    while (n < 4) {  // open the inner child with the largest box until there are four (or only leaves)
      int pick = -1;
      float area = -1.0f;
      for (int k = 0; k < n; k++)
        if (!(ch[k].ref & BVH_LEAF_BIT) && ch[k].box.half_area() > area) {
          area = ch[k].box.half_area();
          pick = k;
        }
      if (pick < 0) break;
      const BvhNode& b = nodes[ch[pick].ref];
      WideChild a0{child_box(b, 0), resolve(b.ref[0])}, a1{child_box(b, 1), resolve(b.ref[1])};
      if (a0.ref == a1.ref) {  // (a wrapped lone leaf)
        ch[pick] = a0;
        continue;
      }
      ch[pick] = a0;
      ch[n++] = a1;
    }
    return emit(i, ch, n);
  }
  uint32_t emit(uint32_t i, const WideChild* ch, int n) {
    const uint32_t w = (uint32_t)out.size();
    out.push_back(WideNode());
    wide_of[i] = w;
    height_of.push_back(1);
    uint32_t below = 0;
    uint32_t refs[4];
    for (int k = 0; k < n; k++) {
      refs[k] = ch[k].ref;
      if (!(ch[k].ref & BVH_LEAF_BIT)) {
        refs[k] = convert(ch[k].ref);
        below = std::max(below, height_of[refs[k]]);
      }
    }
    height_of[w] = 1 + below;
    WideNode wn;
    WideChildBox boxes[4];
    for (int k = 0; k < n; k++) {
      memcpy(boxes[k].lo, ch[k].box.lo, 12);
      memcpy(boxes[k].hi, ch[k].box.hi, 12);
    }
    if (!make_wide_node(boxes, refs, n, wn)) ok = false;  // (non-finite planes: no wide tree, the binary walk keeps its clamped ones)
    out[w] = wn;
    return w;
  }
};
}  // namespace

void build_wide_bvh(BuiltBvh& out) {
  out.wide_nodes.clear();
  out.wide_entries = out.entries;
  out.wide_root_ref = out.root_ref;
  out.wide_stack_depth = 0;
  if (out.dev_nodes || out.root_ref == BVH_INVALID_REF || out.nodes.empty()) return;
  out.wide_nodes.reserve(out.nodes.size() / 2 + 16);
  WideBuilder wb(out.nodes, out.wide_nodes);
  uint32_t blas_height = 0, top_height = 0;
  const bool greedy = getenv("STHIP_WIDE_GREEDY") != nullptr;  // (experiments: the largest-box-first collapse everywhere)
  for (TlasEntry& e : out.wide_entries)
    if (e.identity == TLAS_ENTRY_IDENTITY || e.identity == TLAS_ENTRY_TRANSFORMED) {
      if (!greedy && !(e.root & BVH_LEAF_BIT) && e.root < out.nodes.size() && wb.wide_of[e.root] == 0xFFFFFFFFu) wb.plan(e.root);
      e.root = wb.convert(e.root);
      blas_height = std::max(blas_height, wb.height_of[e.root]);
    }
  if (out.top_is_world_blas) {
    out.wide_root_ref = out.wide_entries[0].root;
  } else if (!(out.root_ref & BVH_LEAF_BIT)) {
    wb.entries = &out.entries;  // (the binary tree's: their roots index out.nodes)
    out.wide_root_ref = wb.convert(out.root_ref);
    top_height = wb.height_of[out.wide_root_ref];  // (with the merged mesh's levels where it is spliced in: an upper bound)
  }
  out.wide_stack_depth = 3 * (top_height + blas_height) + 4;  // three pushes per level at most, the sentinels, a spare level
  if (!wb.ok) {  // (boxes that do not fit the nodes' grid: no wide tree, the caller walks the binary one)
    out.wide_nodes.clear();
    out.wide_entries.clear();
    out.wide_root_ref = BVH_INVALID_REF;
    out.wide_stack_depth = 0;
  }
}

// ---- the 8-wide compressed form (bvh.h: Wide8Node) ----
namespace {
struct Wide8Child {
  Box box;
  uint32_t ref;     // binary reference; for a spliced bottom level: the wide8 node to copy
  uint8_t kind;     // 0 = inner (binary node to collapse), 1 = leaf (triangles / an entry), 2 = a bottom level's root spliced in
};
// Which child goes to which slot: the child lying toward corner (sx, sy, sz) of the node belongs in the slot with those bits,
// so that a ray meets the slots in the order of (slot ^ its octant) — greedy over cost(child, slot) = <centroid offset, corner
// direction>, largest first (Ylitie et al. 2017, section 4.2, with the greedy assignment instead of the auction).
void assign_slots(const Wide8Child* ch, int n, uint8_t* slot) {
  Box all;
  all.reset();
  for (int k = 0; k < n; k++) all.grow(ch[k].box);
  float cost[8][8];
  for (int k = 0; k < n; k++)
    for (int s = 0; s < 8; s++) {
      float c = 0;
      for (int a = 0; a < 3; a++) {
        const float d = 0.5f * (ch[k].box.lo[a] + ch[k].box.hi[a]) - 0.5f * (all.lo[a] + all.hi[a]);
        c += ((s >> a) & 1) ? d : -d;
      }
      cost[k][s] = c;
    }
  bool child_done[8] = {false, false, false, false, false, false, false, false}, slot_taken[8] = {false, false, false, false, false, false, false, false};
  for (int round = 0; round < n; round++) {
    int bk = -1, bs = -1;
    float best = -INFINITY;
    for (int k = 0; k < n; k++) {
      if (child_done[k]) continue;
      for (int s = 0; s < 8; s++)
        if (!slot_taken[s] && (bk < 0 || cost[k][s] > best)) {
          best = cost[k][s];
          bk = k;
          bs = s;
        }
    }
    child_done[bk] = true;
    slot_taken[bs] = true;
    slot[bk] = (uint8_t)bs;
  }
}

struct Wide8Builder {
  std::vector<Wide8Node>& out;
  bool ok = true;
  // bottom levels: the binary nodes and the triangle permutation
  const std::vector<BvhNode>* nodes = nullptr;
  const std::vector<BvhTri>* tris_in = nullptr;
  std::vector<BvhTri> tris_out;
  std::vector<uint32_t> new_index;  // old triangle index -> new
  // top level
  const TopLevelState* st = nullptr;
  const BvhNode* tlas = nullptr;
  uint32_t tlas_base = 0;
  std::vector<TlasEntry>* entries_out = nullptr;

  explicit Wide8Builder(std::vector<Wide8Node>& o) : out(o) {}

  const BvhNode& bin(uint32_t r, bool top) const { return top ? tlas[r - tlas_base] : (*nodes)[r]; }

  // the child a binary reference stands for (top level: the merged mesh's entry is its bottom level's root, spliced in)
  Wide8Child make_child(const Box& box, uint32_t r, bool top) const {
    Wide8Child c;
    c.box = box;
    c.ref = r;
    c.kind = (r & BVH_LEAF_BIT) ? 1 : 0;
    if (top && !(r & BVH_LEAF_BIT) && r < tlas_base) c.kind = 3;  // (never: top-level nodes only refer to top-level nodes and entries; fill() gives up)
    if (top && (r & (BVH_LEAF_BIT | BVH_INST_BIT)) == (BVH_LEAF_BIT | BVH_INST_BIT) && r < 0xFFFFFFFEu) {
      const uint32_t k = r & 0xFFFFu;
      if (st->entries[k].identity == TLAS_ENTRY_IDENTITY && st->wide8_root[k] != BVH_INVALID_REF) {
        c.kind = 2;
        c.ref = st->wide8_root[k];
      }
    }
    return c;
  }

  // ---- bottom levels: the collapse that minimises the surface-area cost of the wide tree over all ways of cutting this binary
  // tree into 8-wide nodes (Ylitie et al. 2017, section 3.1): c(n, i) = the least cost of representing the subtree of n as at
  // most i children of a wide node; a subtree of at most three triangles may become ONE leaf child (its binary leaves' triangles
  // follow one another in the new order); cost of a wide node = its area, of a leaf child = its area x triangles x W8_CPRIM.
  std::vector<float> cost;        // [7 * node + (i - 1)], i = 1..7
  std::vector<uint8_t> split;     // [8 * node + (j - 1)], j = 2..8: how many of j slots the left child gets (c_distribute)
  std::vector<uint8_t> as_leaf;   // [node]: with one slot the subtree is a leaf child
  std::vector<uint8_t> tri_count; // [node]: triangles below, saturating at 255
  float cprim = 0.4f;
  static float child_area(const BvhNode& n, int c) {
    const Box b = child_box(n, c);
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return (dx >= 0 && dy >= 0 && dz >= 0) ? dx * dy + dy * dz + dz * dx : 0.0f;
  }
  static float own_area(const BvhNode& n) {  // of the union of the child boxes
    Box b;
    b.reset();
    for (int c = 0; c < 2; c++)
      if (n.ref[c] != BVH_INVALID_REF) b.grow(child_box(n, c));
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return (dx >= 0 && dy >= 0 && dz >= 0) ? dx * dy + dy * dz + dz * dx : 0.0f;
  }
  float ref_cost(uint32_t r, float area, int i) const {  // c(child, i) of the child a reference stands for
    if (r & BVH_LEAF_BIT) return area * (float)((r & 3u) + 1u) * cprim;
    return cost[7 * (size_t)r + (size_t)(i - 1)];
  }
  uint32_t ref_tris(uint32_t r) const { return (r & BVH_LEAF_BIT) ? (r & 3u) + 1u : tri_count[r]; }
  void plan(uint32_t root) {  // fills cost / split / as_leaf for the subtree of binary inner node `root`
    std::vector<uint32_t> order, todo(1, root);
    while (!todo.empty()) {
      const uint32_t i = todo.back();
      todo.pop_back();
      order.push_back(i);
      const BvhNode& n = (*nodes)[i];
      for (int c = 0; c < 2; c++)
        if (n.ref[c] != BVH_INVALID_REF && !(n.ref[c] & BVH_LEAF_BIT) && !(c == 1 && n.ref[1] == n.ref[0])) todo.push_back(n.ref[c]);
    }
    for (size_t k = order.size(); k-- > 0;) {  // children before parents
      const uint32_t i = order[k];
      const BvhNode& n = (*nodes)[i];
      const bool lone = n.ref[1] == n.ref[0] || n.ref[1] == BVH_INVALID_REF;  // a wrapped lone leaf (or a single child)
      const uint32_t rl = n.ref[0] == BVH_INVALID_REF ? n.ref[1] : n.ref[0], rr = n.ref[1];
      const float al = child_area(n, n.ref[0] == BVH_INVALID_REF ? 1 : 0), ar = child_area(n, 1), area = own_area(n);
      float* c = &cost[7 * (size_t)i];
      if (lone) {
        tri_count[i] = (uint8_t)std::min(255u, ref_tris(rl));
        for (int j = 1; j <= 7; j++) c[j - 1] = ref_cost(rl, al, j);  // nothing to distribute: the node stands for its child
        as_leaf[i] = (rl & BVH_LEAF_BIT) ? 1 : as_leaf[rl];
        for (int j = 2; j <= 8; j++) split[8 * (size_t)i + (size_t)(j - 1)] = 0;  // 0: "pass through to the only child"
        continue;
      }
      const uint32_t tris = std::min(255u, ref_tris(rl) + ref_tris(rr));
      tri_count[i] = (uint8_t)tris;
      float dist[9];  // c_distribute(n, j), j = 2..8
      for (int j = 2; j <= 8; j++) {
        float best = INFINITY;
        int bk = 1;
        for (int kk = 1; kk < j; kk++) {
          const float v = ref_cost(rl, al, std::min(kk, 7)) + ref_cost(rr, ar, std::min(j - kk, 7));
          if (v < best) {
            best = v;
            bk = kk;
          }
        }
        dist[j] = best;
        split[8 * (size_t)i + (size_t)(j - 1)] = (uint8_t)bk;
      }
      const float c_leaf = tris <= 3 ? area * (float)tris * cprim : INFINITY;
      const float c_int = dist[8] + area;
      as_leaf[i] = c_leaf <= c_int ? 1 : 0;
      c[0] = std::min(c_leaf, c_int);
      for (int j = 2; j <= 7; j++) c[j - 1] = std::min(dist[j], c[j - 2]);
    }
  }
  // the children the plan gives binary inner node i when it gets j slots (appended to ch)
  void distribute(uint32_t i, int j, Wide8Child* ch, int& n) const {
    const BvhNode& nd = (*nodes)[i];
    const uint8_t k = split[8 * (size_t)i + (size_t)(j - 1)];
    if (k == 0) {  // a lone child
      const int c = nd.ref[0] == BVH_INVALID_REF ? 1 : 0;
      place(child_box(nd, c), nd.ref[c], j, ch, n);
      return;
    }
    place(child_box(nd, 0), nd.ref[0], k, ch, n);
    place(child_box(nd, 1), nd.ref[1], j - k, ch, n);
  }
  void place(const Box& box, uint32_t r, int slots, Wide8Child* ch, int& n) const {
    if (r & BVH_LEAF_BIT) {
      ch[n++] = Wide8Child{box, r, 1};
      return;
    }
    slots = std::min(slots, 7);
    while (slots > 1 && cost[7 * (size_t)r + (size_t)(slots - 1)] == cost[7 * (size_t)r + (size_t)(slots - 2)]) slots--;  // c(n, i) = c(n, i - 1): the extra slot buys nothing
    if (slots == 1) {
      ch[n++] = Wide8Child{box, r, (uint8_t)(as_leaf[r] ? 4 : 0)};  // 4: a subtree that becomes one leaf child
      return;
    }
    distribute(r, slots, ch, n);
  }
  // the triangles below a binary reference, in leaf order
  void subtree_triangles(uint32_t r, std::vector<uint32_t>& firsts_counts) const {
    if (r & BVH_LEAF_BIT) {
      firsts_counts.push_back(r);
      return;
    }
    const BvhNode& nd = (*nodes)[r];
    for (int c = 0; c < 2; c++) {
      if (nd.ref[c] == BVH_INVALID_REF || (c == 1 && nd.ref[1] == nd.ref[0])) continue;
      subtree_triangles(nd.ref[c], firsts_counts);
    }
  }

  // up to eight children of binary inner node i: the inner child with the largest box is opened until there are eight
  int gather(uint32_t i, bool top, Wide8Child* ch) const {
    if (!top && !cost.empty()) {
      int n = 0;
      distribute(i, 8, ch, n);
      return n;
    }
    int n = 0;
    const BvhNode& root = bin(i, top);
    for (int c = 0; c < 2; c++) {
      if (root.ref[c] == BVH_INVALID_REF) continue;
      if (n == 1 && root.ref[c] == root.ref[0]) continue;  // a wrapped lone leaf fills both slots
      ch[n++] = make_child(child_box(root, c), root.ref[c], top);
    }
    while (n < 8) {
      int pick = -1;
      float area = -1.0f;
      for (int k = 0; k < n; k++)
        if (ch[k].kind == 0 && ch[k].box.half_area() > area) {
          area = ch[k].box.half_area();
          pick = k;
        }
      if (pick < 0) break;
      const BvhNode& b = bin(ch[pick].ref, top);
      if (b.ref[0] == b.ref[1] || b.ref[1] == BVH_INVALID_REF) {
        ch[pick] = make_child(child_box(b, 0), b.ref[0], top);
        continue;
      }
      if (b.ref[0] == BVH_INVALID_REF) {
        ch[pick] = make_child(child_box(b, 1), b.ref[1], top);
        continue;
      }
      const Wide8Child a1 = make_child(child_box(b, 1), b.ref[1], top);
      ch[pick] = make_child(child_box(b, 0), b.ref[0], top);
      ch[n++] = a1;
    }
    return n;
  }

  // makes wide8 node w from binary inner node i; returns the height of the subtree (levels of wide8 nodes, a spliced bottom
  // level counted with blas_height)
  uint32_t fill(uint32_t w, uint32_t i, bool top, uint32_t blas_height) {
    Wide8Child ch[8];
    const int n = gather(i, top, ch);
    uint8_t slot[8];
    assign_slots(ch, n, slot);
    int at[8];  // slot -> child or -1
    for (int s = 0; s < 8; s++) at[s] = -1;
    for (int k = 0; k < n; k++) at[slot[k]] = k;
    Wide8Node wn;
    memset(&wn, 0, sizeof(wn));
    WideChildBox boxes[8];
    for (int k = 0; k < n; k++) {
      memcpy(boxes[k].lo, ch[k].box.lo, 12);
      memcpy(boxes[k].hi, ch[k].box.hi, 12);
    }
    if (n == 0 || !wide8_quantise(boxes, slot, n, wn)) ok = false;
    for (int k = 0; k < n; k++)
      if (ch[k].kind == 3) {
        ok = false;
        ch[k].kind = 1;
        ch[k].ref = BVH_LEAF_BIT | BVH_INST_BIT;
      }
    if (!ok) {  // (the caller drops the whole form)
      out[w] = wn;
      return 1;
    }
    auto is_leaf = [](const Wide8Child& c) { return c.kind == 1 || c.kind == 4; };
    uint32_t inner = 0;
    for (int k = 0; k < n; k++) inner += is_leaf(ch[k]) ? 0u : 1u;
    wn.child_base = (uint32_t)out.size();
    out.resize(out.size() + inner);
    uint32_t items = 0;
    if (top)
      wn.leaf_base = WIDE8_ENTRY_BIT | (uint32_t)entries_out->size();
    else
      wn.leaf_base = (uint32_t)tris_out.size();
    for (int s = 0; s < 8; s++) {
      if (at[s] < 0) continue;
      const Wide8Child& c = ch[at[s]];
      if (!is_leaf(c)) {
        wn.imask |= (uint8_t)(1u << s);
        wn.meta[s] = (uint8_t)(0x20u | (24u + (uint32_t)s));
        continue;
      }
      uint32_t count = 1;
      if (top) {  // a top-level entry: its bottom level starts at a wide8 node
        const uint32_t k = c.ref & 0xFFFFu;
        TlasEntry e = st->entries[k];
        if (e.identity == TLAS_ENTRY_IDENTITY || e.identity == TLAS_ENTRY_TRANSFORMED) {
          e.root = st->wide8_root[k];
          if (e.root == BVH_INVALID_REF) ok = false;
        }
        entries_out->push_back(e);
      } else {
        std::vector<uint32_t> leaves;  // the binary leaves this leaf child holds (one; several for a merged subtree)
        subtree_triangles(c.ref, leaves);
        count = 0;
        for (uint32_t r : leaves) count += (r & 3u) + 1u;
        if (count > 3u) {
          ok = false;
          count = 1;
        } else {
          for (uint32_t r : leaves) {
            const uint32_t first = (r & 0x3FFFFFFFu) >> 2, cnt = (r & 3u) + 1u;
            if (first + cnt > tris_in->size() || new_index[first] != 0xFFFFFFFFu) {
              ok = false;
              break;
            }
            for (uint32_t t = 0; t < cnt; t++) {
              new_index[first + t] = (uint32_t)tris_out.size();
              tris_out.push_back((*tris_in)[first + t]);
            }
          }
        }
      }
      if (items + count > WIDE8_MAX_ITEMS) ok = false;
      wn.meta[s] = (uint8_t)((((1u << count) - 1u) << 5) | (items & 31u));
      items += count;
    }
    out[w] = wn;
    // (stack entries a walk can hold below this node: a group per level; entering an instance leaves the rest of the node
    // group, the rest of the entry group and the exit sentinel behind, then takes a group per level of its bottom level)
    uint32_t below = 0, rank = 0;
    if (top)
      for (int s = 0; s < 8; s++)
        if (at[s] >= 0 && ch[at[s]].kind == 1) {
          const uint32_t k = ch[at[s]].ref & 0xFFFFu;
          const TlasEntry& e = st->entries[k];
          if (e.identity == TLAS_ENTRY_IDENTITY || e.identity == TLAS_ENTRY_TRANSFORMED) below = std::max(below, 2u + (k < st->wide8_height.size() ? st->wide8_height[k] : blas_height));
        }
    for (int s = 0; s < 8; s++) {
      if (at[s] < 0 || is_leaf(ch[at[s]])) continue;
      const Wide8Child& c = ch[at[s]];
      const uint32_t cw = wn.child_base + rank++;
      if (c.kind == 2) {  // the merged mesh's root, copied: its own children and items stay where they are
        out[cw] = out[c.ref];
        uint32_t h = blas_height;
        for (size_t k = 0; k < st->wide8_root.size(); k++)
          if (st->wide8_root[k] == c.ref && k < st->wide8_height.size()) h = st->wide8_height[k];
        below = std::max(below, h);
      } else {
        below = std::max(below, fill(cw, c.ref, top, blas_height));
      }
    }
    return 1 + below;
  }
};
}  // namespace

void build_wide8_bvh(BuiltBvh& out) {
  out.wide8_nodes.clear();
  out.wide8_entries.clear();
  out.wide8_root = BVH_INVALID_REF;
  out.wide8_stack_depth = 0;
  out.top.wide8_root.assign(out.entries.size(), BVH_INVALID_REF);
  out.top.wide8_height.assign(out.entries.size(), 0);
  out.top.wide8_blas_nodes = 0;
  out.top.wide8_blas_height = 0;
  if (out.dev_nodes || out.embedded || out.root_ref == BVH_INVALID_REF || out.nodes.empty()) return;
  std::vector<Wide8Node> nodes;
  nodes.reserve(out.nodes.size() / 3 + 16);
  Wide8Builder wb(nodes);
  wb.nodes = &out.nodes;
  wb.tris_in = &out.tris;
  wb.tris_out.reserve(out.tris.size());
  wb.new_index.assign(out.tris.size(), 0xFFFFFFFFu);
  const bool greedy = getenv("STHIP_W8_GREEDY") != nullptr;  // (experiments: the largest-box-first collapse instead of the planned one)
  if (const char* e = getenv("STHIP_W8_CPRIM")) wb.cprim = (float)atof(e);
  if (!greedy) {
    wb.cost.assign(7 * out.nodes.size(), INFINITY);
    wb.split.assign(8 * out.nodes.size(), 0);
    wb.as_leaf.assign(out.nodes.size(), 0);
    wb.tri_count.assign(out.nodes.size(), 0);
  }
  // bottom levels: one per binary root (instances of a shared mesh share theirs)
  std::unordered_map<uint32_t, std::pair<uint32_t, uint32_t>> of_root;  // binary root -> (wide8 root, height)
  uint32_t blas_height = 0;
  for (size_t k = 0; k < out.entries.size(); k++) {
    const TlasEntry& e = out.entries[k];
    if (e.identity != TLAS_ENTRY_IDENTITY && e.identity != TLAS_ENTRY_TRANSFORMED) continue;
    if (e.root == BVH_INVALID_REF || (e.root & BVH_LEAF_BIT) || e.root >= out.nodes.size()) return;
    auto it = of_root.find(e.root);
    if (it == of_root.end()) {
      const uint32_t w = (uint32_t)nodes.size();
      nodes.push_back(Wide8Node());
      if (!greedy) wb.plan(e.root);
      const uint32_t h = wb.fill(w, e.root, false, 0);
      it = of_root.emplace(e.root, std::make_pair(w, h)).first;
    }
    out.top.wide8_root[k] = it->second.first;
    out.top.wide8_height[k] = it->second.second;
    blas_height = std::max(blas_height, it->second.second);
  }
  if (!wb.ok || nodes.size() >= (1u << 24)) return;
  out.top.wide8_blas_nodes = (uint32_t)nodes.size();
  out.top.wide8_blas_height = blas_height;
  // the permutation is complete only if every triangle was reached (it is: every triangle lies in a leaf of some bottom
  // level); anything else keeps its place behind the others
  for (size_t t = 0; t < out.tris.size(); t++)
    if (wb.new_index[t] == 0xFFFFFFFFu) {
      wb.new_index[t] = (uint32_t)wb.tris_out.size();
      wb.tris_out.push_back(out.tris[t]);
    }
  std::vector<TlasEntry> entries;
  uint32_t root = BVH_INVALID_REF, depth = 0;
  // (the top level's binary nodes are the ones behind the bottom levels': out.nodes itself serves as `tlas` with base 0 —
  // every reference of a top-level node to another one is >= out.top.blas_nodes, which build_wide8_top takes as its base)
  if (!build_wide8_top(out.top, out.nodes.data() + out.top.blas_nodes, out.top.blas_nodes, out.root_ref, out.top_is_world_blas != 0, nodes, entries, root, depth)) {
    out.top.wide8_root.assign(out.entries.size(), BVH_INVALID_REF);
    out.top.wide8_blas_nodes = 0;
    return;
  }
  // commit: the triangles in their new order, the binary leaves pointing at them
  for (BvhNode& nd : out.nodes)
    for (int c = 0; c < 2; c++) {
      const uint32_t r = nd.ref[c];
      if ((r & (BVH_LEAF_BIT | BVH_INST_BIT)) != BVH_LEAF_BIT) continue;
      nd.ref[c] = BVH_LEAF_BIT | (wb.new_index[(r & 0x3FFFFFFFu) >> 2] << 2) | (r & 3u);
    }
  out.tris.swap(wb.tris_out);
  out.wide8_nodes.swap(nodes);
  out.wide8_entries.swap(entries);
  out.wide8_root = root;
  out.wide8_stack_depth = depth;
}

bool build_wide8_top(const TopLevelState& st, const BvhNode* tlas, uint32_t tlas_base, uint32_t root_ref, bool top_is_world_blas, std::vector<Wide8Node>& nodes, std::vector<TlasEntry>& entries,
                     uint32_t& wide8_root, uint32_t& wide8_stack_depth) {
  entries.clear();
  nodes.resize(st.wide8_blas_nodes);
  wide8_root = BVH_INVALID_REF;
  wide8_stack_depth = 0;
  if (st.wide8_root.size() != st.entries.size() || root_ref == BVH_INVALID_REF) return false;
  if (top_is_world_blas) {  // no top level: the walk starts at the merged mesh's root
    if (st.entries.empty() || st.wide8_root[0] == BVH_INVALID_REF) return false;
    entries = st.entries;
    for (size_t k = 0; k < entries.size(); k++)
      if (st.wide8_root[k] != BVH_INVALID_REF) entries[k].root = st.wide8_root[k];
    wide8_root = st.wide8_root[0];
    wide8_stack_depth = (st.wide8_height.empty() ? st.wide8_blas_height : st.wide8_height[0]) + 2;  // a group per level at most, the sentinel at the bottom, the spare slot of the speculative push
    return true;
  }
  if ((root_ref & BVH_LEAF_BIT) || root_ref < tlas_base) return false;
  Wide8Builder wb(nodes);
  wb.st = &st;
  wb.tlas = tlas;
  wb.tlas_base = tlas_base;
  wb.entries_out = &entries;
  const uint32_t w = (uint32_t)nodes.size();
  nodes.push_back(Wide8Node());
  const uint32_t top_height = wb.fill(w, root_ref, true, st.wide8_blas_height);
  if (!wb.ok || nodes.size() >= (1u << 24) || entries.size() >= 0x7FFFFFFFu) return false;
  wide8_root = w;
  // (fill() has counted what lies below every node, instances and the spliced mesh included) + the sentinel at the bottom and
  // the spare slot of the speculative push
  wide8_stack_depth = top_height + 2;
  return true;
}

bool rebuild_top_level(TopLevelState& st, const sthip_TransformData* xf, const sthip_TransformData* inv, uint32_t instance_count, std::vector<BvhNode>& tlas_nodes, uint32_t& root_ref,
                       uint32_t& top_is_world_blas, uint32_t& stack_depth, float scene_center[3], float& scene_radius, std::string& err) {
  const uint32_t TLAS_DEPTH_CAP = 18;
  if (instance_count != st.instance_count) {
    err = "the instance count changed: upload the scene again";
    return false;
  }
  for (uint32_t i = 0; i < instance_count; i++)
    if (st.merged[i] && !(is_identity(xf[i]) && is_identity(inv[i]))) {
      err = "an instance of the merged world-space mesh (identity transform at upload) moved: upload the scene again";
      return false;
    }
  Box scene_box;
  scene_box.reset();
  std::vector<Box> entry_boxes(st.entries.size());
  for (size_t k = 0; k < st.entries.size(); k++) {
    TlasEntry& e = st.entries[k];
    Box wb;
    wb.reset();
    if (e.identity == TLAS_ENTRY_IDENTITY) {
      memcpy(wb.lo, st.merged_box, 12);
      memcpy(wb.hi, st.merged_box + 3, 12);
    } else {
      const uint32_t i = e.id_bits;
      memcpy(e.inv, &inv[i], 48);
      const sthip_TransformData& M = xf[i];
      if (e.identity == TLAS_ENTRY_SPHERE) {
        const float r = fabsf(e.radius);
        for (int a = 0; a < 3; a++) {
          const float c = M.m[a][3];
          const float pad = 1e-3f * r + 1e-5f * fabsf(c);
          wb.lo[a] = c - r - pad;
          wb.hi[a] = c + r + pad;
        }
      } else {
        float corners[8][3];
        if (e.identity == TLAS_ENTRY_VOLUME) {
          const DeviceVolume& v = st.volumes[e.root];
          for (int c = 0; c < 8; c++) {
            const float ip[3] = {(float)((c & 1) ? v.bbox_max[0] + 1 : v.bbox_min[0]), (float)((c & 2) ? v.bbox_max[1] + 1 : v.bbox_min[1]), (float)((c & 4) ? v.bbox_max[2] + 1 : v.bbox_min[2])};
            for (int a = 0; a < 3; a++) corners[c][a] = ip[0] * v.matf[3 * a] + ip[1] * v.matf[3 * a + 1] + ip[2] * v.matf[3 * a + 2] + v.vecf[a];
          }
        } else {  // a transformed mesh: the corners of its object-space bounds (conservative; the full build uses every vertex)
          const float* ob = &st.obj_box[6 * k];
          for (int c = 0; c < 8; c++)
            for (int a = 0; a < 3; a++) corners[c][a] = ((c >> a) & 1) ? ob[3 + a] : ob[a];
        }
        for (int c = 0; c < 8; c++) {
          float w[3];
          for (int r = 0; r < 3; r++) w[r] = M.m[r][0] * corners[c][0] + M.m[r][1] * corners[c][1] + M.m[r][2] * corners[c][2] + M.m[r][3];
          wb.grow(w);
        }
        for (int a = 0; a < 3; a++) {
          const float m = std::max(std::max(fabsf(wb.lo[a]), fabsf(wb.hi[a])), wb.hi[a] - wb.lo[a]);
          const float pad = e.identity == TLAS_ENTRY_VOLUME ? 1e-3f * (wb.hi[a] - wb.lo[a]) + 1e-5f * std::max(fabsf(wb.lo[a]), fabsf(wb.hi[a])) : 2e-5f * m;
          wb.lo[a] -= pad;
          wb.hi[a] += pad;
        }
      }
    }
    entry_boxes[k] = wb;
    scene_box.grow(wb);
  }
  tlas_nodes.clear();
  uint32_t tlas_depth = 0;
  if (st.entries.empty()) {
    root_ref = BVH_INVALID_REF;
    top_is_world_blas = 1;
  } else if (st.entries.size() == 1 && st.entries[0].identity == TLAS_ENTRY_IDENTITY) {
    root_ref = st.entries[0].root;
    top_is_world_blas = 1;
  } else {
    Builder b(entry_boxes, 1, TLAS_DEPTH_CAP);
    tlas_depth = b.max_depth + 1;
    root_ref = flatten(b, 0, tlas_nodes, [&](uint32_t first, uint32_t) { return BVH_LEAF_BIT | BVH_INST_BIT | b.order[first]; });
    if (root_ref & BVH_LEAF_BIT) {
      root_ref = wrap_leaf(tlas_nodes, entry_boxes[0], root_ref);
      tlas_depth += 1;
    }
    for (BvhNode& nd : tlas_nodes)  // the nodes go behind the bottom levels
      for (int c = 0; c < 2; c++)
        if (!(nd.ref[c] & BVH_LEAF_BIT)) nd.ref[c] += st.blas_nodes;
    root_ref += st.blas_nodes;
    top_is_world_blas = 0;
  }
  stack_depth = tlas_depth + st.blas_depth + 3;
  if (scene_box.lo[0] <= scene_box.hi[0]) {
    for (int a = 0; a < 3; a++) scene_center[a] = 0.5f * (scene_box.lo[a] + scene_box.hi[a]);
    scene_radius = 0.5f * sqrtf((scene_box.hi[0] - scene_box.lo[0]) * (scene_box.hi[0] - scene_box.lo[0]) + (scene_box.hi[1] - scene_box.lo[1]) * (scene_box.hi[1] - scene_box.lo[1]) +
                                (scene_box.hi[2] - scene_box.lo[2]) * (scene_box.hi[2] - scene_box.lo[2]));
  }
  return true;
}

namespace {
inline float child_area(const BvhNode& n, int c) {
  const float dx = c == 0 ? n.n0xy[1] - n.n0xy[0] : n.n1xy[1] - n.n1xy[0];
  const float dy = c == 0 ? n.n0xy[3] - n.n0xy[2] : n.n1xy[3] - n.n1xy[2];
  const float dz = c == 0 ? n.nz[1] - n.nz[0] : n.nz[3] - n.nz[2];
  if (!(dx >= 0 && dy >= 0 && dz >= 0)) return 0.0f;  // inverted (empty) box
  return dx * dy + dy * dz + dz * dx;
}
inline float own_area(const BvhNode& n) {  // of the union of the two child boxes
  const float lo[3] = {std::min(n.n0xy[0], n.n1xy[0]), std::min(n.n0xy[2], n.n1xy[2]), std::min(n.nz[0], n.nz[2])};
  const float hi[3] = {std::max(n.n0xy[1], n.n1xy[1]), std::max(n.n0xy[3], n.n1xy[3]), std::max(n.nz[1], n.nz[3])};
  const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
  if (!(dx >= 0 && dy >= 0 && dz >= 0)) return 0.0f;
  return dx * dy + dy * dz + dz * dx;
}
}  // namespace

void build_treetop(const BvhNode* nodes, size_t node_count, const std::vector<TlasEntry>& entries, uint32_t root_ref, uint32_t capacity, Treetop& out) {
  out.nodes.clear();
  out.entries = entries;
  out.root_ref = root_ref;
  if (capacity == 0 || root_ref == BVH_INVALID_REF || (root_ref & BVH_LEAF_BIT) || root_ref >= node_count) return;
  // greedy: always take the candidate with the largest (world-space) box area — the chance that a ray visits a node
  // grows with it. `scale` turns an object-space area of a transformed bottom level into a world-space one.
  struct Cand {
    float priority;
    uint32_t node;
    float scale;
    bool operator<(const Cand& o) const { return priority < o.priority || (priority == o.priority && node > o.node); }
  };
  std::priority_queue<Cand> heap;
  std::unordered_map<uint32_t, uint32_t> slot;  // node index -> treetop index
  std::vector<uint32_t> taken;
  heap.push({std::numeric_limits<float>::infinity(), root_ref, 1.0f});
  while (!heap.empty() && taken.size() < capacity) {
    const Cand c = heap.top();
    heap.pop();
    if (slot.count(c.node)) continue;  // a bottom level shared by several entries
    slot.emplace(c.node, (uint32_t)taken.size());
    taken.push_back(c.node);
    const BvhNode& n = nodes[c.node];
    for (int k = 0; k < 2; k++) {
      const uint32_t ref = n.ref[k];
      const float area = child_area(n, k) * c.scale;
      if (!(ref & BVH_LEAF_BIT)) {
        if (ref < node_count) heap.push({area, ref, c.scale});
      } else if ((ref & BVH_INST_BIT) && ref < 0xFFFFFFFEu) {
        const uint32_t e = ref & 0xFFFFu;
        if (e >= entries.size()) continue;
        const TlasEntry& en = entries[e];
        if (en.identity != TLAS_ENTRY_TRANSFORMED && en.identity != TLAS_ENTRY_IDENTITY) continue;
        if ((en.root & BVH_LEAF_BIT) || en.root >= node_count) continue;
        const float obj = own_area(nodes[en.root]);
        const float sc = en.identity == TLAS_ENTRY_IDENTITY ? 1.0f : (obj > 0 ? area / obj : 0.0f);
        heap.push({area, en.root, sc});
      }
    }
  }
  out.nodes.resize(taken.size());
  for (size_t t = 0; t < taken.size(); t++) {
    BvhNode n = nodes[taken[t]];
    for (int k = 0; k < 2; k++)
      if (!(n.ref[k] & BVH_LEAF_BIT)) {
        auto it = slot.find(n.ref[k]);
        if (it != slot.end()) n.ref[k] = BVH_TOP_BIT | it->second;
      }
    out.nodes[t] = n;
  }
  for (TlasEntry& en : out.entries)
    if ((en.identity == TLAS_ENTRY_TRANSFORMED || en.identity == TLAS_ENTRY_IDENTITY) && !(en.root & BVH_LEAF_BIT)) {
      auto it = slot.find(en.root);
      if (it != slot.end()) en.root = BVH_TOP_BIT | it->second;
    }
  out.root_ref = BVH_TOP_BIT | 0u;
}

}  // namespace sthip
