// wide.hip — the 4-wide form of the acceleration structure (bvh.h: WideNode), made on the device from the packed binary
// nodes as they lie in HBM.
//
// The reference rebuilds its acceleration structures on the GPU whenever the scene is dirty and refits the top level when
// instances move (src/Node/Scene.cpp:345,435-459,614-629; src/Core/AccelerationStructure.cpp:5-27). Here the trees the GPU
// builder makes (lbvh.hip) never visit the host, and a transforms-only update replaces the top level only: in both cases
// the wide form k_trace walks is derived right here, so that a rebuilt or moved scene keeps the walk (and the speed) of a
// freshly uploaded one.
//
// Breadth first from the root: a work item is a binary inner node that becomes a wide node. It gathers up to four children
// by opening the inner child with the largest box until there are four (build_wide_bvh's rule), quantises their boxes
// (make_wide_node, shared with the host builder) and writes the wide node at the slot of ITS binary node in a sparse array;
// inner children are claimed with an atomic exchange on their mark and appended to the next level's queue. The entry of the
// merged world-space mesh is spliced in (its top-level leaf becomes the mesh's root); the root of a transformed instance
// is queued when its top-level leaf is met. Then the marks are scanned and the used slots move to a compact array in
// index order with their references renumbered: the result does not depend on the order the atomics resolved in.
#include <hip/hip_runtime.h>

#include <hipcub/hipcub.hpp>

#include "bvh_build.h"

namespace sthip {

struct DeviceWideScratch {
  WideNode* sparse = nullptr;
  uint32_t* mark = nullptr;   // 1: this binary node is the root of a wide node
  uint32_t* index = nullptr;  // exclusive scan of mark
  uint32_t* queue[2] = {nullptr, nullptr};
  uint32_t* counts = nullptr;  // per level, + [levels]: failure flag
  uint32_t* readback = nullptr;  // pinned
  void* cub_tmp = nullptr;
  size_t cub_bytes = 0;
  size_t node_capacity = 0, level_capacity = 0;
  hipEvent_t ev[2] = {nullptr, nullptr};
};

DeviceWideScratch* device_wide_scratch_create() { return new DeviceWideScratch(); }
void device_wide_scratch_destroy(DeviceWideScratch* s) {
  if (!s) return;
  (void)hipFree(s->sparse);
  (void)hipFree(s->mark);
  (void)hipFree(s->index);
  (void)hipFree(s->queue[0]);
  (void)hipFree(s->queue[1]);
  (void)hipFree(s->counts);
  (void)hipFree(s->cub_tmp);
  if (s->readback) (void)hipHostFree(s->readback);
  for (int k = 0; k < 2; k++)
    if (s->ev[k]) (void)hipEventDestroy(s->ev[k]);
  delete s;
}

namespace {

struct WChild {
  WideChildBox box;
  uint32_t ref;
};

// the two children of a packed node: planes as they are loaded (the reference bytes in their low mantissa byte included:
// pack_plane has rounded every plane outward far enough for any byte), references from those bytes
__device__ inline void unpack_children(const BvhNodePacked& n, WChild c[2]) {
  uint32_t w0[4], w1[4];
  memcpy(w0, n.n0xy, 16);
  memcpy(w1, n.n1xy, 16);
  c[0].box.lo[0] = n.n0xy[0];
  c[0].box.hi[0] = n.n0xy[1];
  c[0].box.lo[1] = n.n0xy[2];
  c[0].box.hi[1] = n.n0xy[3];
  c[0].box.lo[2] = n.nz[0];
  c[0].box.hi[2] = n.nz[1];
  c[1].box.lo[0] = n.n1xy[0];
  c[1].box.hi[0] = n.n1xy[1];
  c[1].box.lo[1] = n.n1xy[2];
  c[1].box.hi[1] = n.n1xy[3];
  c[1].box.lo[2] = n.nz[2];
  c[1].box.hi[2] = n.nz[3];
  c[0].ref = (w0[0] & 0xFFu) | ((w0[1] & 0xFFu) << 8) | ((w0[2] & 0xFFu) << 16) | ((w0[3] & 0xFFu) << 24);
  c[1].ref = (w1[0] & 0xFFu) | ((w1[1] & 0xFFu) << 8) | ((w1[2] & 0xFFu) << 16) | ((w1[3] & 0xFFu) << 24);
}
__device__ inline float half_area(const WideChildBox& b) {
  const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
  return dx * dy + dy * dz + dz * dx;
}
__device__ inline bool is_entry_leaf(uint32_t r) { return (r & (BVH_LEAF_BIT | BVH_INST_BIT)) == (BVH_LEAF_BIT | BVH_INST_BIT) && r < 0xFFFFFFFEu; }

__global__ void __launch_bounds__(256) k_wide_level(const BvhNodeSlot* nodes, uint32_t node_count, const TlasEntry* entries, uint32_t entry_count, uint32_t* mark, const uint32_t* queue_in,
                                                    const uint32_t* count_in, uint32_t* queue_out, uint32_t* count_out, WideNode* sparse, uint32_t* fail) {
  const uint32_t n_items = *count_in;
  for (uint32_t item = blockIdx.x * blockDim.x + threadIdx.x; item < n_items; item += gridDim.x * blockDim.x) {
    const uint32_t i = queue_in[item];
    // the merged world-space mesh needs no change of space: its top-level leaf is replaced by the mesh's root
    auto resolve = [&](uint32_t r) {
      if (is_entry_leaf(r) && (r & 0xFFFFu) < entry_count) {
        const TlasEntry& e = entries[r & 0xFFFFu];
        if (e.identity == TLAS_ENTRY_IDENTITY && !(e.root & BVH_LEAF_BIT)) return e.root;
      }
      return r;
    };
    WChild ch[4];
    int n = 0;
    {
      WChild two[2];
      unpack_children(nodes[i].n, two);
      for (int c = 0; c < 2; c++) {
        if (two[c].ref == BVH_INVALID_REF) continue;
        two[c].ref = resolve(two[c].ref);
        if (n == 1 && two[c].ref == ch[0].ref) continue;  // a wrapped lone leaf fills both slots: once is enough here
        ch[n++] = two[c];
      }
    }
    while (n < 4) {  // open the inner child with the largest box until there are four (or only leaves)
      int pick = -1;
      float area = -1.0f;
      for (int k = 0; k < n; k++)
        if (!(ch[k].ref & BVH_LEAF_BIT) && ch[k].ref < node_count && half_area(ch[k].box) > area) {
          area = half_area(ch[k].box);
          pick = k;
        }
      if (pick < 0) break;
      WChild two[2];
      unpack_children(nodes[ch[pick].ref].n, two);
      two[0].ref = resolve(two[0].ref);
      two[1].ref = resolve(two[1].ref);
      ch[pick] = two[0];
      if (two[0].ref == two[1].ref) continue;  // (a wrapped lone leaf)
      ch[n++] = two[1];
    }
    WideChildBox boxes[4];
    uint32_t refs[4];
    for (int k = 0; k < n; k++) {
      boxes[k] = ch[k].box;
      refs[k] = ch[k].ref;
      uint32_t next = BVH_INVALID_REF;  // the binary node that becomes a wide node of the next level
      if (!(refs[k] & BVH_LEAF_BIT)) {
        next = refs[k];
      } else if (is_entry_leaf(refs[k]) && (refs[k] & 0xFFFFu) < entry_count) {
        const TlasEntry& e = entries[refs[k] & 0xFFFFu];
        if (e.identity == TLAS_ENTRY_TRANSFORMED && !(e.root & BVH_LEAF_BIT)) next = e.root;
      }
      if (next != BVH_INVALID_REF) {
        if (next >= node_count) {
          atomicOr(fail, 2u);
        } else if (atomicExch(&mark[next], 1u) == 0u) {
          queue_out[atomicAdd(count_out, 1u)] = next;
        }
      }
    }
    WideNode wn;
    if (n == 0 || !make_wide_node(boxes, refs, n, wn)) {
      atomicOr(fail, 1u);
      continue;
    }
    sparse[i] = wn;
  }
}

// the used slots of the sparse array, in index order, with their inner references renumbered
__global__ void __launch_bounds__(256) k_wide_compact(const WideNode* sparse, const uint32_t* mark, const uint32_t* index, uint32_t node_count, WideNode* out) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < node_count; i += gridDim.x * blockDim.x) {
    if (!mark[i]) continue;
    WideNode wn = sparse[i];
    for (int k = 0; k < 4; k++)
      if (!(wn.ref[k] & BVH_LEAF_BIT)) wn.ref[k] = index[wn.ref[k]];
    out[index[i]] = wn;
  }
}
__global__ void k_wide_entries(const TlasEntry* entries, uint32_t entry_count, const uint32_t* mark, const uint32_t* index, uint32_t node_count, TlasEntry* out, uint32_t root_ref, uint32_t* result) {
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < entry_count; k += gridDim.x * blockDim.x) {
    TlasEntry e = entries[k];
    if ((e.identity == TLAS_ENTRY_IDENTITY || e.identity == TLAS_ENTRY_TRANSFORMED) && !(e.root & BVH_LEAF_BIT) && e.root < node_count) e.root = mark[e.root] ? index[e.root] : BVH_INVALID_REF;
    out[k] = e;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    result[0] = index[node_count - 1] + mark[node_count - 1];  // wide nodes
    result[1] = root_ref < node_count && mark[root_ref] ? index[root_ref] : BVH_INVALID_REF;
  }
}
__global__ void k_wide_seed(uint32_t* mark, uint32_t* queue, uint32_t* counts, uint32_t root_ref) {
  mark[root_ref] = 1u;
  queue[0] = root_ref;
  counts[0] = 1u;
}

}  // namespace

#define WIDE_TRY(x)                                   \
  do {                                                \
    const hipError_t e_ = (x);                        \
    if (e_ != hipSuccess) {                           \
      err = std::string(#x) + ": " + hipGetErrorString(e_); \
      return false;                                   \
    }                                                 \
  } while (0)

bool collapse_wide_device(DeviceWideScratch* s, const BvhNodeSlot* nodes, uint32_t node_count, const TlasEntry* entries, uint32_t entry_count, uint32_t root_ref, bool top_is_world_blas,
                          uint32_t max_levels, WideNode* wide_nodes, TlasEntry* wide_entries, void* stream_, DeviceWideResult& result, std::string& err) {
  (void)top_is_world_blas;  // (either way the walk starts at root_ref: the top level's root, or the merged mesh's when there is no top level)
  hipStream_t st = (hipStream_t)stream_;
  result = DeviceWideResult();
  if (!s || node_count == 0 || root_ref == BVH_INVALID_REF || (root_ref & BVH_LEAF_BIT) || root_ref >= node_count) {
    err = "collapse_wide_device: nothing to collapse";
    return false;
  }
  max_levels = std::max(max_levels, 2u) + 2u;
  if (node_count > s->node_capacity) {
    (void)hipFree(s->sparse);
    (void)hipFree(s->mark);
    (void)hipFree(s->index);
    (void)hipFree(s->queue[0]);
    (void)hipFree(s->queue[1]);
    s->sparse = nullptr;
    s->mark = s->index = s->queue[0] = s->queue[1] = nullptr;
    s->node_capacity = 0;
    WIDE_TRY(hipMalloc((void**)&s->sparse, (size_t)node_count * sizeof(WideNode)));
    WIDE_TRY(hipMalloc((void**)&s->mark, (size_t)node_count * 4));
    WIDE_TRY(hipMalloc((void**)&s->index, (size_t)node_count * 4));
    WIDE_TRY(hipMalloc((void**)&s->queue[0], (size_t)node_count * 4));
    WIDE_TRY(hipMalloc((void**)&s->queue[1], (size_t)node_count * 4));
    s->node_capacity = node_count;
  }
  if (max_levels + 2 > s->level_capacity) {
    (void)hipFree(s->counts);
    s->counts = nullptr;
    WIDE_TRY(hipMalloc((void**)&s->counts, (size_t)(max_levels + 2) * 4));
    s->level_capacity = max_levels + 2;
  }
  if (!s->readback) WIDE_TRY(hipHostMalloc((void**)&s->readback, (4 + 4096) * 4));
  for (int k = 0; k < 2; k++)
    if (!s->ev[k]) WIDE_TRY(hipEventCreate(&s->ev[k]));
  size_t cub_bytes = 0;
  WIDE_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, cub_bytes, s->mark, s->index, (int)node_count, st));
  if (cub_bytes > s->cub_bytes) {
    (void)hipFree(s->cub_tmp);
    s->cub_tmp = nullptr;
    WIDE_TRY(hipMalloc(&s->cub_tmp, cub_bytes));
    s->cub_bytes = cub_bytes;
  }
  uint32_t* fail = s->counts + max_levels + 1;
  WIDE_TRY(hipEventRecord(s->ev[0], st));
  WIDE_TRY(hipMemsetAsync(s->mark, 0, (size_t)node_count * 4, st));
  WIDE_TRY(hipMemsetAsync(s->counts, 0, (size_t)(max_levels + 2) * 4, st));
  hipLaunchKernelGGL(k_wide_seed, dim3(1), dim3(1), 0, st, s->mark, s->queue[0], s->counts, root_ref);
  const unsigned grid = (unsigned)std::min<size_t>(2048, ((size_t)node_count + 255) / 256);
  for (uint32_t level = 0; level < max_levels; level++)
    hipLaunchKernelGGL(k_wide_level, dim3(grid), dim3(256), 0, st, nodes, node_count, entries, entry_count, s->mark, s->queue[level & 1], s->counts + level, s->queue[(level + 1) & 1],
                       s->counts + level + 1, s->sparse, fail);
  WIDE_TRY(hipGetLastError());
  WIDE_TRY(hipcub::DeviceScan::ExclusiveSum(s->cub_tmp, cub_bytes, s->mark, s->index, (int)node_count, st));
  hipLaunchKernelGGL(k_wide_compact, dim3(grid), dim3(256), 0, st, s->sparse, s->mark, s->index, node_count, wide_nodes);
  uint32_t* dev_result = s->queue[0];  // (free again: the last level's queue has been read)
  hipLaunchKernelGGL(k_wide_entries, dim3(std::max(1u, (entry_count + 255) / 256)), dim3(256), 0, st, entries, entry_count, s->mark, s->index, node_count, wide_entries, root_ref, dev_result);
  WIDE_TRY(hipGetLastError());
  WIDE_TRY(hipEventRecord(s->ev[1], st));
  const uint32_t levels_read = std::min<uint32_t>(max_levels + 2, 4096);
  WIDE_TRY(hipMemcpyAsync(s->readback, dev_result, 8, hipMemcpyDeviceToHost, st));
  WIDE_TRY(hipMemcpyAsync(s->readback + 4, s->counts, (size_t)levels_read * 4, hipMemcpyDeviceToHost, st));
  WIDE_TRY(hipStreamSynchronize(st));
  (void)hipEventElapsedTime(&result.gpu_ms, s->ev[0], s->ev[1]);
  const uint32_t* counts = s->readback + 4;
  if (max_levels + 1 < levels_read && counts[max_levels + 1] != 0) {
    err = counts[max_levels + 1] & 2u ? "collapse_wide_device: a reference outside the node array" : "collapse_wide_device: boxes that do not fit the 8-bit grid";
    return false;
  }
  uint32_t levels = 0;
  while (levels < std::min(max_levels + 1, levels_read) && counts[levels] != 0) levels++;
  if (levels > max_levels || (levels < levels_read && levels == max_levels + 1)) {
    err = "collapse_wide_device: the tree is higher than its bound";
    return false;
  }
  result.node_count = s->readback[0];
  result.root_ref = s->readback[1];
  result.stack_depth = 3 * levels + 4;  // three pushes per level at most, the sentinels, a spare level
  if (result.root_ref == BVH_INVALID_REF || result.node_count == 0) {
    err = "collapse_wide_device: no root";
    return false;
  }
  return true;
}

}  // namespace sthip
