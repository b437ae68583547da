// lbvh.hip — GPU construction of a bottom-level BVH (LBVH): Morton codes of the triangle centroids,
// a device radix sort, the Karras (2012) binary radix tree built fully in parallel, a bottom-up refit, and the
// emission of the 64-byte nodes of bvh.h. This is the "rebuild on dirty" path: the reference rebuilds its
// acceleration structures on the GPU whenever the scene changes (src/Node/Scene.cpp:345,435-459,614-629);
// the binned-SAH host builder of bvh_build.cpp stays the default for static scenes because its trees trace
// faster. The hit contract does not depend on the tree, so both builders give bit-identical images.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <math.h>
#include <string.h>

#include <mutex>

#include "bvh_build.h"

namespace sthip {
namespace {

#define LB_BLOCK 256

struct LBox {
  float lo[3], hi[3];
};

// order-preserving float <-> uint mapping for atomicMin/atomicMax
__device__ __forceinline__ uint32_t f2ord(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ord2f(uint32_t u) {
  const uint32_t v = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
  float f;
  memcpy(&f, &v, 4);
  return f;
}

__global__ void k_lbvh_bounds(const BvhTri* tris, uint32_t n, LBox* boxes, uint32_t* cbounds /* 6 ordered uints */) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const BvhTri t = tris[i];
  LBox b;
  for (int a = 0; a < 3; a++) {
    b.lo[a] = fminf(fminf(t.v0[a], t.v1[a]), t.v2[a]);
    b.hi[a] = fmaxf(fmaxf(t.v0[a], t.v1[a]), t.v2[a]);
  }
  boxes[i] = b;
  for (int a = 0; a < 3; a++) {
    const float c = 0.5f * (b.lo[a] + b.hi[a]);
    atomicMin(&cbounds[a], f2ord(c));
    atomicMax(&cbounds[3 + a], f2ord(c));
  }
}

__device__ __forceinline__ unsigned long long expand21(unsigned long long v) {  // 21 bits -> every third bit
  v &= 0x1FFFFFull;
  v = (v | v << 32) & 0x1F00000000FFFFull;
  v = (v | v << 16) & 0x1F0000FF0000FFull;
  v = (v | v << 8) & 0x100F00F00F00F00Full;
  v = (v | v << 4) & 0x10C30C30C30C30C3ull;
  v = (v | v << 2) & 0x1249249249249249ull;
  return v;
}

__global__ void k_lbvh_morton(const LBox* boxes, uint32_t n, const uint32_t* cbounds, unsigned long long* keys, uint32_t* vals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long code = 0;
  for (int a = 0; a < 3; a++) {
    const float lo = ord2f(cbounds[a]), hi = ord2f(cbounds[3 + a]);
    const float c = 0.5f * (boxes[i].lo[a] + boxes[i].hi[a]);
    const float ext = hi - lo;
    float x = ext > 0 ? (c - lo) / ext : 0.0f;
    x = fminf(fmaxf(x * 2097152.0f, 0.0f), 2097151.0f);
    code |= expand21((unsigned long long)x) << (2 - a);
  }
  keys[i] = code;
  vals[i] = i;
}

// common-prefix length of the keys at sorted positions i and j; equal keys are told apart by position
__device__ __forceinline__ int delta(const unsigned long long* keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const unsigned long long a = keys[i], b = keys[j];
  if (a == b) return 64 + __clz((uint32_t)i ^ (uint32_t)j);
  return __clzll(a ^ b);
}

// Karras 2012, Algorithm: one thread per internal node; children >= n - 1 + ... are encoded as leaf | 0x80000000
__global__ void k_lbvh_hierarchy(const unsigned long long* keys, int n, uint32_t* left, uint32_t* right, uint32_t* parent_of_internal, uint32_t* parent_of_leaf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
  const int dmin = delta(keys, n, i, i - d);
  int lmax = 2;
  while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
  int l = 0;
  for (int t = lmax / 2; t >= 1; t /= 2)
    if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = delta(keys, n, i, j);
  int s = 0;
  int t = l;
  do {
    t = (t + 1) >> 1;
    if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
  } while (t > 1);
  const int gamma = i + s * d + min(d, 0);
  const int lo = min(i, j), hi = max(i, j);
  const uint32_t L = (lo == gamma) ? (0x80000000u | (uint32_t)gamma) : (uint32_t)gamma;
  const uint32_t R = (hi == gamma + 1) ? (0x80000000u | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
  left[i] = L;
  right[i] = R;
  if (L & 0x80000000u)
    parent_of_leaf[gamma] = (uint32_t)i;
  else
    parent_of_internal[gamma] = (uint32_t)i;
  if (R & 0x80000000u)
    parent_of_leaf[gamma + 1] = (uint32_t)i;
  else
    parent_of_internal[gamma + 1] = (uint32_t)i;
}

// One bottom-up pass: an internal node whose children were finished in an EARLIER launch (done < pass) takes
// the union of their boxes. Kernel boundaries order the writes, so no in-kernel fences are needed.
__global__ void k_lbvh_refit_pass(int n, uint32_t pass, const uint32_t* left, const uint32_t* right, const uint32_t* sorted, const LBox* leaf_boxes, LBox* node_boxes,
                                  uint32_t* done, uint32_t* remaining) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1 || done[i]) return;
  const uint32_t L = left[i], R = right[i];
  const bool lready = (L & 0x80000000u) || (done[L] && done[L] < pass);
  const bool rready = (R & 0x80000000u) || (done[R] && done[R] < pass);
  if (!(lready && rready)) {
    atomicAdd(remaining, 1u);
    return;
  }
  const LBox a = (L & 0x80000000u) ? leaf_boxes[sorted[L & 0x7FFFFFFFu]] : node_boxes[L];
  const LBox b = (R & 0x80000000u) ? leaf_boxes[sorted[R & 0x7FFFFFFFu]] : node_boxes[R];
  LBox u;
  for (int k = 0; k < 3; k++) {
    u.lo[k] = fminf(a.lo[k], b.lo[k]);
    u.hi[k] = fmaxf(a.hi[k], b.hi[k]);
  }
  node_boxes[i] = u;
  done[i] = pass;
}

__global__ void k_lbvh_emit(int n, uint32_t node_base, uint32_t tri_base, const uint32_t* left, const uint32_t* right, const uint32_t* sorted, const LBox* leaf_boxes,
                            const LBox* node_boxes, BvhNode* nodes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  const uint32_t c[2] = {left[i], right[i]};
  BvhNode out;
  memset(&out, 0, sizeof(out));
  for (int k = 0; k < 2; k++) {
    const bool leaf = c[k] & 0x80000000u;
    const uint32_t idx = c[k] & 0x7FFFFFFFu;
    const LBox b = leaf ? leaf_boxes[sorted[idx]] : node_boxes[idx];
    float* xy = k == 0 ? out.n0xy : out.n1xy;
    xy[0] = b.lo[0];
    xy[1] = b.hi[0];
    xy[2] = b.lo[1];
    xy[3] = b.hi[1];
    out.nz[2 * k] = b.lo[2];
    out.nz[2 * k + 1] = b.hi[2];
    if (leaf) {
      out.ref[k] = BVH_LEAF_BIT | ((tri_base + idx) << 2);  // one triangle
    } else {
      // An internal child whose own children are both leaves covers two CONSECUTIVE sorted triangles (a radix-tree node
      // spans a contiguous key range): reference it as one two-triangle leaf, as the SAH builder's leaves are
      // (BVH_MAX_LEAF_TRIS), and the ray saves a node visit at the bottom of every descent. The skipped node stays
      // in the array, unreferenced (the root is never skipped: kernels start at an inner node).
      const uint32_t gl = left[idx], gr = right[idx];
      if ((gl & 0x80000000u) && (gr & 0x80000000u))
        out.ref[k] = BVH_LEAF_BIT | ((tri_base + (gl & 0x7FFFFFFFu)) << 2) | 1u;
      else
        out.ref[k] = node_base + idx;
    }
  }
  nodes[i] = out;
}

__global__ void k_lbvh_gather(const BvhTri* in, const uint32_t* sorted, uint32_t n, BvhTri* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[sorted[i]];
}


// ---- the device-resident build (lbvh_build_device): triangles are fetched from the uploaded scene arrays ----

__device__ __forceinline__ float wave_min(float v) {
  for (int o = 32; o >= 1; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// One thread per triangle of the mesh: finds its piece, reads the three indices (scene.h:139-161) and positions, writes
// the BvhTri (unsorted), its box, and folds the centroid into the mesh's centroid bounds (one atomic per wave and
// plane, not per triangle: 6 M atomics on six addresses would take longer than the rest of the build).
__global__ void __launch_bounds__(LB_BLOCK) k_lbvh_fetch(const MeshPiece* pieces, uint32_t piece_count, const sthip_PackedVertexData* vertices, uint32_t vertex_count,
                                                         const uint8_t* indices, uint32_t n, BvhTri* tris, LBox* boxes, uint32_t* cbounds, uint32_t* error) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  float c_lo[3] = {INFINITY, INFINITY, INFINITY}, c_hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  if (i < n) {
    uint32_t lo = 0, hi = piece_count - 1;  // the last piece with prim_begin <= i
    while (lo < hi) {
      const uint32_t mid = (lo + hi + 1) >> 1;
      if (pieces[mid].prim_begin <= i)
        lo = mid;
      else
        hi = mid - 1;
    }
    const MeshPiece pc = pieces[lo];
    const uint32_t prim = i - pc.prim_begin;
    const uint8_t* ib = indices + (size_t)pc.indices_byte_offset + (size_t)prim * 3u * pc.stride;
    uint32_t idx[3];
    for (int k = 0; k < 3; k++) {  // byte loads: an index buffer's byte offset need not be aligned to its stride
      const uint8_t* q = ib + (size_t)k * pc.stride;
      idx[k] = pc.stride == 2 ? ((uint32_t)q[0] | (uint32_t)q[1] << 8) : ((uint32_t)q[0] | (uint32_t)q[1] << 8 | (uint32_t)q[2] << 16 | (uint32_t)q[3] << 24);
      idx[k] += pc.first_vertex;
      if (idx[k] >= vertex_count) {
        atomicOr(error, 1u);
        idx[k] = 0;
      }
    }
    BvhTri t;
    for (int a = 0; a < 3; a++) {
      t.v0[a] = vertices[idx[0]].position[a];
      t.v1[a] = vertices[idx[1]].position[a];
      t.v2[a] = vertices[idx[2]].position[a];
    }
    t.id = (prim << 16) | pc.id_bits;
    t.src_indices = pc.indices_byte_offset + prim * 3u * pc.stride;
    t.src_vertex = pc.first_vertex | (pc.stride == 4u ? 0x80000000u : 0u);
    tris[i] = t;
    LBox b;
    for (int a = 0; a < 3; a++) {
      b.lo[a] = fminf(fminf(t.v0[a], t.v1[a]), t.v2[a]);
      b.hi[a] = fmaxf(fmaxf(t.v0[a], t.v1[a]), t.v2[a]);
      c_lo[a] = c_hi[a] = 0.5f * (b.lo[a] + b.hi[a]);
    }
    boxes[i] = b;
  }
  for (int a = 0; a < 3; a++) {
    const float mn = wave_min(c_lo[a]), mx = wave_max(c_hi[a]);
    if ((threadIdx.x & 63u) == 0 && mn <= mx) {
      atomicMin(&cbounds[a], f2ord(mn));
      atomicMax(&cbounds[3 + a], f2ord(mx));
    }
  }
}

// Bottom-up boxes and heights in ONE launch: every leaf walks towards the root; at an internal node the first arrival
// stops, the second one (both subtrees are complete then) takes the union and goes on. The counter is an acquire-release
// atomic at device scope, which orders the box stores of one thread before the loads of the other. Unions are min / max:
// the result does not depend on who arrives first.
struct LNode {
  float lo[3], hi[3];
  uint32_t height, pad;
};
__global__ void __launch_bounds__(LB_BLOCK) k_lbvh_refit(int n, const uint32_t* left, const uint32_t* right, const uint32_t* parent_of_internal, const uint32_t* parent_of_leaf,
                                                          const uint32_t* sorted, const LBox* leaf_boxes, LNode* node_boxes, uint32_t* visits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t cur = parent_of_leaf[i];
  for (int guard = 0; guard < 4096; guard++) {  // a radix tree over 63-bit keys + 32 index bits is at most ~96 high
    const uint32_t earlier = __hip_atomic_fetch_add(&visits[cur], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (earlier == 0) return;
    const uint32_t L = left[cur], R = right[cur];
    LNode u;
    uint32_t hl = 0, hr = 0;
    LBox a, b;
    if (L & 0x80000000u) {
      a = leaf_boxes[sorted[L & 0x7FFFFFFFu]];
    } else {
      const LNode t = node_boxes[L];
      for (int k = 0; k < 3; k++) {
        a.lo[k] = t.lo[k];
        a.hi[k] = t.hi[k];
      }
      hl = t.height;
    }
    if (R & 0x80000000u) {
      b = leaf_boxes[sorted[R & 0x7FFFFFFFu]];
    } else {
      const LNode t = node_boxes[R];
      for (int k = 0; k < 3; k++) {
        b.lo[k] = t.lo[k];
        b.hi[k] = t.hi[k];
      }
      hr = t.height;
    }
    for (int k = 0; k < 3; k++) {
      u.lo[k] = fminf(a.lo[k], b.lo[k]);
      u.hi[k] = fmaxf(a.hi[k], b.hi[k]);
    }
    u.height = 1u + max(hl, hr);
    u.pad = 0;
    node_boxes[cur] = u;
    if (cur == 0) return;
    cur = parent_of_internal[cur];
  }
}

// k_lbvh_emit for the device-resident build: the node goes out twice, unpacked (what the host keeps for the treetop and
// top-level rebuilds) and packed into its slot of the array the kernels traverse (bvh_build.h: pack_node).
__global__ void __launch_bounds__(LB_BLOCK) k_lbvh_emit_packed(int n, uint32_t node_base, uint32_t tri_base, const uint32_t* left, const uint32_t* right, const uint32_t* sorted,
                                                                const LBox* leaf_boxes, const LNode* node_boxes, BvhNode* raw, BvhNodeSlot* packed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  const uint32_t c[2] = {left[i], right[i]};
  BvhNode out;
  memset(&out, 0, sizeof(out));
  for (int k = 0; k < 2; k++) {
    const bool leaf = c[k] & 0x80000000u;
    const uint32_t idx = c[k] & 0x7FFFFFFFu;
    LBox b;
    if (leaf) {
      b = leaf_boxes[sorted[idx]];
    } else {
      const LNode t = node_boxes[idx];
      for (int a = 0; a < 3; a++) {
        b.lo[a] = t.lo[a];
        b.hi[a] = t.hi[a];
      }
    }
    float* xy = k == 0 ? out.n0xy : out.n1xy;
    xy[0] = b.lo[0];
    xy[1] = b.hi[0];
    xy[2] = b.lo[1];
    xy[3] = b.hi[1];
    out.nz[2 * k] = b.lo[2];
    out.nz[2 * k + 1] = b.hi[2];
    if (leaf) {
      out.ref[k] = BVH_LEAF_BIT | ((tri_base + idx) << 2);
    } else {  // two sibling leaves become one two-triangle leaf (see k_lbvh_emit)
      const uint32_t gl = left[idx], gr = right[idx];
      if ((gl & 0x80000000u) && (gr & 0x80000000u))
        out.ref[k] = BVH_LEAF_BIT | ((tri_base + (gl & 0x7FFFFFFFu)) << 2) | 1u;
      else
        out.ref[k] = node_base + idx;
    }
  }
  raw[node_base + i] = out;
  BvhNodeSlot slot;
  memset(&slot, 0, sizeof(slot));
  slot.n = pack_node(out);
  packed[node_base + i] = slot;
}

struct LResult {
  LNode root;
  uint32_t error, pad[3];
};
__global__ void k_lbvh_result(const LNode* node_boxes, const uint32_t* error, LResult* out) {
  out->root = node_boxes[0];
  out->error = *error;
}

// ---- PLOC (Meister & Bittner 2018, "Parallel locally-ordered clustering"): agglomerative bottom-up construction ----
// Clusters start as the triangles in Morton order. Every round each cluster looks `radius` places to either side for the
// neighbour with which it makes the smallest box (surface area of the union); two clusters that choose each other merge
// into a new node that takes the place of the lower one. The result is much closer to a SAH tree than the radix tree of
// the Morton codes (which only ever splits at the spatial median): 0.93x instead of 0.86x of the SAH tree's trace rate
// on the bench scene (radius 4; larger radii lower the SAH cost a little but trace no faster), 0.96x with the host's SAH top
// over its subtrees (DeviceBuildTarget::sah_top_size). Everything is deterministic: ties prefer the parity partner i ^ 1 (so a run of identical boxes —
// 60 k copies of one triangle — pairs up completely every round instead of merging one pair per round), then the lower
// index; new nodes are numbered by a prefix sum over the merging pairs, not by an atomic counter.
struct PCluster {
  float lo[3];
  uint32_t id;  // 0x80000000 | sorted position of a triangle, or the index of an internal node
  float hi[3];
  uint32_t pad;
};
#define PLOC_MAX_RADIUS 32

__global__ void __launch_bounds__(LB_BLOCK) k_ploc_init(uint32_t n, const uint32_t* sorted, const LBox* leaf_boxes, PCluster* cl) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const LBox b = leaf_boxes[sorted[i]];
  PCluster c;
  for (int a = 0; a < 3; a++) {
    c.lo[a] = b.lo[a];
    c.hi[a] = b.hi[a];
  }
  c.id = 0x80000000u | i;
  c.pad = 0;
  cl[i] = c;
}

__device__ __forceinline__ float union_area(const PCluster& a, const PCluster& b) {
  const float dx = fmaxf(a.hi[0], b.hi[0]) - fminf(a.lo[0], b.lo[0]);
  const float dy = fmaxf(a.hi[1], b.hi[1]) - fminf(a.lo[1], b.lo[1]);
  const float dz = fmaxf(a.hi[2], b.hi[2]) - fminf(a.lo[2], b.lo[2]);
  return dx * dy + dy * dz + dz * dx;
}

__global__ void __launch_bounds__(LB_BLOCK) k_ploc_nn(const PCluster* cl, uint32_t c, int radius, uint32_t* nn) {
  __shared__ PCluster tile[LB_BLOCK + 2 * PLOC_MAX_RADIUS];
  const int base = (int)(blockIdx.x * blockDim.x) - radius;
  for (int t = threadIdx.x; t < (int)blockDim.x + 2 * radius; t += blockDim.x) {
    const int g = base + t;
    if (g >= 0 && g < (int)c) tile[t] = cl[g];
  }
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int)c) return;
  const PCluster me = tile[threadIdx.x + radius];
  float best = INFINITY;
  int best_j = -1;
  const int partner = i ^ 1;  // examined first: wins exact ties
  if (partner < (int)c) {
    best = union_area(me, tile[partner - base]);
    best_j = partner;
  }
  const int lo = max(0, i - radius), hi = min((int)c - 1, i + radius);
  for (int j = lo; j <= hi; j++) {
    if (j == i || j == partner) continue;
    const float d = union_area(me, tile[j - base]);
    if (d < best) {
      best = d;
      best_j = j;
    }
  }
  nn[i] = (uint32_t)best_j;
}

// keep: the slot survives the round (everything but the upper cluster of a merging pair); merge: the lower cluster of one
__global__ void __launch_bounds__(LB_BLOCK) k_ploc_flags(const uint32_t* nn, uint32_t c, unsigned long long* flags) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  const uint32_t j = nn[i];
  const bool mutual = j < c && nn[j] == i;
  const unsigned long long keep = !(mutual && i > j), merge = mutual && i < j;
  flags[i] = keep | (merge << 32);
}

__global__ void __launch_bounds__(LB_BLOCK) k_ploc_apply(const PCluster* cl, uint32_t c, const uint32_t* nn, const unsigned long long* flags, const unsigned long long* ranks,
                                                          uint32_t next_node, PCluster* out, uint32_t* left, uint32_t* right, LNode* nodes, uint32_t* parent_of_internal,
                                                          uint32_t* parent_of_leaf, uint32_t* eff, unsigned long long* totals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  const unsigned long long f = flags[i], r = ranks[i];
  if (i == c - 1) *totals = r + f;  // low half: clusters of the next round; high half: nodes made in this one
  if (!(f & 1ull)) return;
  const uint32_t pos = (uint32_t)(r & 0xFFFFFFFFull);
  PCluster me = cl[i];
  if (f >> 32) {
    const PCluster other = cl[nn[i]];
    const uint32_t k = next_node + (uint32_t)(r >> 32);
    uint32_t h[2], cnt[2], ef[2];
    const uint32_t ids[2] = {me.id, other.id};
    for (int s = 0; s < 2; s++) {
      if (ids[s] & 0x80000000u) {
        h[s] = 0;
        cnt[s] = 1;
        ef[s] = 0;
        parent_of_leaf[ids[s] & 0x7FFFFFFFu] = k | ((uint32_t)s << 31);
      } else {
        h[s] = nodes[ids[s]].height;
        cnt[s] = nodes[ids[s]].pad;
        ef[s] = eff[ids[s]];
        parent_of_internal[ids[s]] = k | ((uint32_t)s << 31);
      }
    }
    // nodes of the final array below (and including) this one: a subtree of <= BVH_MAX_LEAF_TRIS triangles becomes a leaf
    eff[k] = ef[0] + ef[1] + (cnt[0] + cnt[1] > BVH_MAX_LEAF_TRIS ? 1u : 0u);
    left[k] = me.id;
    right[k] = other.id;
    LNode nd;
    for (int a = 0; a < 3; a++) {
      nd.lo[a] = me.lo[a] = fminf(me.lo[a], other.lo[a]);
      nd.hi[a] = me.hi[a] = fmaxf(me.hi[a], other.hi[a]);
    }
    // the stack the subtree needs = its height in nodes of the final array (a subtree that becomes a leaf needs none)
    nd.height = cnt[0] + cnt[1] > BVH_MAX_LEAF_TRIS ? 1u + max(h[0], h[1]) : 0u;
    nd.pad = cnt[0] + cnt[1];  // triangles below
    nodes[k] = nd;
    me.id = k;
  }
  out[pos] = me;
}

// Depth-first position of every leaf, first-leaf position and depth-first (pre-order) index of every internal node:
// walking up, every time the path comes out of a RIGHT child the triangles (and nodes) of the left sibling lie before it.
// Makes the triangles of a subtree contiguous, so a subtree of <= 2 triangles can be referenced as one leaf, and lays the
// nodes out as the host SAH builder does — a node next to its left child, the skipped nodes (subtrees that became leaves)
// not in the array at all: half the bytes and far better cache lines than the order the nodes were made in.
__global__ void __launch_bounds__(LB_BLOCK) k_ploc_positions(uint32_t n, const uint32_t* left, const LNode* nodes, const uint32_t* eff, const uint32_t* parent_of_internal,
                                                              const uint32_t* parent_of_leaf, uint32_t* leaf_pos, uint32_t* node_start, uint32_t* node_index) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * n - 1) return;
  uint32_t link = t < n ? parent_of_leaf[t] : parent_of_internal[t - n];
  uint32_t pos = 0, index = 0;
  for (int guard = 0; guard < 100000 && link != 0xFFFFFFFFu; guard++) {
    const uint32_t p = link & 0x7FFFFFFFu;
    index += 1;  // the ancestor itself comes before everything below it
    if (link >> 31) {
      const uint32_t l = left[p];
      if (l & 0x80000000u) {
        pos += 1u;
      } else {
        pos += nodes[l].pad;
        index += eff[l];
      }
    }
    link = parent_of_internal[p];
  }
  if (t < n) {
    leaf_pos[t] = pos;
  } else {
    node_start[t - n] = pos;
    node_index[t - n] = index;  // meaningful for nodes with more than BVH_MAX_LEAF_TRIS triangles (all their ancestors are such nodes too)
  }
}

__global__ void __launch_bounds__(LB_BLOCK) k_ploc_emit(uint32_t n, uint32_t node_base, uint32_t tri_base, const uint32_t* left, const uint32_t* right, const uint32_t* sorted,
                                                         const LBox* leaf_boxes, const LNode* nodes, const uint32_t* leaf_pos, const uint32_t* node_start, const uint32_t* node_index,
                                                         BvhNode* raw, BvhNodeSlot* packed) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  if (nodes[i].pad <= BVH_MAX_LEAF_TRIS) return;  // became a leaf of its parent
  const uint32_t c[2] = {left[i], right[i]};
  BvhNode out;
  memset(&out, 0, sizeof(out));
  for (int k = 0; k < 2; k++) {
    const bool leaf = c[k] & 0x80000000u;
    const uint32_t idx = c[k] & 0x7FFFFFFFu;
    LBox b;
    if (leaf) {
      b = leaf_boxes[sorted[idx]];
      out.ref[k] = BVH_LEAF_BIT | ((tri_base + leaf_pos[idx]) << 2);
    } else {
      const LNode t = nodes[idx];
      for (int a = 0; a < 3; a++) {
        b.lo[a] = t.lo[a];
        b.hi[a] = t.hi[a];
      }
      if (t.pad <= BVH_MAX_LEAF_TRIS)
        out.ref[k] = BVH_LEAF_BIT | ((tri_base + node_start[idx]) << 2) | (t.pad - 1u);
      else
        out.ref[k] = node_base + node_index[idx];
    }
    float* xy = k == 0 ? out.n0xy : out.n1xy;
    xy[0] = b.lo[0];
    xy[1] = b.hi[0];
    xy[2] = b.lo[1];
    xy[3] = b.hi[1];
    out.nz[2 * k] = b.lo[2];
    out.nz[2 * k + 1] = b.hi[2];
  }
  const uint32_t at = node_base + node_index[i];
  raw[at] = out;
  BvhNodeSlot slot;
  memset(&slot, 0, sizeof(slot));
  slot.n = pack_node(out);
  packed[at] = slot;
}

__global__ void __launch_bounds__(LB_BLOCK) k_ploc_gather(const BvhTri* in, const uint32_t* sorted, const uint32_t* leaf_pos, uint32_t n, BvhTri* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[leaf_pos[i]] = in[sorted[i]];
}

__global__ void k_ploc_result(const LNode* nodes, uint32_t root, const uint32_t* error, LResult* out) {
  out->root = nodes[root];
  out->error = *error;
}

// ---- the frontier for a host-built SAH top (DeviceBuildTarget::sah_top_size): every child slot (node k, side s) whose
// parent holds more than `limit` triangles and whose own subtree holds at most `limit`. The slots are flagged, ranked by a
// prefix sum (deterministic order) and written out with the reference k_ploc_emit gives that child.
__global__ void __launch_bounds__(LB_BLOCK) k_ploc_frontier_flags(uint32_t n, uint32_t limit, const uint32_t* left, const uint32_t* right, const LNode* nodes, uint32_t* flags) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * (n - 1)) return;
  const uint32_t k = t >> 1;
  const uint32_t c = (t & 1u) ? right[k] : left[k];
  const uint32_t child_count = (c & 0x80000000u) ? 1u : nodes[c].pad;
  flags[t] = (nodes[k].pad > limit && child_count <= limit) ? 1u : 0u;
}
__global__ void __launch_bounds__(LB_BLOCK) k_ploc_frontier_emit(uint32_t n, uint32_t node_base, uint32_t tri_base, const uint32_t* left, const uint32_t* right, const uint32_t* sorted,
                                                                  const LBox* leaf_boxes, const LNode* nodes, const uint32_t* leaf_pos, const uint32_t* node_start,
                                                                  const uint32_t* node_index, const uint32_t* flags, const uint32_t* ranks, FrontierEntry* out, uint32_t* total) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * (n - 1)) return;
  if (t == 2 * (n - 1) - 1) *total = ranks[t] + flags[t];
  if (!flags[t]) return;
  const uint32_t k = t >> 1;
  const uint32_t c = (t & 1u) ? right[k] : left[k];
  FrontierEntry e;
  if (c & 0x80000000u) {
    const uint32_t idx = c & 0x7FFFFFFFu;
    const LBox b = leaf_boxes[sorted[idx]];
    for (int a = 0; a < 3; a++) {
      e.lo[a] = b.lo[a];
      e.hi[a] = b.hi[a];
    }
    e.ref = BVH_LEAF_BIT | ((tri_base + leaf_pos[idx]) << 2);
    e.height = 0;
  } else {
    const LNode nd = nodes[c];
    for (int a = 0; a < 3; a++) {
      e.lo[a] = nd.lo[a];
      e.hi[a] = nd.hi[a];
    }
    e.ref = nd.pad <= BVH_MAX_LEAF_TRIS ? (BVH_LEAF_BIT | ((tri_base + node_start[c]) << 2) | (nd.pad - 1u)) : node_base + node_index[c];
    e.height = nd.height;
  }
  out[ranks[t]] = e;
}

// scratch arena of lbvh_build_gpu, kept for the life of the process (one build at a time: the mutex)
struct Arena {
  char* base = nullptr;
  size_t bytes = 0;
  int device = -1;
  std::mutex mutex;
};
Arena& arena(int device) {  // one per GPU: contexts on different GPUs build concurrently (the multi-GPU host uploads in parallel)
  static Arena a[64];
  return a[device & 63];
}

#define LB_TRY(expr)                                                            \
  do {                                                                          \
    hipError_t _e = (expr);                                                     \
    if (_e != hipSuccess) {                                                     \
      err = std::string("lbvh: ") + #expr + ": " + hipGetErrorString(_e);       \
      ok = false;                                                               \
      goto done;                                                                \
    }                                                                           \
  } while (0)

}  // namespace

// tris_in: n >= 2 triangles (object space, ids filled in). Appends n - 1 nodes and n triangles to the outputs;
// node and triangle references are made relative to the current sizes of nodes_out / tris_out.
bool lbvh_build_gpu(const std::vector<BvhTri>& tris_in, std::vector<BvhNode>& nodes_out, std::vector<BvhTri>& tris_out, uint32_t& root_ref, uint32_t& stack_need, float& gpu_ms,
                    std::string& err) {
  const uint32_t n = (uint32_t)tris_in.size();
  const uint32_t node_base = (uint32_t)nodes_out.size(), tri_base = (uint32_t)tris_out.size();
  bool ok = true;
  BvhTri *d_in = nullptr, *d_out = nullptr;
  LBox *d_leaf = nullptr, *d_node = nullptr;
  uint32_t *d_cb = nullptr, *d_vals = nullptr, *d_sorted = nullptr, *d_left = nullptr, *d_right = nullptr, *d_pi = nullptr, *d_pl = nullptr, *d_done = nullptr, *d_rem = nullptr;
  unsigned long long *d_keys = nullptr, *d_keys_sorted = nullptr;
  BvhNode* d_nodes = nullptr;
  void* d_tmp = nullptr;
  size_t tmp_bytes = 0;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  std::unique_lock<std::mutex> arena_lock;
  const uint32_t grid = (n + LB_BLOCK - 1) / LB_BLOCK;
  uint32_t passes = 0;
  {
    // every scratch array comes out of one arena that is kept between builds (grown when a larger mesh arrives):
    // a rebuild then costs no hipMalloc / hipFree at all (18 of them used to dominate the wall time of a 7 ms build)
    size_t tmp_need = 0;
    LB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_need, d_keys, d_keys_sorted, d_vals, d_sorted, (int)n, 0, 63));
    tmp_bytes = tmp_need;
    {
      int dev = 0;
      LB_TRY(hipGetDevice(&dev));
      Arena& A = arena(dev);
      arena_lock = std::unique_lock<std::mutex>(A.mutex);  // held until the results have been copied out
      size_t total = 0;
      auto reserve = [&](size_t bytes) {
        const size_t at = total;
        total += (bytes + 255) & ~(size_t)255;
        return at;
      };
      const size_t o_in = reserve((size_t)n * sizeof(BvhTri)), o_out = reserve((size_t)n * sizeof(BvhTri)), o_leaf = reserve((size_t)n * sizeof(LBox)), o_node = reserve((size_t)n * sizeof(LBox));
      const size_t o_cb = reserve(8 * 4), o_keys = reserve((size_t)n * 8), o_keys2 = reserve((size_t)n * 8), o_vals = reserve((size_t)n * 4), o_sorted = reserve((size_t)n * 4);
      const size_t o_left = reserve((size_t)n * 4), o_right = reserve((size_t)n * 4), o_pi = reserve((size_t)n * 4), o_pl = reserve((size_t)n * 4), o_done = reserve((size_t)n * 4), o_rem = reserve(4);
      const size_t o_nodes = reserve((size_t)n * sizeof(BvhNode)), o_tmp = reserve(tmp_bytes);
      if (A.device != dev || A.bytes < total) {
        if (A.base) (void)hipFree(A.base);
        A.base = nullptr;
        A.bytes = 0;
        LB_TRY(hipMalloc((void**)&A.base, total));
        A.bytes = total;
        A.device = dev;
      }
      char* b = A.base;
      d_in = (BvhTri*)(b + o_in);
      d_out = (BvhTri*)(b + o_out);
      d_leaf = (LBox*)(b + o_leaf);
      d_node = (LBox*)(b + o_node);
      d_cb = (uint32_t*)(b + o_cb);
      d_keys = (unsigned long long*)(b + o_keys);
      d_keys_sorted = (unsigned long long*)(b + o_keys2);
      d_vals = (uint32_t*)(b + o_vals);
      d_sorted = (uint32_t*)(b + o_sorted);
      d_left = (uint32_t*)(b + o_left);
      d_right = (uint32_t*)(b + o_right);
      d_pi = (uint32_t*)(b + o_pi);
      d_pl = (uint32_t*)(b + o_pl);
      d_done = (uint32_t*)(b + o_done);
      d_rem = (uint32_t*)(b + o_rem);
      d_nodes = (BvhNode*)(b + o_nodes);
      d_tmp = b + o_tmp;
    }
    LB_TRY(hipMemcpy(d_in, tris_in.data(), (size_t)n * sizeof(BvhTri), hipMemcpyHostToDevice));
    LB_TRY(hipEventCreate(&e0));
    LB_TRY(hipEventCreate(&e1));
    LB_TRY(hipEventRecord(e0, nullptr));
    const uint32_t init[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0, 0};
    LB_TRY(hipMemcpyAsync(d_cb, init, sizeof(init), hipMemcpyHostToDevice, nullptr));
    hipLaunchKernelGGL(k_lbvh_bounds, dim3(grid), dim3(LB_BLOCK), 0, nullptr, d_in, n, d_leaf, d_cb);
    hipLaunchKernelGGL(k_lbvh_morton, dim3(grid), dim3(LB_BLOCK), 0, nullptr, d_leaf, n, d_cb, d_keys, d_vals);
    LB_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_keys, d_keys_sorted, d_vals, d_sorted, (int)n, 0, 63));
    hipLaunchKernelGGL(k_lbvh_hierarchy, dim3(grid), dim3(LB_BLOCK), 0, nullptr, d_keys_sorted, (int)n, d_left, d_right, d_pi, d_pl);
    LB_TRY(hipMemsetAsync(d_done, 0, (size_t)n * 4, nullptr));
    for (;;) {  // bottom-up, one tree level (at least) per launch; the remaining-node count is read every 8 passes
      for (int k = 0; k < 8; k++) {
        passes++;
        if (k == 7) LB_TRY(hipMemsetAsync(d_rem, 0, 4, nullptr));
        hipLaunchKernelGGL(k_lbvh_refit_pass, dim3(grid), dim3(LB_BLOCK), 0, nullptr, (int)n, passes, d_left, d_right, d_sorted, d_leaf, d_node, d_done, d_rem);
      }
      uint32_t rem = 0;
      LB_TRY(hipMemcpy(&rem, d_rem, 4, hipMemcpyDeviceToHost));
      if (rem == 0) break;
      if (passes > 4096) {
        err = "lbvh: refit did not converge";
        ok = false;
        goto done;
      }
    }
    hipLaunchKernelGGL(k_lbvh_emit, dim3(grid), dim3(LB_BLOCK), 0, nullptr, (int)n, node_base, tri_base, d_left, d_right, d_sorted, d_leaf, d_node, d_nodes);
    hipLaunchKernelGGL(k_lbvh_gather, dim3(grid), dim3(LB_BLOCK), 0, nullptr, d_in, d_sorted, n, d_out);
    LB_TRY(hipEventRecord(e1, nullptr));
    LB_TRY(hipEventSynchronize(e1));
    LB_TRY(hipGetLastError());
    LB_TRY(hipEventElapsedTime(&gpu_ms, e0, e1));
    nodes_out.resize((size_t)node_base + n - 1);
    tris_out.resize((size_t)tri_base + n);
    LB_TRY(hipMemcpy(nodes_out.data() + node_base, d_nodes, (size_t)(n - 1) * sizeof(BvhNode), hipMemcpyDeviceToHost));
    LB_TRY(hipMemcpy(tris_out.data() + tri_base, d_out, (size_t)n * sizeof(BvhTri), hipMemcpyDeviceToHost));
    root_ref = node_base;  // internal node 0 is the root
    // done[root] is the pass in which the root was finished = the height of the tree = the bound of the stack
    LB_TRY(hipMemcpy(&stack_need, d_done, 4, hipMemcpyDeviceToHost));
  }
done:
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  return ok;
}


bool lbvh_build_device(const DeviceBuildTarget& tgt, const std::vector<MeshPiece>& pieces, uint32_t node_base, uint32_t tri_base, uint32_t& root_ref, uint32_t& height, float bounds[6],
                       float& gpu_ms, std::string& err, std::vector<FrontierEntry>* frontier, uint32_t frontier_size) {
  if (frontier) frontier->clear();
  uint32_t n = 0;
  for (const MeshPiece& pc : pieces) n += pc.prim_count;
  if (n < 2 || pieces.empty()) {
    err = "lbvh: a device build needs at least two triangles";
    return false;
  }
  bool ok = true;
  hipStream_t st = (hipStream_t)tgt.stream;
  BvhTri* d_in = nullptr;
  LBox* d_leaf = nullptr;
  LNode* d_node = nullptr;
  uint32_t *d_cb = nullptr, *d_vals = nullptr, *d_sorted = nullptr, *d_left = nullptr, *d_right = nullptr, *d_pi = nullptr, *d_pl = nullptr, *d_visits = nullptr, *d_err = nullptr;
  unsigned long long *d_keys = nullptr, *d_keys_sorted = nullptr;
  MeshPiece* d_pieces = nullptr;
  LResult* d_res = nullptr;
  PCluster* d_cl[2] = {nullptr, nullptr};
  uint32_t *d_nn = nullptr, *d_leaf_pos = nullptr, *d_node_start = nullptr, *d_node_index = nullptr, *d_eff = nullptr;
  unsigned long long *d_flags = nullptr, *d_ranks = nullptr, *d_totals = nullptr;
  void* d_scan_tmp = nullptr;
  size_t scan_tmp_bytes = 0;
  const bool ploc = tgt.algorithm == 1;
  const int radius = std::min(std::max(tgt.ploc_radius, 1), PLOC_MAX_RADIUS);
  void* d_tmp = nullptr;
  size_t tmp_bytes = 0;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  std::unique_lock<std::mutex> arena_lock;
  const uint32_t grid = (n + LB_BLOCK - 1) / LB_BLOCK;
  {
    size_t tmp_need = 0;
    LB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_need, d_keys, d_keys_sorted, d_vals, d_sorted, (int)n, 0, 63, st));
    tmp_bytes = tmp_need;
    if (ploc) {
      LB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_tmp_bytes, d_flags, d_ranks, (int)n, st));
      size_t frontier_scan = 0;  // the frontier's scan runs over 2 (n - 1) 32-bit flags in the same scratch
      LB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, frontier_scan, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)(2 * n), st));
      scan_tmp_bytes = std::max(scan_tmp_bytes, frontier_scan);
    }
    int dev = 0;
    LB_TRY(hipGetDevice(&dev));
    Arena& A = arena(dev);
    arena_lock = std::unique_lock<std::mutex>(A.mutex);
    size_t total = 0;
    auto reserve = [&](size_t bytes) {
      const size_t at = total;
      total += (bytes + 255) & ~(size_t)255;
      return at;
    };
    const size_t o_in = reserve((size_t)n * sizeof(BvhTri)), o_leaf = reserve((size_t)n * sizeof(LBox)), o_node = reserve((size_t)n * sizeof(LNode));
    const size_t o_cb = reserve(8 * 4), o_keys = reserve((size_t)n * 8), o_keys2 = reserve((size_t)n * 8), o_vals = reserve((size_t)n * 4), o_sorted = reserve((size_t)n * 4);
    const size_t o_left = reserve((size_t)n * 4), o_right = reserve((size_t)n * 4), o_pi = reserve((size_t)n * 4), o_pl = reserve((size_t)n * 4), o_visits = reserve((size_t)n * 4);
    const size_t o_pieces = reserve(pieces.size() * sizeof(MeshPiece)), o_res = reserve(sizeof(LResult)), o_tmp = reserve(tmp_bytes);
    const size_t pn = ploc ? n : 0;
    const size_t o_cl0 = reserve(pn * sizeof(PCluster)), o_cl1 = reserve(pn * sizeof(PCluster)), o_nn = reserve(pn * 4), o_flags = reserve(pn * 8), o_ranks = reserve(pn * 8), o_totals = reserve(8);
    const size_t o_lpos = reserve(pn * 4), o_nstart = reserve(pn * 4), o_nindex = reserve(pn * 4), o_eff = reserve(pn * 4), o_scan = reserve(ploc ? scan_tmp_bytes : 0);
    if (A.device != dev || A.bytes < total) {
      if (A.base) (void)hipFree(A.base);
      A.base = nullptr;
      A.bytes = 0;
      LB_TRY(hipMalloc((void**)&A.base, total));
      A.bytes = total;
      A.device = dev;
    }
    char* b = A.base;
    d_in = (BvhTri*)(b + o_in);
    d_leaf = (LBox*)(b + o_leaf);
    d_node = (LNode*)(b + o_node);
    d_cb = (uint32_t*)(b + o_cb);
    d_err = d_cb + 6;
    d_keys = (unsigned long long*)(b + o_keys);
    d_keys_sorted = (unsigned long long*)(b + o_keys2);
    d_vals = (uint32_t*)(b + o_vals);
    d_sorted = (uint32_t*)(b + o_sorted);
    d_left = (uint32_t*)(b + o_left);
    d_right = (uint32_t*)(b + o_right);
    d_pi = (uint32_t*)(b + o_pi);
    d_pl = (uint32_t*)(b + o_pl);
    d_visits = (uint32_t*)(b + o_visits);
    d_pieces = (MeshPiece*)(b + o_pieces);
    d_res = (LResult*)(b + o_res);
    d_tmp = b + o_tmp;
    d_cl[0] = (PCluster*)(b + o_cl0);
    d_cl[1] = (PCluster*)(b + o_cl1);
    d_nn = (uint32_t*)(b + o_nn);
    d_flags = (unsigned long long*)(b + o_flags);
    d_ranks = (unsigned long long*)(b + o_ranks);
    d_totals = (unsigned long long*)(b + o_totals);
    d_leaf_pos = (uint32_t*)(b + o_lpos);
    d_node_start = (uint32_t*)(b + o_nstart);
    d_node_index = (uint32_t*)(b + o_nindex);
    d_eff = (uint32_t*)(b + o_eff);
    d_scan_tmp = b + o_scan;
    LB_TRY(hipEventCreate(&e0));
    LB_TRY(hipEventCreate(&e1));
    LB_TRY(hipEventRecord(e0, st));
    const uint32_t init[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0, 0};  // centroid bounds (ordered), then the error word
    LB_TRY(hipMemcpyAsync(d_cb, init, sizeof(init), hipMemcpyHostToDevice, st));
    LB_TRY(hipMemcpyAsync(d_pieces, pieces.data(), pieces.size() * sizeof(MeshPiece), hipMemcpyHostToDevice, st));
    LB_TRY(hipMemsetAsync(d_visits, 0, (size_t)n * 4, st));
    hipLaunchKernelGGL(k_lbvh_fetch, dim3(grid), dim3(LB_BLOCK), 0, st, d_pieces, (uint32_t)pieces.size(), tgt.vertices, tgt.vertex_count, tgt.indices, n, d_in, d_leaf, d_cb, d_err);
    hipLaunchKernelGGL(k_lbvh_morton, dim3(grid), dim3(LB_BLOCK), 0, st, d_leaf, n, d_cb, d_keys, d_vals);
    LB_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_keys, d_keys_sorted, d_vals, d_sorted, (int)n, 0, 63, st));
    uint32_t root_node = 0;  // Karras: internal node 0 is the root
    if (!ploc) {
      hipLaunchKernelGGL(k_lbvh_hierarchy, dim3(grid), dim3(LB_BLOCK), 0, st, d_keys_sorted, (int)n, d_left, d_right, d_pi, d_pl);
      hipLaunchKernelGGL(k_lbvh_refit, dim3(grid), dim3(LB_BLOCK), 0, st, (int)n, d_left, d_right, d_pi, d_pl, d_sorted, d_leaf, d_node, d_visits);
      hipLaunchKernelGGL(k_lbvh_emit_packed, dim3(grid), dim3(LB_BLOCK), 0, st, (int)n, node_base, tri_base, d_left, d_right, d_sorted, d_leaf, d_node, tgt.raw_nodes, tgt.nodes);
      hipLaunchKernelGGL(k_lbvh_gather, dim3(grid), dim3(LB_BLOCK), 0, st, d_in, d_sorted, n, tgt.tris + tri_base);
      hipLaunchKernelGGL(k_lbvh_result, dim3(1), dim3(1), 0, st, d_node, d_err, d_res);
    } else {
      LB_TRY(hipMemsetAsync(d_pi, 0xFF, (size_t)n * 4, st));  // the root keeps "no parent"
      hipLaunchKernelGGL(k_ploc_init, dim3(grid), dim3(LB_BLOCK), 0, st, n, d_sorted, d_leaf, d_cl[0]);
      uint32_t c = n, next_node = 0;
      int cur = 0;
      for (uint32_t round = 0; c > 1; round++) {
        if (round > 20000) {  // (a chain of ever closer neighbours merges one pair per round; never seen, bounded anyway)
          err = "lbvh: PLOC did not converge";
          ok = false;
          goto done;
        }
        const uint32_t g = (c + LB_BLOCK - 1) / LB_BLOCK;
        hipLaunchKernelGGL(k_ploc_nn, dim3(g), dim3(LB_BLOCK), 0, st, d_cl[cur], c, radius, d_nn);
        hipLaunchKernelGGL(k_ploc_flags, dim3(g), dim3(LB_BLOCK), 0, st, d_nn, c, d_flags);
        size_t tb = scan_tmp_bytes;
        LB_TRY(hipcub::DeviceScan::ExclusiveSum(d_scan_tmp, tb, d_flags, d_ranks, (int)c, st));
        hipLaunchKernelGGL(k_ploc_apply, dim3(g), dim3(LB_BLOCK), 0, st, d_cl[cur], c, d_nn, d_flags, d_ranks, next_node, d_cl[cur ^ 1], d_left, d_right, d_node, d_pi, d_pl, d_eff, d_totals);
        unsigned long long totals = 0;
        LB_TRY(hipMemcpyAsync(&totals, d_totals, 8, hipMemcpyDeviceToHost, st));
        LB_TRY(hipStreamSynchronize(st));
        const uint32_t kept = (uint32_t)(totals & 0xFFFFFFFFull), made = (uint32_t)(totals >> 32);
        if (made == 0 || kept + made != c) {
          err = "lbvh: PLOC round made no progress";
          ok = false;
          goto done;
        }
        c = kept;
        next_node += made;
        cur ^= 1;
      }
      root_node = next_node - 1;  // = n - 2: the last node made
      hipLaunchKernelGGL(k_ploc_positions, dim3((2 * n + LB_BLOCK - 1) / LB_BLOCK), dim3(LB_BLOCK), 0, st, n, d_left, d_node, d_eff, d_pi, d_pl, d_leaf_pos, d_node_start, d_node_index);
      hipLaunchKernelGGL(k_ploc_emit, dim3(grid), dim3(LB_BLOCK), 0, st, n, node_base, tri_base, d_left, d_right, d_sorted, d_leaf, d_node, d_leaf_pos, d_node_start, d_node_index, tgt.raw_nodes, tgt.nodes);
      hipLaunchKernelGGL(k_ploc_gather, dim3(grid), dim3(LB_BLOCK), 0, st, d_in, d_sorted, d_leaf_pos, n, tgt.tris + tri_base);
      hipLaunchKernelGGL(k_ploc_result, dim3(1), dim3(1), 0, st, d_node, root_node, d_err, d_res);
      if (frontier && frontier_size >= BVH_MAX_LEAF_TRIS && n > 2 * frontier_size) {
        // the child slots under which the host rebuilds the top (the cluster buffers are free by now: flags and ranks take
        // the scan arrays' place — 2 (n - 1) words fit their n * 8 bytes — and the entries go where the clusters were)
        uint32_t* f_flags = reinterpret_cast<uint32_t*>(d_flags);
        uint32_t* f_ranks = reinterpret_cast<uint32_t*>(d_ranks);
        FrontierEntry* f_out = reinterpret_cast<FrontierEntry*>(d_cl[0]);  // n * 32 bytes: room for n entries; there are at most 2 n / frontier_size
        const uint32_t slots = 2 * (n - 1);
        const uint32_t fg = (slots + LB_BLOCK - 1) / LB_BLOCK;
        hipLaunchKernelGGL(k_ploc_frontier_flags, dim3(fg), dim3(LB_BLOCK), 0, st, n, frontier_size, d_left, d_right, d_node, f_flags);
        size_t tb = scan_tmp_bytes;
        LB_TRY(hipcub::DeviceScan::ExclusiveSum(d_scan_tmp, tb, f_flags, f_ranks, (int)slots, st));
        hipLaunchKernelGGL(k_ploc_frontier_emit, dim3(fg), dim3(LB_BLOCK), 0, st, n, node_base, tri_base, d_left, d_right, d_sorted, d_leaf, d_node, d_leaf_pos, d_node_start, d_node_index, f_flags,
                           f_ranks, f_out, reinterpret_cast<uint32_t*>(d_totals));
        uint32_t count = 0;
        LB_TRY(hipMemcpyAsync(&count, d_totals, 4, hipMemcpyDeviceToHost, st));
        LB_TRY(hipStreamSynchronize(st));
        if (count >= 2 && count <= n) {
          frontier->resize(count);
          LB_TRY(hipMemcpyAsync(frontier->data(), f_out, (size_t)count * sizeof(FrontierEntry), hipMemcpyDeviceToHost, st));
        }
      }
    }
    LB_TRY(hipEventRecord(e1, st));
    LResult res;
    LB_TRY(hipMemcpyAsync(&res, d_res, sizeof(res), hipMemcpyDeviceToHost, st));
    LB_TRY(hipStreamSynchronize(st));
    LB_TRY(hipGetLastError());
    LB_TRY(hipEventElapsedTime(&gpu_ms, e0, e1));
    if (res.error) {
      err = "vertex index exceeds gVertices";
      ok = false;
      goto done;
    }
    root_ref = ploc ? node_base : node_base + root_node;  // PLOC lays its nodes out depth-first: the root comes first
    height = res.root.height;
    memcpy(bounds, res.root.lo, 12);
    memcpy(bounds + 3, res.root.hi, 12);
  }
done:
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  return ok;
}

}  // namespace sthip
