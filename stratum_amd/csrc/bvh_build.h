// bvh_build.h — host-side acceleration-structure build (see bvh.h for the layout)
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/sthip.h"
#include "bvh.h"

#if defined(__HIPCC__)
#define STHIP_BVH_HD __host__ __device__
#else
#define STHIP_BVH_HD
#endif

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
  // the 8-wide form (build_wide8_bvh): the bottom levels' nodes are [0, wide8_blas_nodes) of the wide8 array and stay; a
  // transforms-only update makes the top level's nodes again behind them (build_wide8_top)
  std::vector<uint32_t> wide8_root;  // per entry: the wide8 node its bottom level starts at (BVH_INVALID_REF: none)
  std::vector<uint32_t> wide8_height;  // ... and that bottom level's height in wide8 nodes
  uint32_t wide8_blas_nodes = 0;
  uint32_t wide8_blas_height = 0;
};

struct BuiltBvh {
  std::vector<BvhNode> nodes;
  std::vector<BvhTri> tris;
  bool any_alpha = false;  // some material has an alpha mask: the uploader keeps the leaf triangles' uvs beside them (BvhTriUv, filled on the device)
  std::vector<uint32_t> inst_alpha;  // per instance: gImage1s index of its material's alpha mask or BVH_NO_ALPHA
  std::vector<TlasEntry> entries;
  std::vector<DeviceVolume> volumes;  // parsed headers of gVolumes (first_word is filled by the uploader)
  uint32_t root_ref = BVH_INVALID_REF;  // inner-node index traversal starts at; BVH_INVALID_REF: empty scene
  uint32_t top_is_world_blas = 1;       // 1: root_ref is the merged world-space mesh, no top level
  uint32_t stack_depth = 4;             // upper bound of the traversal stack height
  float scene_center[3] = {0, 0, 0};
  float scene_radius = 0;
  float gpu_build_ms = 0;  // device time of the LBVH kernels (0 for the host builder)
  // Device-resident build (DeviceBuildTarget): the GPU builder has written nodes [0, dev_nodes) and triangles
  // [0, dev_tris) of the final arrays in place; `nodes` / `tris` then hold only what the host built (small meshes, SAH
  // fallbacks, the top level) and belong at [dev_nodes, ...) / [dev_tris, ...): their references are already final.
  uint32_t dev_nodes = 0, dev_tris = 0;
  std::vector<uint8_t> absolute_ref;  // device builds: [2 * node + child] = 1 for a reference of a host node that is final already (into the device region)
  // Embedded leaves (host builds without a device target): the triangles of a leaf lie in the NODE array, in the 48-byte
  // units right behind the node that refers to them, so the dependent fetch of a leaf visit goes to the cache line(s) the
  // node itself just came from. unit_tri[u] = index into `tris` of the triangle in unit u, or 0xFFFFFFFF for a node;
  // leaf references then count units of the node array (DeviceBvh::tris = nodes).
  bool embedded = false;
  std::vector<uint32_t> unit_tri;
  TopLevelState top;
  // The 4-wide form of the same tree for k_trace (build_wide_bvh; empty unless asked for): its nodes, a copy of `entries`
  // whose roots index wide_nodes, the reference traversal starts at, and the bound of the stack it needs (up to three
  // pushes per level).
  std::vector<WideNode> wide_nodes;
  std::vector<TlasEntry> wide_entries;
  uint32_t wide_root_ref = BVH_INVALID_REF;
  uint32_t wide_stack_depth = 0;
  // The 8-wide compressed form (build_wide8_bvh; bvh.h: Wide8Node): nodes, the entry table in the order the top level's
  // nodes refer to it (roots index wide8_nodes), the node the walk starts at, the number of 64-bit stack entries it needs.
  std::vector<Wide8Node> wide8_nodes;
  std::vector<TlasEntry> wide8_entries;
  uint32_t wide8_root = BVH_INVALID_REF;
  uint32_t wide8_stack_depth = 0;
};

// One run of triangles of a bottom level, as the scene arrays describe it (scene.h:139-161)
struct MeshPiece {
  uint32_t id_bits;              // OR-ed into BvhTri::id below the primitive index: the instance for the merged mesh, else 0
  uint32_t first_vertex;
  uint32_t indices_byte_offset;
  uint32_t stride;               // 2 or 4
  uint32_t prim_count;
  uint32_t prim_begin;           // running sum of prim_count over the pieces before this one
};

// The device-resident arrays a GPU build reads (the scene, uploaded before the build) and writes in place (lbvh.hip):
// no triangle or node crosses PCIe except the unpacked nodes the host keeps for the treetop / top-level rebuilds.
struct DeviceBuildTarget {
  const sthip_PackedVertexData* vertices = nullptr;  // device
  uint32_t vertex_count = 0;
  const uint8_t* indices = nullptr;                  // device, padded by 8 bytes
  BvhNodeSlot* nodes = nullptr;                      // device: the packed node array the kernels traverse
  BvhNode* raw_nodes = nullptr;                      // device: the same nodes unpacked (copied to the host once, pinned)
  BvhTri* tris = nullptr;                            // device: leaf triangles
  void* stream = nullptr;                            // hipStream_t
  int algorithm = 1;                                 // 0: Karras radix tree over the Morton codes; 1: PLOC (better trees, ~2x the build time)
  int ploc_radius = 4;                               // neighbours examined on either side (<= 32)
  // > 0: the subtrees of at most this many triangles keep their PLOC shape, and the tree ABOVE them is rebuilt on the host
  // with the binned-SAH builder over their boxes (a few thousand boxes: ~1 ms). PLOC's merges are greedy and local; the
  // levels a ray spends most visits in are the ones where that costs most.
  uint32_t sah_top_size = 0;
  // called once the sizes are known, before the first build: makes nodes / raw_nodes / tris at least this large
  bool (*reserve)(void* user, size_t node_capacity, size_t tri_capacity, DeviceBuildTarget& self) = nullptr;
  void* user = nullptr;
};
// A subtree of the device-built tree the host may put a better top over (PLOC only): its box, the reference a parent uses for
// it (a node of the final array, or a leaf), its stack need. See DeviceBuildTarget::sah_top_size.
struct FrontierEntry {
  float lo[3];
  uint32_t ref;
  float hi[3];
  uint32_t height;
};
// Builds one bottom level over `pieces` on the device: n - 1 nodes at [node_base, ...), n triangles at [tri_base, ...).
// bounds[6] = lo xyz, hi xyz of the mesh; height = the stack the tree needs. false + err on a HIP error or a vertex index
// outside gVertices.
bool lbvh_build_device(const DeviceBuildTarget& tgt, const std::vector<MeshPiece>& pieces, uint32_t node_base, uint32_t tri_base, uint32_t& root_ref, uint32_t& height, float bounds[6],
                       float& gpu_ms, std::string& err, std::vector<FrontierEntry>* frontier = nullptr, uint32_t frontier_size = 0);

enum BvhBuilderKind { BVH_BUILDER_SAH_HOST = 0, BVH_BUILDER_LBVH_GPU = 1 };

// Validates the scene arrays and builds. Returns false and sets `err` on malformed input.
// `builder`: binned SAH on the host (default; best traversal) or LBVH on the GPU (fastest build; lbvh.hip).
// With a DeviceBuildTarget (and the LBVH builder) bottom levels of >= 64 triangles are built in place on the device.
bool build_scene_bvh(const sthip_scene_desc& scene, BuiltBvh& out, std::string& err, int builder = BVH_BUILDER_SAH_HOST, DeviceBuildTarget* device = nullptr, bool embed_leaves = false);
// Collapses the finished binary tree of a host build (out.dev_nodes == 0) into 4-wide nodes with quantised child boxes
// (bvh.h: WideNode): out.wide_*. Every leaf reference, and with it every triangle and entry, stays what it is, and every
// decoded box contains the box it stands for: a traversal of the wide tree finds the hits the binary one finds.
void build_wide_bvh(BuiltBvh& out);

// Collapses the finished binary tree of a host build (out.dev_nodes == 0, no embedded leaves) into 8-wide compressed nodes
// (bvh.h: Wide8Node): out.wide8_*. The items of a node's leaf children must be consecutive, so this PERMUTES out.tris
// — every binary leaf keeps its triangles together and in order — and rewrites the binary nodes' leaf
// references to match: call it before anything is uploaded. Leaves nothing (wide8_nodes empty) when the tree cannot take
// the form: a box that fits no grid, a leaf of more than three triangles. out.top.wide8_* keeps what build_wide8_top needs.
void build_wide8_bvh(BuiltBvh& out);
// The top level of the 8-wide form over `tlas` (binary top-level nodes whose inner references count from tlas_base =
// st.blas_nodes, as rebuild_top_level returns them) and st.entries: appends to `nodes` (which holds the bottom levels'
// st.wide8_blas_nodes nodes), makes `entries`. root_ref: the binary tree's root (rebuild_top_level). false: no 8-wide form.
bool build_wide8_top(const TopLevelState& st, const BvhNode* tlas, uint32_t tlas_base, uint32_t root_ref, bool top_is_world_blas, std::vector<Wide8Node>& nodes, std::vector<TlasEntry>& entries,
                     uint32_t& wide8_root, uint32_t& wide8_stack_depth);

// Transforms-only update: new entry matrices and world boxes, a new top level. `tlas_nodes` come out with their child
// references already offset by st.blas_nodes (they go to nodes[st.blas_nodes ...]). Fails (false) when an instance of the
// merged world-space mesh no longer has the identity transform: that needs a full build.
bool rebuild_top_level(TopLevelState& st, const sthip_TransformData* xf, const sthip_TransformData* inv, uint32_t instance_count, std::vector<BvhNode>& tlas_nodes, uint32_t& root_ref,
                       uint32_t& top_is_world_blas, uint32_t& stack_depth, float scene_center[3], float& scene_radius, std::string& err);

// The "treetop": a copy of the inner nodes a ray is most likely to visit (greedy by box surface area from the root down,
// through top-level entries into the bottom levels they refer to), at most `capacity` of them, in an index space of its
// own. The persistent trace kernel keeps it in LDS: a visit to one of these nodes costs an LDS read instead of four
// divergent vector loads. Child references between treetop nodes carry BVH_TOP_BIT | treetop index; every other
// reference is unchanged (it leaves the treetop for the node array in HBM). `entries` is a copy of the entry table whose
// `root` is redirected into the treetop where that bottom-level root was taken. The node array itself is not touched.
struct Treetop {
  std::vector<BvhNode> nodes;
  std::vector<TlasEntry> entries;
  uint32_t root_ref = BVH_INVALID_REF;  // BVH_TOP_BIT | 0 when the treetop is not empty, else the scene's root_ref
};
void build_treetop(const BvhNode* nodes, size_t node_count, const std::vector<TlasEntry>& entries, uint32_t root_ref, uint32_t capacity, Treetop& out);

// BvhNode -> BvhNodePacked (bvh.h). A plane whose low mantissa byte is replaced by an arbitrary byte b reads
// sign * (M + b) in units of its ulp, M being the stored magnitude with a zero low byte. A lower plane must stay <= the true
// one for every b, an upper plane >= it:
//   lower, v >= 0: M = (m - 255) & ~255 (for m < 255: -0 with M = 0, i.e. values in [-255 ulp, -0]);  lower, v < 0: M = (m + 255) & ~255
//   upper: mirrored.  Infinite / NaN planes (empty children) are clamped to +-3.4e38 first so that no byte makes a NaN.
STHIP_BVH_HD inline uint32_t pack_plane(float v, bool upper, uint32_t byte) {
  if (!(v > -3.0e38f)) v = -3.0e38f;  // also NaN
  if (v > 3.0e38f) v = 3.0e38f;
  uint32_t u;
  memcpy(&u, &v, 4);
  const bool negative = (u >> 31) != 0;
  const uint32_t m = u & 0x7FFFFFFFu;
  uint32_t sign, M;
  const bool outward_grows = upper != negative;  // rounding away from the box means a larger magnitude
  if (outward_grows) {
    sign = negative ? 0x80000000u : 0u;
    M = (m + 255u) & ~255u;
  } else if (m >= 255u) {
    sign = negative ? 0x80000000u : 0u;
    M = (m - 255u) & ~255u;
  } else {  // within 255 ulp of zero: step across zero, every byte then lies on the safe side
    sign = negative ? 0u : 0x80000000u;
    M = 0u;
  }
  return sign | M | (byte & 0xFFu);
}
STHIP_BVH_HD inline BvhNodePacked pack_node(const BvhNode& n) {
  BvhNodePacked q;
  uint32_t w[8];
  for (int k = 0; k < 4; k++) {
    w[k] = pack_plane(n.n0xy[k], (k & 1) != 0, n.ref[0] >> (8 * k));      // lo.x, hi.x, lo.y, hi.y of child 0
    w[4 + k] = pack_plane(n.n1xy[k], (k & 1) != 0, n.ref[1] >> (8 * k));  // of child 1
  }
  memcpy(q.n0xy, w, 16);
  memcpy(q.n1xy, w + 4, 16);
  for (int k = 0; k < 4; k++) {  // z planes are exact; only non-finite ones are clamped (they must not turn into NaN in the slab test)
    float z = n.nz[k];
    if (!(z > -3.0e38f)) z = -3.0e38f;
    if (z > 3.0e38f) z = 3.0e38f;
    q.nz[k] = z;
  }
  return q;
}
inline void pack_nodes(const BvhNode* in, size_t count, std::vector<BvhNodePacked>& out) {
  out.resize(count);
  for (size_t i = 0; i < count; i++) out[i] = pack_node(in[i]);
}

// ---- the 4-wide node (bvh.h: WideNode), made the same way on the host (build_wide_bvh) and on the device (wide.hip) ----
struct WideChildBox {
  float lo[3], hi[3];
};
// Quantises the boxes of n <= 4 children onto the node's 8-bit grid — origin = the lower corner of their union, a power-of-two
// step per axis, lower planes rounded down and upper planes up, checked in double — and fills in the references (an unused
// slot: entry planes behind exit planes and the first child's reference). false: the boxes do not fit any grid (a plane that
// is not finite): the node must not be used.
STHIP_BVH_HD inline bool make_wide_node(const WideChildBox* ch, const uint32_t* refs, int n, WideNode& wn) {
  bool ok = true;
  memset(&wn, 0, sizeof(wn));
  for (int a = 0; a < 3; a++) {
    float lo = ch[0].lo[a], hi = ch[0].hi[a];
    for (int k = 1; k < n; k++) {
      lo = ch[k].lo[a] < lo ? ch[k].lo[a] : lo;
      hi = ch[k].hi[a] > hi ? ch[k].hi[a] : hi;
    }
    wn.origin[a] = lo;
    const double ext = (double)hi - (double)lo;
    int e = 1;  // biased; the step is 2^(e - 127)
    if (ext > 0) {
      int x;
      (void)frexp(ext / 255.0, &x);  // ext / 255 = m * 2^x, m in [0.5, 1): 2^x >= ext / 255
      e = x + 127 < 1 ? 1 : (x + 127 > 254 ? 254 : x + 127);
    }
    for (;;) {
      const double step = ldexp(1.0, e - 127);
      bool fits = true;
      for (int k = 0; k < n && fits; k++) {
        double ql = floor(((double)ch[k].lo[a] - (double)lo) / step), qh = ceil(((double)ch[k].hi[a] - (double)lo) / step);
        while (ql > 0 && (double)lo + ql * step > (double)ch[k].lo[a]) ql -= 1;
        while ((double)lo + qh * step < (double)ch[k].hi[a]) qh += 1;
        if (ql < 0) ql = 0;
        if (qh > 255 || !(qh >= 0)) {
          fits = false;
          break;
        }
        wn.q[2 * a][k] = (uint8_t)ql;
        wn.q[2 * a + 1][k] = (uint8_t)qh;
      }
      if (fits) break;
      if (e >= 254) {  // no power of two spans it (or a plane is not finite)
        ok = false;
        break;
      }
      e++;
    }
    if (!(lo > -3.4e38f && lo < 3.4e38f)) ok = false;
    wn.exp[a] = (uint8_t)(int8_t)(e - 127);  // signed: the kernel sign-extends the byte and scales with v_ldexp_f32
  }
  wn.exp[3] = (uint8_t)n;
  for (int k = 0; k < 4; k++) {
    if (k < n) {
      wn.ref[k] = refs[k];
    } else {
      wn.ref[k] = refs[0];  // never followed unless the node is point-sized (the walk has no test for unused slots); then the first child twice, which changes no hit
      for (int a = 0; a < 3; a++) {
        wn.q[2 * a][k] = 255;
        wn.q[2 * a + 1][k] = 0;
      }
    }
  }
  return ok;
}

// ---- the 8-wide compressed node (bvh.h: Wide8Node) ----
// Quantises the boxes of n <= 8 children, child k in slot slot[k], onto the node's 8-bit grid exactly as make_wide_node does
// (origin = the lower corner of their union, a power-of-two step per axis, lower planes rounded down and upper planes up,
// checked in double); unused slots get entry planes behind exit planes. Fills origin, exp and q only. false: no grid fits.
STHIP_BVH_HD inline bool wide8_quantise(const WideChildBox* ch, const uint8_t* slot, int n, Wide8Node& wn) {
  bool ok = true;
  for (int a = 0; a < 3; a++) {
    for (int s = 0; s < 8; s++) {
      wn.q[2 * a][s] = 255;
      wn.q[2 * a + 1][s] = 0;
    }
    float lo = ch[0].lo[a], hi = ch[0].hi[a];
    for (int k = 1; k < n; k++) {
      lo = ch[k].lo[a] < lo ? ch[k].lo[a] : lo;
      hi = ch[k].hi[a] > hi ? ch[k].hi[a] : hi;
    }
    wn.origin[a] = lo;
    const double ext = (double)hi - (double)lo;
    int e = 1;  // biased; the step is 2^(e - 127)
    if (ext > 0) {
      int x;
      (void)frexp(ext / 255.0, &x);
      e = x + 127 < 1 ? 1 : (x + 127 > 254 ? 254 : x + 127);
    }
    for (;;) {
      const double step = ldexp(1.0, e - 127);
      bool fits = true;
      for (int k = 0; k < n && fits; k++) {
        double ql = floor(((double)ch[k].lo[a] - (double)lo) / step), qh = ceil(((double)ch[k].hi[a] - (double)lo) / step);
        while (ql > 0 && (double)lo + ql * step > (double)ch[k].lo[a]) ql -= 1;
        while ((double)lo + qh * step < (double)ch[k].hi[a]) qh += 1;
        if (ql < 0) ql = 0;
        if (qh > 255 || !(qh >= 0)) {
          fits = false;
          break;
        }
        wn.q[2 * a][slot[k]] = (uint8_t)ql;
        wn.q[2 * a + 1][slot[k]] = (uint8_t)qh;
      }
      if (fits) break;
      if (e >= 254) {
        ok = false;
        break;
      }
      e++;
    }
    if (!(lo > -3.4e38f && lo < 3.4e38f)) ok = false;
    wn.exp[a] = (uint8_t)(int8_t)(e - 127);
  }
  return ok;
}

// The 4-wide form made ON THE DEVICE from the packed binary nodes as they lie in HBM (wide.hip): for trees the GPU builder
// made (no node of theirs ever visits the host) and after a transforms-only update (new top level). Same collapse rule as
// build_wide_bvh (the inner child with the largest box is opened until there are four), the planes the packed nodes hold
// (rounded outward by pack_plane: still conservative). Deterministic: a wide node is first written at the index of the binary
// node it stands for, then the used slots are compacted in index order.
struct DeviceWideResult {
  uint32_t node_count = 0;
  uint32_t root_ref = BVH_INVALID_REF;
  uint32_t stack_depth = 0;
  float gpu_ms = 0;
};
struct DeviceWideScratch;  // (wide.hip: the buffers a collapse needs, kept between calls)
DeviceWideScratch* device_wide_scratch_create();
void device_wide_scratch_destroy(DeviceWideScratch*);
// nodes: node_count packed nodes (device); entries: entry_count entries (device); wide_nodes / wide_entries: device outputs
// (wide_nodes must hold node_count entries, wide_entries entry_count). max_levels: a bound of the binary tree's height.
bool collapse_wide_device(DeviceWideScratch* scratch, const BvhNodeSlot* nodes, uint32_t node_count, const TlasEntry* entries, uint32_t entry_count, uint32_t root_ref, bool top_is_world_blas,
                          uint32_t max_levels, WideNode* wide_nodes, TlasEntry* wide_entries, void* stream, DeviceWideResult& result, std::string& err);

// GPU LBVH of one mesh (lbvh.hip): appends nodes and leaf-ordered triangles to the outputs.
bool lbvh_build_gpu(const std::vector<BvhTri>& tris_in, std::vector<BvhNode>& nodes_out, std::vector<BvhTri>& tris_out, uint32_t& root_ref, uint32_t& stack_need, float& gpu_ms,
                    std::string& err);

}  // namespace sthip
