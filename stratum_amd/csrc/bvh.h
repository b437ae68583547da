// bvh.h — acceleration-structure layout in HBM (product-internal; replaces the driver-built
// BLAS/TLAS of src/Core/AccelerationStructure.cpp:5-27 and src/Node/Scene.cpp:435-459,614-629).
//
// One flat array of BVH2 nodes holds the top level and every bottom level. The builders work on the 64-byte BvhNode
// below; what is uploaded is its 48-byte packed form (BvhNodePacked further down). A node stores
// the boxes of its two children in the Aila-Laine arrangement:
//     n0xy = (c0.lo.x, c0.hi.x, c0.lo.y, c0.hi.y)
//     n1xy = (c1.lo.x, c1.hi.x, c1.lo.y, c1.hi.y)
//     nz   = (c0.lo.z, c0.hi.z, c1.lo.z, c1.hi.z)
//     ref  = (child0, child1, 0, 0)
// Child references: bit31 = 0 -> index of an inner node; bit31 = 1 -> leaf:
//     bit30 = 0: triangle leaf, bits[29:2] = first triangle, bits[1:0] = count-1
//     bit30 = 1: instance leaf, bits[15:0] = index into the TLAS entry table
// Leaf triangles are stored in leaf order, 48 bytes each: three float4 (v0, v1, v2 in the object
// space of their mesh); v0.w carries the hit id bits: instance | primitive << 16 for the merged
// world-space mesh, primitive << 16 for a shared mesh (the instance comes from the TLAS entry).
#pragma once
#include <stdint.h>

#define BVH_LEAF_BIT 0x80000000u
#define BVH_INST_BIT 0x40000000u
#define BVH_TOP_BIT 0x40000000u  // on an INNER reference (bit31 = 0): index into the treetop copy (bvh_build.h), not the node array
#define BVH_MAX_LEAF_TRIS 2  // measured on the 1M-triangle atrium: 2 beats 1, 3 and 4 (the leaf reference can encode up to 4)
#define BVH_INVALID_REF 0xFFFFFFFFu  // empty child (never intersected: box is inverted)

struct BvhNode {
  float n0xy[4];
  float n1xy[4];
  float nz[4];
  uint32_t ref[4];
};

struct BvhTri {
  float v0[3];
  uint32_t id;
  float v1[3];
  uint32_t src_indices;  // where the triangle came from: byte offset of its index triple in gIndices
  float v2[3];
  uint32_t src_vertex;   // ... first_vertex of its mesh, bit31 = 32-bit indices (k_fill_tri_shade reads the vertices again through these)
};
// Beside every leaf triangle, in the same order: what shading needs of its three vertices apart from the positions BvhTri
// holds — normal and v of each, the three u (sthip_PackedVertexData: position.xyz u | normal.xyz v). Filled on the device
// once the tree is resident (api.hip: k_fill_tri_shade), whoever built it.
struct BvhTriShade {
  float n0[3], v0;
  float n1[3], v1;
  float n2[3], v2;
  float u[3], pad;
};
// The node as it lies in HBM (and in the LDS treetop): 48 bytes = three float4, so a lane fetches it with THREE 16-byte
// loads. The traversal kernel is bound by the rate at which the vector-memory front end takes divergent lane loads
// (about one lane-instruction per clock and CU, whatever its width), so the fourth load that a 64-byte node needs for
// its two child references costs a quarter of the node bandwidth. Here the references ride in the low mantissa byte of
// the eight x / y planes: byte k of ref[0] is the low byte of n0xy[k], byte k of ref[1] the low byte of n1xy[k]. The
// planes are used as they are, reference byte included: pack_node() (bvh_build.h) rounds every plane outward far enough
// (<= 511 ulp, 6e-5 relative) that the box stays conservative whatever that byte holds. z planes are exact.
struct BvhNodePacked {
  float n0xy[4];
  float n1xy[4];
  float nz[4];
};
// The slot a packed node occupies in HBM. 48: nodes back to back (a node then straddles a 64-byte boundary every other
// time and a 128-byte line 3 times in 8). 64: one node per half line, 16 bytes of padding.
#ifndef STHIP_NODE_STRIDE
#define STHIP_NODE_STRIDE 48
#endif
#define BVH_NODE_BYTES ((unsigned)STHIP_NODE_STRIDE)
struct BvhNodeSlot {
  BvhNodePacked n;
#if STHIP_NODE_STRIDE > 48
  uint32_t pad[(STHIP_NODE_STRIDE - 48) / 4];
#endif
};
// The 4-wide node k_trace can walk instead ("wide_bvh"; bvh_build.h: build_wide_bvh collapses the binary tree into it): 64
// aligned bytes = four 16-byte loads per lane. The boxes of up to four children as 8-bit planes on a per-node grid:
// plane = origin[axis] + q * 2^exp[axis] (exp a signed byte), lower planes rounded down and upper planes up on that grid (exactly: the
// builder checks them in double), so a decoded box contains the child's box. An unused child slot (index >= exp[3])
// has its lower planes at 255 and its upper planes at 0 — the entry plane lies behind the exit plane on every axis, a miss
// unless the node is point-sized — and a copy of the first child's reference, so that even then nothing but a repeated
// visit can happen. References are the binary tree's, except that an inner
// reference indexes THIS array.
struct WideNode {
  float origin[3];
  uint8_t exp[4];    // exponents of the x, y, z plane step as signed bytes (-126 .. 127); [3] = number of children (not read by the kernel)
  uint8_t q[6][4];   // [lo.x, hi.x, lo.y, hi.y, lo.z, hi.z][child]
  uint32_t pad[2];
  uint32_t ref[4];
};
// The 8-wide compressed node ("wide_bvh" = 3; bvh_build.h: build_wide8_bvh): 80 bytes = five 16-byte loads per lane, after
// Ylitie, Karras, Laine, "Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs" (HPG 2017). The boxes of
// up to eight children as 8-bit planes on the node's grid (as WideNode). No child references: the INNER children of a node
// are consecutive nodes of this array from child_base on, in slot order (child of slot s = child_base + popcount(imask &
// ((1 << s) - 1))); its LEAF children's items (triangles; top-level entries in a node of the top level) are consecutive
// from leaf_base on. So what a walk keeps of a visited node is a GROUP — a base and a bit mask — and its stack holds groups.
//   meta[s]: 0 = unused slot; inner child: 0x20 | (24 + s); leaf child: (unary item count: 1, 3 or 7) << 5 | first item
//   (relative to leaf_base, < 24). A hit child contributes (meta >> 5) << (meta & 31) to the 32-bit hit mask of the node:
//   items in bits 0..23, inner children in bits 24..31 — the latter at 24 + (s ^ octinv), octinv = 7 ^ (sign bits of the
//   ray direction), so that the highest set bit is the child to visit first: the builder puts the child lying toward the
//   (+x, +y, +z) corner of the node into slot 7, toward (-x, -y, -z) into slot 0, and so on.
// Slots are filled sparsely (by position); an unused slot also has entry planes behind its exit planes.
struct Wide8Node {
  float origin[3];
  uint8_t exp[3];       // signed exponents of the x, y, z plane step
  uint8_t imask;        // slots that hold inner children
  uint32_t child_base;  // index of the first inner child in this array
  uint32_t leaf_base;   // first item of the leaf children: a triangle index, or WIDE8_ENTRY_BIT | index into the entry table
  uint8_t meta[8];
  uint8_t q[6][8];      // [lo.x, hi.x, lo.y, hi.y, lo.z, hi.z][slot]
};
#define WIDE8_ENTRY_BIT 0x80000000u
#define WIDE8_MAX_ITEMS 24
#define BVH_NO_ALPHA 0xFFFFFFFFu  // DeviceBvh::inst_alpha entry of an instance whose material has no alpha mask
// uv of the three vertices of a leaf triangle, in leaf order next to BvhTri; only built for scenes with alpha masks
struct BvhTriUv {
  float uv[3][2];
};

// One entry of the top level: the merged world-space mesh (all instances whose transform is the identity,
// flattened into one BLAS), one transformed triangle instance, or one sphere instance (tested in place, no BLAS).
#define TLAS_ENTRY_TRANSFORMED 0u
#define TLAS_ENTRY_IDENTITY 1u
#define TLAS_ENTRY_SPHERE 2u
#define TLAS_ENTRY_VOLUME 3u  // root = index into DeviceBvh::volumes; tested in place by the instantiations that carry it
struct TlasEntry {
  float inv[12];      // gInstanceInverseTransforms[instance], row-major 3x4
  uint32_t root;      // index of the BLAS root (always an inner node); unused for a sphere
  uint32_t id_bits;   // OR-ed into BvhTri::id: 0 for the merged mesh, the instance index otherwise
  uint32_t identity;  // TLAS_ENTRY_*
  uint32_t pad;
  float center[3];    // object-space bounding sphere of the mesh: sizes the per-ray box padding
  float radius;       // ... or the radius of the sphere instance (InstanceData::radius, scene.h:43)
};

// The header of one NanoVDB float grid of gVolumes[], parsed on the host at upload (media.h reads the tree through it)
struct DeviceVolume {
  uint32_t first_word;  // the grid's first 32-bit word in DeviceVolumes::words
  uint32_t bytes;
  uint32_t root;        // byte address of the root node inside the grid
  uint32_t pad;
  int32_t bbox_min[3], bbox_max[3];
  float root_max, pad1;
  float matf[9], invmatf[9], vecf[3];
  float pad2[3];
};

#ifdef __cplusplus
static_assert(sizeof(BvhNode) == 64, "BvhNode");
static_assert(sizeof(BvhNodePacked) == 48, "BvhNodePacked");
static_assert(sizeof(BvhNodeSlot) == BVH_NODE_BYTES, "BvhNodeSlot");
static_assert(sizeof(BvhTri) == 48, "BvhTri");
static_assert(sizeof(BvhTriShade) == 64, "BvhTriShade");
static_assert(sizeof(WideNode) == 64, "WideNode");
static_assert(sizeof(Wide8Node) == 80, "Wide8Node");
static_assert(sizeof(TlasEntry) == 80, "TlasEntry");
#endif
