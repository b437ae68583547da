// traverse.h — BVH traversal and ray/triangle intersection for gfx950 (wave64).
//
// Replaces rayQuery.TraceRayInline/Proceed (src/Shaders/common/intersection.hlsli:68-75), which the
// reference leaves to the Vulkan driver. The hit contract (object-space watertight triangle test,
// closest hit over all triangles with (t, instance, primitive) ordering, any-hit = existence) is the
// one stated in include/sthip.h and oracle/stratum_oracle.cpp; the acceleration structure must never
// change the answer, so every box test is conservative:
//   * per ray and per axis the origin is padded by P_k = 4e-6 * (L1 distance to the bounding-sphere
//     centre + radius) + 3e-7 * |o_k| of the space being traversed. The first term bounds the triangle
//     test's own tolerance (~3e-7 * distance) for every box the ray can reach; the second bounds the
//     rounding of the precomputed o_k / d_k, which lets the slab distance be ONE fma per plane
//     (plane * inv_d - (o -+ P) * inv_d) without the cancellation that form has for rays whose origin
//     is large compared with the box;
//   * |dir| is clamped away from 0 and comparisons are inclusive.
// One lane = one ray. The traversal stack is in LDS, laid out [level][lane] so that a wave's pushes
// and pops of one level hit 64 consecutive banks.
#pragma once

#include "bvh.h"
#include "device_math.h"
#include "media.h"

struct DeviceBvh {
  const float4* nodes;  // BvhNodePacked = 3 x float4 (48 bytes; the child references ride in the low bytes of the x / y planes)
  const float4* tris;   // BvhTri = 3 x float4
  const TlasEntry* entries;
  uint32_t root_ref;
  uint32_t top_is_world_blas;
  uint32_t stack_depth;
  float scene_cx, scene_cy, scene_cz, scene_radius;
  // alpha masks (gAlphaTest, intersection.hlsli:117-131); alpha_test = 0 unless the flag is set AND a material has a mask
  const float2* tri_uv;         // BvhTriUv: 3 x float2 per leaf triangle
  const uint32_t* inst_alpha;   // per instance: gImage1s index of its material's mask, BVH_NO_ALPHA if none
  const struct DeviceImage1* images1;
  const float* image1_texels;
  uint32_t alpha_test;
  uint32_t flip_uvs;            // gFlipTriangleUVs for the mask lookup
  const DeviceVolume* volumes;  // gVolumes headers (volume instances are top-level entries tested in place, like spheres)
  // the treetop (bvh_build.h): the most-visited inner nodes as a copy of their own that the persistent trace kernel holds
  // in LDS; a reference with BVH_TOP_BIT indexes it. Only k_trace follows these (it starts at top_root_ref and reads
  // top_entries); every other kernel walks the node array from root_ref and never sees the bit.
  const float4* top_nodes;
  const TlasEntry* top_entries;
  uint32_t top_root_ref;
  uint32_t top_count;
  // Bounded LDS stacks (trees higher than the LDS affords at full occupancy, e.g. Morton-built ones): the per-lane LDS stack
  // has lds_levels levels (= stack_depth when unbounded); a ray that needs more is traced again with a stack of stack_depth
  // words in global memory (`spill`: one column per resident lane of the persistent grid). nullptr: unbounded.
  uint32_t lds_levels;
  uint32_t* spill;
  // The 4-wide form of the tree (bvh.h: WideNode; "wide_bvh"), walked by k_trace only: nullptr when not built. Inner
  // references of wide_entries' roots, of wide_root_ref and of the wide nodes index wide_nodes; leaves are the binary tree's.
  const uint4* wide_nodes;
  const TlasEntry* wide_entries;
  uint32_t wide_root_ref;
  uint32_t wide_stack_depth;
  // The 8-wide compressed form (bvh.h: Wide8Node; "wide_bvh" = 3), walked by k_trace only: nullptr when not built. Its entry
  // table is in the order its top-level nodes refer to (roots index wide8_nodes); its stack entries are 64-bit groups.
  const uint4* wide8_nodes;
  const TlasEntry* wide8_entries;
  uint32_t wide8_root;
  uint32_t wide8_stack_depth;
  uint32_t wide8_tri_min;  // the leaf phase of the 8-wide walk goes on while at least this many lanes of the wave hold a triangle ("tri_min_lanes")
};
struct DeviceImage1 {
  uint32_t offset, w, h, pad;
};
// level-0 lookup of a one-channel image: bilinear, repeat addressing (the same arithmetic as DisneyMaterial::bilinear)
DEV float sample_image1(const DeviceBvh& bvh, uint32_t index, float u, float v) {
  const DeviceImage1 im = bvh.images1[index];
  const float x = u * (float)im.w - 0.5f, y = v * (float)im.h - 0.5f;
  const float x0 = floorf(x), y0 = floorf(y);
  const float fx = x - x0, fy = y - y0;
  const int w = (int)im.w, h = (int)im.h;
  const int ix = (int)x0, iy = (int)y0;
  const int xa = ((ix % w) + w) % w, xb = (((ix + 1) % w) + w) % w, ya = ((iy % h) + h) % h, yb = (((iy + 1) % h) + h) % h;
  const float* t = bvh.image1_texels + im.offset;
  const float a = lerp1(t[(size_t)ya * w + xa], t[(size_t)ya * w + xb], fx), b = lerp1(t[(size_t)yb * w + xa], t[(size_t)yb * w + xb], fx);
  return lerp1(a, b, fy);
}

// a float4 in LDS, typed as such: a pointer that keeps its address space makes the treetop read a ds_read_b128; a generic
// one would merge with the global path into flat loads, which take the vector-memory path the treetop exists to avoid
typedef float lds_f4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) lds_f4v LdsFloat4;
DEV float4 lds_load4(const LdsFloat4* p) {
  const lds_f4v v = *p;
  return make_float4(v.x, v.y, v.z, v.w);
}

// child reference k of a packed node: the low bytes of the four floats of n0xy (k = 0) or n1xy (k = 1), bvh.h
DEV uint32_t packed_ref(float4 q) {
  return __builtin_amdgcn_perm(__float_as_uint(q.y), __float_as_uint(q.x), 0x0c0c0400u) | __builtin_amdgcn_perm(__float_as_uint(q.w), __float_as_uint(q.z), 0x04000c0cu);
}

struct RayHit {
  float t, b1, b2;
  uint32_t ip;    // instance | primitive << 16; 0xFFFFFFFF = miss
  uint32_t leaf;  // index of the hit triangle in the leaf-triangle array (k_shade finds its vertices there: BvhTri / BvhTriShade); unset for a sphere / volume / miss
};

// Diagnostics of the COUNT instantiations. *_slots count 64 per wave-level iteration (added by the first active
// lane), so nodes / inner_slots and tris / tri_slots are the lane utilisations of the two loops.
struct TraverseCounters {
  uint32_t nodes, tris;
  uint32_t inner_slots, tri_slots;
  uint32_t st[8];  // sthip_stats::lane_states (lane 0 of a wave counts for the wave)
  DEV void clear() {
    nodes = tris = inner_slots = tri_slots = 0;
    for (int i = 0; i < 8; i++) st[i] = 0;
  }
};
DEV bool first_active_lane() { return (threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)__ballot(1)) - 1); }

// per-space ray constants
struct RaySpace {
  f3 o;               // origin in this space (triangle test)
  f3 idir;            // 1 / d, |d| clamped away from 0
  f3 noodL, noodH;    // -(o + P) * idir and -(o - P) * idir: slab distance = fma(plane, idir, nood)
  float Sx, Sy, Sz;   // watertight shear constants
  int k;              // kx | ky << 2 | kz << 4: the watertight test's axis permutation in one register
};

// 1 / d for the slab tests, |d| clamped away from 0. The hardware's reciprocal (1 ulp) instead of the correctly rounded
// division (one instruction for eleven): box tests are not part of the hit contract, they only have to be conservative, and
// both terms of a slab distance are multiplied by the SAME value, so its error is that of a ray whose direction differs by
// 1.2e-7 relative per axis — a sideways shift of 1.2e-7 x distance against the 4e-6 x distance the ray's origin is padded by.
DEV float safe_rcp_dir(float d) {
  const float eps = 1e-30f;
  const float dd = fabsf(d) < eps ? copysignf(eps, d) : d;
  return __builtin_amdgcn_rcpf(dd);
}

DEV void setup_space(RaySpace& s, f3 o, f3 d, float cx, float cy, float cz, float radius) {
  s.o = o;
  s.idir = F3(safe_rcp_dir(d.x), safe_rcp_dir(d.y), safe_rcp_dir(d.z));
  const float P = 4e-6f * (fabsf(o.x - cx) + fabsf(o.y - cy) + fabsf(o.z - cz) + radius);
  const float Px = P + 3e-7f * fabsf(o.x), Py = P + 3e-7f * fabsf(o.y), Pz = P + 3e-7f * fabsf(o.z);
  s.noodL = F3(-(o.x + Px) * s.idir.x, -(o.y + Py) * s.idir.y, -(o.z + Pz) * s.idir.z);
  s.noodH = F3(-(o.x - Px) * s.idir.x, -(o.y - Py) * s.idir.y, -(o.z - Pz) * s.idir.z);
  // watertight shear constants (Woop, Benthin, Wald 2013)
  int kz = 0;
  float m = fabsf(d.x);
  if (fabsf(d.y) > m) {
    kz = 1;
    m = fabsf(d.y);
  }
  if (fabsf(d.z) > m) kz = 2;
  int kx = kz + 1;
  if (kx == 3) kx = 0;
  int ky = kx + 1;
  if (ky == 3) ky = 0;
  const float dz = comp3(d, kz);
  if (dz < 0.0f) {
    const int t = kx;
    kx = ky;
    ky = t;
  }
  s.k = kx | (ky << 2) | (kz << 4);
  s.Sx = comp3(d, kx) / dz;
  s.Sy = comp3(d, ky) / dz;
  s.Sz = 1.0f / dz;
}

// object-space ray of an instance: fmaf chains, see the contract
DEV f3 obj_point(const float* m, f3 p) {
  return F3(fmaf(m[2], p.z, fmaf(m[1], p.y, m[0] * p.x)) + m[3], fmaf(m[6], p.z, fmaf(m[5], p.y, m[4] * p.x)) + m[7],
            fmaf(m[10], p.z, fmaf(m[9], p.y, m[8] * p.x)) + m[11]);
}
DEV f3 obj_vector(const float* m, f3 p) {
  return F3(fmaf(m[2], p.z, fmaf(m[1], p.y, m[0] * p.x)), fmaf(m[6], p.z, fmaf(m[5], p.y, m[4] * p.x)), fmaf(m[10], p.z, fmaf(m[9], p.y, m[8] * p.x)));
}

// Sphere instances (ray_sphere, common.h:163-173, as used by intersection.hlsli:79-89), arithmetic exactly as the
// contract states: products rounded separately, near root if it lies beyond tmin, else the far root
DEV bool sphere_test(f3 o, f3 d, float r, float tmin, float tmax, float& t) {
  const float a = dot3(d, d);
  const float b = dot3(o, d);
  const f3 l = F3(a * o.x - d.x * b, a * o.y - d.y * b, a * o.z - d.z * b);
  float det = (a * r) * (a * r) - dot3(l, l);
  if (det < 0) return false;
  const float inv_a = 1 / a;
  det = sqrtf(det * inv_a) * inv_a;
  const float e = (-b) * inv_a;
  const float t0 = e - det, t1 = e + det;
  if (!(t0 < t1)) return false;
  const float tt = t0 > tmin ? t0 : t1;
  if (!(tt > tmin && tt < tmax)) return false;
  t = tt;
  return true;
}

// closest-hit ordering key: instance first, then primitive
DEV uint32_t hit_key(uint32_t ip) { return (ip << 16) | (ip >> 16); }

// Watertight ray/triangle test, arithmetic exactly as the contract states. Edge functions are
// products rounded separately (no fma) so that a shared edge evaluates antisymmetrically.
DEV bool tri_test(const RaySpace& s, f3 p0, f3 p1, f3 p2, float tmin, float tmax, float& t, float& b1, float& b2) {
  const f3 A = p0 - s.o, B = p1 - s.o, C = p2 - s.o;
  const int kx = s.k & 3, ky = (s.k >> 2) & 3, kz = s.k >> 4;
  const float Akz = comp3(A, kz), Bkz = comp3(B, kz), Ckz = comp3(C, kz);
  const float Ax = fmaf(-s.Sx, Akz, comp3(A, kx)), Ay = fmaf(-s.Sy, Akz, comp3(A, ky));
  const float Bx = fmaf(-s.Sx, Bkz, comp3(B, kx)), By = fmaf(-s.Sy, Bkz, comp3(B, ky));
  const float Cx = fmaf(-s.Sx, Ckz, comp3(C, kx)), Cy = fmaf(-s.Sy, Ckz, comp3(C, ky));
  const float U = Cx * By - Cy * Bx;
  const float V = Ax * Cy - Ay * Cx;
  const float W = Bx * Ay - By * Ax;
  // No early exits: in a wave every lane waits for the slowest one anyway, and each `return` would cost an exec-mask
  // region (the tests are the same, in the same arithmetic; a zero determinant makes rcp infinite and tt fail its range
  // test, or NaN, which fails it too — det != 0 is tested all the same).
  const bool edges_disagree = ((U < 0.0f) | (V < 0.0f) | (W < 0.0f)) & ((U > 0.0f) | (V > 0.0f) | (W > 0.0f));
  const float det = U + V + W;
  const float Az = s.Sz * Akz, Bz = s.Sz * Bkz, Cz = s.Sz * Ckz;
  const float T = fmaf(W, Cz, fmaf(V, Bz, U * Az));
  const float rcp = 1.0f / det;
  const float tt = T * rcp;
  t = tt;
  b1 = V * rcp;
  b2 = W * rcp;
  return !edges_disagree & (det != 0.0f) & (tt > tmin) & (tt < tmax);
}

// stack sentinels (both carry the leaf bits, so the inner-node loop hands them to the leaf handler)
#define TRAV_DONE 0xFFFFFFFFu           // bottom of the stack: the ray is finished
#define TRAV_EXIT_INSTANCE 0xFFFFFFFEu  // pushed when a transformed instance is entered
#define TRAV_CANARY 0xFFFFFFFDu         // never a reference (entry indices stay below 0xFFFE): marks an untouched slot, see BOUNDED

// Per-lane traversal state machine. `ref` is the next thing to process: an inner-node index, a leaf
// reference or a sentinel. It runs "while-while": every lane of the wave walks inner nodes until it
// holds a leaf, then the wave processes leaves together, so that the long triangle code is not executed
// once per inner-node step of some other lane. The inner loop is kept minimal: no emptiness or
// instance checks on pop (sentinels on the stack do that), one fma per slab plane, 32-bit offsets.
// MODE: which kind of query the lanes run
#define TRAV_CLOSEST 0  // closest hit
#define TRAV_ANY 1      // occlusion: stop at the first accepted triangle (hit.ip = 0)
#define TRAV_MIXED 2    // per lane, member `any` (the persistent kernel feeds closest-hit and shadow rays to one wave)
// TOP: inner references may point into the treetop held in LDS at `top_lds` (BVH_TOP_BIT), see DeviceBvh
// BOUNDED: the stack has fewer levels than the tree is high. `top` never passes `limit` (one v_min per step; a push there
// lands in the spare slot and is lost). Whether that happened is read off a canary: start() puts TRAV_CANARY into the slot
// at `limit`, and every step in which `top` stands there overwrites it with its speculative push — so overflowed() is true
// for every ray that lost a push (and for the few that merely filled the stack to the brim). Such a traversal still
// terminates (every reference on the stack is one this ray wrote), but its result is void: the caller traces the ray again
// with a full-height stack.
#ifndef STHIP_ENTRY_BATCH
#define STHIP_ENTRY_BATCH 12u
#endif
template <int MODE, bool COUNT, uint32_t STRIDE, bool ALPHA = false, bool TOP = false, bool BOUNDED = false, bool SAVE_WORLD = false, uint32_t ENTRY_BATCH = 1, bool WIDE = false>  // ALPHA: gAlphaTest is compiled in (scenes with alpha masks)
struct Traversal {
  const LdsFloat4* top_lds;  // TOP only
  uint32_t limit;            // BOUNDED only: (levels - 1) * STRIDE
  DEV bool overflowed(const uint32_t* stack) const { return BOUNDED && stack[limit] != TRAV_CANARY; }
  bool any;  // TRAV_MIXED only
  DEV bool is_any() const { return MODE == TRAV_ANY || (MODE == TRAV_MIXED && any); }
  f3 o, d;  // world-space ray
  float tmin, tmax;
  RaySpace sp;  // the space currently being traversed (world, or the object space of an instance)
  // The world-space constants, kept while an instance is traversed (SAVE_WORLD): leaving an instance then copies 13
  // registers instead of running setup_space again (6 correctly rounded divisions, ~130 instructions) — the values are
  // the ones start() computed, so nothing changes but the cost. Only where registers allow (the persistent kernel has 23
  // to spare below its occupancy step).
  f3 w_idir, w_noodL, w_noodH;
  float w_Sx, w_Sy, w_Sz;
  int w_k;
  RayHit hit;
  uint32_t ref;
  uint32_t top;  // stack height in entries * STRIDE (an LDS word offset)
  uint32_t id_bits;

  // WIDE: the wide walk picks the entry and the exit plane of each axis by the sign of the direction; the origin terms are
  // stored to match — noodL with the entry planes, noodH with the exit planes — right after every setup_space.
  DEV void orient_space() {
    if (!WIDE) return;
    const float lx = sp.noodL.x, ly = sp.noodL.y, lz = sp.noodL.z;
    const bool nx = sp.idir.x < 0.0f, ny = sp.idir.y < 0.0f, nz = sp.idir.z < 0.0f;
    sp.noodL.x = nx ? sp.noodH.x : lx;
    sp.noodH.x = nx ? lx : sp.noodH.x;
    sp.noodL.y = ny ? sp.noodH.y : ly;
    sp.noodH.y = ny ? ly : sp.noodH.y;
    sp.noodL.z = nz ? sp.noodH.z : lz;
    sp.noodH.z = nz ? lz : sp.noodH.z;
  }
  DEV bool active() const { return ref != TRAV_DONE; }
  DEV void reset() { ref = TRAV_DONE; }
  // a triangle leaf (not a sentinel, not an instance entry)
  static DEV bool is_tri_leaf(uint32_t r) { return (r & (BVH_LEAF_BIT | BVH_INST_BIT)) == BVH_LEAF_BIT; }

  DEV void start(const DeviceBvh& bvh, uint32_t* stack, f3 ro, f3 rd, float t0, float t1) {
    o = ro;
    d = rd;
    tmin = t0;
    tmax = t1;
    hit.t = t1;
    hit.b1 = hit.b2 = 0.0f;
    hit.ip = 0xFFFFFFFFu;
    hit.leaf = 0xFFFFFFFFu;
    stack[0] = TRAV_DONE;
    top = STRIDE;
    if (BOUNDED) stack[limit] = TRAV_CANARY;
    id_bits = 0;
    ref = bvh.root_ref;  // BVH_INVALID_REF == TRAV_DONE for an empty scene
    setup_space(sp, ro, rd, bvh.scene_cx, bvh.scene_cy, bvh.scene_cz, bvh.scene_radius);
    orient_space();
    if (SAVE_WORLD) {
      w_idir = sp.idir;
      w_noodL = sp.noodL;
      w_noodH = sp.noodH;
      w_Sx = sp.Sx;
      w_Sy = sp.Sy;
      w_Sz = sp.Sz;
      w_k = sp.k;
    }
  }

  DEV void pop(const uint32_t* stack) {
    top -= STRIDE;
    ref = stack[top];
  }

  // Walks inner nodes until `ref` is a leaf or a sentinel, or until fewer than `min_lanes` lanes of the
  // wave are still walking (the others would only wait for them).
  DEV void inner_loop(const DeviceBvh& bvh, uint32_t* stack, uint32_t min_lanes, TraverseCounters& cnt) {
    const char* base = reinterpret_cast<const char*>(bvh.nodes);
    const float tbest = hit.t;  // does not change while inner nodes are walked (an occlusion lane stops at its first hit, so for it hit.t stays tmax)
    // Every lane that holds an inner node takes at least one step per call (progress), then the wave goes on while
    // at least `min_lanes` lanes (>= 1) still hold an inner node: one wave-uniform test per step.
    if (WIDE) {
      wide_loop(bvh, stack, min_lanes, cnt);
      return;
    }
    for (;;) {
      if (!(ref & BVH_LEAF_BIT)) {
      float4 n0, n1, nz;
      if (TOP && (ref & BVH_TOP_BIT)) {  // a treetop node: three LDS reads instead of three divergent vector loads
        const LdsFloat4* n = top_lds + (uint32_t)((ref & ~BVH_TOP_BIT) * 3u);
        n0 = lds_load4(n);
        n1 = lds_load4(n + 1);
        nz = lds_load4(n + 2);
      } else {
        // 32-bit byte offset (the upload checks the array stays below 4 GiB): scalar base + vector offset addressing, no
        // 64-bit multiply; 48 = 32 + 16 as shifts and an add (a 32-bit v_mul_lo is a quarter-rate instruction)
        uint32_t offset;
        if (BVH_NODE_BYTES == 48u) {
          uint32_t r32;  // (written as ref * 48 or as two shifts the compiler emits v_mul_lo_u32; the asm keeps the shift)
          asm("v_lshlrev_b32 %0, 5, %1" : "=v"(r32) : "v"(ref));
          offset = r32 + (ref << 4);
        } else {
          offset = ref * BVH_NODE_BYTES;
        }
        const float4* n = reinterpret_cast<const float4*>(base + offset);
        n0 = n[0];
        n1 = n[1];
        nz = n[2];
      }
      const uint2 cr = make_uint2(packed_ref(n0), packed_ref(n1));
      uint32_t* slot = stack + (top - STRIDE);  // the pop slot; the push slot is one level above it (one address, two offsets)
      const uint32_t popped = slot[0];
      if (COUNT) {
        cnt.nodes++;
        if (first_active_lane()) cnt.inner_slots += 64;
      }
      const float a0x = fmaf(n0.x, sp.idir.x, sp.noodL.x), b0x = fmaf(n0.y, sp.idir.x, sp.noodH.x);
      const float a0y = fmaf(n0.z, sp.idir.y, sp.noodL.y), b0y = fmaf(n0.w, sp.idir.y, sp.noodH.y);
      const float a0z = fmaf(nz.x, sp.idir.z, sp.noodL.z), b0z = fmaf(nz.y, sp.idir.z, sp.noodH.z);
      const float a1x = fmaf(n1.x, sp.idir.x, sp.noodL.x), b1x = fmaf(n1.y, sp.idir.x, sp.noodH.x);
      const float a1y = fmaf(n1.z, sp.idir.y, sp.noodL.y), b1y = fmaf(n1.w, sp.idir.y, sp.noodH.y);
      const float a1z = fmaf(nz.z, sp.idir.z, sp.noodL.z), b1z = fmaf(nz.w, sp.idir.z, sp.noodH.z);
      const float tn0 = fmaxf(fmaxf(fminf(a0x, b0x), fminf(a0y, b0y)), fmaxf(fminf(a0z, b0z), tmin));
      const float tf0 = fminf(fminf(fmaxf(a0x, b0x), fmaxf(a0y, b0y)), fmaxf(a0z, b0z));
      const float tn1 = fmaxf(fmaxf(fminf(a1x, b1x), fminf(a1y, b1y)), fmaxf(fminf(a1z, b1z), tmin));
      const float tf1 = fminf(fminf(fmaxf(a1x, b1x), fmaxf(a1y, b1y)), fmaxf(a1z, b1z));
      // (every node has two valid children: a single-leaf tree is wrapped with the leaf in both slots, bvh_build.cpp)
      // Branch-free step: the pop is read speculatively (the slot below `top` always exists: the DONE sentinel sits
      // at the bottom) next to the node loads, the push is written speculatively (the slot at `top` is free; the LDS
      // stack has one spare level for it), and selects pick what applies — no exec-mask regions in the loop.
      // the far bound tbest as a compare of its own (tn <= min(tf, tbest) is tn <= tf and tn <= tbest): a register that comes
      // from outside the loop would have to be quieted (v_max x, x) before it may enter a min, in every step
      const bool h0 = (tn0 <= tf0) & (tn0 <= tbest);
      const bool h1 = (tn1 <= tf1) & (tn1 <= tbest);
      const bool first1 = h1 & (!h0 | (tn1 < tn0));  // descend into child 1 first (bitwise: a short-circuit here compiles to an exec-mask region)
      slot[STRIDE] = first1 ? cr.x : cr.y;
      ref = (h0 || h1) ? (first1 ? cr.y : cr.x) : popped;
      const uint32_t next_top = (h0 && h1) ? top + STRIDE : ((h0 || h1) ? top : top - STRIDE);
      top = BOUNDED ? min(next_top, limit) : next_top;
      }
      if ((uint32_t)__popcll(__ballot(!(ref & BVH_LEAF_BIT))) < min_lanes) break;
    }
  }

  // The same walk over 4-wide nodes (bvh.h: WideNode): one 64-byte node = four 16-byte loads, the boxes of up to four children
  // decoded from 8-bit planes — plane = origin + q * 2^e, so t = q * (2^e * idir) + (origin * idir + nood): the scaling by
  // the power of two is exact, the two fmas round once each, well inside the padding of the ray's origin (setup_space) —
  // the hit children ordered by entry distance with a five-comparator network on (distance bits, child), the nearest
  // followed and the others pushed nearest on top. Branch-free like the binary step: the pop is read speculatively, the
  // three pushes are always written, at fixed places above `top` (see below).
  DEV void wide_loop(const DeviceBvh& bvh, uint32_t* stack, uint32_t min_lanes, TraverseCounters& cnt) {
    const char* base = reinterpret_cast<const char*>(bvh.wide_nodes);
    const float tbest = hit.t;
    // The entry plane of an axis is the lower one where the ray runs up and the upper one where it runs down: picked once per
    // node for all four children (their plane bytes share a word), together with the padded origin term that goes with it
    // (noodL belongs to lower planes, noodH to upper ones, both pad outward whatever the direction: setup_space). Same
    // boxes as min / max over both planes, twelve instructions instead of twenty-four.
    // (orient_space has put the entry planes' term into noodL and the exit planes' into noodH already)
    const bool nx = sp.idir.x < 0.0f, ny = sp.idir.y < 0.0f, nz = sp.idir.z < 0.0f;
    const float cnx = sp.noodL.x, cfx = sp.noodH.x, cny = sp.noodL.y, cfy = sp.noodH.y, cnz = sp.noodL.z, cfz = sp.noodH.z;
    for (;;) {
      if (COUNT) {  // where the lanes are (the lanes without a ray are not in this loop)
        const unsigned long long here = __ballot(true), walking = __ballot(!(ref & BVH_LEAF_BIT)), leaf = __ballot(is_tri_leaf(ref));
        if (first_active_lane()) {
          cnt.st[0] += (uint32_t)__popcll(walking);
          cnt.st[1] += (uint32_t)__popcll(leaf);
          cnt.st[2] += (uint32_t)__popcll(here & ~walking & ~leaf);
          cnt.st[3] += 64u - (uint32_t)__popcll(here);
        }
      }
      if (!(ref & BVH_LEAF_BIT)) {
        const uint4* n = reinterpret_cast<const uint4*>(base + (ref << 6));
        const uint4 q0 = n[0], q1 = n[1], q2 = n[2];
        uint4 q3 = n[3];
        const uint32_t popped = stack[top - STRIDE];
        // all four loads in flight before anything is computed: left alone the scheduler sinks the references' load behind the
        // box tests — a second memory latency in every step
        asm volatile("" : "+v"(q3.x), "+v"(q3.y), "+v"(q3.z), "+v"(q3.w));
        if (COUNT) {
          cnt.nodes++;
          if (first_active_lane()) cnt.inner_slots += 64;
        }
        // 2^e * idir: the exponents are signed bytes (v_bfe_i32 sign-extends, v_ldexp_f32 scales exactly)
        const float ax = __builtin_amdgcn_ldexpf(sp.idir.x, (int)(q0.w << 24) >> 24);
        const float ay = __builtin_amdgcn_ldexpf(sp.idir.y, (int)(q0.w << 16) >> 24);
        const float az = __builtin_amdgcn_ldexpf(sp.idir.z, (int)(q0.w << 8) >> 24);
        const float ox = __uint_as_float(q0.x), oy = __uint_as_float(q0.y), oz = __uint_as_float(q0.z);
        const float enx = fmaf(ox, sp.idir.x, cnx), efx = fmaf(ox, sp.idir.x, cfx);
        const float eny = fmaf(oy, sp.idir.y, cny), efy = fmaf(oy, sp.idir.y, cfy);
        const float enz = fmaf(oz, sp.idir.z, cnz), efz = fmaf(oz, sp.idir.z, cfz);
        const uint32_t qnx = nx ? q1.y : q1.x, qfx = nx ? q1.x : q1.y;
        const uint32_t qny = ny ? q1.w : q1.z, qfy = ny ? q1.z : q1.w;
        const uint32_t qnz = nz ? q2.y : q2.x, qfz = nz ? q2.x : q2.y;
        uint32_t key[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const float tnx = fmaf((float)((qnx >> (8 * c)) & 0xFFu), ax, enx), tfx = fmaf((float)((qfx >> (8 * c)) & 0xFFu), ax, efx);
          const float tny = fmaf((float)((qny >> (8 * c)) & 0xFFu), ay, eny), tfy = fmaf((float)((qfy >> (8 * c)) & 0xFFu), ay, efy);
          const float tnz = fmaf((float)((qnz >> (8 * c)) & 0xFFu), az, enz), tfz = fmaf((float)((qfz >> (8 * c)) & 0xFFu), az, efz);
          const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin));
          const float tf = fminf(fminf(tfx, tfy), tfz);
          const bool h = (tn <= tf) & (tn <= tbest);  // (an unused slot: entry planes behind exit planes, and a copy of the first child's reference should a point-sized node let it through)
          key[c] = h ? ((__float_as_uint(tn) & 0x7FFFFFF8u) | (4u + (uint32_t)c)) : 0u;  // (tn >= tmin >= 0: its bits order like the number; a hit's key is never 0)
        }
        // descending — farthest first, misses (0) last: (0,1) (2,3) (0,2) (1,3) (1,2)
        uint32_t k0 = max(key[0], key[1]), k1 = min(key[0], key[1]), k2 = max(key[2], key[3]), k3 = min(key[2], key[3]);
        uint32_t t0 = max(k0, k2), t2 = min(k0, k2), t1 = max(k1, k3), t3 = min(k1, k3);
        k0 = t0;
        k1 = max(t1, t2);
        k2 = min(t1, t2);
        k3 = t3;
        auto child_ref = [&](uint32_t k) {  // reference number (k & 3): two bit masks and three bit selects, written as the
          // instructions (the compiler makes and / not / or triples of the C form) and as one block (it pads every asm
          // statement with wait states of its own)
          uint32_t m0, m1, r;
          asm("v_bfe_i32 %0, %3, 0, 1\n\tv_bfe_i32 %1, %3, 1, 1\n\tv_bfi_b32 %2, %0, %5, %4\n\tv_bfi_b32 %0, %0, %7, %6\n\tv_bfi_b32 %2, %1, %0, %2"
              : "=&v"(m0), "=&v"(m1), "=&v"(r)
              : "v"(k), "v"(q3.x), "v"(q3.y), "v"(q3.z), "v"(q3.w));
          return r;
        };
        const uint32_t r0 = child_ref(k0), r1 = child_ref(k1), r2 = child_ref(k2), r3 = child_ref(k3);
        // The pushes need no addresses of their own: the three farthest candidates go to top, top + 1, top + 2 in this order
        // whatever was hit. With h hits the nearest one is number h - 1 of the order: it is followed, the new top is
        // top + h - 1, and what lies at or above it — the followed child, the references of missed slots — is never read (the
        // stack has two spare levels for it; BOUNDED: the canary sits at `limit`, which a step with top >= limit - 2 writes
        // over, so overflowed() errs on the safe side).
        stack[top] = r0;
        stack[top + STRIDE] = r1;
        stack[top + 2u * STRIDE] = r2;
        const bool h1 = k0 != 0u, h2 = k1 != 0u, h3 = k2 != 0u, h4 = k3 != 0u;  // at least 1, 2, 3, 4 children hit
        ref = h4 ? r3 : (h3 ? r2 : (h2 ? r1 : (h1 ? r0 : popped)));
        const uint32_t next_top = top - STRIDE + ((h1 ? STRIDE : 0u) + (h2 ? STRIDE : 0u) + (h3 ? STRIDE : 0u) + (h4 ? STRIDE : 0u));
        top = BOUNDED ? min(next_top, limit) : next_top;
      }
      if ((uint32_t)__popcll(__ballot(!(ref & BVH_LEAF_BIT))) < min_lanes) break;
    }
  }

  // ref has the leaf bit: a sentinel, an instance, or up to 4 triangles
  DEV void leaf_step(const DeviceBvh& bvh, uint32_t* stack, TraverseCounters& cnt) {
    // (lanes here hold a leaf reference; "triangles" = a leaf that is neither a sentinel nor an instance entry)
    const bool entries_due = ENTRY_BATCH > 1 ? !__any(ref < TRAV_EXIT_INSTANCE && !(ref & BVH_INST_BIT)) : true;
    if (COUNT) {
      const unsigned long long tri = __ballot(is_tri_leaf(ref));
      if (first_active_lane()) {
        cnt.st[4] += 64;
        cnt.st[5] += (uint32_t)__popcll(tri);
        cnt.st[6] += (uint32_t)__popcll(__ballot(true) & ~tri);
      }
    }
    if (special_step(bvh, stack, cnt, entries_due)) return;
    triangles_step(bvh, stack, cnt);
  }

  // ref is a sentinel or an instance entry: handled here (returns true); a triangle leaf: returns false
  DEV bool special_step(const DeviceBvh& bvh, uint32_t* stack, TraverseCounters& cnt, bool entries_due) {
    if (ref >= TRAV_EXIT_INSTANCE) {
      if (ref == TRAV_EXIT_INSTANCE) {  // everything pushed inside the instance is consumed: back to world space
        if (SAVE_WORLD) {
          sp.o = o;
          sp.idir = w_idir;
          sp.noodL = w_noodL;
          sp.noodH = w_noodH;
          sp.Sx = w_Sx;
          sp.Sy = w_Sy;
          sp.Sz = w_Sz;
          sp.k = w_k;
        } else {
          setup_space(sp, o, d, bvh.scene_cx, bvh.scene_cy, bvh.scene_cz, bvh.scene_radius);
          orient_space();
        }
        id_bits = 0;
        pop(stack);
      }
      return true;  // TRAV_DONE stays
    }
    if (ref & BVH_INST_BIT) {
      // ENTRY_BATCH (the persistent kernel): entering an instance is ~250 instructions and a dependent load, run by the few
      // lanes that hold an entry while the rest of the wave waits (9 lanes on average on the bench scene: 8 % of the
      // kernel's instructions). An entry is therefore put off — the lane keeps its reference and comes back next round —
      // until at least ENTRY_BATCH lanes hold one or no lane of this round has triangles left to test.
      if (ENTRY_BATCH > 1) {
        const uint32_t entering = (uint32_t)__popcll(__ballot(true));  // (the lanes in this branch)
        if (entering < ENTRY_BATCH && !entries_due) return true;
      }
      const TlasEntry* e = bvh.entries + (ref & 0xFFFFu);
      const float4* ev = reinterpret_cast<const float4*>(e);
      const uint4 info = *reinterpret_cast<const uint4*>(ev + 3);
      if (info.z != TLAS_ENTRY_IDENTITY) {
        const float4 r0 = ev[0], r1 = ev[1], r2 = ev[2];
        const float4 sph = ev[4];
        const float m[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
        if (info.z == TLAS_ENTRY_SPHERE) {  // a sphere instance is tested right here (intersection.hlsli:79-89)
          if (COUNT) {
            cnt.tris++;
            if (first_active_lane()) cnt.tri_slots += 64;
          }
          float t;
          if (sphere_test(obj_point(m, o), obj_vector(m, d), sph.w, tmin, tmax, t)) {
            if (is_any()) {
              hit.ip = 0;
              ref = TRAV_DONE;
              return true;
            }
            const uint32_t ip = info.y | 0xFFFF0000u;  // instance | INVALID_PRIMITIVE << 16
            if (t < hit.t || (t == hit.t && hit.ip != 0xFFFFFFFFu && hit_key(ip) < hit_key(hit.ip))) {
              hit.t = t;
              hit.b1 = hit.b2 = 0.0f;
              hit.ip = ip;
            }
          }
          pop(stack);
          return true;
        }
        if (ALPHA && info.z == TLAS_ENTRY_VOLUME) {  // a volume instance: the slabs of its grid's bounding box (intersection.hlsli:93-113)
          if (COUNT) {
            cnt.tris++;
            if (first_active_lane()) cnt.tri_slots += 64;
          }
          float t;
          if (volume_test(bvh.volumes[info.x], obj_point(m, o), obj_vector(m, d), tmin, tmax, t)) {
            if (is_any()) {
              hit.ip = 0;
              ref = TRAV_DONE;
              return true;
            }
            const uint32_t ip = info.y | 0xFFFF0000u;
            if (t < hit.t || (t == hit.t && hit.ip != 0xFFFFFFFFu && hit_key(ip) < hit_key(hit.ip))) {
              hit.t = t;
              hit.b1 = hit.b2 = 0.0f;
              hit.ip = ip;
            }
          }
          pop(stack);
          return true;
        }
        setup_space(sp, obj_point(m, o), obj_vector(m, d), sph.x, sph.y, sph.z, sph.w);
        orient_space();
        id_bits = info.y;
        stack[top] = TRAV_EXIT_INSTANCE;
        top = BOUNDED ? min(top + STRIDE, limit) : top + STRIDE;
      }
      // identity entry (the merged world-space mesh): same ray, same (larger, still conservative) padding,
      // id_bits stays 0 because its triangles carry their instance index themselves
      ref = info.x;
      return true;
    }
    return false;
  }

  // One triangle of a leaf against this lane's ray, folded into the hit record with selects: no exec-mask regions (each costs
  // more than the arithmetic it would skip: the lanes of a wave wait for one another anyway). Returns whether an occlusion
  // lane found its hit.
  DEV bool one_triangle(const DeviceBvh& bvh, uint32_t index, TraverseCounters& cnt) {
    const float4* tv = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(bvh.tris) + (size_t)(index * 48u));
    const float4 v0 = tv[0], v1 = tv[1], v2 = tv[2];
    if (COUNT) {
      cnt.tris++;
      if (first_active_lane()) cnt.tri_slots += 64;
    }
    float t, b1, b2;
    bool candidate = tri_test(sp, xyz(v0), xyz(v1), xyz(v2), tmin, tmax, t, b1, b2);
    if (ALPHA && candidate) {
      // gAlphaTest: the candidate must pass the mask of its instance's material (instances that share a mesh may
      // have different materials, so the mask comes from the instance, the uvs from the leaf triangle)
      const uint32_t mask = bvh.alpha_test ? bvh.inst_alpha[(__float_as_uint(v0.w) | id_bits) & 0xFFFFu] : BVH_NO_ALPHA;
      if (mask != BVH_NO_ALPHA) {
        const float2* q = bvh.tri_uv + (size_t)index * 3u;
        const float2 u0 = q[0], u1 = q[1], u2 = q[2];
        const float u = u0.x + (u1.x - u0.x) * b1 + (u2.x - u0.x) * b2;  // shading_data.hlsli:2-6
        float v = u0.y + (u1.y - u0.y) * b1 + (u2.y - u0.y) * b2;
        if (bvh.flip_uvs) v = 1 - v;
        candidate = sample_image1(bvh, mask, u, v) >= 0.75f;
      }
    }
    const bool any_lane = is_any();
    const uint32_t ip = __float_as_uint(v0.w) | id_bits;
    const bool closer = candidate & !any_lane & ((t < hit.t) | ((t == hit.t) & (hit.ip != 0xFFFFFFFFu) & (hit_key(ip) < hit_key(hit.ip))));
    hit.t = closer ? t : hit.t;
    hit.b1 = closer ? b1 : hit.b1;
    hit.b2 = closer ? b2 : hit.b2;
    hit.ip = closer ? ip : hit.ip;
    hit.leaf = closer ? index : hit.leaf;
    return candidate & any_lane;
  }

  // ref is a triangle leaf: all its triangles, then the pop
  DEV void triangles_step(const DeviceBvh& bvh, uint32_t* stack, TraverseCounters& cnt) {
    const uint32_t first = (ref & 0x3FFFFFFFu) >> 2;
    const uint32_t count = (ref & 3u) + 1u;
    // an occlusion lane's verdict is collected in `occluded` and applied behind the loop
    bool occluded = false;
    for (uint32_t i = 0; i < count; i++) occluded |= one_triangle(bvh, first + i, cnt);
    // pop — or, for an occlusion lane that found its hit, the end of the ray (hit.ip = 0 says "occluded")
    const uint32_t popped = stack[top - STRIDE];
    hit.ip = occluded ? 0u : hit.ip;
    ref = occluded ? TRAV_DONE : popped;
    top = occluded ? top : top - STRIDE;
  }

  // one wave-synchronous round: inner nodes until (almost) every lane holds a leaf, then the leaves
  DEV void round(const DeviceBvh& bvh, uint32_t* stack, uint32_t min_lanes, TraverseCounters& cnt) {
    if (active()) inner_loop(bvh, stack, min_lanes, cnt);
    if (active() && (ref & BVH_LEAF_BIT)) leaf_step(bvh, stack, cnt);
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// The walk over the 8-wide compressed tree (bvh.h: Wide8Node), k_trace's "wide_bvh" = 3. What a lane holds of the tree is
// not a reference but two GROUPS: the node group G = (base index of a node's inner children, their hit bits in 31..24 in
// visiting order | the node's imask in 7..0) and the item group T = (first item of the node's leaf children, the items
// still to test in bits 23..0) — after Ylitie, Karras, Laine (HPG 2017). The stack holds groups (64-bit entries, LDS
// [level][lane]): a node step pushes at most ONE entry — the rest of the group it took its node from — whatever the
// number of children hit, needs no sort (the hit bits come out in the ray's octant order) and no child references (an
// inner child's index is base + popcount(imask below its slot)). States of a lane:
//   walking   gy > 0x00FFFFFF and ty == 0: it visits the node of the highest inner bit
//   items     ty != 0: it waits for the leaf phase, which tests one item per lane and iteration (a triangle; or, for a
//             top-level node's group, tx has WIDE8_ENTRY_BIT: enters / tests one top-level entry)
//   sentinel  neither: the last pop brought W8_EXIT (back to world space, pop again) or W8_DONE (finished) in gx
// Whenever a lane has nothing pending (no inner bits, no items) it pops at once; a popped entry with no inner bits is an
// item group (pushed when an instance is entered) or a sentinel (y == 0). Hits do not depend on the order of any of
// this (the contract), so frames are the ones the binary walk gives, bit for bit.
#define W8_DONE 0xFFFFFFFFu
#define W8_EXIT 0xFFFFFFFEu
#define W8_CANARY 0xFFFFFFFDu
#ifndef STHIP_WIDE8_STRIDE
#define STHIP_WIDE8_STRIDE 80u  // bytes from node to node in HBM
#endif
// SAVE_WORLD: the world-space constants are kept while an instance is walked (13 registers: 143 instead of 125, a wave per
// SIMD less) instead of being made again when it is left (~130 instructions per instance a ray enters)
template <bool COUNT, uint32_t STRIDE, bool ALPHA, bool BOUNDED, uint32_t ENTRY_BATCH, bool SAVE_WORLD = false>
struct Traversal8 {
  uint32_t limit;  // BOUNDED only: (levels - 1) * STRIDE
  bool any;
  DEV bool is_any() const { return any; }
  f3 o, d;
  float tmin, tmax;
  RaySpace sp;
  f3 w_idir, w_noodL, w_noodH;  // the world-space constants while an instance is walked (as Traversal's SAVE_WORLD)
  float w_Sx, w_Sy, w_Sz;
  int w_k;
  RayHit hit;
  uint32_t gx, gy, tx, ty;
  uint32_t top;  // stack height in entries * STRIDE
  uint32_t id_bits;
  uint32_t tri_min;

  static DEV uint2* column(uint32_t* stack) { return reinterpret_cast<uint2*>(stack); }
  DEV bool overflowed(const uint32_t* stack) const { return BOUNDED && reinterpret_cast<const uint2*>(stack)[limit].x != W8_CANARY; }
  DEV bool walking() const { return (gy > 0x00FFFFFFu) & (ty == 0u); }
  DEV bool active() const { return (ty != 0u) | (gy > 0x00FFFFFFu) | (gx != W8_DONE); }
  DEV void reset() {
    gx = W8_DONE;
    gy = ty = tx = 0u;
  }
  // entry planes' origin term into noodL, exit planes' into noodH (as Traversal::orient_space for the wide walk)
  DEV void orient_space() {
    const float lx = sp.noodL.x, ly = sp.noodL.y, lz = sp.noodL.z;
    const bool nx = sp.idir.x < 0.0f, ny = sp.idir.y < 0.0f, nz = sp.idir.z < 0.0f;
    sp.noodL.x = nx ? sp.noodH.x : lx;
    sp.noodH.x = nx ? lx : sp.noodH.x;
    sp.noodL.y = ny ? sp.noodH.y : ly;
    sp.noodH.y = ny ? ly : sp.noodH.y;
    sp.noodL.z = nz ? sp.noodH.z : lz;
    sp.noodH.z = nz ? lz : sp.noodH.z;
  }
  DEV void start(const DeviceBvh& bvh, uint32_t* stack, f3 ro, f3 rd, float t0, float t1) {
    o = ro;
    d = rd;
    tmin = t0;
    tmax = t1;
    hit.t = t1;
    hit.b1 = hit.b2 = 0.0f;
    hit.ip = 0xFFFFFFFFu;
    hit.leaf = 0xFFFFFFFFu;
    uint2* st = column(stack);
    st[0] = make_uint2(W8_DONE, 0u);
    top = STRIDE;
    if (BOUNDED) st[limit] = make_uint2(W8_CANARY, 0u);
    id_bits = 0;
    gx = bvh.root_ref;  // (an empty scene: BVH_INVALID_REF == W8_DONE)
    gy = bvh.root_ref == BVH_INVALID_REF ? 0u : 0x80000000u;  // a group of one: the root, whatever the octant (imask 0: index = base)
    tx = ty = 0u;
    setup_space(sp, ro, rd, bvh.scene_cx, bvh.scene_cy, bvh.scene_cz, bvh.scene_radius);
    orient_space();
    if (SAVE_WORLD) {
      w_idir = sp.idir;
      w_noodL = sp.noodL;
      w_noodH = sp.noodH;
      w_Sx = sp.Sx;
      w_Sy = sp.Sy;
      w_Sz = sp.Sz;
      w_k = sp.k;
    }
  }
  // BOUNDED: a write at `limit` (the canary's slot) makes the ray's result void (k_trace hands it to k_trace_deep), and the
  // walk must END there: a group's hit bits are in the visiting order of the space it was made in, so a walk that has lost
  // an exit sentinel and reads a world-space group inside an instance would compute child indices the node does not have.
  // Returns true when that happened.
  DEV bool push(uint2* st, uint32_t x, uint32_t y, bool keep) {
    const bool lost = BOUNDED && top == limit;
    st[top] = make_uint2(x, y);
    top = keep && !lost ? top + STRIDE : top;
    return lost;
  }
  DEV void give_up() {  // (BOUNDED, after a write at `limit`)
    gx = W8_DONE;
    gy = ty = 0u;
  }
  // the popped entry becomes the lane's state: a node group, or (no inner bits) an item group / a sentinel
  DEV void take(uint2 P) {
    gx = P.x;
    gy = P.y;
    tx = P.x;
    ty = P.y <= 0x00FFFFFFu ? P.y : 0u;
  }
  DEV void finish_occluded() {
    hit.ip = 0u;
    gx = W8_DONE;
    gy = ty = 0u;
  }

  DEV void node_loop(const DeviceBvh& bvh, uint32_t* stack, uint32_t min_lanes, TraverseCounters& cnt) {
    const char* base = reinterpret_cast<const char*>(bvh.wide8_nodes);
    uint2* st = column(stack);
    const float tbest = hit.t;
    const bool nx = sp.idir.x < 0.0f, ny = sp.idir.y < 0.0f, nz = sp.idir.z < 0.0f;
    const uint32_t octinv = (nx ? 0u : 1u) | (ny ? 0u : 2u) | (nz ? 0u : 4u);
    const uint32_t oct4 = octinv * 0x01010101u;
    const float cnx = sp.noodL.x, cfx = sp.noodH.x, cny = sp.noodL.y, cfy = sp.noodH.y, cnz = sp.noodL.z, cfz = sp.noodH.z;
    for (;;) {
      if (COUNT) {
        const unsigned long long here = __ballot(true), walk = __ballot(walking()), leaf = __ballot(ty != 0u && !(tx & WIDE8_ENTRY_BIT));
        if (first_active_lane()) {
          cnt.st[0] += (uint32_t)__popcll(walk);
          cnt.st[1] += (uint32_t)__popcll(leaf);
          cnt.st[2] += (uint32_t)__popcll(here & ~walk & ~leaf);
          cnt.st[3] += 64u - (uint32_t)__popcll(here);
        }
      }
      if (walking()) {
        const uint32_t bit = 31u - (uint32_t)__clz((int)gy);
        const uint32_t rest = gy & ~(1u << bit);
        const uint32_t slot = (bit - 24u) ^ octinv;
        const uint32_t index = gx + (uint32_t)__popc(gy & ((1u << slot) - 1u));  // (slot <= 7: only imask bits are counted)
        const bool lost = BOUNDED && top == limit;  // (this write takes the canary: see push())
        st[top] = make_uint2(gx, rest);  // the rest of this group: kept if it still has inner bits
        const uint32_t ntop = rest > 0x00FFFFFFu && !lost ? top + STRIDE : top;
        const uint4* n = reinterpret_cast<const uint4*>(base + __umul24(index, STHIP_WIDE8_STRIDE));
        const uint4 n0 = n[0], n1 = n[1], n2 = n[2], n3 = n[3];
        uint4 n4 = n[4];
        const uint2 P = st[ntop - STRIDE];  // what a node without hits falls back to (the DONE sentinel at the very bottom)
        asm volatile("" : "+v"(n4.x), "+v"(n4.y), "+v"(n4.z), "+v"(n4.w));  // all five loads in flight before anything is computed
        if (COUNT) {
          cnt.nodes++;
          if (first_active_lane()) cnt.inner_slots += 64;
        }
        const float ax = __builtin_amdgcn_ldexpf(sp.idir.x, (int)(n0.w << 24) >> 24);
        const float ay = __builtin_amdgcn_ldexpf(sp.idir.y, (int)(n0.w << 16) >> 24);
        const float az = __builtin_amdgcn_ldexpf(sp.idir.z, (int)(n0.w << 8) >> 24);
        const float ox = __uint_as_float(n0.x), oy = __uint_as_float(n0.y), oz = __uint_as_float(n0.z);
        const float enx = fmaf(ox, sp.idir.x, cnx), efx = fmaf(ox, sp.idir.x, cfx);
        const float eny = fmaf(oy, sp.idir.y, cny), efy = fmaf(oy, sp.idir.y, cfy);
        const float enz = fmaf(oz, sp.idir.z, cnz), efz = fmaf(oz, sp.idir.z, cfz);
        // entry / exit plane words of each axis (slots 0-3, slots 4-7), picked by the sign of the direction
        const uint32_t qnx[2] = {nx ? n2.z : n2.x, nx ? n2.w : n2.y}, qfx[2] = {nx ? n2.x : n2.z, nx ? n2.y : n2.w};
        const uint32_t qny[2] = {ny ? n3.z : n3.x, ny ? n3.w : n3.y}, qfy[2] = {ny ? n3.x : n3.z, ny ? n3.y : n3.w};
        const uint32_t qnz[2] = {nz ? n4.z : n4.x, nz ? n4.w : n4.y}, qfz[2] = {nz ? n4.x : n4.z, nz ? n4.y : n4.w};
        // per slot (byte-parallel over four): what a hit adds to the mask (meta >> 5) and where (meta & 31, inner children
        // moved to the ray's visiting order: ^ octinv)
        uint32_t bits4[2], at4[2];
#pragma unroll
        for (int w = 0; w < 2; w++) {
          const uint32_t m = w ? n1.w : n1.z;
          const uint32_t is_inner = (m & (m << 1)) & 0x10101010u;      // both of bits 3 and 4: a position of 24 or more
          const uint32_t low3 = (is_inner - (is_inner >> 3)) >> 1;     // 0x07 in the bytes of inner children
          at4[w] = (m ^ (oct4 & low3)) & 0x1F1F1F1Fu;
          bits4[w] = (m >> 5) & 0x07070707u;
        }
        uint32_t hitmask = 0u;
#pragma unroll
        for (int c = 0; c < 8; c++) {
          const int w = c >> 2, b = 8 * (c & 3);
          const float tnx = fmaf((float)((qnx[w] >> b) & 0xFFu), ax, enx), tfx = fmaf((float)((qfx[w] >> b) & 0xFFu), ax, efx);
          const float tny = fmaf((float)((qny[w] >> b) & 0xFFu), ay, eny), tfy = fmaf((float)((qfy[w] >> b) & 0xFFu), ay, efy);
          const float tnz = fmaf((float)((qnz[w] >> b) & 0xFFu), az, enz), tfz = fmaf((float)((qfz[w] >> b) & 0xFFu), az, efz);
          const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin));
          const float tf = fminf(fminf(tfx, tfy), tfz);
          const bool h = (tn <= tf) & (tn <= tbest);
          const uint32_t add = ((bits4[w] >> b) & 0xFFu) << ((at4[w] >> b) & 0xFFu);
          hitmask |= h ? add : 0u;
        }
        const bool nohit = hitmask == 0u;
        gx = nohit ? P.x : n1.x;
        gy = nohit ? P.y : ((hitmask & 0xFF000000u) | (n0.w >> 24));
        tx = nohit ? P.x : n1.y;
        ty = nohit ? (P.y <= 0x00FFFFFFu ? P.y : 0u) : (hitmask & 0x00FFFFFFu);
        top = nohit ? ntop - STRIDE : ntop;
        if (lost) give_up();
      }
      if ((uint32_t)__popcll(__ballot(walking())) < min_lanes) break;
    }
  }

  // One triangle against this lane's ray (as Traversal::one_triangle)
  DEV bool one_triangle(const DeviceBvh& bvh, uint32_t index, TraverseCounters& cnt) {
    const float4* tv = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(bvh.tris) + (size_t)(index * 48u));
    const float4 v0 = tv[0], v1 = tv[1], v2 = tv[2];
    if (COUNT) {
      cnt.tris++;
      if (first_active_lane()) cnt.tri_slots += 64;
    }
    float t, b1, b2;
    bool candidate = tri_test(sp, xyz(v0), xyz(v1), xyz(v2), tmin, tmax, t, b1, b2);
    if (ALPHA && candidate) {
      const uint32_t mask = bvh.alpha_test ? bvh.inst_alpha[(__float_as_uint(v0.w) | id_bits) & 0xFFFFu] : BVH_NO_ALPHA;
      if (mask != BVH_NO_ALPHA) {
        const float2* q = bvh.tri_uv + (size_t)index * 3u;
        const float2 u0 = q[0], u1 = q[1], u2 = q[2];
        const float u = u0.x + (u1.x - u0.x) * b1 + (u2.x - u0.x) * b2;  // shading_data.hlsli:2-6
        float v = u0.y + (u1.y - u0.y) * b1 + (u2.y - u0.y) * b2;
        if (bvh.flip_uvs) v = 1 - v;
        candidate = sample_image1(bvh, mask, u, v) >= 0.75f;
      }
    }
    const uint32_t ip = __float_as_uint(v0.w) | id_bits;
    const bool closer = candidate & !any & ((t < hit.t) | ((t == hit.t) & (hit.ip != 0xFFFFFFFFu) & (hit_key(ip) < hit_key(hit.ip))));
    hit.t = closer ? t : hit.t;
    hit.b1 = closer ? b1 : hit.b1;
    hit.b2 = closer ? b2 : hit.b2;
    hit.ip = closer ? ip : hit.ip;
    hit.leaf = closer ? index : hit.leaf;
    return candidate & any;
  }
  // a sphere or volume entry's verdict folded into the hit record; true: an occlusion lane found its hit
  DEV bool fold_entry_hit(bool found, float t, uint32_t instance) {
    if (!found) return false;
    if (any) return true;
    const uint32_t ip = instance | 0xFFFF0000u;  // instance | INVALID_PRIMITIVE << 16
    if (t < hit.t || (t == hit.t && hit.ip != 0xFFFFFFFFu && hit_key(ip) < hit_key(hit.ip))) {
      hit.t = t;
      hit.b1 = hit.b2 = 0.0f;
      hit.ip = ip;
    }
    return false;
  }

  // the lanes that are not walking: sentinels, one top-level entry per lane, then triangles
  DEV void leaf_step(const DeviceBvh& bvh, uint32_t* stack, TraverseCounters& cnt) {
    uint2* st = column(stack);
    if (COUNT) {
      const unsigned long long tri = __ballot(ty != 0u && !(tx & WIDE8_ENTRY_BIT));
      if (first_active_lane()) {
        cnt.st[4] += 64;
        cnt.st[5] += (uint32_t)__popcll(tri);
        cnt.st[6] += (uint32_t)__popcll(__ballot(true) & ~tri);
      }
    }
    if (ty == 0u && gx == W8_EXIT) {  // (not walking and no items: a sentinel) everything pushed inside the instance is consumed
      if (SAVE_WORLD) {
        sp.o = o;
        sp.idir = w_idir;
        sp.noodL = w_noodL;
        sp.noodH = w_noodH;
        sp.Sx = w_Sx;
        sp.Sy = w_Sy;
        sp.Sz = w_Sz;
        sp.k = w_k;
      } else {
        setup_space(sp, o, d, bvh.scene_cx, bvh.scene_cy, bvh.scene_cz, bvh.scene_radius);
        orient_space();
      }
      id_bits = 0;
      top -= STRIDE;
      take(st[top]);
    }
    const bool has_entry = ty != 0u && (tx & WIDE8_ENTRY_BIT);
    // entering an instance is ~250 instructions and a dependent load run by the few lanes that hold an entry: put off until
    // ENTRY_BATCH lanes hold one or no lane of this round has triangles to test (as Traversal::special_step)
    const bool entries_due = ENTRY_BATCH > 1 ? !__any(ty != 0u && !(tx & WIDE8_ENTRY_BIT)) : true;
    if (has_entry && (entries_due || (uint32_t)__popcll(__ballot(true)) >= ENTRY_BATCH)) {
      const uint32_t bit = (uint32_t)__ffs((int)ty) - 1u;
      ty &= ty - 1u;
      const TlasEntry* e = bvh.entries + ((tx & ~WIDE8_ENTRY_BIT) + bit);
      const float4* ev = reinterpret_cast<const float4*>(e);
      const uint4 info = *reinterpret_cast<const uint4*>(ev + 3);
      bool in_place = false, occluded = false, lost = false;
      if (info.z != TLAS_ENTRY_IDENTITY) {
        const float4 r0 = ev[0], r1 = ev[1], r2 = ev[2];
        const float4 sph = ev[4];
        const float m[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
        if (info.z == TLAS_ENTRY_SPHERE) {  // tested right here (intersection.hlsli:79-89)
          if (COUNT) {
            cnt.tris++;
            if (first_active_lane()) cnt.tri_slots += 64;
          }
          float t = 0.0f;
          const bool found = sphere_test(obj_point(m, o), obj_vector(m, d), sph.w, tmin, tmax, t);
          occluded = fold_entry_hit(found, t, info.y);
          in_place = true;
        } else if (ALPHA && info.z == TLAS_ENTRY_VOLUME) {  // the slabs of its grid's bounding box (intersection.hlsli:93-113)
          if (COUNT) {
            cnt.tris++;
            if (first_active_lane()) cnt.tri_slots += 64;
          }
          float t = 0.0f;
          const bool found = volume_test(bvh.volumes[info.x], obj_point(m, o), obj_vector(m, d), tmin, tmax, t);
          occluded = fold_entry_hit(found, t, info.y);
          in_place = true;
        } else {
          // what is pending at this level waits on the stack: the rest of the node group, the rest of the entry group
          lost = push(st, gx, gy, gy > 0x00FFFFFFu);
          lost |= push(st, tx, ty, ty != 0u);
          lost |= push(st, W8_EXIT, 0u, true);
          setup_space(sp, obj_point(m, o), obj_vector(m, d), sph.x, sph.y, sph.z, sph.w);
          orient_space();
          id_bits = info.y;
        }
      } else {  // (the merged mesh is spliced into the top level; an entry of it all the same: same space, no id bits)
        lost = push(st, gx, gy, gy > 0x00FFFFFFu);
        lost |= push(st, tx, ty, ty != 0u);
      }
      if (lost) {
        give_up();
      } else if (!in_place) {
        gx = info.x;
        gy = 0x80000000u;
        ty = 0u;
      } else if (occluded) {
        finish_occluded();
      } else if (ty == 0u && gy <= 0x00FFFFFFu) {
        top -= STRIDE;
        take(st[top]);
      }
    }
    // triangles: one per lane and iteration, while enough lanes hold one (at least once)
    for (;;) {
      const bool has = ty != 0u && !(tx & WIDE8_ENTRY_BIT);
      if (has) {
        const uint32_t bit = (uint32_t)__ffs((int)ty) - 1u;
        ty &= ty - 1u;
        const uint2 P = st[top - STRIDE];
        const bool occluded = one_triangle(bvh, tx + bit, cnt);
        const bool pop = !occluded & (ty == 0u) & (gy <= 0x00FFFFFFu);
        hit.ip = occluded ? 0u : hit.ip;
        gx = occluded ? W8_DONE : (pop ? P.x : gx);
        gy = occluded ? 0u : (pop ? P.y : gy);
        tx = pop ? P.x : tx;
        ty = occluded ? 0u : (pop ? (P.y <= 0x00FFFFFFu ? P.y : 0u) : ty);
        top = pop ? top - STRIDE : top;
      }
      if ((uint32_t)__popcll(__ballot(ty != 0u && !(tx & WIDE8_ENTRY_BIT))) < tri_min) break;
    }
  }

  DEV void round(const DeviceBvh& bvh, uint32_t* stack, uint32_t min_lanes, TraverseCounters& cnt) {
    if (active()) node_loop(bvh, stack, min_lanes, cnt);
    if (active() && !walking()) leaf_step(bvh, stack, cnt);
  }
};

// One ray to completion (ray batches, tests). Returns true if something was hit; ANY_HIT stops at the
// first accepted triangle (hit.ip = 0). `stack` points at this lane's column of the LDS stack.
template <int MODE, bool COUNT, uint32_t STRIDE, bool ALPHA = false>
DEV bool traverse(const DeviceBvh& bvh, f3 o, f3 d, float tmin, float tmax, uint32_t* stack, RayHit& hit, TraverseCounters& cnt) {
  Traversal<MODE, COUNT, STRIDE, ALPHA> tr;
  tr.any = MODE == TRAV_ANY;
  tr.start(bvh, stack, o, d, tmin, tmax);
  while (tr.active()) tr.round(bvh, stack, 1, cnt);
  hit = tr.hit;
  return hit.ip != 0xFFFFFFFFu;
}

// Work queues of the wavefront pipeline are cut into QUEUE_SEGMENTS segments, each with its own control words
// (entries produced; dequeue head — on separate 128-byte lines). One word serialises at ~88 atomics/us on MI355X
// (device-scope atomics resolve outside the per-XCD L2), which bounds a producer kernel that appends once per wave
// and forces a consumer into big chunks with an uneven tail; eight lines give eight times that. Workgroups go
// round-robin over the 8 XCDs, so segment = blockIdx % 8 keeps a path on "its" XCD from kernel to kernel.
#define QUEUE_SEGMENTS 8u
#define QCTL_STRIDE 32u  // 64-bit words per segment: two 128-byte lines, so that reading the size (constant while a
                         // consumer runs) never touches the line its dequeue atomics keep busy
#define QCTL_SIZE 0u     // word: entries produced
#define QCTL_KEPT 1u     // word: entries k_cull_terminal kept for k_shade (kernels.h)
#define QCTL_ANSWERED 2u // word: rays of this bounce that k_shade answered instead of queueing them (aims_at_emitter, kernels.h)
#define QCTL_HEAD 16u    // word: dequeue head of the consuming trace kernel
// control line of (kind: 0 = path queue entering bounce `depth`, 1 = shadow rays of bounce `depth`; depth < 64; segment)
DEV unsigned long long* queue_ctl(unsigned long long* qctl, uint32_t kind, uint32_t depth, uint32_t seg) {
  return qctl + (size_t)((kind * 64u + depth) * QUEUE_SEGMENTS + seg) * QCTL_STRIDE;
}

// Wave-level work distribution for the persistent trace kernels. A wave starts on segment blockIdx % 8 and moves
// on to the next when its own is dry (a dry segment stays dry, so each is probed at most once); it takes WORK_CHUNK
// entries per atomic and hands them to idle lanes ranked with ballot + popcount.
#ifndef WORK_CHUNK
#define WORK_CHUNK 64u
#endif
struct WaveWork {
  uint32_t next, end;
  uint32_t cur, dry;  // segment in use, segments found empty
  bool exhausted;
  DEV void init() {
    next = end = 0;
    cur = blockIdx.x % QUEUE_SEGMENTS;
    dry = 0;
    exhausted = false;
  }
  // ctl: control line of segment 0. Segment s holds entries [s * stride, s * stride + len_s) where len_s is the
  // produced count (stride != 0) or, for the un-queued first bounce (stride == 0), an equal share of n.
  // For the lanes in `want`: returns the entry index each one takes, or 0xFFFFFFFF.
  DEV uint32_t take(bool want, unsigned long long* ctl, uint32_t stride, uint32_t n) {
    const unsigned long long mask = __ballot(want);
    if (!mask) return 0xFFFFFFFFu;
    const uint32_t lane = threadIdx.x & 63u;
    if (next >= end && !exhausted) {
      const uint32_t per = (((n + QUEUE_SEGMENTS - 1u) / QUEUE_SEGMENTS) + 63u) & ~63u;
      for (;;) {
        unsigned long long* line = ctl + (size_t)cur * QCTL_STRIDE;
        uint32_t lo, len;
        if (stride) {
          lo = cur * stride;
          len = (uint32_t)line[QCTL_SIZE];
        } else {
          lo = cur * per;
          len = lo < n ? (lo + per < n ? per : n - lo) : 0u;
        }
        uint32_t got = 0xFFFFFFFFu;
        if (len) {
          unsigned long long b = 0;
          if (lane == 0) b = atomicAdd(&line[QCTL_HEAD], (unsigned long long)WORK_CHUNK);
          // take() runs in wave-uniform control flow (all 64 lanes), so the first lane is lane 0; a scalar keeps the
          // chunk bounds out of the vector registers
          const uint32_t b32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b > 0xFFFFFFFFull ? 0xFFFFFFFFull : b));
          if (b32 < len) got = b32;
        }
        if (got != 0xFFFFFFFFu) {
          next = lo + got;
          end = lo + (got + WORK_CHUNK < len ? got + WORK_CHUNK : len);
          break;
        }
        cur = cur + 1u == QUEUE_SEGMENTS ? 0u : cur + 1u;
        if (++dry == QUEUE_SEGMENTS) {
          exhausted = true;
          break;
        }
      }
    }
    const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    const uint32_t idx = next + rank;
    const uint32_t avail = end - next;
    const uint32_t wanted = (uint32_t)__popcll(mask);
    next += wanted < avail ? wanted : avail;
    return (want && rank < avail) ? idx : 0xFFFFFFFFu;
  }
};
