// traverse.h — BVH traversal and ray/triangle intersection for gfx950 (wave64).
//
// Replaces rayQuery.TraceRayInline/Proceed (src/Shaders/common/intersection.hlsli:68-75), which the
// reference leaves to the Vulkan driver. The hit contract (object-space watertight triangle test,
// closest hit over all triangles with (t, instance, primitive) ordering, any-hit = existence) is the
// one stated in include/sthip.h and oracle/stratum_oracle.cpp; the acceleration structure must never
// change the answer, so every box test is conservative:
//   * the ray's origin is padded per ray by P = 4e-6 * (L1 distance to the bounding sphere centre +
//     radius) of the space it is traversing, which bounds the rounding of the slab test and of the
//     triangle test for every box the ray can reach;
//   * slab distances use (plane - origin) * inv_dir (no fused form: it cancels catastrophically for
//     near-axis-parallel rays), |dir| is clamped away from 0, comparisons are inclusive.
// One lane = one ray. The traversal stack is in LDS, laid out [level][lane] so that a wave's pushes
// and pops of one level hit 64 consecutive banks.
#pragma once

#include "bvh.h"
#include "device_math.h"

struct DeviceBvh {
  const float4* nodes;  // BvhNode = 4 x float4
  const float4* tris;   // BvhTri = 3 x float4
  const TlasEntry* entries;
  uint32_t root_ref;
  uint32_t top_is_world_blas;
  uint32_t stack_depth;
  float scene_cx, scene_cy, scene_cz, scene_radius;
};

struct RayHit {
  float t, b1, b2;
  uint32_t ip;  // instance | primitive << 16; 0xFFFFFFFF = miss
};

struct TraverseCounters {
  uint32_t nodes, tris;
};

// per-space ray constants
struct RaySpace {
  f3 o, d;
  f3 idir;
  f3 oL, oH;  // origin +/- padding: (lo - oL) and (hi - oH) are the padded slab numerators
  float Sx, Sy, Sz;
  int kx, ky, kz;
};

DEV float safe_rcp_dir(float d) {
  const float eps = 1e-30f;
  const float dd = fabsf(d) < eps ? copysignf(eps, d) : d;
  return 1.0f / dd;
}

DEV void setup_space(RaySpace& s, f3 o, f3 d, float cx, float cy, float cz, float radius) {
  s.o = o;
  s.d = d;
  s.idir = F3(safe_rcp_dir(d.x), safe_rcp_dir(d.y), safe_rcp_dir(d.z));
  const float P = 4e-6f * (fabsf(o.x - cx) + fabsf(o.y - cy) + fabsf(o.z - cz) + radius);
  s.oL = F3(o.x + P, o.y + P, o.z + P);
  s.oH = F3(o.x - P, o.y - P, o.z - P);
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
  s.kx = kx;
  s.ky = ky;
  s.kz = kz;
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

// closest-hit ordering key: instance first, then primitive
DEV uint32_t hit_key(uint32_t ip) { return (ip << 16) | (ip >> 16); }

// Watertight ray/triangle test, arithmetic exactly as the contract states. Edge functions are
// products rounded separately (no fma) so that a shared edge evaluates antisymmetrically.
DEV bool tri_test(const RaySpace& s, f3 p0, f3 p1, f3 p2, float tmin, float tmax, float& t, float& b1, float& b2) {
  const f3 A = p0 - s.o, B = p1 - s.o, C = p2 - s.o;
  const float Akz = comp3(A, s.kz), Bkz = comp3(B, s.kz), Ckz = comp3(C, s.kz);
  const float Ax = fmaf(-s.Sx, Akz, comp3(A, s.kx)), Ay = fmaf(-s.Sy, Akz, comp3(A, s.ky));
  const float Bx = fmaf(-s.Sx, Bkz, comp3(B, s.kx)), By = fmaf(-s.Sy, Bkz, comp3(B, s.ky));
  const float Cx = fmaf(-s.Sx, Ckz, comp3(C, s.kx)), Cy = fmaf(-s.Sy, Ckz, comp3(C, s.ky));
  const float U = Cx * By - Cy * Bx;
  const float V = Ax * Cy - Ay * Cx;
  const float W = Bx * Ay - By * Ax;
  if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
  const float det = U + V + W;
  if (det == 0.0f) return false;
  const float Az = s.Sz * Akz, Bz = s.Sz * Bkz, Cz = s.Sz * Ckz;
  const float T = fmaf(W, Cz, fmaf(V, Bz, U * Az));
  const float rcp = 1.0f / det;
  const float tt = T * rcp;
  if (!(tt > tmin && tt < tmax)) return false;
  t = tt;
  b1 = V * rcp;
  b2 = W * rcp;
  return true;
}

#define TRAV_DONE 0xFFFFFFFFu

// Per-lane traversal state machine. `ref` is the next thing to process: an inner-node index, a leaf
// reference, or TRAV_DONE. The drivers below run it "while-while": every lane of the wave walks inner
// nodes until it holds a leaf (or is done), then the wave processes leaves together, so that the long
// triangle code is not executed once per inner-node step of some other lane.
template <bool ANY_HIT, bool COUNT>
struct Traversal {
  f3 o, d;  // world-space ray
  float tmin, tmax;
  RaySpace sp;  // the space currently being traversed (world, or the object space of an instance)
  RayHit hit;
  uint32_t ref;
  int top;      // stack height
  int inst_sp;  // stack height at which the current instance was entered, -1 in world space
  uint32_t id_bits;

  DEV bool active() const { return ref != TRAV_DONE; }
  DEV void reset() { ref = TRAV_DONE; }

  DEV void start(const DeviceBvh& bvh, f3 ro, f3 rd, float t0, float t1) {
    o = ro;
    d = rd;
    tmin = t0;
    tmax = t1;
    hit.t = t1;
    hit.b1 = hit.b2 = 0.0f;
    hit.ip = 0xFFFFFFFFu;
    top = 0;
    inst_sp = -1;
    id_bits = 0;
    ref = bvh.root_ref;  // BVH_INVALID_REF == TRAV_DONE for an empty scene
    setup_space(sp, ro, rd, bvh.scene_cx, bvh.scene_cy, bvh.scene_cz, bvh.scene_radius);
  }

  DEV void pop(const DeviceBvh& bvh, const uint32_t* stack, uint32_t stride) {
    if (top == inst_sp) {  // everything pushed inside the instance is consumed: back to world space
      setup_space(sp, o, d, bvh.scene_cx, bvh.scene_cy, bvh.scene_cz, bvh.scene_radius);
      id_bits = 0;
      inst_sp = -1;
    }
    if (top == 0) {
      ref = TRAV_DONE;
    } else {
      top--;
      ref = stack[(uint32_t)top * stride];
    }
  }

  // ref is an inner node: test both children, descend into the nearer, push the farther
  DEV void inner_step(const DeviceBvh& bvh, uint32_t* stack, uint32_t stride, TraverseCounters& cnt) {
    const float4* n = bvh.nodes + (size_t)ref * 4;
    const float4 n0 = n[0], n1 = n[1], nz = n[2];
    const uint4 cr = *reinterpret_cast<const uint4*>(n + 3);
    if (COUNT) cnt.nodes++;
    const float tbest = ANY_HIT ? tmax : hit.t;
    const float a0x = (n0.x - sp.oL.x) * sp.idir.x, b0x = (n0.y - sp.oH.x) * sp.idir.x;
    const float a0y = (n0.z - sp.oL.y) * sp.idir.y, b0y = (n0.w - sp.oH.y) * sp.idir.y;
    const float a0z = (nz.x - sp.oL.z) * sp.idir.z, b0z = (nz.y - sp.oH.z) * sp.idir.z;
    const float a1x = (n1.x - sp.oL.x) * sp.idir.x, b1x = (n1.y - sp.oH.x) * sp.idir.x;
    const float a1y = (n1.z - sp.oL.y) * sp.idir.y, b1y = (n1.w - sp.oH.y) * sp.idir.y;
    const float a1z = (nz.z - sp.oL.z) * sp.idir.z, b1z = (nz.w - sp.oH.z) * sp.idir.z;
    const float tn0 = fmaxf(fmaxf(fminf(a0x, b0x), fminf(a0y, b0y)), fmaxf(fminf(a0z, b0z), tmin));
    const float tf0 = fminf(fminf(fmaxf(a0x, b0x), fmaxf(a0y, b0y)), fminf(fmaxf(a0z, b0z), tbest));
    const float tn1 = fmaxf(fmaxf(fminf(a1x, b1x), fminf(a1y, b1y)), fmaxf(fminf(a1z, b1z), tmin));
    const float tf1 = fminf(fminf(fmaxf(a1x, b1x), fmaxf(a1y, b1y)), fminf(fmaxf(a1z, b1z), tbest));
    const bool h0 = (tn0 <= tf0) && (cr.x != BVH_INVALID_REF);
    const bool h1 = (tn1 <= tf1) && (cr.y != BVH_INVALID_REF);
    if (h0 && h1) {
      const bool swap = tn1 < tn0;
      const uint32_t nearc = swap ? cr.y : cr.x;
      const uint32_t farc = swap ? cr.x : cr.y;
      // the builder caps tree depth at stack_depth - 2, so this bound is never reached; it only
      // keeps a malformed structure from writing outside the LDS allocation
      if (top < (int)bvh.stack_depth) {
        stack[(uint32_t)top * stride] = farc;
        top++;
      }
      ref = nearc;
    } else if (h0) {
      ref = cr.x;
    } else if (h1) {
      ref = cr.y;
    } else {
      pop(bvh, stack, stride);
    }
  }

  // ref is a leaf: an instance (move the ray into its object space) or up to 4 triangles
  DEV void leaf_step(const DeviceBvh& bvh, uint32_t* stack, uint32_t stride, TraverseCounters& cnt) {
    if (ref & BVH_INST_BIT) {
      const TlasEntry* e = bvh.entries + (ref & 0xFFFFu);
      const float4* ev = reinterpret_cast<const float4*>(e);
      const float4 r0 = ev[0], r1 = ev[1], r2 = ev[2];
      const uint4 info = *reinterpret_cast<const uint4*>(ev + 3);
      const float4 sph = ev[4];
      const float m[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
      f3 oo = o, od = d;
      if (!info.z) {
        oo = obj_point(m, o);
        od = obj_vector(m, d);
      }
      setup_space(sp, oo, od, sph.x, sph.y, sph.z, sph.w);
      id_bits = info.y;
      inst_sp = top;
      ref = info.x;
      return;
    }
    const uint32_t first = (ref & 0x3FFFFFFFu) >> 2;
    const uint32_t count = (ref & 3u) + 1u;
    for (uint32_t i = 0; i < count; i++) {
      const float4* tv = bvh.tris + (size_t)(first + i) * 3;
      const float4 v0 = tv[0], v1 = tv[1], v2 = tv[2];
      if (COUNT) cnt.tris++;
      float t, b1, b2;
      if (tri_test(sp, xyz(v0), xyz(v1), xyz(v2), tmin, tmax, t, b1, b2)) {
        if (ANY_HIT) {
          hit.ip = 0;
          ref = TRAV_DONE;
          return;
        }
        const uint32_t ip = __float_as_uint(v0.w) | id_bits;
        if (t < hit.t || (t == hit.t && hit.ip != 0xFFFFFFFFu && hit_key(ip) < hit_key(hit.ip))) {
          hit.t = t;
          hit.b1 = b1;
          hit.b2 = b2;
          hit.ip = ip;
        }
      }
    }
    pop(bvh, stack, stride);
  }

  // one wave-synchronous round: inner nodes until every lane holds a leaf, then the leaves
  DEV void round(const DeviceBvh& bvh, uint32_t* stack, uint32_t stride, TraverseCounters& cnt) {
    while (active() && !(ref & BVH_LEAF_BIT)) inner_step(bvh, stack, stride, cnt);
    if (active()) leaf_step(bvh, stack, stride, cnt);
  }
};

// One ray to completion (ray batches, tests). Returns true if something was hit; ANY_HIT stops at the
// first accepted triangle (hit.ip = 0). `stack` points at this lane's column of the LDS stack.
template <bool ANY_HIT, bool COUNT>
DEV bool traverse(const DeviceBvh& bvh, f3 o, f3 d, float tmin, float tmax, uint32_t* stack, uint32_t stride, RayHit& hit, TraverseCounters& cnt) {
  Traversal<ANY_HIT, COUNT> tr;
  tr.start(bvh, o, d, tmin, tmax);
  while (tr.active()) tr.round(bvh, stack, stride, cnt);
  hit = tr.hit;
  return hit.ip != 0xFFFFFFFFu;
}

// Wave-level work distribution for the persistent trace kernels. Every wave owns a private chunk
// [next, end) of the queue, refilled with ONE global atomic per WORK_CHUNK rays (a single head word
// saturates at ~88 dequeues/us on MI355X); idle lanes take consecutive entries of the chunk, ranked
// with ballot + popcount.
#define WORK_CHUNK 256u
struct WaveWork {
  uint32_t next, end;
  bool exhausted;
  DEV void init() {
    next = end = 0;
    exhausted = false;
  }
  // For the lanes in `want`: returns the queue index each one takes, or 0xFFFFFFFF.
  DEV uint32_t take(bool want, unsigned long long* head, uint32_t n) {
    const unsigned long long mask = __ballot(want);
    if (!mask) return 0xFFFFFFFFu;
    const uint32_t lane = threadIdx.x & 63u;
    if (next >= end && !exhausted) {
      uint32_t base = 0;
      if (lane == 0) base = (uint32_t)atomicAdd(head, (unsigned long long)WORK_CHUNK);
      base = (uint32_t)__shfl((int)base, 0, 64);
      if (base >= n) {
        exhausted = true;
      } else {
        next = base;
        end = base + WORK_CHUNK < n ? base + WORK_CHUNK : n;
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
