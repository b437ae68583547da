// stratum_oracle.cpp — CPU restatement of Stratum's path-tracing hot path.
//
// TEST INFRASTRUCTURE ONLY. Nothing in the product (stratum_amd/, libstratum_hip.so)
// may include, link or call this file; only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg use it, as the checker.
//
// What it restates (paths relative to /root/reference/src/Shaders):
//   kernels/renderers/bdpt.hlsl   sample_visibility (:149-300), trace_shadows (:302-326)
//   common/path.hlsli             PathIntegrator: trace (:1003-1044), next_vertex (:955-998,1048-1075),
//                                 connect_light (:311-366), sample_Le (:141-164), DirectLightSample (:166-222),
//                                 eval_emission (:847-894), sample_direction (:898-952),
//                                 russian_roulette (:829-845), mis (:8-15), shading_normal_correction (:67-98)
//   common/intersection.hlsli     trace_ray contract (:65-191), trace_visibility_ray (:192-239), ray_offset (:44-62)
//   common/light.hlsli            sample_point_on_light (:37-152), point_on_light_pdf (:154-174)
//   common/shading_data.hlsli     make_triangle_shading_data (:2-73)
//   common/rng.hlsli              pcg4d and the counter RNG (:22-47)
//   materials/disney_*.hlsli      DisneyMaterial load/eval/sample and its four lobes
//   microfacet.h, common.h, transform.h, bitfield.h, scene.h, bdpt.h, shading_data.h
//   kernels/temporal_accumulation.hlsl:102-131  running mean that defines N samples per pixel
//   and, from SURVEY.md §8f: environment.h / dist2.h, light.hlsli sphere lights, image_value.h, alpha test
//   (intersection.hlsli:117-131), presample_lights / sample_photons / add_light_trace (bdpt.hlsl:84-147,328-338),
//   connect_view / connect_light_subpath / connect_light_vertex / connect_light_reservoir (path.hlsli:368-822),
//   sample_texel (bdpt_util.hlsli:85-180), materials/medium.hlsli + the medium-aware trace_ray / trace_visibility_ray
//   (intersection.hlsli:192-285) over NanoVDB grids (the PNanoVDB.h subset those call)
//
// PARITY PINNING: the reference ships no tests, golden vectors or fixtures,
// cannot be built here (needs Vulkan, Eigen, Slang fetched from the network) and has no CPU
// path. This oracle is therefore pinned by (i) published known answers for the integer
// hashes, (ii) analytic checks (furnace test, BSDF/pdf consistency, brute-force traversal),
// (iii) the two pieces of the reference that DO compile from their own sources here (oracle/_ref,
// `make -C oracle ref`): the vendored stb_image_write (HDR export, byte-identical files) and the
// vendored NanoVDB 32.3 (the grid reader: 6000 values, bounds, maxima and the map functions read back
// through the reference's own PNanoVDB.h; tests/golden/fog_sphere.npz) — see tests/. The BVH build, ray/triangle test, texture filtering and the transcendental
// intrinsics live in the Vulkan driver / shader compiler and are "parity unpinned" against
// the reference; for those the contract is the one written here and in include/sthip_detmath.h.
//
// Arithmetic: IEEE binary32, no contraction (-ffp-contract=off), fmaf only where written.

#include <stdint.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <atomic>
#include <functional>
#include <thread>
#include <vector>
#include <map>
#include <tuple>

#include "../include/sthip.h"
#include "../include/sthip_detmath.h"

namespace {

// ---------------------------------------------------------------------------------------------
// small vector algebra. Every operation is written out; evaluation order is part of the contract.
// ---------------------------------------------------------------------------------------------
struct v3 {
  float x, y, z;
};
inline v3 V3(float x, float y, float z) { return v3{x, y, z}; }
inline v3 V3(float s) { return v3{s, s, s}; }
inline v3 operator+(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline v3 operator-(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline v3 operator*(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline v3 operator*(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
inline v3 operator*(float s, v3 a) { return V3(s * a.x, s * a.y, s * a.z); }
inline v3 operator/(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
inline v3 operator/(v3 a, v3 b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline v3 operator-(v3 a) { return V3(-a.x, -a.y, -a.z); }
inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline v3 cross(v3 a, v3 b) { return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float len_sqr(v3 a) { return dot(a, a); }
inline float length(v3 a) { return sqrtf(dot(a, a)); }
// HLSL normalize: v * rsqrt(dot(v,v)); pinned here as one division and three multiplies
inline v3 normalize(v3 a) {
  const float inv = 1.0f / sqrtf(dot(a, a));
  return a * inv;
}
inline float pow2(float x) { return x * x; }
inline bool all_le0(v3 a) { return a.x <= 0 && a.y <= 0 && a.z <= 0; }
inline bool any_gt0(v3 a) { return a.x > 0 || a.y > 0 || a.z > 0; }
inline bool any_nan(v3 a) { return a.x != a.x || a.y != a.y || a.z != a.z; }
inline float sgn(float x) { return (float)((x > 0) - (x < 0)); }  // HLSL sign()
inline float lerpf(float a, float b, float t) { return a + t * (b - a); }  // HLSL lerp
inline v3 lerp3(v3 a, v3 b, float t) { return a + (b - a) * t; }
inline float luminance(v3 c) { return dot(c, V3(0.2126f, 0.7152f, 0.0722f)); }  // common.h:66-68

const float POS_INF = INFINITY;

// transform.h:9-23 — mul(m, float4(v,0|1)) written as a left-to-right dot per row
inline v3 transform_vector(const sthip_TransformData& t, v3 v) {
  return V3(t.m[0][0] * v.x + t.m[0][1] * v.y + t.m[0][2] * v.z, t.m[1][0] * v.x + t.m[1][1] * v.y + t.m[1][2] * v.z,
            t.m[2][0] * v.x + t.m[2][1] * v.y + t.m[2][2] * v.z);
}
inline v3 transform_point(const sthip_TransformData& t, v3 v) {
  return V3(t.m[0][0] * v.x + t.m[0][1] * v.y + t.m[0][2] * v.z + t.m[0][3],
            t.m[1][0] * v.x + t.m[1][1] * v.y + t.m[1][2] * v.z + t.m[1][3],
            t.m[2][0] * v.x + t.m[2][1] * v.y + t.m[2][2] * v.z + t.m[2][3]);
}
// transform.h:88-104 tmul: lhs * [rhs; 0 0 0 1]
inline sthip_TransformData tmul(const sthip_TransformData& a, const sthip_TransformData& b) {
  sthip_TransformData r;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 4; j++) {
      float s = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
      if (j == 3) s = s + a.m[i][3];
      r.m[i][j] = s;
    }
  }
  return r;
}

// ---------------------------------------------------------------------------------------------
// R1/R2 — rng.hlsli:22-47
// ---------------------------------------------------------------------------------------------
inline void pcg4d(uint32_t v[4]) {
  for (int i = 0; i < 4; i++) v[i] = v[i] * 1664525u + 1013904223u;
  v[0] += v[1] * v[3];
  v[1] += v[2] * v[0];
  v[2] += v[0] * v[1];
  v[3] += v[1] * v[2];
  for (int i = 0; i < 4; i++) v[i] ^= v[i] >> 16;
  v[0] += v[1] * v[3];
  v[1] += v[2] * v[0];
  v[2] += v[0] * v[1];
  v[3] += v[1] * v[2];
}
inline uint32_t pcg(uint32_t v) {  // rng.hlsli:17-21
  const uint32_t state = v * 747796405u + 2891336453u;
  const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}
inline uint32_t xxhash32(uint32_t p) {  // rng.hlsli:6-15
  const uint32_t PRIME32_2 = 2246822519U, PRIME32_3 = 3266489917U;
  const uint32_t PRIME32_4 = 668265263U, PRIME32_5 = 374761393U;
  uint32_t h32 = p + PRIME32_5;
  h32 = PRIME32_4 * ((h32 << 17) | (h32 >> (32 - 17)));
  h32 = PRIME32_2 * (h32 ^ (h32 >> 15));
  h32 = PRIME32_3 * (h32 ^ (h32 >> 13));
  return h32 ^ (h32 >> 16);
}
struct Rng {
  uint32_t v[4];  // (pixel.x, pixel.y, gRandomSeed, counter)
  uint32_t next_uint() {
    v[3]++;
    uint32_t t[4] = {v[0], v[1], v[2], v[3]};
    pcg4d(t);
    return t[0];
  }
  float next_float() { return det_u2f(0x3f800000u | (next_uint() >> 9)) - 1.0f; }
};

// ---------------------------------------------------------------------------------------------
// R3 — bitfield.h:56-93 octahedral fp16x2 normals
// ---------------------------------------------------------------------------------------------
inline uint32_t pack_normal_octahedron(v3 v) {
  const float s = 1.0f / (fabsf(v.x) + fabsf(v.y) + fabsf(v.z));
  float px = v.x * s, py = v.y * s;
  if (v.z <= 0) {
    const float qx = (1.0f - fabsf(py)) * (px >= 0 ? 1.0f : -1.0f);
    const float qy = (1.0f - fabsf(px)) * (py >= 0 ? 1.0f : -1.0f);
    px = qx;
    py = qy;
  }
  return det_f32tof16(px) | (det_f32tof16(py) << 16);
}
inline v3 unpack_normal_octahedron(uint32_t packed) {
  const float px = det_f16tof32(packed & 0xFFFFu), py = det_f16tof32(packed >> 16);
  v3 v = V3(px, py, 1.0f - (fabsf(px) + fabsf(py)));
  if (v.z < 0) {
    const float qx = (1.0f - fabsf(v.y)) * (v.x >= 0 ? 1.0f : -1.0f);
    const float qy = (1.0f - fabsf(v.x)) * (v.y >= 0 ? 1.0f : -1.0f);
    v.x = qx;
    v.y = qy;
  }
  return normalize(v);
}

// ---------------------------------------------------------------------------------------------
// T3 — intersection.hlsli:44-62 (Waechter-Binder)
// ---------------------------------------------------------------------------------------------
inline v3 ray_offset(v3 pos, v3 n) {
  const float int_scale = 256.0f;
  const float origin = 1 / 32.0f;
  const float float_scale = 1 / 65536.0f;
  int32_t ox = (int32_t)(int_scale * n.x), oy = (int32_t)(int_scale * n.y), oz = (int32_t)(int_scale * n.z);
  if (pos.x < 0) ox = -ox;
  if (pos.y < 0) oy = -oy;
  if (pos.z < 0) oz = -oz;
  const float pix = det_u2f((uint32_t)((int32_t)det_f2u(pos.x) + ox));
  const float piy = det_u2f((uint32_t)((int32_t)det_f2u(pos.y) + oy));
  const float piz = det_u2f((uint32_t)((int32_t)det_f2u(pos.z) + oz));
  return V3(fabsf(pos.x) < origin ? pos.x + n.x * float_scale : pix, fabsf(pos.y) < origin ? pos.y + n.y * float_scale : piy,
            fabsf(pos.z) < origin ? pos.z + n.z * float_scale : piz);
}

// common.h:125-132
inline void make_orthonormal(v3 N, v3& T, v3& B) {
  if (N.x != N.y || N.x != N.z)
    T = V3(N.z - N.y, N.x - N.z, N.y - N.x);
  else
    T = V3(N.z - N.y, N.x + N.z, -N.y - N.x);
  T = normalize(T);
  B = cross(N, T);
}
// common.h:154-161
inline v3 sample_cos_hemisphere(float u1, float u2) {
  const float phi = DET_2PI * u2;
  float s, c;
  det_sincosf(phi, &s, &c);
  const float r = sqrtf(u1);
  const float x = r * c, y = r * s;
  return V3(x, y, sqrtf(fmaxf(0.f, 1.0f - (x * x + y * y))));
}
inline float cosine_hemisphere_pdfW(float cos_theta) { return fmaxf(cos_theta, 0.f) / DET_PI; }
// common.h:184-190
inline float ray_plane(v3 origin, v3 dir, v3 normal) {
  const float denom = dot(normal, dir);
  if (fabsf(denom) > 0)
    return -dot(origin, normal) / denom;
  else
    return POS_INF;
}

// ---------------------------------------------------------------------------------------------
// scene
// ---------------------------------------------------------------------------------------------
inline uint32_t bf_get(uint32_t y, uint32_t start, uint32_t len) { return (y >> start) & ((1u << len) - 1u); }
struct Inst {
  sthip_InstanceData d;
  uint32_t type() const { return bf_get(d.packed[0], 0, 4); }
  uint32_t material_address() const { return bf_get(d.packed[0], 4, 28); }
  uint32_t light_index() const { return bf_get(d.packed[1], 0, 12); }
  uint32_t prim_count() const { return bf_get(d.packed[1], 12, 16); }
  uint32_t index_stride() const { return bf_get(d.packed[1], 28, 4); }
  uint32_t first_vertex() const { return d.packed[2]; }
  uint32_t indices_byte_offset() const { return d.packed[3]; }
  float radius() const { return det_u2f(d.packed[2]); }  // sphere instances, scene.h:43
  uint32_t volume_index() const { return d.packed[2]; }  // volume instances, scene.h:46
};

struct Aabb {
  float lo[3], hi[3];
  void reset() {
    for (int i = 0; i < 3; i++) {
      lo[i] = INFINITY;
      hi[i] = -INFINITY;
    }
  }
  void grow(const float* p) {
    for (int i = 0; i < 3; i++) {
      lo[i] = std::min(lo[i], p[i]);
      hi[i] = std::max(hi[i], p[i]);
    }
  }
  void grow(const Aabb& b) {
    for (int i = 0; i < 3; i++) {
      lo[i] = std::min(lo[i], b.lo[i]);
      hi[i] = std::max(hi[i], b.hi[i]);
    }
  }
  float half_area() const {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
  }
  void pad() {  // conservative traversal: see trace contract below
    for (int i = 0; i < 3; i++) {
      const float m = std::max(std::max(fabsf(lo[i]), fabsf(hi[i])), hi[i] - lo[i]);
      const float p = 4e-5f * m + 1e-30f;
      lo[i] -= p;
      hi[i] += p;
    }
  }
};

// A plain BVH2 over boxes; leaves hold [first, first+count) into `order`.
struct Bvh {
  struct Node {
    Aabb box;
    int32_t left, right;    // children (inner) or -1
    uint32_t first, count;  // leaf range
  };
  std::vector<Node> nodes;
  std::vector<uint32_t> order;
  // 4e-6 x (largest coordinate magnitude + diagonal) of the root box: the absolute padding of box_test in this space
  float abs_pad() const {
    if (nodes.empty()) return 0;
    const Aabb& b = nodes[0].box;
    float m = 0, d = 0;
    for (int a = 0; a < 3; a++) {
      if (!(b.hi[a] >= b.lo[a])) continue;
      m = std::max(m, std::max(fabsf(b.lo[a]), fabsf(b.hi[a])));
      d += b.hi[a] - b.lo[a];
    }
    return 4e-6f * (m + d) + 1e-30f;
  }

  void build(const std::vector<Aabb>& boxes, uint32_t leaf_size) {
    const uint32_t n = (uint32_t)boxes.size();
    order.resize(n);
    for (uint32_t i = 0; i < n; i++) order[i] = i;
    nodes.clear();
    nodes.reserve(2 * n / std::max(1u, leaf_size) + 8);
    std::vector<float> cen(3 * (size_t)n);
    for (uint32_t i = 0; i < n; i++)
      for (int a = 0; a < 3; a++) cen[3 * (size_t)i + a] = 0.5f * (boxes[i].lo[a] + boxes[i].hi[a]);
    if (n) build_range(boxes, cen, 0, n, leaf_size);
    for (auto& nd : nodes) nd.box.pad();
  }

 private:
  int32_t build_range(const std::vector<Aabb>& boxes, const std::vector<float>& cen, uint32_t lo, uint32_t hi, uint32_t leaf_size) {
    const int32_t idx = (int32_t)nodes.size();
    nodes.push_back(Node());
    Aabb box, cbox;
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
    if (hi - lo <= leaf_size) return idx;
    // binned SAH over the widest centroid axis, 16 bins
    int axis = 0;
    float ext = cbox.hi[0] - cbox.lo[0];
    for (int a = 1; a < 3; a++)
      if (cbox.hi[a] - cbox.lo[a] > ext) {
        ext = cbox.hi[a] - cbox.lo[a];
        axis = a;
      }
    uint32_t mid = (lo + hi) / 2;
    if (ext > 0) {
      const int NB = 16;
      Aabb bb[NB];
      uint32_t bc[NB];
      for (int b = 0; b < NB; b++) {
        bb[b].reset();
        bc[b] = 0;
      }
      const float k = NB / ext;
      auto bin_of = [&](uint32_t p) {
        int b = (int)((cen[3 * (size_t)p + axis] - cbox.lo[axis]) * k);
        return std::min(std::max(b, 0), NB - 1);
      };
      for (uint32_t i = lo; i < hi; i++) {
        const int b = bin_of(order[i]);
        bb[b].grow(boxes[order[i]]);
        bc[b]++;
      }
      float best = INFINITY;
      int best_split = -1;
      Aabb r;
      float ra[NB];
      uint32_t rc[NB];
      r.reset();
      uint32_t c = 0;
      for (int b = NB - 1; b > 0; b--) {
        r.grow(bb[b]);
        c += bc[b];
        ra[b] = r.half_area();
        rc[b] = c;
      }
      Aabb l;
      l.reset();
      c = 0;
      for (int b = 0; b < NB - 1; b++) {
        l.grow(bb[b]);
        c += bc[b];
        if (c == 0 || rc[b + 1] == 0) continue;
        const float cost = l.half_area() * c + ra[b + 1] * rc[b + 1];
        if (cost < best) {
          best = cost;
          best_split = b;
        }
      }
      if (best_split >= 0) {
        uint32_t* first = order.data() + lo;
        uint32_t* last = order.data() + hi;
        uint32_t* m = std::partition(first, last, [&](uint32_t p) { return bin_of(p) <= best_split; });
        mid = (uint32_t)(m - order.data());
      }
      if (mid == lo || mid == hi) {
        mid = (lo + hi) / 2;
        std::nth_element(order.begin() + lo, order.begin() + mid, order.begin() + hi,
                         [&](uint32_t a, uint32_t b) { return cen[3 * (size_t)a + axis] < cen[3 * (size_t)b + axis]; });
      }
    }
    const int32_t l = build_range(boxes, cen, lo, mid, leaf_size);
    const int32_t r = build_range(boxes, cen, mid, hi, leaf_size);
    nodes[idx].left = l;
    nodes[idx].right = r;
    nodes[idx].count = 0;
    return idx;
  }
};

struct Mesh {  // one BLAS: unique (first_vertex, indices_byte_offset, prim_count, stride)
  uint32_t first_vertex, indices_byte_offset, prim_count, stride;
  Bvh bvh;  // over triangles, object space
};

struct Ray {
  v3 o, d;
  float tmin, tmax;
  bool alpha_test = false;  // gAlphaTest: triangles of masked materials are hit only where the mask is >= 0.75
  bool flip_uvs = false;    // gFlipTriangleUVs for the mask lookup
};
struct Hit {
  float t, b1, b2;
  uint32_t ip;  // instance | primitive << 16 ; 0xFFFFFFFF = miss
};

// Per-ray constants of the watertight test (object space)
struct RayShear {
  int kx, ky, kz;
  float Sx, Sy, Sz;
};

}  // namespace

// Texture2D<float4> with its mip chain (2x2 box filter, level k+1 = max(1, floor(dim / 2)))
struct OrcImage {
  std::vector<uint32_t> w, h;
  std::vector<std::vector<float>> mip;  // RGBA32F per level
};

// A NanoVDB float grid as bound to gVolumes[] (ByteAddressBuffer, bdpt.hlsl:35): the subset of PNanoVDB.h (NanoVDB
// 32.3, the version the reference vendors under src/extern/nanovdb) that medium.hlsli:58-71,85-88 and
// intersection.hlsli:93-113 call — root bounding box and maximum, the value at an index coordinate through
// root tile -> upper (32^3) -> lower (16^3) -> leaf (8^3), and the grid's affine map. Layout constants from the
// published format (PNanoVDB.h:702-711,761-777,904-916,964-1111 and the FLOAT row of pnanovdb_grid_type_constants).
// Reads outside the buffer return zero. Pinned by tests/golden/fog_sphere.npz (answers of the reference's own reader).
struct NvdbGrid {
  std::vector<uint8_t> bytes;
  size_t root = 0;
  int32_t bbox_min[3] = {0, 0, 0}, bbox_max[3] = {0, 0, 0};
  float matf[9], invmatf[9], vecf[3];
  template <class T>
  T rd(size_t off) const {
    T v{};
    if (off + sizeof(T) <= bytes.size()) memcpy(&v, bytes.data() + off, sizeof(T));
    return v;
  }
  bool init() {
    if (bytes.size() < 672 + 64 + 64) return false;
    if (rd<uint64_t>(0) != 0x304244566f6e614eull) return false;  // "NanoVDB0"
    if (rd<uint32_t>(636) != 1u) return false;                    // GRID_TYPE_FLOAT
    const size_t tree = 672;                                      // PNANOVDB_GRID_SIZE
    root = tree + (size_t)rd<uint64_t>(tree + 24);                // TREE_OFF_NODE_OFFSET_ROOT
    for (int k = 0; k < 3; k++) {
      bbox_min[k] = rd<int32_t>(root + 4 * k);
      bbox_max[k] = rd<int32_t>(root + 12 + 4 * k);
      vecf[k] = rd<float>(296 + 72 + 4 * k);
    }
    for (int k = 0; k < 9; k++) {
      matf[k] = rd<float>(296 + 4 * k);
      invmatf[k] = rd<float>(296 + 36 + 4 * k);
    }
    return true;
  }
  float root_max() const { return rd<float>(root + 36); }
  bool bit(size_t mask, uint32_t n) const { return (rd<uint32_t>(mask + 4 * (n >> 5)) >> (n & 31u)) & 1u; }
  float value(int32_t x, int32_t y, int32_t z) const {
    const uint64_t key = (uint64_t)((uint32_t)z >> 12) | ((uint64_t)((uint32_t)y >> 12) << 21) | ((uint64_t)((uint32_t)x >> 12) << 42);
    const uint32_t tiles = rd<uint32_t>(root + 24);
    for (uint32_t i = 0; i < tiles; i++) {
      const size_t tile = root + 64 + 32 * (size_t)i;
      if (rd<uint64_t>(tile) != key) continue;
      const int64_t child = rd<int64_t>(tile + 8);
      if (child == 0) return rd<float>(tile + 20);
      const size_t upper = root + (size_t)child;
      const uint32_t n = ((((uint32_t)x & 4095u) >> 7) << 10) + ((((uint32_t)y & 4095u) >> 7) << 5) + (((uint32_t)z & 4095u) >> 7);
      if (!bit(upper + 4128, n)) return rd<float>(upper + 8256 + 8 * (size_t)n);
      const size_t lower = upper + (size_t)rd<int64_t>(upper + 8256 + 8 * (size_t)n);
      const uint32_t n2 = ((((uint32_t)x & 127u) >> 3) << 8) + ((((uint32_t)y & 127u) >> 3) << 4) + (((uint32_t)z & 127u) >> 3);
      if (!bit(lower + 544, n2)) return rd<float>(lower + 1088 + 8 * (size_t)n2);
      const size_t leaf = lower + (size_t)rd<int64_t>(lower + 1088 + 8 * (size_t)n2);
      const uint32_t n3 = (((uint32_t)x & 7u) << 6) + (((uint32_t)y & 7u) << 3) + ((uint32_t)z & 7u);
      return rd<float>(leaf + 96 + 4 * (size_t)n3);
    }
    return rd<float>(root + 28);  // background
  }
  // pnanovdb_map_apply / _inverse / _jacobi / _inverse_jacobi, PNanoVDB.h:1988-2034 (left-to-right sums)
  v3 index_to_world(v3 s) const {
    return V3(s.x * matf[0] + s.y * matf[1] + s.z * matf[2] + vecf[0], s.x * matf[3] + s.y * matf[4] + s.z * matf[5] + vecf[1], s.x * matf[6] + s.y * matf[7] + s.z * matf[8] + vecf[2]);
  }
  v3 index_to_world_dir(v3 s) const {
    return V3(s.x * matf[0] + s.y * matf[1] + s.z * matf[2], s.x * matf[3] + s.y * matf[4] + s.z * matf[5], s.x * matf[6] + s.y * matf[7] + s.z * matf[8]);
  }
  v3 world_to_index_dir(v3 s) const {
    return V3(s.x * invmatf[0] + s.y * invmatf[1] + s.z * invmatf[2], s.x * invmatf[3] + s.y * invmatf[4] + s.z * invmatf[5], s.x * invmatf[6] + s.y * invmatf[7] + s.z * invmatf[8]);
  }
  v3 world_to_index(v3 p) const { return world_to_index_dir(V3(p.x - vecf[0], p.y - vecf[1], p.z - vecf[2])); }
};

struct OrcImage1 {  // Texture2D<float>: one coverage value per texel (alpha masks)
  uint32_t w = 0, h = 0;
  std::vector<float> px;
};

struct orc_scene {
  std::vector<OrcImage> images;
  std::vector<OrcImage1> images1;
  std::vector<sthip_PackedVertexData> vertices;
  std::vector<uint8_t> indices;
  std::vector<Inst> instances;
  std::vector<sthip_TransformData> xf, inv_xf, motion_xf;
  std::vector<uint8_t> materials;
  std::vector<uint32_t> lights;
  std::vector<float> distributions;  // gDistributions (dist2.h tables of the environment map)
  std::vector<NvdbGrid> volumes;     // gVolumes
  std::vector<Mesh> meshes;
  std::vector<uint32_t> inst_mesh;
  std::vector<uint8_t> inst_identity;
  Bvh tlas;
  // traversal statistics of orc_trace_rays / orc_render (relaxed atomics)
  std::atomic<uint64_t> stat_nodes{0}, stat_tris{0};

  // scene.h:139-161 load_tri
  void load_tri(const Inst& in, uint32_t prim, uint32_t tri[3]) const {
    const uint32_t stride = in.index_stride();
    const uint32_t off = in.indices_byte_offset() + prim * 3 * stride;
    if (stride == 2) {
      const uint32_t aligned = off & ~3u;
      uint32_t w[2];
      memcpy(w, &indices[aligned], 8);
      if (aligned == off) {
        tri[0] = w[0] & 0xffff;
        tri[1] = (w[0] >> 16) & 0xffff;
        tri[2] = w[1] & 0xffff;
      } else {
        tri[0] = (w[0] >> 16) & 0xffff;
        tri[1] = w[1] & 0xffff;
        tri[2] = (w[1] >> 16) & 0xffff;
      }
    } else {
      memcpy(tri, &indices[off], 12);
    }
    for (int i = 0; i < 3; i++) tri[i] += in.first_vertex();
  }
};

namespace {

// ---------------------------------------------------------------------------------------------
// T1/T2 — the traversal contract (the reference delegates this to the Vulkan driver,
// intersection.hlsli:68-75). Definition used on both sides of the ABI:
//
//   * A ray is tested against a triangle in the OBJECT space of its instance: o' = Minv*(o,1),
//     d' = Minv*(d,0) with Minv = gInstanceInverseTransforms[i] (fmaf chains below); if Minv is
//     bit-for-bit the identity, o' = o and d' = d. d' is not renormalised, so t is the world t.
//   * Ray/triangle: Woop-Benthin-Wald watertight test (JCGT 2013) in single precision without
//     the double fallback: edge values that are exactly zero count as inside. Arithmetic exactly
//     as written in tri_test(). A hit needs tmin < t < tmax. Both faces hit.
//   * Closest hit = the accepted triangle with the smallest t over ALL triangles of the scene;
//     equal t: the smaller instance index, then the smaller primitive index wins.
//     Any-hit (occlusion) = whether any triangle is accepted.
//   * The acceleration structure is an implementation detail that must never change that answer:
//     box tests are conservative (padded boxes, inclusive comparisons).
// ---------------------------------------------------------------------------------------------
inline bool is_identity(const sthip_TransformData& t) {
  static const float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  return memcmp(&t, I, sizeof(I)) == 0;
}
inline v3 obj_point(const sthip_TransformData& m, v3 p) {
  return V3(fmaf(m.m[0][2], p.z, fmaf(m.m[0][1], p.y, m.m[0][0] * p.x)) + m.m[0][3],
            fmaf(m.m[1][2], p.z, fmaf(m.m[1][1], p.y, m.m[1][0] * p.x)) + m.m[1][3],
            fmaf(m.m[2][2], p.z, fmaf(m.m[2][1], p.y, m.m[2][0] * p.x)) + m.m[2][3]);
}
inline v3 obj_vector(const sthip_TransformData& m, v3 p) {
  return V3(fmaf(m.m[0][2], p.z, fmaf(m.m[0][1], p.y, m.m[0][0] * p.x)), fmaf(m.m[1][2], p.z, fmaf(m.m[1][1], p.y, m.m[1][0] * p.x)),
            fmaf(m.m[2][2], p.z, fmaf(m.m[2][1], p.y, m.m[2][0] * p.x)));
}
inline float comp(v3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

inline RayShear make_shear(v3 d) {
  RayShear s;
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
  if (comp(d, kz) < 0.0f) std::swap(kx, ky);
  s.kx = kx;
  s.ky = ky;
  s.kz = kz;
  const float dz = comp(d, kz);
  s.Sx = comp(d, kx) / dz;
  s.Sy = comp(d, ky) / dz;
  s.Sz = 1.0f / dz;
  return s;
}

// returns true and fills t,b1,b2 when the triangle is accepted for (tmin, tmax)
inline bool tri_test(v3 o, const RayShear& s, v3 p0, v3 p1, v3 p2, float tmin, float tmax, float& t, float& b1, float& b2) {
  const v3 A = p0 - o, B = p1 - o, C = p2 - o;
  const float Akz = comp(A, s.kz), Bkz = comp(B, s.kz), Ckz = comp(C, s.kz);
  const float Ax = fmaf(-s.Sx, Akz, comp(A, s.kx)), Ay = fmaf(-s.Sy, Akz, comp(A, s.ky));
  const float Bx = fmaf(-s.Sx, Bkz, comp(B, s.kx)), By = fmaf(-s.Sy, Bkz, comp(B, s.ky));
  const float Cx = fmaf(-s.Sx, Ckz, comp(C, s.kx)), Cy = fmaf(-s.Sy, Ckz, comp(C, s.ky));
  // Edge functions: both products rounded separately (NOT an fma), so that the value for a shared
  // edge is exactly antisymmetric between the two triangles that share it — that is what makes the
  // test watertight.
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

// conservative slab test against a padded box; NaNs (0*inf) fall out of fminf/fmaxf
// The box is padded per axis by 1e-5 of the coordinates involved: the triangle test accepts hits whose rounded t lies a
// few ulp off the plane (e.g. a ray that starts exactly on a flat, axis-aligned mesh with a tiny direction), and the
// acceleration structure must never lose a hit the contract accepts (tools/fuzz_parity.py rays found that case).
// `abs_pad` = 4e-6 x the size of the space being traversed (Bvh::abs_pad): the position error of an accepted hit does not
// shrink with the coordinates (a ray lying in the plane y = 0 of a mesh around the origin).
inline bool box_test(const Aabb& b, v3 o, v3 inv_d, float tmin, float tmax, float& tn, float abs_pad) {
  const float px = 1e-5f * (fabsf(o.x) + fmaxf(fabsf(b.lo[0]), fabsf(b.hi[0]))) + abs_pad;
  const float py = 1e-5f * (fabsf(o.y) + fmaxf(fabsf(b.lo[1]), fabsf(b.hi[1]))) + abs_pad;
  const float pz = 1e-5f * (fabsf(o.z) + fmaxf(fabsf(b.lo[2]), fabsf(b.hi[2]))) + abs_pad;
  const float tx0 = (b.lo[0] - px - o.x) * inv_d.x, tx1 = (b.hi[0] + px - o.x) * inv_d.x;
  const float ty0 = (b.lo[1] - py - o.y) * inv_d.y, ty1 = (b.hi[1] + py - o.y) * inv_d.y;
  const float tz0 = (b.lo[2] - pz - o.z) * inv_d.z, tz1 = (b.hi[2] + pz - o.z) * inv_d.z;
  float t0 = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), tmin));
  float t1 = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), tmax));
  tn = t0;
  return t0 <= t1 * 1.00001f;
}
inline v3 safe_inv(v3 d) {
  const float eps = 1e-30f;
  const float x = fabsf(d.x) < eps ? copysignf(eps, d.x) : d.x;
  const float y = fabsf(d.y) < eps ? copysignf(eps, d.y) : d.y;
  const float z = fabsf(d.z) < eps ? copysignf(eps, d.z) : d.z;
  return V3(1.0f / x, 1.0f / y, 1.0f / z);
}

inline v3 vpos(const orc_scene& sc, uint32_t vi) {
  const float* p = sc.vertices[vi].position;
  return V3(p[0], p[1], p[2]);
}

// closest-hit ordering: smaller t, then smaller instance index, then smaller primitive index
inline uint32_t hit_key(uint32_t ip) { return ((ip & 0xFFFFu) << 16) | (ip >> 16); }
inline void accept(Hit& h, float t, float b1, float b2, uint32_t ip) {
  if (t < h.t || (t == h.t && h.ip != 0xFFFFFFFFu && hit_key(ip) < hit_key(h.ip))) {
    h.t = t;
    h.b1 = b1;
    h.b2 = b2;
    h.ip = ip;
  }
}

// Sphere instances (intersection.hlsli:79-89 with ray_sphere, common.h:163-173), part of the traversal contract:
// the ray goes to object space through Minv like a triangle ray, ray_sphere's arithmetic is taken as written
// (products rounded separately), t is the near root if it lies beyond tmin, else the far root, and a hit needs
// tmin < t < tmax. Its id is instance | INVALID_PRIMITIVE << 16; b1, b2 are 0.
inline bool sphere_test(v3 o, v3 d, float r, float tmin, float tmax, float& t) {
  const float a = dot(d, d);
  const float b = dot(o, d);
  const v3 l = V3(a * o.x - d.x * b, a * o.y - d.y * b, a * o.z - d.z * b);
  float det = (a * r) * (a * r) - dot(l, l);
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

// Volume instances (intersection.hlsli:93-113), part of the traversal contract: the ray goes to the object space of the
// instance, then to the index space of its NanoVDB grid (world_to_indexf / world_to_index_dirf); slabs of the root
// bounding box [bbox_min, bbox_max + 1]; t is the entry if it lies beyond tmin, else the exit, and a hit needs
// tmin < t < tmax. The reference runs this for every candidate the driver reports and never checks that the slabs
// overlap (a ray whose bounding-box test was a conservative false positive would get a hit at the entry slab); the
// contract requires entry <= exit. Its id is instance | INVALID_PRIMITIVE << 16. `axis_sign` returns the hit face:
// (t == t1) - (t == t0) per axis, the index-space normal the reference packs into the shading data.
inline bool volume_test(const NvdbGrid& g, v3 o, v3 d, float tmin, float tmax, float& t, v3* face = nullptr) {
  const v3 io = g.world_to_index(o), id = g.world_to_index_dir(d);
  const v3 lo = V3((float)g.bbox_min[0], (float)g.bbox_min[1], (float)g.bbox_min[2]);
  const v3 hi = V3((float)(g.bbox_max[0] + 1), (float)(g.bbox_max[1] + 1), (float)(g.bbox_max[2] + 1));
  const v3 t0 = V3((lo.x - io.x) / id.x, (lo.y - io.y) / id.y, (lo.z - io.z) / id.z);
  const v3 t1 = V3((hi.x - io.x) / id.x, (hi.y - io.y) / id.y, (hi.z - io.z) / id.z);
  const float near = fmaxf(fmaxf(fminf(t0.x, t1.x), fminf(t0.y, t1.y)), fminf(t0.z, t1.z));
  const float far = fminf(fminf(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y)), fmaxf(t0.z, t1.z));
  if (!(near <= far)) return false;
  const float tt = near > tmin ? near : far;
  if (!(tt > tmin && tt < tmax)) return false;
  t = tt;
  if (face) *face = V3((tt == t1.x ? 1.0f : 0.0f) - (tt == t0.x ? 1.0f : 0.0f), (tt == t1.y ? 1.0f : 0.0f) - (tt == t0.y ? 1.0f : 0.0f), (tt == t1.z ? 1.0f : 0.0f) - (tt == t0.z ? 1.0f : 0.0f));
  return true;
}

// The alpha test of a candidate triangle hit (intersection.hlsli:117-131): the mask of the instance's material
// (MaterialRecord.alpha_mask_index, read at material_address + 60) sampled at the hit's uv
// (make_triangle_shading_data's interpolation, shading_data.hlsli:2-6) at level 0 — bilinear, repeat addressing — must
// be >= 0.75. Materials without a mask are opaque.
inline float sample_image1(const OrcImage1& im, float u, float v) {
  const float x = u * (float)im.w - 0.5f, y = v * (float)im.h - 0.5f;
  const float x0 = floorf(x), y0 = floorf(y);
  const float fx = x - x0, fy = y - y0;
  const int w = (int)im.w, h = (int)im.h;
  auto at = [&](int xi, int yi) {
    xi = ((xi % w) + w) % w;
    yi = ((yi % h) + h) % h;
    return im.px[(size_t)yi * w + xi];
  };
  const int ix = (int)x0, iy = (int)y0;
  const float a = lerpf(at(ix, iy), at(ix + 1, iy), fx), b = lerpf(at(ix, iy + 1), at(ix + 1, iy + 1), fx);
  return lerpf(a, b, fy);
}
inline bool alpha_passes(const orc_scene& sc, const Inst& in, const uint32_t tri[3], float b1, float b2, bool flip_uvs) {
  uint32_t mask_index;
  memcpy(&mask_index, &sc.materials[in.material_address() + 60], 4);
  if (mask_index >= sc.images1.size()) return true;
  const sthip_PackedVertexData &q0 = sc.vertices[tri[0]], &q1 = sc.vertices[tri[1]], &q2 = sc.vertices[tri[2]];
  const float u = q0.u + (q1.u - q0.u) * b1 + (q2.u - q0.u) * b2;
  float v = q0.v + (q1.v - q0.v) * b1 + (q2.v - q0.v) * b2;
  if (flip_uvs) v = 1 - v;
  return sample_image1(sc.images1[mask_index], u, v) >= 0.75f;
}

// one instance, object-space ray; returns true as soon as something is hit when any_hit
bool trace_instance(const orc_scene& sc, uint32_t inst_index, const Ray& wr, Hit& h, bool any_hit, bool brute, uint64_t& n_nodes, uint64_t& n_tris) {
  const Inst& in = sc.instances[inst_index];
  if (in.type() == STHIP_INSTANCE_TYPE_SPHERE) {
    float t;
    n_tris++;
    if (!sphere_test(obj_point(sc.inv_xf[inst_index], wr.o), obj_vector(sc.inv_xf[inst_index], wr.d), in.radius(), wr.tmin, wr.tmax, t)) return false;
    if (any_hit) {
      h.ip = 0;
      return true;
    }
    accept(h, t, 0.0f, 0.0f, inst_index | (STHIP_INVALID_PRIMITIVE << 16));
    return false;
  }
  if (in.type() == STHIP_INSTANCE_TYPE_VOLUME) {
    float t;
    n_tris++;
    if (in.volume_index() >= sc.volumes.size()) return false;
    if (!volume_test(sc.volumes[in.volume_index()], obj_point(sc.inv_xf[inst_index], wr.o), obj_vector(sc.inv_xf[inst_index], wr.d), wr.tmin, wr.tmax, t)) return false;
    if (any_hit) {
      h.ip = 0;
      return true;
    }
    accept(h, t, 0.0f, 0.0f, inst_index | (STHIP_INVALID_PRIMITIVE << 16));
    return false;
  }
  if (in.type() != STHIP_INSTANCE_TYPE_TRIANGLES) return false;
  v3 o = wr.o, d = wr.d;
  if (!sc.inst_identity[inst_index]) {
    o = obj_point(sc.inv_xf[inst_index], wr.o);
    d = obj_vector(sc.inv_xf[inst_index], wr.d);
  }
  const RayShear sh = make_shear(d);
  const Mesh& mesh = sc.meshes[sc.inst_mesh[inst_index]];
  auto test_prim = [&](uint32_t prim) -> bool {
    uint32_t tri[3];
    sc.load_tri(in, prim, tri);
    float t, b1, b2;
    n_tris++;
    if (tri_test(o, sh, vpos(sc, tri[0]), vpos(sc, tri[1]), vpos(sc, tri[2]), wr.tmin, wr.tmax, t, b1, b2)) {
      if (wr.alpha_test && !alpha_passes(sc, in, tri, b1, b2, wr.flip_uvs)) return false;  // intersection.hlsli:117-131
      if (any_hit) {
        h.ip = 0;
        return true;
      }
      accept(h, t, b1, b2, inst_index | (prim << 16));
    }
    return false;
  };
  if (brute) {
    for (uint32_t p = 0; p < in.prim_count(); p++)
      if (test_prim(p)) return true;
    return false;
  }
  const v3 inv_d = safe_inv(d);
  int32_t stack[128];
  int sp = 0;
  if (mesh.bvh.nodes.empty()) return false;
  stack[sp++] = 0;
  while (sp) {
    const Bvh::Node& nd = mesh.bvh.nodes[stack[--sp]];
    float tn;
    n_nodes++;
    if (!box_test(nd.box, o, inv_d, wr.tmin, std::min(wr.tmax, h.t), tn, mesh.bvh.abs_pad())) continue;
    if (nd.left < 0) {
      for (uint32_t i = 0; i < nd.count; i++)
        if (test_prim(mesh.bvh.order[nd.first + i])) return true;
    } else {
      // near child last on the stack (popped first): order by the box centres along the ray
      const Bvh::Node& L = mesh.bvh.nodes[nd.left];
      const Bvh::Node& R = mesh.bvh.nodes[nd.right];
      const float cl = (L.box.lo[0] + L.box.hi[0]) * d.x + (L.box.lo[1] + L.box.hi[1]) * d.y + (L.box.lo[2] + L.box.hi[2]) * d.z;
      const float cr = (R.box.lo[0] + R.box.hi[0]) * d.x + (R.box.lo[1] + R.box.hi[1]) * d.y + (R.box.lo[2] + R.box.hi[2]) * d.z;
      if (sp + 2 > 128) return false;  // cannot happen for the depths built here
      if (cl < cr) {
        stack[sp++] = nd.right;
        stack[sp++] = nd.left;
      } else {
        stack[sp++] = nd.left;
        stack[sp++] = nd.right;
      }
    }
  }
  return false;
}

Hit trace(const orc_scene& sc, const Ray& r, bool any_hit, bool brute, uint64_t* counters) {
  Hit h;
  h.t = r.tmax;
  h.b1 = h.b2 = 0;
  h.ip = 0xFFFFFFFFu;
  uint64_t nn = 0, nt = 0;
  if (brute || sc.tlas.nodes.empty()) {
    for (uint32_t i = 0; i < sc.instances.size(); i++)
      if (trace_instance(sc, i, r, h, any_hit, brute, nn, nt)) break;
  } else {
    const v3 inv_d = safe_inv(r.d);
    int32_t stack[128];
    int sp = 0;
    stack[sp++] = 0;
    bool done = false;
    while (sp && !done) {
      const Bvh::Node& nd = sc.tlas.nodes[stack[--sp]];
      float tn;
      nn++;
      if (!box_test(nd.box, r.o, inv_d, r.tmin, std::min(r.tmax, h.t), tn, sc.tlas.abs_pad())) continue;
      if (nd.left < 0) {
        for (uint32_t i = 0; i < nd.count && !done; i++) done = trace_instance(sc, sc.tlas.order[nd.first + i], r, h, any_hit, false, nn, nt);
      } else {
        stack[sp++] = nd.right;
        stack[sp++] = nd.left;
      }
    }
  }
  if (counters) {
    counters[0] += nn;
    counters[1] += nt;
  }
  return h;
}

// ---------------------------------------------------------------------------------------------
// W7/S1 — ShadingData and make_triangle_shading_data (shading_data.hlsli:2-73).
// uv_screen_size / mean_curvature feed the texture LOD through ray cones (path.hlsli:224-244).
// ---------------------------------------------------------------------------------------------
struct ShadingData {
  v3 position;
  uint32_t flags;
  uint32_t packed_geometry_normal, packed_shading_normal, packed_tangent;
  float shape_area;
  float u, v;
  float uv_screen_size, mean_curvature;
  v3 geometry_normal() const { return unpack_normal_octahedron(packed_geometry_normal); }
  v3 shading_normal() const { return unpack_normal_octahedron(packed_shading_normal); }
  v3 tangent() const { return unpack_normal_octahedron(packed_tangent); }
  // shading_data.h:29-37; SHADING_FLAG_FLIP_BITANGENT is never set on this path (flags = 0 | FRONT_FACE)
  v3 to_world(v3 w) const {
    const v3 n = shading_normal(), t = tangent();
    return w.x * t + w.y * cross(n, t) + w.z * n;
  }
  v3 to_local(v3 w) const {
    const v3 n = shading_normal(), t = tangent();
    return V3(dot(w, t), dot(w, cross(n, t)), dot(w, n));
  }
};

void make_triangle_shading_data(const orc_scene& sc, ShadingData& r, uint32_t inst_index, uint32_t prim, float b1, float b2, bool flip_uvs = false) {
  const Inst& in = sc.instances[inst_index];
  const sthip_TransformData& xf = sc.xf[inst_index];
  uint32_t tri[3];
  sc.load_tri(in, prim, tri);
  const sthip_PackedVertexData &q0 = sc.vertices[tri[0]], &q1 = sc.vertices[tri[1]], &q2 = sc.vertices[tri[2]];
  const v3 p0 = V3(q0.position[0], q0.position[1], q0.position[2]);
  const v3 p1 = V3(q1.position[0], q1.position[1], q1.position[2]);
  const v3 p2 = V3(q2.position[0], q2.position[1], q2.position[2]);
  const v3 n0 = V3(q0.normal[0], q0.normal[1], q0.normal[2]);
  const v3 n1 = V3(q1.normal[0], q1.normal[1], q1.normal[2]);
  const v3 n2 = V3(q2.normal[0], q2.normal[1], q2.normal[2]);
  // :64-73
  const v3 v1v0 = p1 - p0, v2v0 = p2 - p0;
  const v3 local_position = p0 + v1v0 * b1 + v2v0 * b2;
  r.position = transform_point(xf, local_position);
  // :2-63
  r.u = q0.u + (q1.u - q0.u) * b1 + (q2.u - q0.u) * b2;
  r.v = q0.v + (q1.v - q0.v) * b1 + (q2.v - q0.v) * b2;
  if (flip_uvs) r.v = 1 - r.v;  // gFlipTriangleUVs, shading_data.hlsli:5-6
  const v3 dPds = transform_vector(xf, p0 - p2);
  const v3 dPdt = transform_vector(xf, p1 - p2);
  v3 geometry_normal = cross(dPds, dPdt);
  const float area2 = length(geometry_normal);
  geometry_normal = geometry_normal / area2;
  r.packed_geometry_normal = pack_normal_octahedron(geometry_normal);
  r.shape_area = area2 / 2;

  const float duvds0 = q2.u - q0.u, duvds1 = q2.v - q0.v;
  const float duvdt0 = q2.u - q1.u, duvdt1 = q2.v - q1.v;
  const float det = duvds0 * duvdt1 - duvdt0 * duvds1;
  const float inv_det = 1 / det;
  const float dsdu = duvdt1 * inv_det;
  const float dtdu = -duvds1 * inv_det;
  const float dsdv = duvdt0 * inv_det;
  const float dtdv = -duvds0 * inv_det;
  v3 dPdu, dPdv;
  if (det != 0) {
    dPdu = -(dPds * dsdu + dPdt * dtdu);
    dPdv = -(dPds * dsdv + dPdt * dtdv);
    r.uv_screen_size = 1 / fmaxf(length(dPdu), length(dPdv));
  } else {
    make_orthonormal(geometry_normal, dPdu, dPdv);
    r.uv_screen_size = 1;
  }

  v3 shading_normal = n0 + (n1 - n0) * b1 + (n2 - n0) * b2;
  if ((shading_normal.x == 0 && shading_normal.y == 0 && shading_normal.z == 0) || any_nan(shading_normal)) {
    r.packed_shading_normal = r.packed_geometry_normal;
    r.packed_tangent = pack_normal_octahedron(normalize(dPdu));
    r.mean_curvature = 0;
  } else {
    shading_normal = normalize(transform_vector(xf, shading_normal));
    const v3 tangent = normalize(dPdu - shading_normal * dot(shading_normal, dPdu));
    r.packed_shading_normal = pack_normal_octahedron(shading_normal);
    r.packed_tangent = pack_normal_octahedron(tangent);
    if (dot(shading_normal, geometry_normal) < 0) r.packed_geometry_normal = pack_normal_octahedron(-geometry_normal);
    // :56-61
    const v3 dNds = n2 - n0, dNdt = n2 - n1;
    const v3 dNdu = dNds * dsdu + dNdt * dtdu;
    const v3 dNdv = dNds * dsdv + dNdt * dtdv;
    const v3 bitangent = normalize(cross(shading_normal, tangent));
    r.mean_curvature = (dot(dNdu, tangent) + dot(dNdv, bitangent)) / 2;
  }
  r.flags = 0;
}

// ---------------------------------------------------------------------------------------------
// S3/M1-M4 — DisneyMaterial (materials/disney_material.hlsli, disney_*.hlsli, microfacet.h)
// ---------------------------------------------------------------------------------------------
// common.h:134-147
inline void cartesian_to_spherical_uv(v3 v, float& u, float& vv) {
  const float theta = det_atan2f(v.z, v.x);
  u = theta * DET_INV_PI * .5f + .5f;
  vv = det_acosf(fminf(fmaxf(v.y, -1.f), 1.f)) * DET_INV_PI;
}
inline v3 spherical_uv_to_cartesian(float u, float v) {
  u = u * 2 - 1;
  u *= DET_PI;
  v *= DET_PI;
  float su, cu, sv, cv;
  det_sincosf(u, &su, &cu);
  det_sincosf(v, &sv, &cv);
  return V3(sv * cu, cv, sv * su);
}

// shading_data.hlsli:93-105 (the angles of dpdu / dpdv are the uv themselves, as the reference writes them)
void make_sphere_shading_data(const orc_scene& sc, ShadingData& r, uint32_t inst_index, v3 local_position) {
  const sthip_TransformData& t = sc.xf[inst_index];
  const v3 normal = normalize(transform_vector(t, local_position));
  r.position = transform_point(t, local_position);
  r.packed_geometry_normal = r.packed_shading_normal = pack_normal_octahedron(normal);
  const float radius = sc.instances[inst_index].radius();
  r.shape_area = 4 * DET_PI * radius * radius;
  r.mean_curvature = 1 / radius;
  cartesian_to_spherical_uv(normalize(local_position), r.u, r.v);
  float su, cu, sv, cv;
  det_sincosf(r.u, &su, &cu);
  det_sincosf(r.v, &sv, &cv);
  const v3 dpdu = transform_vector(t, V3(-su * sv, 0, cu * sv));
  const v3 dpdv = transform_vector(t, V3(cu * cv, -sv, su * cv));
  r.packed_tangent = pack_normal_octahedron(normalize(dpdu - normal * dot(normal, dpdu)));
  r.uv_screen_size = 1 / fmaxf(length(dpdu), length(dpdv));
}

struct MaterialEvalRecord {
  v3 f;
  float pdf_fwd, pdf_rev;
};
struct MaterialSampleRecord {
  v3 dir_out;
  float pdf_fwd, pdf_rev, eta, roughness;
};

inline float schlick_fresnel1(float F0, float cos_theta) { return F0 + (1 - F0) * det_pow5f(fmaxf(1.0f - cos_theta, 0.0f)); }
inline v3 schlick_fresnel3(v3 F0, float cos_theta) { return F0 + (V3(1.0f) - F0) * det_pow5f(fmaxf(1.0f - cos_theta, 0.0f)); }
inline float fresnel_dielectric3(float n_dot_i, float n_dot_t, float eta) {  // microfacet.h:33-38
  const float rs = (n_dot_i - eta * n_dot_t) / (n_dot_i + eta * n_dot_t);
  const float rp = (eta * n_dot_i - n_dot_t) / (eta * n_dot_i + n_dot_t);
  return (rs * rs + rp * rp) / 2;
}
inline float fresnel_dielectric(float n_dot_i, float eta) {  // microfacet.h:45-53
  const float n_dot_t_sq = 1 - (1 - n_dot_i * n_dot_i) / (eta * eta);
  if (n_dot_t_sq < 0) return 1;
  const float n_dot_t = sqrtf(n_dot_t_sq);
  return fresnel_dielectric3(fabsf(n_dot_i), n_dot_t, eta);
}
inline float Dm(float ax, float ay, v3 h) {  // disney_material.hlsli:4-10
  const float ax2 = ax * ax, ay2 = ay * ay;
  const v3 h2 = h * h;
  const float hh = h2.x / ax2 + h2.y / ay2 + h2.z;
  return 1 / (DET_PI * ax * ay * hh * hh);
}
inline float G1(float ax, float ay, v3 w) {  // :11-17
  const float ax2 = ax * ax, ay2 = ay * ay;
  const v3 w2 = w * w;
  const float lambda = (sqrtf(1 + (w2.x * ax2 + w2.y * ay2) / w2.z) - 1) / 2;
  return 1 / (1 + lambda);
}
inline float R0(float eta) {
  const float num = eta - 1, denom = eta + 1;
  return (num * num) / (denom * denom);
}
inline float Dc(float alpha_g, float h_lz) {  // :24-27
  const float a2 = alpha_g * alpha_g;
  return (a2 - 1) / (DET_PI * det_logf(a2) * (1 + (a2 - 1) * h_lz * h_lz));
}
inline float Gc(v3 w) {  // :28-33
  const float wx = w.x * 0.25f, wy = w.y * 0.25f;
  const float lambda = (sqrtf(1 + (wx * wx + wy * wy) / (w.z * w.z)) - 1) / 2;
  return 1 / (1 + lambda);
}
inline v3 reflect(v3 i, v3 n) { return i - 2 * dot(n, i) * n; }
inline v3 refract(v3 i, v3 n, float eta) {
  const float ni = dot(n, i);
  const float k = 1 - eta * eta * (1 - ni * ni);
  if (k < 0) return V3(0.0f);
  return eta * i - (eta * ni + sqrtf(k)) * n;
}
// microfacet.h:76-106 (Heitz 2018)
v3 sample_visible_normals(v3 local_dir_in, float ax, float ay, float r0, float r1) {
  const bool inside = local_dir_in.z < 0;
  if (inside) local_dir_in = -local_dir_in;
  const v3 hemi_dir_in = normalize(V3(ax * local_dir_in.x, ay * local_dir_in.y, local_dir_in.z));
  const float r = sqrtf(r0);
  const float phi = DET_2PI * r1;
  float sphi, cphi;
  det_sincosf(phi, &sphi, &cphi);
  const float t1 = r * cphi;
  float t2 = r * sphi;
  const float s = (1 + hemi_dir_in.z) / 2;
  t2 = (1 - s) * sqrtf(1 - t1 * t1) + s * t2;
  const v3 disk_N = V3(t1, t2, sqrtf(fmaxf(0.0f, 1 - t1 * t1 - t2 * t2)));
  v3 T1, T2;
  make_orthonormal(hemi_dir_in, T1, T2);
  const v3 hemi_N = disk_N.x * T1 + disk_N.y * T2 + disk_N.z * hemi_dir_in;
  v3 N = normalize(V3(ax * hemi_N.x, ay * hemi_N.y, fmaxf(0.f, hemi_N.z)));
  if (inside) N = -N;
  return N;
}

struct DisneyMaterial {
  float data[3][4];
  v3 base_color() const { return V3(data[0][0], data[0][1], data[0][2]); }
  float emission() const { return data[0][3]; }
  float metallic() const { return data[1][0]; }
  float roughness() const { return data[1][1]; }
  float anisotropic() const { return data[1][2]; }
  float subsurface() const { return data[1][3]; }
  float clearcoat() const { return data[2][0]; }
  float clearcoat_gloss() const { return data[2][1]; }
  float transmission() const { return data[2][2]; }
  float eta() const { return data[2][3]; }
  float alpha() const { return roughness() * roughness(); }

  // disney_material.hlsli:46-79; image values: image_value.h:183-207; flags: BDPTFlagBits
  void load(const orc_scene& sc, uint32_t address, float u, float v, float uv_screen_size, uint32_t& packed_shading_normal, uint32_t& packed_tangent, uint32_t sampling_flags) {
    const sthip_MaterialRecord* rec = (const sthip_MaterialRecord*)&sc.materials[address];
    const bool ray_cones = (sampling_flags >> STHIP_eRayCones) & 1u;
    for (int i = 0; i < 3; i++) {
      const sthip_ImageValue4& iv = rec->values[i];
      float out[4] = {iv.value[0], iv.value[1], iv.value[2], iv.value[3]};
      if (iv.image_index < STHIP_IMAGE_COUNT) {  // ImageValue4::eval, image_value.h:194-198
        if (!(iv.value[0] > 0 || iv.value[1] > 0 || iv.value[2] > 0 || iv.value[3] > 0)) {
          out[0] = out[1] = out[2] = out[3] = 0;
        } else {
          float t[4];
          sample_image(sc, iv.image_index, u, v, uv_screen_size, ray_cones, t);
          for (int j = 0; j < 4; j++) out[j] = iv.value[j] * t[j];
        }
      }
      for (int j = 0; j < 4; j++) data[i][j] = out[j];
    }
    // normal map, disney_material.hlsli:55-74 (flip_bitangent is never set on this path)
    if (((sampling_flags >> STHIP_eNormalMaps) & 1u) && rec->bump_index < STHIP_IMAGE_COUNT && rec->bump_strength > 0) {
      float t[4];
      sample_image(sc, rec->bump_index, u, v, uv_screen_size, ray_cones, t);  // ImageValue3 with value = 1
      v3 bump = V3(1.0f * t[0], 1.0f * t[1], 1.0f * t[2]) * 2 - V3(1.0f);
      if ((sampling_flags >> STHIP_eFlipNormalMaps) & 1u) bump.y = -bump.y;
      bump = normalize(V3(bump.x * rec->bump_strength, bump.y * rec->bump_strength, bump.z > 0 ? bump.z : 1.0f));
      v3 n = unpack_normal_octahedron(packed_shading_normal);
      v3 t3 = unpack_normal_octahedron(packed_tangent);
      n = normalize(t3 * bump.x + cross(n, t3) * bump.y + n * bump.z);
      t3 = normalize(t3 - n * dot(n, t3));
      packed_shading_normal = pack_normal_octahedron(n);
      packed_tangent = pack_normal_octahedron(t3);
    }
  }
  void load(const orc_scene& sc, uint32_t address, ShadingData& sd, uint32_t sampling_flags) {
    load(sc, address, sd.u, sd.v, sd.uv_screen_size, sd.packed_shading_normal, sd.packed_tangent, sampling_flags);
  }

  // sample_image, image_value.h:81-97: SampleLevel(gStaticSampler, uv, lod) restated as repeat addressing +
  // trilinear filtering over the box-filtered mip chain (the reference's 8x anisotropy is hardware-defined)
  static void texel(const OrcImage& im, uint32_t level, int x, int y, float out[4]) {
    const int w = (int)im.w[level], h = (int)im.h[level];
    x = ((x % w) + w) % w;
    y = ((y % h) + h) % h;
    const float* p = &im.mip[level][4 * ((size_t)y * w + x)];
    out[0] = p[0];
    out[1] = p[1];
    out[2] = p[2];
    out[3] = p[3];
  }
  static void bilinear(const OrcImage& im, uint32_t level, float u, float v, float out[4]) {
    const float x = u * (float)im.w[level] - 0.5f, y = v * (float)im.h[level] - 0.5f;
    const float x0 = floorf(x), y0 = floorf(y);
    const float fx = x - x0, fy = y - y0;
    const int ix = (int)x0, iy = (int)y0;
    float c00[4], c10[4], c01[4], c11[4];
    texel(im, level, ix, iy, c00);
    texel(im, level, ix + 1, iy, c10);
    texel(im, level, ix, iy + 1, c01);
    texel(im, level, ix + 1, iy + 1, c11);
    for (int k = 0; k < 4; k++) {
      const float a = lerpf(c00[k], c10[k], fx), b = lerpf(c01[k], c11[k], fx);
      out[k] = lerpf(a, b, fy);
    }
  }
  static void sample_image(const orc_scene& sc, uint32_t index, float u, float v, float uv_screen_size, bool ray_cones, float out[4]) {
    const OrcImage& im = sc.images[index];
    float lod = 0;
    if (ray_cones && uv_screen_size > 0) lod = det_log2f(fmaxf(uv_screen_size * fmaxf((float)im.w[0], (float)im.h[0]), 1e-6f));
    const float top = (float)(im.w.size() - 1);
    lod = fminf(fmaxf(lod, 0.0f), top);
    const float l0 = floorf(lod);
    const uint32_t i0 = (uint32_t)l0, i1 = std::min<uint32_t>(i0 + 1, (uint32_t)im.w.size() - 1);
    const float f = lod - l0;
    float a[4], b[4];
    bilinear(im, i0, u, v, a);
    bilinear(im, i1, u, v, b);
    for (int k = 0; k < 4; k++) out[k] = lerpf(a[k], b[k], f);
  }
  v3 Le() const { return base_color() * emission(); }  // :81
  v3 albedo() const { return base_color(); }
  bool can_eval() const { return emission() <= 0 && any_gt0(base_color()); }  // :83
  bool is_specular() const { return (metallic() > 0.999f || transmission() > 0.999f) && roughness() <= 1e-2f; }  // :125

  // disney_diffuse.hlsli:1-17
  v3 diffuse_eval(v3 dir_in, v3 dir_out) const {
    const float hdotwo = fabsf(dot(normalize(dir_in + dir_out), dir_out));
    const float FSS90 = roughness() * hdotwo * hdotwo;
    const float FD90 = 0.5f + 2 * FSS90;
    const float ndotwi5 = det_pow5f(1 - fabsf(dir_in.z));
    const float ndotwo5 = det_pow5f(1 - fabsf(dir_out.z));
    const float FDwi = 1 + (FD90 - 1) * ndotwi5;
    const float FDwo = 1 + (FD90 - 1) * ndotwo5;
    const v3 f_base_diffuse = (base_color() / DET_PI) * FDwi * FDwo;
    const float FSSwi = 1 + (FSS90 - 1) * ndotwi5;
    const float FSSwo = 1 + (FSS90 - 1) * ndotwo5;
    const v3 f_subsurface = (1.25f * base_color() / DET_PI) * (FSSwi * FSSwo * (1 / (fabsf(dir_in.z) + fabsf(dir_out.z)) - 0.5f) + 0.5f);
    return lerp3(f_base_diffuse, f_subsurface, subsurface()) * fabsf(dir_out.z);
  }
  void alphas(float& ax, float& ay) const {
    const float aspect = sqrtf(1 - 0.9f * anisotropic());
    ax = fmaxf(0.0001f, alpha() / aspect);
    ay = fmaxf(0.0001f, alpha() * aspect);
  }
  // disney_glass.hlsli:1-29
  static float glass_reflect_pdf(float F, float D, float G_in, float cos_theta_in) { return (F * D * G_in) / (4 * fabsf(cos_theta_in)); }
  static float glass_refract_pdf(float F, float D, float G_in, float cos_theta_in, float h_dot_in, float h_dot_out, float eta) {
    const float sqrt_denom = h_dot_in + eta * h_dot_out;
    const float dh_dout = eta * eta * h_dot_out / (sqrt_denom * sqrt_denom);
    return (1 - F) * D * G_in * fabsf(dh_dout * h_dot_in / cos_theta_in);
  }
  static v3 glass_eval_reflect(v3 base_color, float F, float D, float G, float cos_theta_in) { return base_color * (F * D * G) / (4 * fabsf(cos_theta_in)); }
  static v3 glass_eval_refract(v3 base_color, float F, float D, float G, float cos_theta_in, float h_dot_in, float h_dot_out, float local_eta, bool adjoint) {
    const float sqrt_denom = h_dot_in + local_eta * h_dot_out;
    const float eta_factor = adjoint ? (1 / (local_eta * local_eta)) : 1;
    const v3 sq = V3(sqrtf(base_color.x), sqrtf(base_color.y), sqrtf(base_color.z));
    return sq * (eta_factor * (1 - F) * D * G * fabsf(h_dot_out * h_dot_in)) / (fabsf(cos_theta_in) * sqrt_denom * sqrt_denom);
  }
  // disney_metal.hlsli:1-7
  static float metal_eval_pdf(float D, float G_in, float cos_theta_in) { return D * G_in / (4 * fabsf(cos_theta_in)); }
  static v3 metal_eval(v3 base_color, float D, float G, v3 dir_in, float h_dot_out) {
    return base_color * schlick_fresnel3(base_color, fabsf(h_dot_out)) * D * G / (4 * fabsf(dir_in.z));
  }
  // disney_clearcoat.hlsli:1-9
  static float clearcoat_eval_pdf(float D, v3 h, float hdotwo) { return D * fabsf(h.z) / (4 * fabsf(hdotwo)); }
  static float clearcoat_eval(float D, v3 dir_in, v3 dir_out, v3 /*h*/, float hdotwo) {
    const float Fc = schlick_fresnel1(R0(1.5f), hdotwo);
    return Fc * D * Gc(dir_in) * Gc(dir_out) / (4 * fabsf(dir_in.z));
  }

  // disney_material.hlsli:141-200
  void eval(MaterialEvalRecord& r, v3 dir_in, v3 dir_out, bool adjoint) const {
    r.f = V3(0.0f);
    r.pdf_fwd = r.pdf_rev = 0;
    if (emission() > 0) return;
    const float one_minus_metallic = 1 - metallic();
    const float w_diffuse = (1 - transmission()) * one_minus_metallic;
    const float w_metal = metallic();
    const float w_glass = transmission() * one_minus_metallic;
    const float w_clearcoat = 0.25f * clearcoat();
    const float local_eta = dir_in.z < 0 ? 1 / eta() : eta();
    const bool transmit = dir_in.z * dir_out.z < 0;
    v3 h = normalize(transmit ? (dir_in + dir_out * local_eta) : (dir_in + dir_out));
    if (h.z * dir_in.z < 0) h = -h;
    const float h_dot_in = dot(h, dir_in);
    const float h_dot_out = dot(h, dir_out);
    float ax, ay;
    alphas(ax, ay);
    const float D = Dm(ax, ay, h);
    const float G_in = G1(ax, ay, dir_in);
    const float G_out = G1(ax, ay, dir_out);
    const float F = fresnel_dielectric(h_dot_in, local_eta);
    if (transmit) {
      if (w_glass > 0) {
        r.f = w_glass * glass_eval_refract(base_color(), F, D, G_in * G_out, dir_in.z, h_dot_in, h_dot_out, local_eta, adjoint);
        r.pdf_fwd = w_glass * glass_refract_pdf(F, D, G_in, dir_in.z, h_dot_in, h_dot_out, local_eta);
        r.pdf_rev = w_glass * glass_refract_pdf(fresnel_dielectric(h_dot_out, 1 / local_eta), D, G_out, dir_out.z, h_dot_out, h_dot_in, 1 / local_eta);
      }
    } else {
      if (w_glass > 0) {
        r.f = r.f + w_glass * glass_eval_reflect(base_color(), F, D, G_in * G_out, dir_in.z);
        r.pdf_fwd += w_glass * glass_reflect_pdf(F, D, G_in, dir_in.z);
        r.pdf_rev += w_glass * glass_reflect_pdf(fresnel_dielectric(h_dot_out, local_eta), D, G_out, dir_out.z);
      }
      if (w_metal > 0) {
        r.f = r.f + w_metal * metal_eval(base_color(), D, G_in * G_out, dir_in, dot(h, dir_out));
        r.pdf_fwd += w_metal * metal_eval_pdf(D, G_in, dir_in.z);
        r.pdf_rev += w_metal * metal_eval_pdf(D, G_out, dir_out.z);
      }
      if (w_clearcoat > 0) {
        const float D_c = Dc((1 - clearcoat_gloss()) * 0.1f + clearcoat_gloss() * 0.001f, h.z);
        r.f = r.f + V3(w_clearcoat * clearcoat_eval(D_c, dir_in, dir_out, h, h_dot_out));
        r.pdf_fwd += w_clearcoat * clearcoat_eval_pdf(D_c, h, h_dot_out);
        r.pdf_rev += w_clearcoat * clearcoat_eval_pdf(D_c, h, h_dot_in);
      }
      if (w_diffuse > 0) {
        r.pdf_fwd += w_diffuse * cosine_hemisphere_pdfW(fabsf(dir_out.z));
        r.pdf_rev += w_diffuse * cosine_hemisphere_pdfW(fabsf(dir_in.z));
        r.f = r.f + w_diffuse * diffuse_eval(dir_in, dir_out);
      }
    }
  }

  // disney_material.hlsli:201-315; returns f, multiplies beta by f/pdf_fwd
  v3 sample(MaterialSampleRecord& r, v3 rnd, v3 dir_in, v3& beta, bool adjoint) const {
    if (emission() > 0) {
      beta = V3(0.0f);
      r.pdf_fwd = r.pdf_rev = 0;
      r.eta = 0;
      r.roughness = 0;
      r.dir_out = V3(0.0f);
      return V3(0.0f);
    }
    const float one_minus_metallic = 1 - metallic();
    const float w_diffuse = (1 - transmission()) * one_minus_metallic;
    const float w_metal = metallic();
    const float w_glass = transmission() * one_minus_metallic;
    const float w_clearcoat = 0.25f * clearcoat();
    float ax, ay;
    alphas(ax, ay);
    const float alpha_c = (1 - clearcoat_gloss()) * 0.1f + clearcoat_gloss() * 0.001f;
    const float local_eta = dir_in.z < 0 ? 1 / eta() : eta();
    const float G_in = G1(ax, ay, dir_in);
    v3 h;
    float h_dot_in, D, F;
    r.eta = 0;
    r.roughness = roughness();
    if (rnd.z < w_glass + w_metal) {
      h = sample_visible_normals(dir_in, ax, ay, rnd.x, rnd.y);
      h_dot_in = dot(h, dir_in);
      D = Dm(ax, ay, h);
      F = fresnel_dielectric(h_dot_in, local_eta);
      if (rnd.z < w_glass) {
        const float h_dot_out_sq = 1 - (1 - h_dot_in * h_dot_in) / (local_eta * local_eta);
        if (h_dot_out_sq <= 0 || rnd.z / w_glass <= F) {
          r.dir_out = reflect(-dir_in, h);
        } else {
          r.dir_out = refract(-dir_in, h, 1 / local_eta);
          r.eta = local_eta;
          const float G_out = G1(ax, ay, r.dir_out);
          const float h_dot_out = dot(h, r.dir_out);
          r.pdf_fwd = w_glass * glass_refract_pdf(F, D, G_in, dir_in.z, h_dot_in, h_dot_out, local_eta);
          r.pdf_rev = w_glass * glass_refract_pdf(fresnel_dielectric(h_dot_out, 1 / local_eta), D, G_out, r.dir_out.z, h_dot_out, h_dot_in, 1 / local_eta);
          const v3 f = w_glass * glass_eval_refract(base_color(), F, D, G_in * G_out, dir_in.z, h_dot_in, h_dot_out, local_eta, adjoint);
          beta = beta * (f / r.pdf_fwd);
          return f;
        }
      } else {
        r.dir_out = reflect(-dir_in, h);
      }
    } else {
      if (rnd.z < w_glass + w_metal + w_clearcoat) {
        const float alpha2 = alpha_c * alpha_c;
        const float cos_phi = sqrtf((1 - det_powf(alpha2, 1 - rnd.x)) / (1 - alpha2));
        const float sin_phi = sqrtf(1 - fmaxf(cos_phi * cos_phi, 0.0f));
        const float theta = DET_2PI * rnd.y;
        float st, ct;
        det_sincosf(theta, &st, &ct);
        h = V3(sin_phi * ct, sin_phi * st, cos_phi);
        if (dir_in.z < 0) h = -h;
        r.dir_out = reflect(-dir_in, h);
        r.roughness = alpha_c;
      } else {
        r.dir_out = sample_cos_hemisphere(rnd.x, rnd.y);
        if (dir_in.z < 0) r.dir_out = -r.dir_out;
        r.roughness = 1;
        h = normalize(dir_in + r.dir_out);
      }
      h_dot_in = dot(h, dir_in);
      D = Dm(ax, ay, h);
      F = fresnel_dielectric(h_dot_in, local_eta);
    }
    const float G_out = G1(ax, ay, r.dir_out);
    const float h_dot_out = dot(h, r.dir_out);
    r.pdf_fwd = 0;
    r.pdf_rev = 0;
    v3 f = V3(0.0f);
    if (w_glass > 0) {
      r.pdf_fwd += w_glass * glass_reflect_pdf(F, D, G_in, dir_in.z);
      r.pdf_rev += w_glass * glass_reflect_pdf(fresnel_dielectric(h_dot_out, local_eta), D, G_out, r.dir_out.z);
      f = f + w_glass * glass_eval_reflect(base_color(), F, D, G_in * G_out, dir_in.z);
    }
    if (w_metal > 0) {
      r.pdf_fwd += w_metal * metal_eval_pdf(D, G_in, dir_in.z);
      r.pdf_rev += w_metal * metal_eval_pdf(D, G_out, r.dir_out.z);
      f = f + w_metal * metal_eval(base_color(), D, G_in * G_out, dir_in, h_dot_out);
    }
    if (w_clearcoat > 0) {
      const float D_c = Dc(alpha_c, h.z);
      r.pdf_fwd += w_clearcoat * clearcoat_eval_pdf(D_c, h, h_dot_out);
      r.pdf_rev += w_clearcoat * clearcoat_eval_pdf(D_c, h, h_dot_in);
      f = f + V3(w_clearcoat * clearcoat_eval(D_c, dir_in, r.dir_out, h, h_dot_out));
    }
    if (w_diffuse > 0) {
      r.pdf_fwd += w_diffuse * cosine_hemisphere_pdfW(fabsf(r.dir_out.z));
      r.pdf_rev += w_diffuse * cosine_hemisphere_pdfW(fabsf(dir_in.z));
      f = f + w_diffuse * diffuse_eval(dir_in, r.dir_out);
    }
    beta = beta * (f / r.pdf_fwd);
    return f;
  }
};

// ---------------------------------------------------------------------------------------------
// frame state shared by all pixels
// ---------------------------------------------------------------------------------------------
struct PresampledLightPoint {  // bdpt.h:92-100
  v3 position;
  uint32_t packed_geometry_normal;
  v3 Le;
  float pdfA;  // negative for environment map samples
};
// PathVertex, bdpt.h:102-155: a stored light-subpath vertex (eConnectToLightPaths), 64 bytes
// reservoir.h:4-27
struct Reservoir {
  float total_weight;
  uint32_t M;
  float W(float sample_target_pdf) const { return (sample_target_pdf > 0 && M > 0) ? total_weight / ((float)M * sample_target_pdf) : 0.0f; }
  void init() {
    total_weight = 0;
    M = 0;
  }
  bool update(float rnd, float w) {
    M++;
    total_weight += w;
    return rnd * total_weight <= w;
  }
};

// NEEReservoir, bdpt.h:157-165 (48 bytes): what a view vertex appends to the NEE hash grid
struct NEEReservoir {
  Reservoir r;
  uint32_t packed_geometry_normal;
  float W;
  PresampledLightPoint y;
};
static_assert(sizeof(NEEReservoir) == 48, "NEEReservoir is 48 bytes");

// hashgrid.hlsli:22-89. Upstream builds the grid with atomics (compare-exchange probing, per-bucket counters, a global
// append counter), so which bucket a cell gets when two cells compete, and the order of a bucket's records, depend on
// thread scheduling. DEFINED here: records are appended in (path index, diffuse vertex) order — the result of a serial
// run of upstream's kernel. Probing does not wrap upstream (it runs off the buffer, where robust buffer access turns
// the operations into no-ops of unspecified effect); here the table simply has 32 slots more than gHashGridBucketCount.
struct HashGridTable {
  uint32_t bucket_count = 0;
  std::vector<uint32_t> checksums, counters, indices;
  void reset(uint32_t n) {
    bucket_count = n;
    checksums.assign((size_t)n + 32, 0u);
    counters.assign((size_t)n + 32, 0u);
    indices.assign((size_t)n + 32, 0u);
  }
  static uint32_t f2u_sat(float f) {  // float -> uint as the hardware converts: NaN and negatives to 0, too large to 0xFFFFFFFF
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
  }
  // hashgrid_bucket_index, hashgrid.hlsli:15-21
  uint32_t bucket_index(v3 pos, float cell_size, uint32_t& checksum) const {
    const float q[3] = {floorf(pos.x / cell_size) + 0.5f, floorf(pos.y / cell_size) + 0.5f, floorf(pos.z / cell_size) + 0.5f};
    int32_t pi[3];
    for (int k = 0; k < 3; k++) pi[k] = q[k] >= 2147483648.0f ? 0x7FFFFFFF : (q[k] <= -2147483648.0f ? (int32_t)0x80000000 : (q[k] == q[k] ? (int32_t)q[k] : 0));
    const uint32_t px = (uint32_t)pi[0], py = (uint32_t)pi[1], pz = (uint32_t)pi[2];
    checksum = std::max(1u, xxhash32(f2u_sat(cell_size + (float)xxhash32(pz + xxhash32(py + xxhash32(px))))));
    return pcg(f2u_sat(cell_size + (float)pcg(pz + pcg(py + pcg(px))))) % bucket_count;
  }
  uint32_t find(v3 pos, float cell_size) const {
    uint32_t checksum;
    uint32_t b = bucket_index(pos, cell_size, checksum);
    for (uint32_t i = 0; i < 32; i++, b++)
      if (checksums[b] == checksum) return b;
    return 0xFFFFFFFFu;
  }
  uint32_t find_or_insert(v3 pos, float cell_size) {
    uint32_t checksum;
    uint32_t b = bucket_index(pos, cell_size, checksum);
    for (uint32_t i = 0; i < 32; i++, b++) {
      if (checksums[b] == 0) checksums[b] = checksum;
      if (checksums[b] == checksum) return b;
    }
    return 0xFFFFFFFFu;
  }
};
template <typename T>
struct HashGridOf {
  HashGridTable table;
  std::vector<T> data;
  // one frame's appends, in the defined order: (position, cell size, record); builds counters, indices and data
  struct Append {
    v3 pos;
    float cell_size;
    T y;
  };
  void build(uint32_t bucket_count, const std::vector<Append>& appends) {
    table.reset(bucket_count);
    std::vector<std::pair<uint32_t, uint32_t>> where;  // (bucket, index in bucket) of every stored append
    std::vector<const T*> what;
    for (const Append& a : appends) {
      const uint32_t b = table.find_or_insert(a.pos, a.cell_size);
      if (b == 0xFFFFFFFFu) continue;  // 32 probes without a free or matching slot: the record is dropped (hashgrid.hlsli:56-58)
      where.emplace_back(b, table.counters[b]++);
      what.push_back(&a.y);
    }
    uint32_t running = 0;  // compute_indices, hashgrid.hlsli:72-79 (upstream: in whatever order the atomics land; here by bucket index)
    for (size_t b = 0; b < table.counters.size(); b++) {
      table.indices[b] = running;
      running += table.counters[b];
    }
    data.assign(running, T{});
    for (size_t k = 0; k < where.size(); k++) data[table.indices[where[k].first] + where[k].second] = *what[k];  // swizzle, :81-88
  }
};

struct PathVertex {
  float position[3];
  uint32_t packed_geometry_normal;
  uint32_t material_address;
  uint32_t packed_local_dir_in;
  uint32_t packed_shading_normal;
  uint32_t packed_tangent;
  float uv[2];
  uint32_t packed_beta[2];  // f16 beta rgb | subpath_length:7 | diffuse_vertices:5 | flags:4
  float prev_dVC, G_rev, prev_pdfA_fwd, path_pdf;
  void pack_beta(v3 b, uint32_t subpath_length, uint32_t diffuse_vertices, uint32_t flags) {
    packed_beta[0] = det_f32tof16(b.x) | (det_f32tof16(b.y) << 16);
    packed_beta[1] = det_f32tof16(b.z) | ((subpath_length & 0x7Fu) << 16) | ((diffuse_vertices & 0x1Fu) << 23) | ((flags & 0xFu) << 28);
  }
  v3 beta() const { return V3(det_f16tof32(packed_beta[0] & 0xFFFFu), det_f16tof32(packed_beta[0] >> 16), det_f16tof32(packed_beta[1] & 0xFFFFu)); }
  uint32_t subpath_length() const { return (packed_beta[1] >> 16) & 0x7Fu; }
  uint32_t diffuse_vertices() const { return (packed_beta[1] >> 23) & 0x1Fu; }
  bool is_prev_delta() const { return (packed_beta[1] >> 28) & 8u; }  // PATH_VERTEX_FLAG_IS_PREV_DELTA
  bool is_medium() const { return (packed_beta[1] >> 28) & 4u; }      // PATH_VERTEX_FLAG_IS_MEDIUM
  bool flip_bitangent() const { return (packed_beta[1] >> 28) & 1u; }
};
static_assert(sizeof(PathVertex) == 64, "PathVertex is 64 bytes");
// PathVertexReservoir, bdpt.h:166-174 (80 bytes): what a view vertex appends to the LVC hash grid
struct PathVertexReservoir {
  Reservoir r;
  uint32_t packed_geometry_normal;
  float W;
  PathVertex y;
};
static_assert(sizeof(PathVertexReservoir) == 80, "PathVertexReservoir is 80 bytes");

struct Frame {
  // gLightPathVertices of the seed being traced (eConnectToLightPaths): [diffuse_vertices - 1][W * H], zero-filled
  // before sample_photons (BDPT.cpp:569-572,655-659)
  PathVertex* light_vertices = nullptr;
  size_t light_vertex_count = 0;
  // eLVC (the light vertex cache, path.hlsli:523-527,683-800). Upstream hands out cache slots with an atomic counter, so
  // its cache order — and with it which vertex `li % n` names — depends on thread scheduling. The order DEFINED here is
  // one upstream's scheduler may produce: light paths in path-index order, each path's vertices in the order it stores
  // them. While the light paths are traced the vertices go to lvc_staging[path_index * (gMaxDiffuseVertices - 1) +
  // diffuse_vertices - 1]; compacting that array gives gLightPathVertices, lvc_count = gLightPathVertexCount[0].
  PathVertex* lvc_staging = nullptr;
  uint32_t lvc_count = 0;
  // eNEEReservoirReuse: the previous seed's grid (null for the first seed of a call: gReservoirSpatialM = 0 then,
  // BDPT.cpp:482-483) and this seed's appends, staged at [path_index * gMaxDiffuseVertices + diffuse_vertices - 1]
  const HashGridOf<NEEReservoir>* prev_nee_grid = nullptr;
  HashGridOf<NEEReservoir>::Append* nee_appends = nullptr;
  uint8_t* nee_append_valid = nullptr;
  // eLVCReservoirReuse: the same for the reservoirs of connect_lvc (path.hlsli:727-768)
  const HashGridOf<PathVertexReservoir>* prev_lvc_grid = nullptr;
  HashGridOf<PathVertexReservoir>::Append* lvc_appends = nullptr;
  uint8_t* lvc_append_valid = nullptr;
  bool lvc() const { return flag(STHIP_eConnectToLightPaths) && flag(STHIP_eLVC); }
  bool bdpt() const { return flag(STHIP_eConnectToViews) || flag(STHIP_eConnectToLightPaths); }
  const orc_scene* sc;
  sthip_BDPTPushConstants pc;
  uint32_t sampling_flags, scene_flags;
  sthip_frame_desc fd;
  // gPresampledLights of every seed of the call (ePresampleLights): [seed - seed_begin][gLightPresampleTileSize * TileCount]
  const std::vector<PresampledLightPoint>* presampled = nullptr;
  uint32_t seed_begin = 0;
  // gLightTraceSamples of the seed being traced (eConnectToViews): uint4 per pixel = quantised rgb sums + overflow bits
  std::atomic<uint32_t>* light_trace = nullptr;
  uint32_t light_trace_quantization = 65536;  // BDPT.hpp:55
  bool flag(int b) const { return (sampling_flags >> b) & 1u; }
  // BDPTDebugMode (bdpt.h:177-193): a specialisation constant upstream (gDebugMode), here sthip_outputs::debug_mode
  uint32_t debug_mode = 0;
  bool debug(uint32_t m) const { return debug_mode == m; }
};

// dist2.h:6-20 (upper_bound), :29-57 (dist2d_pdf / dist2d_sample) over gDistributions
inline uint32_t dist_upper_bound(const std::vector<float>& data, uint32_t first, uint32_t last, float value) {
  int count = (int)(last - first);
  while (count > 0) {
    uint32_t it = first;
    const int step = count / 2;
    it += (uint32_t)step;
    if (value >= data[it]) {
      first = ++it;
      count -= step + 1;
    } else
      count = step;
  }
  return first;
}
inline int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
inline float dist2d_pdf(const std::vector<float>& data, uint32_t pdf_marginals, uint32_t pdf_rows, uint32_t w, uint32_t h, float u, float v) {
  const int x = (int)fminf(fmaxf(u * (float)w, 0.0f), (float)(w - 1));
  const int y = (int)fminf(fmaxf(v * (float)h, 0.0f), (float)(h - 1));
  const float pdf_y = data[pdf_marginals + y];
  const float pdf_x = data[pdf_rows + y * w + x];
  return pdf_y * pdf_x * (float)w * (float)h;
}
inline void dist2d_sample(const std::vector<float>& data, uint32_t cdf_marginals, uint32_t cdf_rows, uint32_t w, uint32_t h, float rx, float ry, float& u, float& v) {
  const uint32_t y_ptr = dist_upper_bound(data, cdf_marginals, cdf_marginals + h + 1, ry) - cdf_marginals;
  const int y_offset = clampi((int)y_ptr - 1, 0, (int)h - 1);
  float dy = ry - data[cdf_marginals + y_offset];
  if ((data[cdf_marginals + y_offset + 1] - data[cdf_marginals + y_offset]) > 0) dy /= (data[cdf_marginals + y_offset + 1] - data[cdf_marginals + y_offset]);
  const int row_offset = y_offset * ((int)w + 1);
  const uint32_t x_ptr = dist_upper_bound(data, cdf_rows + row_offset, cdf_rows + row_offset + w + 1, rx) - cdf_rows;
  const int x_offset = clampi((int)x_ptr - row_offset - 1, 0, (int)w - 1);
  float dx = rx - data[cdf_rows + row_offset + x_offset];
  if (data[cdf_rows + row_offset + x_offset + 1] - data[cdf_rows + row_offset + x_offset] > 0) dx /= (data[cdf_rows + row_offset + x_offset + 1] - data[cdf_rows + row_offset + x_offset]);
  u = ((float)x_offset + dx) / (float)w;
  v = ((float)y_offset + dy) / (float)h;
}

// environment.h:8-95. Record in gMaterialData: ImageValue3 (float3 value, uint image_index) and, when an image is
// bound, the offsets of marginal_pdf, row_pdf, marginal_cdf, row_cdf in gDistributions (:17-22,37-45).
// eSampleEnvironmentMapDirectly (sample_texel) is not restated.
struct Environment {
  v3 value;
  uint32_t image_index;
  uint32_t marginal_pdf, row_pdf, marginal_cdf, row_cdf;
  const orc_scene* sc;
  bool has_image() const { return image_index < sc->images.size(); }
  void load(const orc_scene& scene, uint32_t address) {
    sc = &scene;
    float f[3];
    memcpy(f, &scene.materials[address], 12);
    value = V3(f[0], f[1], f[2]);
    memcpy(&image_index, &scene.materials[address + 12], 4);
    marginal_pdf = row_pdf = marginal_cdf = row_cdf = 0;
    if (has_image()) {
      uint32_t d[4];
      memcpy(d, &scene.materials[address + 16], 16);
      marginal_pdf = d[0];
      row_pdf = d[1];
      marginal_cdf = d[2];
      row_cdf = d[3];
    }
  }
  v3 lookup(float u, float v) const {
    float c[4];
    DisneyMaterial::sample_image(*sc, image_index, u, v, 0.0f, false, c);
    return V3(c[0], c[1], c[2]);
  }
  v3 eval(v3 dir_out) const {
    if (!has_image()) return value;
    float u, v;
    cartesian_to_spherical_uv(dir_out, u, v);
    return lookup(u, v) * value;
  }
  // sample_texel / sample_texel_pdf, bdpt_util.hlsli:85-180 (eSampleEnvironmentMapDirectly): a descent through the mip
  // chain from the 2 x 1 level towards level 1 (at most 10 levels), at each level one of the 2 x 2 children in
  // proportion to luminance x sin(theta). Texel loads outside a level return zero, as Texture2D::Load does.
  float texel_weight(uint32_t level, uint32_t x, uint32_t y, float inv_h) const {
    const OrcImage& im = sc->images[image_index];
    if (x >= im.w[level] || y >= im.h[level]) return 0.0f * det_sin(DET_PI * ((float)y + 0.5f) * inv_h);
    const float* t = &im.mip[level][4 * ((size_t)y * im.w[level] + x)];
    return luminance(V3(t[0], t[1], t[2])) * det_sin(DET_PI * ((float)y + 0.5f) * inv_h);
  }
  static float det_sin(float x) {
    float sn, cs;
    det_sincosf(x, &sn, &cs);
    return sn;
  }
  bool texel_level(uint32_t level, uint32_t cx, uint32_t cy, float p[4]) const {
    const OrcImage& im = sc->images[image_index];
    const float inv_h = 1 / (float)im.h[level];
    p[0] = p[1] = p[2] = p[3] = 0;
    if (im.w[level] > 1) {
      p[0] = texel_weight(level, cx, cy, inv_h);
      p[1] = texel_weight(level, cx + 1, cy, inv_h);
    }
    if (im.h[level] > 1) {
      p[2] = texel_weight(level, cx, cy + 1, inv_h);
      p[3] = texel_weight(level, cx + 1, cy + 1, inv_h);
    }
    const float sum = ((p[0] + p[1]) + p[2]) + p[3];  // dot(p, 1)
    if (sum < 1e-6f) return false;
    for (int j = 0; j < 4; j++) p[j] /= sum;
    return true;
  }
  void sample_texel(float rx, float ry, float& pdf, float& u, float& v) const {
    const OrcImage& im = sc->images[image_index];
    const uint32_t level_count = (uint32_t)im.w.size();
    pdf = 1;
    uint32_t cx = 0, cy = 0, lw = 1, lh = 1;
    for (uint32_t i = 1; i < std::min(10u + 1u, level_count - 1); i++) {
      const uint32_t level = level_count - 1 - i;
      const uint32_t w = im.w[level], h = im.h[level];
      cx *= w / lw;
      cy *= h / lh;
      float p[4];
      if (!texel_level(level, cx, cy, p)) continue;
      for (int j = 0; j < 4; j++) {
        if (rx < p[j]) {
          cx += (uint32_t)(j & 1);
          cy += (uint32_t)(j >> 1);
          pdf *= p[j];
          rx /= p[j];
          break;
        }
        rx -= p[j];
      }
      lw = w;
      lh = h;
    }
    pdf *= (float)(lw * lh);
    u = ((float)cx + rx) / (float)lw;
    v = ((float)cy + ry) / (float)lh;
  }
  float sample_texel_pdf(float u, float v) const {
    const OrcImage& im = sc->images[image_index];
    const uint32_t level_count = (uint32_t)im.w.size();
    float pdf = 1;
    uint32_t lw = 1, lh = 1;
    for (uint32_t i = 1; i < std::min(10u + 1u, level_count - 1); i++) {
      const uint32_t level = level_count - 1 - i;
      const uint32_t w = im.w[level], h = im.h[level];
      const uint32_t cx = (uint32_t)(floorf((float)w * u / 2) * 2), cy = (uint32_t)(floorf((float)h * v / 2) * 2);
      float p[4];
      if (!texel_level(level, cx, cy, p)) continue;
      // saturate(uint2(uv * size) - coord): unsigned, clamped to 0 .. 1
      const uint32_t dx = (uint32_t)(u * (float)w) - cx, dy = (uint32_t)(v * (float)h) - cy;
      const uint32_t ox = dx > 1 ? 1 : dx, oy = dy > 1 ? 1 : dy;
      pdf *= p[oy * 2 + ox];
      lw = w;
      lh = h;
    }
    return pdf * (float)(lw * lh);
  }
  v3 sample(float rx, float ry, v3& dir_out, float& pdf, bool direct = false) const {
    if (has_image() && direct) {  // environment.h:66-67
      float u, v;
      sample_texel(rx, ry, pdf, u, v);
      dir_out = spherical_uv_to_cartesian(u, v);
      pdf /= (DET_2PI2 * sqrtf(1 - dir_out.y * dir_out.y));
      return value * lookup(u, v);
    }
    if (!has_image()) {
      // sample_uniform_sphere's (phi, theta) go through spherical_uv_to_cartesian as if they were uv (as upstream)
      dir_out = spherical_uv_to_cartesian(2 * DET_PI * ry, det_acosf(2 * rx - 1));
      pdf = DET_INV_4PI;
      return value;
    }
    const uint32_t w = sc->images[image_index].w[0], h = sc->images[image_index].h[0];
    float u, v;
    dist2d_sample(sc->distributions, marginal_cdf, row_cdf, w, h, rx, ry, u, v);
    pdf = dist2d_pdf(sc->distributions, marginal_pdf, row_pdf, w, h, u, v);
    dir_out = spherical_uv_to_cartesian(u, v);
    pdf /= (DET_2PI2 * sqrtf(1 - dir_out.y * dir_out.y));
    return value * lookup(u, v);
  }
  float eval_pdf(v3 dir_out, bool direct = false) const {
    if (!has_image()) return DET_INV_4PI;
    float u, v;
    cartesian_to_spherical_uv(dir_out, u, v);
    if (direct) return sample_texel_pdf(u, v) / (DET_2PI2 * sqrtf(1 - dir_out.y * dir_out.y));  // environment.h:84-85
    const uint32_t w = sc->images[image_index].w[0], h = sc->images[image_index].h[0];
    const float pdf = dist2d_pdf(sc->distributions, marginal_pdf, row_pdf, w, h, u, v);
    return pdf / (DET_2PI2 * sqrtf(1 - dir_out.y * dir_out.y));
  }
};

struct IntersectionVertex {
  ShadingData sd;
  uint32_t instance_primitive_index;
  float shape_pdf;
  bool shape_pdf_area_measure;
  uint32_t instance_index() const { return instance_primitive_index & 0xFFFF; }
  uint32_t primitive_index() const { return instance_primitive_index >> 16; }
};

// intersection.hlsli:65-191 around the traversal contract
float trace_ray(const Frame& fr, v3 origin, v3 direction, float t_max, IntersectionVertex& isect, bool accept_first, uint64_t* counters) {
  Ray r;
  r.o = origin;
  r.d = direction;
  r.tmin = 0;
  r.tmax = t_max;
  r.alpha_test = fr.flag(STHIP_eAlphaTest);
  r.flip_uvs = fr.flag(STHIP_eFlipTriangleUVs);
  const Hit h = trace(*fr.sc, r, accept_first, false, counters);
  if (h.ip != 0xFFFFFFFFu) {
    isect.instance_primitive_index = h.ip;
    if (accept_first) return h.t;  // occlusion query: nothing else is consumed (intersection.hlsli:198-233)
    const Inst& in = fr.sc->instances[isect.instance_index()];
    if (in.type() == STHIP_INSTANCE_TYPE_SPHERE) {  // intersection.hlsli:140-159
      const sthip_TransformData& inv = fr.sc->inv_xf[isect.instance_index()];
      const v3 local_hit_pos = obj_point(inv, origin) + obj_vector(inv, direction) * h.t;
      make_sphere_shading_data(*fr.sc, isect.sd, isect.instance_index(), local_hit_pos);
      if (fr.flag(STHIP_eUniformSphereSampling)) {
        isect.shape_pdf = 1 / isect.sd.shape_area;
        isect.shape_pdf_area_measure = true;
      } else {
        const sthip_TransformData& t = fr.sc->xf[isect.instance_index()];
        const v3 to_center = V3(t.m[0][3], t.m[1][3], t.m[2][3]) - origin;
        const float sin_elevation_max_sq = pow2(in.radius()) / dot(to_center, to_center);
        const float cos_elevation_max = sqrtf(fmaxf(0.0f, 1 - sin_elevation_max_sq));
        isect.shape_pdf = 1 / (DET_2PI * (1 - cos_elevation_max));
        isect.shape_pdf_area_measure = false;
      }
    } else if (in.type() == STHIP_INSTANCE_TYPE_VOLUME) {  // intersection.hlsli:93-113,160-165; make_volume_shading_data, shading_data.hlsli:106-110
      const uint32_t ii = isect.instance_index();
      const sthip_TransformData& inv = fr.sc->inv_xf[ii];
      const v3 oo = obj_point(inv, origin), od = obj_vector(inv, direction);
      const NvdbGrid& g = fr.sc->volumes[in.volume_index()];
      float tt;
      v3 face = V3(0.0f);
      volume_test(g, oo, od, 0.0f, POS_INF, tt, &face);  // the committed candidate again, for its face
      const v3 vol_normal = normalize(transform_vector(fr.sc->xf[ii], g.index_to_world_dir(face)));
      isect.sd.packed_geometry_normal = isect.sd.packed_shading_normal = pack_normal_octahedron(vol_normal);
      isect.sd.position = transform_point(fr.sc->xf[ii], oo + od * h.t);
      isect.sd.shape_area = 0;
      isect.sd.uv_screen_size = 0;
      isect.shape_pdf = 1;
      isect.shape_pdf_area_measure = false;
    } else {
      make_triangle_shading_data(*fr.sc, isect.sd, isect.instance_index(), isect.primitive_index(), h.b1, h.b2, fr.flag(STHIP_eFlipTriangleUVs));
      isect.shape_pdf = 1 / (isect.sd.shape_area * (float)in.prim_count());
      isect.shape_pdf_area_measure = true;
    }
    isect.sd.flags = 0;
    if (dot(direction, isect.sd.geometry_normal()) < 0) isect.sd.flags |= STHIP_SHADING_FLAG_FRONT_FACE;
    return h.t;
  } else {
    isect.instance_primitive_index = 0xFFFFFFFFu;
    isect.sd.shape_area = 0;
    isect.sd.position = direction;
    isect.shape_pdf = 0;
    isect.shape_pdf_area_measure = false;
    return t_max;
  }
}

// light.hlsli:6-152 (uniform light choice; B4: power sampling is broken upstream)
struct LightSampleRecord {
  v3 radiance;
  float pdf;
  bool pdf_area_measure;
  bool is_environment;
  v3 to_light;
  float dist;
  v3 position;
  v3 normal;
};
inline bool has_environment(const Frame& fr) { return (fr.scene_flags & STHIP_BDPT_FLAG_HAS_ENVIRONMENT) != 0; }
inline bool has_emissives(const Frame& fr) { return (fr.scene_flags & STHIP_BDPT_FLAG_HAS_EMISSIVES) != 0; }
inline bool has_media(const Frame& fr) { return (fr.scene_flags & STHIP_BDPT_FLAG_HAS_MEDIA) != 0; }
void sample_point_on_light(const Frame& fr, LightSampleRecord& ls, const float rnd[4], v3 ref_pos) {
  const orc_scene& sc = *fr.sc;
  ls.is_environment = false;
  ls.radiance = V3(0.0f);
  ls.position = ls.normal = V3(0.0f);
  if (has_environment(fr) && (!has_emissives(fr) || rnd[3] <= fr.pc.gEnvironmentSampleProbability)) {
    Environment env;
    env.load(sc, fr.pc.gEnvironmentMaterialAddress);
    ls.radiance = env.sample(rnd[0], rnd[1], ls.to_light, ls.pdf, fr.flag(STHIP_eSampleEnvironmentMapDirectly));
    if (has_emissives(fr)) ls.pdf *= fr.pc.gEnvironmentSampleProbability;
    ls.is_environment = true;
    ls.dist = POS_INF;
    ls.pdf_area_measure = false;
    return;
  }
  if (!has_emissives(fr)) {
    ls.pdf = 0;
    ls.pdf_area_measure = true;
    ls.to_light = V3(0.0f);
    ls.dist = 0;
    return;
  }
  const float rw = has_environment(fr) ? (rnd[3] - fr.pc.gEnvironmentSampleProbability) / (1 - fr.pc.gEnvironmentSampleProbability) : rnd[3];
  const int li = (int)(rw * ((float)fr.pc.gLightCount * .9999f));
  ls.pdf = 1 / (float)fr.pc.gLightCount;
  const uint32_t light_instance_index = sc.lights[li];
  if (has_environment(fr)) ls.pdf *= 1 - fr.pc.gEnvironmentSampleProbability;
  const Inst& in = sc.instances[light_instance_index];
  float u, v;
  if (in.type() == STHIP_INSTANCE_TYPE_SPHERE) {  // light.hlsli:58-121
    const float r = in.radius();
    const sthip_TransformData& t = sc.xf[light_instance_index];
    if (fr.flag(STHIP_eUniformSphereSampling)) {
      ls.pdf /= 4 * DET_PI * r * r;
      ls.pdf_area_measure = true;
      const float z = 1 - 2 * rnd[0];
      const float r_ = sqrtf(fmaxf(0.0f, 1 - z * z));
      const float phi = DET_2PI * rnd[1];
      float sp, cp;
      det_sincosf(phi, &sp, &cp);
      const v3 local_normal = V3(r_ * cp, z, r_ * sp);
      cartesian_to_spherical_uv(local_normal, u, v);
      ls.position = transform_point(t, r * local_normal);
      ls.normal = normalize(transform_vector(t, local_normal));
      ls.to_light = ls.position - ref_pos;
      ls.dist = length(ls.to_light);
      ls.to_light = ls.to_light / ls.dist;
    } else {
      const v3 center = V3(t.m[0][3], t.m[1][3], t.m[2][3]);
      v3 to_center = center - ref_pos;
      const float dist = length(to_center);
      to_center = to_center / dist;
      const float sinThetaMax = r / dist;
      const float sinThetaMax2 = sinThetaMax * sinThetaMax;
      const float invSinThetaMax = 1 / sinThetaMax;
      const float cosThetaMax = sqrtf(fmaxf(0.0f, 1 - sinThetaMax2));
      ls.pdf /= DET_2PI * (1 - cosThetaMax);
      ls.pdf_area_measure = false;
      float cosTheta = (cosThetaMax - 1) * rnd[0] + 1;
      float sinTheta2 = 1 - cosTheta * cosTheta;
      if (sinThetaMax2 < 0.00068523f) {
        sinTheta2 = sinThetaMax2 * rnd[0];
        cosTheta = sqrtf(1 - sinTheta2);
      }
      const float cosAlpha = sinTheta2 * invSinThetaMax + cosTheta * sqrtf(fmaxf(0.0f, 1 - sinTheta2 * invSinThetaMax * invSinThetaMax));
      const float sinAlpha = sqrtf(fmaxf(0.0f, 1 - cosAlpha * cosAlpha));
      const float phi = rnd[1] * 2 * DET_PI;
      float sp, cp;
      det_sincosf(phi, &sp, &cp);
      v3 T, B;
      make_orthonormal(to_center, T, B);
      ls.normal = -(T * sinAlpha * cp + B * sinAlpha * sp + to_center * cosAlpha);
      ls.position = center + r * ls.normal;
      ls.to_light = ls.position - ref_pos;
      ls.dist = length(ls.to_light);
      ls.to_light = ls.to_light / ls.dist;
      const v3 local_normal = transform_vector(sc.inv_xf[light_instance_index], ls.normal);
      cartesian_to_spherical_uv(local_normal, u, v);
    }
  } else {
    const uint32_t pc = in.prim_count();
    const uint32_t prim_index = (uint32_t)fminf(rnd[2] * (float)pc, (float)(pc - 1));
    const float a = sqrtf(rnd[0]);
    const float b1 = 1 - a, b2 = a * rnd[1];
    ShadingData sd;
    make_triangle_shading_data(sc, sd, light_instance_index, prim_index, b1, b2, fr.flag(STHIP_eFlipTriangleUVs));
    u = sd.u;
    v = sd.v;
    ls.position = sd.position;
    ls.normal = sd.geometry_normal();
    ls.to_light = sd.position - ref_pos;
    ls.dist = length(ls.to_light);
    ls.to_light = ls.to_light / ls.dist;
    ls.pdf /= sd.shape_area * (float)pc;
    ls.pdf_area_measure = true;
  }
  if (ls.pdf > 0) {
    DisneyMaterial m;
    // light.hlsli:143-150: only uv and uv_screen_size = 0 of the ShadingData are set for this lookup
    uint32_t dummy_n = 0, dummy_t = 0;
    m.load(sc, in.material_address(), u, v, 0.0f, dummy_n, dummy_t, fr.sampling_flags & ~(1u << STHIP_eNormalMaps));
    ls.radiance = m.Le();
  }
}

// light.hlsli:154-174
inline float point_on_light_pdf(const Frame& fr, const IntersectionVertex& isect, v3 direction, bool& area_measure) {
  if (isect.instance_index() == STHIP_INVALID_INSTANCE) {
    area_measure = false;
    if (!has_environment(fr)) return 0;
    Environment env;
    env.load(*fr.sc, fr.pc.gEnvironmentMaterialAddress);
    float pdf = env.eval_pdf(direction, fr.flag(STHIP_eSampleEnvironmentMapDirectly));  // _isect.sd.position holds the ray direction on a miss (intersection.hlsli:183)
    if (has_emissives(fr)) pdf *= fr.pc.gEnvironmentSampleProbability;
    return pdf;
  }
  area_measure = isect.shape_pdf_area_measure;
  if (!has_emissives(fr)) return 0;
  float pdf = isect.shape_pdf;
  pdf /= (float)fr.pc.gLightCount;
  if (has_environment(fr)) pdf *= 1 - fr.pc.gEnvironmentSampleProbability;
  return pdf;
}

// path.hlsli:8-15
inline float mis2(const Frame& fr, float a, float b) {
  if (!fr.flag(STHIP_eMIS)) return 0.5f;
  const float a2 = a * a;
  return a2 / (a2 + b * b);
}
// path.hlsli:67-98 (gShadingNormalFix off, adjoint=false on view paths)
// path.hlsli:67-98, view paths (adjoint = false): the light-leak test, and with eShadingNormalShadowFix the shadow
// terminator term G = min(1, |ngdotout / (ndotout ngdotns)|), G <- -G^3 + G^2 + G
inline float shading_normal_correction(float ndotin, float ndotout, float ngdotin, float ngdotout, float ngdotns = 1.0f, bool terminator_fix = false, bool adjoint = false) {
  if (sgn(ngdotout * ngdotin) != sgn(ndotin * ndotout)) return 0;
  float G = 1;
  if (terminator_fix) {
    G = fminf(1.0f, fabsf(adjoint ? ngdotin / (ndotin * ngdotns) : ngdotout / (ndotout * ngdotns)));
    G = -(pow2(G) * G) + pow2(G) + G;
  }
  if (adjoint) {  // light paths: the non-symmetry of shading normals (Veach), path.hlsli:90-95
    const float num = ngdotout * ndotin;
    const float denom = ndotout * ngdotin;
    if (fabsf(denom) > 1e-5f) G *= fabsf(num / denom);
  }
  return G;
}
inline void project_point(const sthip_ProjectionData& p, v3 v, float r[4]);
// path.hlsli:29-36: dE (or dL) of a vertex from the previous vertex's
inline float connection_dVC(float dVC, float pdfA_rev, float prev_pdfA_fwd, bool specular) { return ((specular ? 0.0f : 1.0f) + dVC * pow2(pdfA_rev)) / pow2(prev_pdfA_fwd); }

// ---------------------------------------------------------------------------------------------
// Medium (materials/medium.hlsli): a heterogeneous participating medium over NanoVDB grids — Henyey-Greenstein phase
// function and delta tracking against the density grid's root maximum
// ---------------------------------------------------------------------------------------------
struct Medium {
  v3 density_scale, albedo_scale;
  float anisotropy, attenuation_unit;
  uint32_t density_volume_index, albedo_volume_index;
  void load(const orc_scene& sc, uint32_t address) {  // medium.hlsli:12-19, Material.hpp:80-87
    float f[8];
    uint32_t u[2];
    memcpy(f, &sc.materials[address], 32);
    memcpy(u, &sc.materials[address + 32], 8);
    density_scale = V3(f[0], f[1], f[2]);
    anisotropy = f[3];
    albedo_scale = V3(f[4], f[5], f[6]);
    attenuation_unit = f[7];
    density_volume_index = u[0];
    albedo_volume_index = u[1];
  }
  bool can_eval() const { return density_scale.x > 0 || density_scale.y > 0 || density_scale.z > 0; }
  bool is_specular() const { return fabsf(anisotropy) > 0.999f; }
  float phase(v3 dir_in, v3 dir_out) const {  // medium.hlsli:26-34
    return DET_INV_4PI * (1 - anisotropy * anisotropy) / det_powf(1 + anisotropy * anisotropy + 2 * anisotropy * dot(dir_in, dir_out), 1.5f);
  }
  // medium.hlsli:35-56; returns dir_out (world space: dir_in is), pdf_fwd = pdf_rev = f
  v3 sample(float r0, float r1, v3 dir_in, float& pdf, float& roughness) const {
    v3 dir_out;
    if (fabsf(anisotropy) < 1e-3f) {
      const float z = 1 - 2 * r0;
      const float phi = DET_2PI * r1;
      float sn, cs;
      det_sincosf(phi, &sn, &cs);
      const float rr = sqrtf(fmaxf(0.0f, 1 - z * z));
      dir_out = V3(rr * cs, rr * sn, z);
    } else {
      const float tmp = (anisotropy * anisotropy - 1) / (2 * r0 * anisotropy - (anisotropy + 1));
      const float cos_elevation = (tmp * tmp - (1 + anisotropy * anisotropy)) / (2 * anisotropy);
      const float sin_elevation = sqrtf(fmaxf(1 - cos_elevation * cos_elevation, 0.0f));
      const float azimuth = DET_2PI * r1;
      float sn, cs;
      det_sincosf(azimuth, &sn, &cs);
      v3 t, b;
      make_orthonormal(dir_in, t, b);
      dir_out = t * (sin_elevation * cs) + b * (sin_elevation * sn) + dir_in * cos_elevation;
    }
    pdf = phase(dir_in, dir_out);
    roughness = 1 - fabsf(anisotropy);
    return dir_out;
  }
  float density_at(const orc_scene& sc, v3 pos_index) const {
    return sc.volumes[density_volume_index].value((int32_t)floorf(pos_index.x), (int32_t)floorf(pos_index.y), (int32_t)floorf(pos_index.z));
  }
  float albedo_at(const orc_scene& sc, v3 pos_index) const {
    if (albedo_volume_index == 0xFFFFFFFFu) return 1;
    return sc.volumes[albedo_volume_index].value((int32_t)floorf(pos_index.x), (int32_t)floorf(pos_index.y), (int32_t)floorf(pos_index.z));
  }
  // delta_track, medium.hlsli:74-127. origin / direction in the object space of the volume instance (= the grid's world
  // space). Returns true with the scatter position (grid world space, as upstream returns it) when a real collision
  // happens; a null collision ends the walk (upstream returns after the first one).
  bool delta_track(const orc_scene& sc, Rng& rng, v3 origin, v3 direction, float t_max, v3& beta, v3& dir_pdf, v3& nee_pdf, bool can_scatter, uint32_t max_null_collisions, v3& scatter_p) const {
    const NvdbGrid& g = sc.volumes[density_volume_index];
    const v3 majorant = density_scale * g.root_max();
    const uint32_t channel = rng.next_uint() % 3u;
    const float maj_c = channel == 0 ? majorant.x : (channel == 1 ? majorant.y : majorant.z);
    if (maj_c < 1e-6f) return false;
    origin = g.world_to_index(origin);
    direction = g.world_to_index_dir(direction);
    for (uint32_t iteration = 0; iteration < max_null_collisions && any_gt0(beta); iteration++) {
      const float r0 = rng.next_float(), r1 = rng.next_float();
      const float t = attenuation_unit * -det_logf(1 - r0) / maj_c;
      if (t < t_max) {
        origin = origin + direction * t;
        t_max -= t;
        const v3 local_density = density_scale * density_at(sc, origin);
        const v3 local_albedo = albedo_scale * albedo_at(sc, origin);
        const v3 local_sigma_s = local_density * local_albedo;
        const v3 local_sigma_a = local_density * (V3(1.0f) - local_albedo);
        const v3 local_sigma_t = local_sigma_s + local_sigma_a;
        const v3 real_prob = V3(local_sigma_t.x / majorant.x, local_sigma_t.y / majorant.y, local_sigma_t.z / majorant.z);
        const float max_maj = fmaxf(fmaxf(majorant.x, majorant.y), majorant.z);
        const v3 tr = V3(det_expf(-majorant.x * t), det_expf(-majorant.y * t), det_expf(-majorant.z * t)) / max_maj;
        const float rp_c = channel == 0 ? real_prob.x : (channel == 1 ? real_prob.y : real_prob.z);
        if (can_scatter && r1 < rp_c) {  // real particle
          beta = beta * (tr * local_sigma_s);
          dir_pdf = dir_pdf * (tr * majorant * real_prob);
          scatter_p = g.index_to_world(origin);
          return true;
        } else {  // fake particle
          beta = beta * (tr * (majorant - local_sigma_t));
          dir_pdf = dir_pdf * (tr * majorant * (V3(1.0f) - real_prob));
          nee_pdf = nee_pdf * (tr * majorant);
          return false;
        }
      } else {  // transmitted without scattering
        const v3 tr = V3(det_expf(-majorant.x * t_max), det_expf(-majorant.y * t_max), det_expf(-majorant.z * t_max));
        beta = beta * tr;
        nee_pdf = nee_pdf * tr;
        dir_pdf = dir_pdf * tr;
        break;
      }
    }
    return false;
  }
};
inline float average3(v3 x) { return (x.x + x.y + x.z) / 3; }

// ---------------------------------------------------------------------------------------------
// P1-P8 — PathIntegrator (path.hlsli:248-1075), view paths
// ---------------------------------------------------------------------------------------------
struct PathIntegrator {
  const Frame& fr;
  uint32_t px, py;
  uint32_t diffuse_vertices, path_length;
  Rng rng;
  v3 beta;
  float eta_scale;
  float bsdf_pdf;
  v3 origin, direction;
  float prev_cos_out;
  IntersectionVertex isect;
  v3 local_dir_in;
  float ngdotin, G;
  float rd_radius, rd_spread;               // RayDifferential (path.hlsli:224-244), only with eRayCones
  // BDPT quantities (path.hlsli:262-267), used when eConnectToViews is on
  float path_pdf, path_pdf_rev, dVC;
  bool prev_specular;
  v3 path_contrib = V3(1.0f);  // the light path's unweighted contribution (path.hlsli:258,901,1043): what eLVCReservoirs store instead of beta
  bool trace_light;                         // gTraceLight: this is a light subpath (sample_photons)
  uint32_t medium = STHIP_INVALID_INSTANCE; // _medium: the volume instance the path is inside of
  float T_nee_pdf = 1;                      // path.hlsli:282
  v3 radiance;                              // accumulate_contribution target (path.hlsli:300-304)
  float* dbg = nullptr;                     // gDebugImage[pixel_coord] (rgba), or null: no debug mode
  // accumulate_contribution, path.hlsli:300-304
  void accumulate_contribution(v3 contrib, float weight, uint32_t light_length) {
    radiance = radiance + contrib * weight;
    if (dbg && fr.debug(STHIP_DEBUG_PATH_LENGTH_CONTRIBUTION) && fr.pc.gDebugLightPathLength == light_length && path_length == fr.pc.gDebugViewPathLength) debug_add(contrib);
  }
  void debug_set(v3 c) {
    dbg[0] = c.x;
    dbg[1] = c.y;
    dbg[2] = c.z;
    dbg[3] = 1;
  }
  void debug_add(v3 c) {
    dbg[0] = dbg[0] + c.x;
    dbg[1] = dbg[1] + c.y;
    dbg[2] = dbg[2] + c.z;
  }
  sthip_ShadowRayData shadow_rays[32];      // this pixel's gShadowRays slots (path.hlsli:65,355-364)
  uint32_t max_shadow;
  uint64_t counters[2];                      // nodes, tris
  uint64_t rays_total, rays_path;

  uint32_t seed;
  // map_pixel_coord, bdpt_util.hlsli:76-83, with the 8x4 groups of bdpt.hlsl:11-12
  uint32_t path_index() const {
    const uint32_t W = fr.pc.gOutputExtent[0];
    if (fr.flag(STHIP_eRemapThreads)) {
      const uint32_t dispatch_w = (W + 7) / 8;
      const uint32_t group_index = (py / 4) * dispatch_w + (px / 8);
      return group_index * 32 + (py % 4) * 8 + (px % 8);
    }
    return py * W + px;
  }

  PathIntegrator(const Frame& f, uint32_t x, uint32_t y, uint32_t seed_, bool light = false) : fr(f), px(x), py(y), seed(seed_) {
    const uint32_t seed = seed_;
    trace_light = light;
    path_pdf = path_pdf_rev = dVC = 1;
    prev_specular = false;
    diffuse_vertices = 0;
    path_length = 1;
    eta_scale = 1;
    rng.v[0] = x;
    rng.v[1] = y;
    rng.v[2] = seed;
    rng.v[3] = light ? 0xFFFFFFu : 0u;  // path.hlsli:295-297, no media
    radiance = V3(0.0f);
    max_shadow = std::min(32u, std::max(fr.pc.gMaxDiffuseVertices, 1u));
    memset(shadow_rays, 0, sizeof(shadow_rays));  // BDPT.cpp:672-676 fill(0)
    counters[0] = counters[1] = 0;
    rays_total = rays_path = 0;
    beta = V3(0.0f);
    bsdf_pdf = 1;
    G = 1;
    ngdotin = 1;
    prev_cos_out = 1;
    rd_radius = rd_spread = 0;
  }

  // path.hlsli:1003-1044
  void trace() {
    rays_path++;
    float T_dir_pdf = 1;
    T_nee_pdf = 1;
    if (has_media(fr)) {
      trace_ray_media(T_dir_pdf);
    } else {
      rays_total++;
      trace_ray(fr, origin, direction, POS_INF, isect, false, counters);
    }
    if (T_dir_pdf <= 0 || all_le0(beta)) {
      beta = V3(0.0f);
      return;
    }
    if (has_media(fr)) {  // path.hlsli:1009-1011 (T_dir_pdf = 1 without media)
      beta = beta / T_dir_pdf;
      if (!fr.flag(STHIP_eDeferShadowRays)) bsdf_pdf *= T_dir_pdf;
      path_pdf *= T_dir_pdf;
    }
    path_length++;
    if (isect.instance_index() == STHIP_INVALID_INSTANCE) {
      G = 1;
      ngdotin = 1;
      return;
    }
    const float dist2 = len_sqr(isect.sd.position - origin);
    if (fr.flag(STHIP_eRayCones)) {  // path.hlsli:1026-1029
      rd_radius += rd_spread * sqrtf(dist2);
      isect.sd.uv_screen_size *= rd_radius;
    }
    G = 1 / dist2;
    if (has_media(fr) && isect.sd.shape_area == 0) {  // a vertex inside a medium, path.hlsli:1035-1036
      ngdotin = 1;
    } else {
      ngdotin = -dot(direction, isect.sd.geometry_normal());
      G *= fabsf(ngdotin);
    }
    path_pdf *= bsdf_pdf * G;  // pdfWtoA, path.hlsli:1042
    path_contrib = path_contrib * G;  // path.hlsli:1043
  }

  // the medium-aware trace_ray, intersection.hlsli:240-285: up to 64 segments between volume boundaries, delta tracking
  // inside the current medium over each segment
  void trace_ray_media(float& T_dir_pdf) {
    v3 o = origin;
    Medium m;
    if (medium != STHIP_INVALID_INSTANCE) m.load(*fr.sc, fr.sc->instances[medium].material_address());
    for (uint32_t steps = 0; steps < 64; steps++) {
      rays_total++;
      const float dt = trace_ray(fr, o, direction, POS_INF, isect, false, counters);
      if (medium != STHIP_INVALID_INSTANCE) {
        const sthip_TransformData& inv = fr.sc->inv_xf[medium];
        v3 dir_pdf = V3(1.0f), nee_pdf = V3(1.0f), scatter_p;
        const bool scattered = m.delta_track(*fr.sc, rng, transform_point(inv, o), transform_vector(inv, direction), dt, beta, dir_pdf, nee_pdf, true, fr.pc.gMaxNullCollisions, scatter_p);
        T_dir_pdf *= average3(dir_pdf);
        T_nee_pdf *= average3(nee_pdf);
        if (scattered && std::isfinite(scatter_p.x) && std::isfinite(scatter_p.y) && std::isfinite(scatter_p.z)) {
          isect.instance_primitive_index = medium | (STHIP_INVALID_PRIMITIVE << 16);
          isect.sd.position = scatter_p;  // grid world space, as upstream stores it
          isect.sd.shape_area = 0;
          break;
        }
      }
      if (isect.instance_index() == STHIP_INVALID_INSTANCE || steps == 63) break;
      const Inst& in = fr.sc->instances[isect.instance_index()];
      if (in.type() != STHIP_INSTANCE_TYPE_VOLUME) break;
      if (isect.sd.flags & STHIP_SHADING_FLAG_FRONT_FACE) {  // entering the volume
        medium = isect.instance_index();
        m.load(*fr.sc, in.material_address());
        o = ray_offset(isect.sd.position, -isect.sd.geometry_normal());
      } else {  // leaving it
        medium = STHIP_INVALID_INSTANCE;
        o = ray_offset(isect.sd.position, isect.sd.geometry_normal());
      }
    }
  }

  // trace_visibility_ray with media, intersection.hlsli:192-239: surfaces block, volume boundaries are crossed, the
  // medium in between attenuates (delta tracking that cannot scatter)
  // max_segments: the inline walks (no eDeferShadowRays) are pinned to at most 64 closest-hit queries — upstream's loop has no
  // bound, its path walk has this one (intersection.hlsli:247); the product must not let a thread spin on a degenerate boundary
  void trace_visibility_media(Rng& r, v3 o, v3 d, float t_max, uint32_t cur_medium, v3& contribution, float& T_dir, float& T_nee, uint32_t max_segments = 0xFFFFFFFFu) {
    Medium m;
    if (cur_medium != STHIP_INVALID_INSTANCE) m.load(*fr.sc, fr.sc->instances[cur_medium].material_address());
    for (uint32_t segment = 0; t_max > 1e-6f && segment < max_segments; segment++) {
      IntersectionVertex sh = isect;  // a scratch vertex (its untouched fields do not matter)
      rays_total++;
      const float dt = trace_ray(fr, o, d, t_max, sh, false, counters);
      if (!std::isinf(t_max)) t_max -= dt;
      if (sh.instance_index() == STHIP_INVALID_INSTANCE) break;
      const Inst& in = fr.sc->instances[sh.instance_index()];
      if (in.type() != STHIP_INSTANCE_TYPE_VOLUME) {  // a surface
        contribution = V3(0.0f);
        T_dir = 0;
        T_nee = 0;
        break;
      }
      if (cur_medium != STHIP_INVALID_INSTANCE) {
        const sthip_TransformData& inv = fr.sc->inv_xf[cur_medium];
        v3 dir_pdf = V3(1.0f), nee_pdf = V3(1.0f), scatter_p;
        m.delta_track(*fr.sc, r, transform_point(inv, o), transform_vector(inv, d), dt, contribution, dir_pdf, nee_pdf, false, fr.pc.gMaxNullCollisions, scatter_p);
        T_dir *= average3(dir_pdf);
        T_nee *= average3(nee_pdf);
      }
      if (sh.sd.flags & STHIP_SHADING_FLAG_FRONT_FACE) {
        cur_medium = sh.instance_index();
        m.load(*fr.sc, in.material_address());
        o = ray_offset(sh.sd.position, -sh.sd.geometry_normal());
      } else {
        cur_medium = STHIP_INVALID_INSTANCE;
        o = ray_offset(sh.sd.position, sh.sd.geometry_normal());
      }
    }
  }

  // path.hlsli:847-894
  void eval_emission(v3 Le) {
    if (all_le0(Le)) return;
    const v3 contrib = beta * Le;
    float cos_theta_light;
    if (isect.instance_index() == STHIP_INVALID_INSTANCE || isect.sd.shape_area == 0) {
      cos_theta_light = 1;  // background
    } else {
      cos_theta_light = -dot(isect.sd.geometry_normal(), direction);
      if (cos_theta_light < 0) return;
    }
    bool area_measure;
    float light_pdfA = point_on_light_pdf(fr, isect, direction, area_measure);
    if (!area_measure) light_pdfA = light_pdfA * G;  // pdfWtoA
    if (!fr.flag(STHIP_eDeferShadowRays)) light_pdfA *= T_nee_pdf;  // path.hlsli:866 (1 without media)
    float weight = 1;
    if (path_length > 2) {
      if (fr.bdpt()) {  // path.hlsli:870-880
        if (fr.flag(STHIP_eMIS)) {
          const float p_rev_k = cosine_hemisphere_pdfW(fabsf(cos_theta_light)) * (fabsf(prev_cos_out) / len_sqr(origin - isect.sd.position));
          if (fr.flag(STHIP_eConnectToLightPaths)) {
            const float dE_k = connection_dVC(dVC, p_rev_k, bsdf_pdf * G, prev_specular);
            weight = 1 / (1 + dE_k * pow2(light_pdfA));
          } else
            weight = prev_specular ? 0.0f : mis2(fr, path_pdf, path_pdf_rev * p_rev_k * light_pdfA);
        } else {
          weight = path_weight(path_length, 0);
        }
      } else if (fr.flag(STHIP_eNEE))
        weight = fr.flag(STHIP_eNEEReservoirs) ? 0.5f : mis2(fr, bsdf_pdf * G, light_pdfA);  // path.hlsli:881-886
    }
    if (dbg && fr.debug(STHIP_DEBUG_VIEW_TRACE_CONTRIBUTION)) debug_add(contrib);  // path.hlsli:890-891
    accumulate_contribution(contrib, weight, 0);
  }

  // path.hlsli:829-845, non-coherent form (eCoherentRR is wave-scope and implementation-defined)
  // eCoherentRR (the reference's default, BDPT.cpp:58): p = WaveActiveMax(p), the decision is the first lane's
  // (path.hlsli:831,836) — over the lanes of the 8x4 workgroup that execute this call in this iteration of the
  // vertex loop (a wave of 32 on the hardware the reference targets; all live lanes of a group are at the same
  // path_length). The group's (p_max, decision) per path length come from the driver (orc_render_window), which finds
  // them by replaying the group's paths: a call that finds no decision for its path length records this lane's p and the
  // random number it would draw, and ends the path. With media the non-coherent form stays (walks through volumes break
  // the lockstep).
  struct GroupRR {
    bool valid = false;
    float p_max = 0;
    bool kill = false;
  };
  const GroupRR* group_rr = nullptr;  // indexed by path_length (null: not run as part of a group -> the non-coherent form)
  int rr_missing_length = -1;         // the path length at which this path needed a group decision that is not known yet
  float rr_missing_p = 0, rr_missing_rnd = 0;
  // eCoherentSampling (path.hlsli:317-318,378-387,688,703): an index drawn at random becomes WaveReadLaneFirst(index) +
  // WaveGetLaneIndex(), i.e. the 32 lanes of a workgroup read 32 CONSECUTIVE presampled lights / cached light vertices. The
  // wave is given the same defined meaning as for eCoherentRR: the 8x4 workgroup, lane = (y & 3) * 8 + (x & 7), lane count
  // 32, "first lane" = the lowest lane whose path executes that statement at that path length. Two sites per vertex, in
  // execution order: the NEE index (connect_light's tile index with ePresampleLights, or connect_light_reservoir's), then
  // connect_lvc's. Same replay protocol as the roulette: a path that finds no group value reports its own draw and stops.
  struct GroupValue {
    bool valid = false;
    uint32_t value = 0;
  };
  const GroupValue* group_nee = nullptr;  // indexed by path_length; null: not run as part of a group -> own draws
  const GroupValue* group_lvc = nullptr;
  uint32_t lane = 0;
  int cs_missing_site = -1, cs_missing_length = -1;  // site: 0 = NEE index, 1 = connect_lvc's index
  uint32_t cs_missing_value = 0;
  bool aborted = false;  // the replay protocol stopped this path in the middle of a vertex
  // own_draw -> the index this lane uses. False: the group's value is not known yet (reported; the path stops).
  bool coherent_index(int site, uint32_t& index) {
    const GroupValue* g = site == 0 ? group_nee : group_lvc;
    if (!fr.flag(STHIP_eCoherentSampling) || !g) return true;
    if (!g[path_length].valid) {
      cs_missing_site = site;
      cs_missing_length = (int)path_length;
      cs_missing_value = index;
      aborted = true;
      beta = V3(0.0f);
      return false;
    }
    index = g[path_length].value + lane;
    return true;
  }
  bool coherent_sampling() const { return fr.flag(STHIP_eCoherentSampling) && group_nee != nullptr; }
  bool russian_roulette() {
    float p = luminance(beta) / eta_scale * 0.95f;
    const bool coherent = fr.flag(STHIP_eCoherentRR) && !has_media(fr) && group_rr;
    bool kill = false;
    if (coherent) {
      if (!group_rr[path_length].valid) {  // replay protocol: report and stop; the driver works the decision out and runs the group again
        rr_missing_length = (int)path_length;
        rr_missing_p = p;
        Rng peek = rng;
        rr_missing_rnd = peek.next_float();
        beta = V3(0.0f);
        return false;
      }
      p = group_rr[path_length].p_max;  // >= this lane's own p
      kill = group_rr[path_length].kill;
    }
    if (p >= 1) return true;
    const bool v = rng.next_float() > p;  // every lane draws; with eCoherentRR only the first lane's comparison counts
    if (coherent ? kill : v) return false;
    beta = beta / p;
    path_pdf *= p;  // path.hlsli:842
    return true;
  }

  // one light candidate: DirectLightSample's two constructors (path.hlsli:179-201) in front of setup()
  struct LightCandidate {
    v3 Le, ray_direction;
    float pdfA, ray_distance, G;
    v3 position;                      // DirectLightSample::p.position / packed_geometry_normal: what a reservoir stores
    uint32_t packed_geometry_normal;
  };
  // DirectLightSample(_isect, PresampledLightPoint), path.hlsli:184-201 (surface points: environment samples are not reused)
  LightCandidate light_candidate_from(const PresampledLightPoint& lp) const {
    LightCandidate c;
    c.Le = lp.Le;
    c.pdfA = lp.pdfA;
    c.position = lp.position;
    c.packed_geometry_normal = lp.packed_geometry_normal;
    c.ray_direction = lp.position - isect.sd.position;
    const float dist2 = len_sqr(c.ray_direction);
    c.ray_distance = sqrtf(dist2);
    c.ray_direction = c.ray_direction / c.ray_distance;
    c.G = fabsf(dot(c.ray_direction, unpack_normal_octahedron(lp.packed_geometry_normal))) / dist2;
    return c;
  }
  // hashgrid_cell_size, hashgrid.hlsli:4-14
  float hashgrid_cell_size(v3 pos) const {
    if (fr.pc.gHashGridBucketPixelRadius < 0) return fr.pc.gHashGridMinBucketRadius;
    const sthip_TransformData& t = fr.fd.gViewTransforms[0];
    const float dist = length(pos - V3(t.m[0][3], t.m[1][3], t.m[2][3]));
    const sthip_ViewData& view = fr.fd.gViews[0];
    const float ex = (float)(view.image_max[0] - view.image_min[0]), ey = (float)(view.image_max[1] - view.image_min[1]);
    const float step = dist * det_tanf(fr.pc.gHashGridBucketPixelRadius * view.projection.vertical_fov * fmaxf(1 / ey, ey / pow2(ex)));
    const uint32_t level = std::min(HashGridTable::f2u_sat(det_log2f(step / fr.pc.gHashGridMinBucketRadius)), 31u);
    return fr.pc.gHashGridMinBucketRadius * (float)(int32_t)(1u << level);
  }
  // presampled = true: `ti` picks the tile's point; else four randoms are drawn (sample_Le, path.hlsli:141-164)
  LightCandidate light_candidate(bool presampled, uint32_t ti) {
    LightCandidate c;
    if (presampled) {
      // path.hlsli:313-320: one of the tile's presampled points; DirectLightSample(_isect, PresampledLightPoint) :184-201
      const uint32_t tile_size = fr.pc.gLightPresampleTileSize;
      const uint32_t tile_offset = ((path_index() / tile_size) % fr.pc.gLightPresampleTileCount) * tile_size;
      const PresampledLightPoint& lp = fr.presampled[seed - fr.seed_begin][tile_offset + ti % tile_size];
      c.Le = lp.Le;
      c.pdfA = lp.pdfA;
      c.position = lp.position;
      c.packed_geometry_normal = lp.packed_geometry_normal;
      c.ray_direction = lp.position - isect.sd.position;
      const float dist2 = len_sqr(c.ray_direction);
      c.ray_distance = sqrtf(dist2);
      c.ray_direction = c.ray_direction / c.ray_distance;
      c.G = fabsf(dot(c.ray_direction, unpack_normal_octahedron(lp.packed_geometry_normal))) / dist2;
    } else {
      float rnd[4];
      rnd[0] = rng.next_float();
      rnd[1] = rng.next_float();
      rnd[2] = rng.next_float();
      rnd[3] = rng.next_float();
      LightSampleRecord ls;
      sample_point_on_light(fr, ls, rnd, isect.sd.position);
      c.Le = ls.radiance;
      c.pdfA = ls.pdf;
      c.position = ls.position;
      c.packed_geometry_normal = pack_normal_octahedron(ls.normal);
      c.ray_direction = ls.to_light;
      c.ray_distance = ls.dist;
      if (ls.is_environment) {  // sample_Le, path.hlsli:156-162
        c.G = 1;
      } else {
        c.G = fabsf(dot(ls.to_light, ls.normal)) / pow2(ls.dist);
        if (!ls.pdf_area_measure) c.pdfA = c.pdfA * c.G;
      }
    }
    return c;
  }

  // path.hlsli:311-366 + sample_Le :141-164 + DirectLightSample :166-222
  void connect_light(const DisneyMaterial& m) {
    if (fr.flag(STHIP_eNEEReservoirs)) {
      connect_light_reservoir(m);
      return;
    }
    const bool presampled = fr.flag(STHIP_ePresampleLights);
    uint32_t ti = 0;
    if (presampled) {
      ti = rng.next_uint() % fr.pc.gLightPresampleTileSize;  // :316
      if (!coherent_index(0, ti)) return;                  // :317-318 (light_candidate takes it modulo the tile size)
    }
    const LightCandidate cand = light_candidate(presampled, ti);
    v3 Le = cand.Le, ray_direction = cand.ray_direction;
    float pdfA = cand.pdfA, ray_distance = cand.ray_distance, cG = cand.G;
    // setup()
    v3 ray_origin = isect.sd.position;
    const v3 local_to_light = normalize(isect.sd.to_local(ray_direction));
    const v3 geometry_normal = isect.sd.geometry_normal();
    const float ngdotout = dot(geometry_normal, ray_direction);
    ray_origin = ray_offset(ray_origin, ngdotout > 0 ? geometry_normal : -geometry_normal);
    ray_distance = ray_distance * 0.999f;

    if (all_le0(Le) && pdfA < 1e-6f) return;
    MaterialEvalRecord ev;
    m.eval(ev, local_dir_in, local_to_light, false);
    float pdfA_fwd = ev.pdf_fwd * cG;
    if (pdfA_fwd < 1e-6f) return;
    const bool defer = fr.flag(STHIP_eDeferShadowRays);
    if (!defer) {  // path.hlsli:329-332: with media the walk attenuates Le, scales both pdfs and draws from the path's own stream
      if (has_media(fr)) trace_visibility_media(rng, ray_origin, ray_direction, ray_distance, medium, Le, pdfA_fwd, pdfA, 64);
      else if (occluded(ray_origin, ray_direction, ray_distance)) Le = V3(0.0f);
      if (all_le0(Le)) return;
    }
    cG *= shading_normal_correction(local_dir_in.z, local_to_light.z, ngdotin, ngdotout, dot(geometry_normal, isect.sd.shading_normal()), fr.flag(STHIP_eShadingNormalShadowFix));
    const v3 contrib = Le * ev.f * cG / pdfA;
    if (all_le0(contrib)) return;
    float weight = 1;
    if (fr.bdpt()) {  // BDPT MIS, path.hlsli:341-351
      if (fr.flag(STHIP_eMIS)) {
        const float emission_pdfA = cosine_hemisphere_pdfW(ngdotout) * (ngdotout / pow2(ray_distance));  // setup(), :219 (after the distance epsilon)
        const float dL = connection_dVC(1 / pdfA, emission_pdfA, pdfA, false);
        const float G_rev = prev_cos_out / len_sqr(origin - isect.sd.position);
        const float dE = connection_dVC(dVC, ev.pdf_rev * G_rev, bsdf_pdf * G, prev_specular);
        weight = 1 / (1 + dE * pow2(emission_pdfA) + dL * pow2(pdfA_fwd));
      } else {
        weight = path_weight(path_length, 1);
      }
    } else if (fr.flag(STHIP_eSampleBSDFs))
      weight = mis2(fr, pdfA, pdfA_fwd);
    if (defer) {
      const v3 c = beta * contrib * weight;
      if (diffuse_vertices >= 1 && diffuse_vertices <= max_shadow) {
        sthip_ShadowRayData& rd = shadow_rays[diffuse_vertices - 1];
        rd.contribution[0] = c.x;
        rd.contribution[1] = c.y;
        rd.contribution[2] = c.z;
        rd.rng_offset = rng.v[3];
        rd.ray_origin[0] = ray_origin.x;
        rd.ray_origin[1] = ray_origin.y;
        rd.ray_origin[2] = ray_origin.z;
        rd.medium = medium;
        rd.ray_direction[0] = ray_direction.x;
        rd.ray_direction[1] = ray_direction.y;
        rd.ray_direction[2] = ray_direction.z;
        rd.ray_distance = ray_distance;
      }
    } else {
      accumulate_contribution(beta * contrib, weight, 1);  // path.hlsli:365
    }
  }

  // connect_light_reservoir, path.hlsli:368-486, without spatial reuse (eNEEReservoirReuse is not restated):
  // resampled importance sampling over gReservoirM candidates, target = luminance(Le) G |cos|
  void connect_light_reservoir(const DisneyMaterial& m) {
    const bool presampled = fr.flag(STHIP_ePresampleLights);
    const v3 geometry_normal = isect.sd.geometry_normal();
    LightCandidate c;
    memset(&c, 0, sizeof(c));
    v3 c_local_to_light = V3(0.0f);
    float total_weight = 0, r_target_pdf = 0;
    uint32_t M = 0;
    uint32_t ti = rng.next_uint();  // :378 (drawn in either mode)
    const bool coherent = presampled && coherent_sampling();  // (without presampled lights the index is never used)
    if (coherent && !coherent_index(0, ti)) return;  // :379-380
    for (uint32_t i = 0; i < fr.pc.gReservoirM; i++) {
      if (presampled && !coherent) ti = rng.next_uint();  // :385
      const LightCandidate c_i = light_candidate(presampled, ti);
      if (coherent) ti += 32;  // :387, WaveGetLaneCount()
      if (c_i.pdfA <= 0 || all_le0(c_i.Le)) continue;
      const v3 local_to_light = normalize(isect.sd.to_local(c_i.ray_direction));  // setup(), :209
      const float target_pdf_i = luminance(c_i.Le) * c_i.G * fabsf(local_to_light.z);
      const float w = target_pdf_i / c_i.pdfA;
      M++;  // Reservoir::update, reservoir.h:22-26
      total_weight += w;
      if (rng.next_float() * total_weight <= w) {
        r_target_pdf = target_pdf_i;
        c = c_i;
        c_local_to_light = local_to_light;
      }
    }
    // spatial reuse through the previous frame's hash grid, path.hlsli:402-428
    const bool reuse = fr.flag(STHIP_eNEEReservoirReuse);
    v3 t = V3(0.0f), b = V3(0.0f);
    float cell_size = 0;
    auto jittered = [&]() {  // the position a lookup / an append hashes: jittered in the tangent plane with eHashGridJitter
      const float phi = rng.next_float() * 2 * DET_PI;
      if (!fr.flag(STHIP_eHashGridJitter)) return isect.sd.position;
      const float radius = cell_size * rng.next_float();
      float sn, cs;
      det_sincosf(phi, &sn, &cs);
      return isect.sd.position + (t * cs + b * sn) * radius;
    };
    if (reuse) {
      make_orthonormal(geometry_normal, t, b);
      cell_size = hashgrid_cell_size(isect.sd.position);
      if (fr.prev_nee_grid && fr.pc.gReservoirSpatialM > 0) {
        const v3 at = jittered();
        const uint32_t bucket = fr.prev_nee_grid->table.find(at, cell_size);
        if (bucket != 0xFFFFFFFFu) {
          const uint32_t bucket_start = fr.prev_nee_grid->table.indices[bucket], bucket_size = fr.prev_nee_grid->table.counters[bucket];
          uint32_t Msum = M;
          for (uint32_t i = 0; i < fr.pc.gReservoirSpatialM; i++) {
            const NEEReservoir& prev = fr.prev_nee_grid->data[bucket_start + rng.next_uint() % bucket_size];
            const LightCandidate c_i = light_candidate_from(prev.y);
            if (c_i.pdfA <= 0 || all_le0(c_i.Le)) continue;
            Msum += prev.r.M;
            const v3 local_to_light = normalize(isect.sd.to_local(c_i.ray_direction));
            const float target_pdf_i = luminance(c_i.Le) * c_i.G * fabsf(local_to_light.z);
            const float w = target_pdf_i * prev.W * (float)prev.r.M;
            M++;
            total_weight += w;
            if (rng.next_float() * total_weight <= w) {
              r_target_pdf = target_pdf_i;
              c = c_i;
              c_local_to_light = local_to_light;
            }
          }
          M = Msum;
        }
      }
    }
    const float W = (r_target_pdf > 0 && M > 0) ? total_weight / ((float)M * r_target_pdf) : 0;  // reservoir.h:8-13
    if (W <= 1e-6f || W != W) return;
    if (reuse) {  // path.hlsli:434-439: this vertex's reservoir goes into the grid the NEXT frame looks up
      const v3 at = jittered();
      const size_t k = (size_t)path_index() * fr.pc.gMaxDiffuseVertices + (diffuse_vertices - 1);
      if (fr.nee_appends && diffuse_vertices >= 1 && diffuse_vertices <= fr.pc.gMaxDiffuseVertices) {
        HashGridOf<NEEReservoir>::Append& a = fr.nee_appends[k];
        a.pos = at;
        a.cell_size = cell_size;
        a.y.r.total_weight = total_weight;
        a.y.r.M = std::min(M, fr.pc.gReservoirMaxM);
        a.y.packed_geometry_normal = isect.sd.packed_geometry_normal;
        a.y.W = W;
        a.y.y.position = c.position;
        a.y.y.packed_geometry_normal = c.packed_geometry_normal;
        a.y.y.Le = c.Le;
        a.y.y.pdfA = c.pdfA;
        fr.nee_append_valid[k] = 1;
      }
    }
    // setup() of the chosen candidate
    const float ngdotout = dot(geometry_normal, c.ray_direction);
    const v3 ray_origin = ray_offset(isect.sd.position, ngdotout > 0 ? geometry_normal : -geometry_normal);
    const float ray_distance = c.ray_distance * 0.999f;
    MaterialEvalRecord ev;
    m.eval(ev, local_dir_in, c_local_to_light, false);
    float cG = c.G * shading_normal_correction(local_dir_in.z, c_local_to_light.z, ngdotin, ngdotout, dot(geometry_normal, isect.sd.shading_normal()), fr.flag(STHIP_eShadingNormalShadowFix));
    v3 contrib = c.Le * ev.f * cG * W;
    if (all_le0(contrib) || c.pdfA < 1e-6f) return;
    float weight = 1;
    if (fr.bdpt()) {  // path.hlsli:458-465 (no eMIS test upstream; c.G already carries the shading-normal term)
      const float emission_pdfA = cosine_hemisphere_pdfW(ngdotout) * (ngdotout / pow2(ray_distance));  // setup(), :219
      const float dL = connection_dVC(W, emission_pdfA, 1 / W, false);
      const float G_rev = prev_cos_out / len_sqr(origin - isect.sd.position);
      const float dE = connection_dVC(dVC, ev.pdf_rev * G_rev, bsdf_pdf * G, prev_specular);
      weight = 1 / (1 + dE * pow2(emission_pdfA) + dL * pow2(ev.pdf_fwd * cG));
    } else if (fr.flag(STHIP_eSampleBSDFs))
      weight = 1 - 0.5f;  // DirectLightSample::reservoir_bsdf_mis, path.hlsli:175-177
    if (fr.flag(STHIP_eDeferShadowRays)) {
      const v3 cc = beta * contrib * weight;
      if (diffuse_vertices >= 1 && diffuse_vertices <= max_shadow) {
        sthip_ShadowRayData& rd = shadow_rays[diffuse_vertices - 1];
        rd.contribution[0] = cc.x;
        rd.contribution[1] = cc.y;
        rd.contribution[2] = cc.z;
        rd.rng_offset = rng.v[3];
        rd.ray_origin[0] = ray_origin.x;
        rd.ray_origin[1] = ray_origin.y;
        rd.ray_origin[2] = ray_origin.z;
        rd.medium = medium;
        rd.ray_direction[0] = c.ray_direction.x;
        rd.ray_direction[1] = c.ray_direction.y;
        rd.ray_direction[2] = c.ray_direction.z;
        rd.ray_distance = ray_distance;
      }
    } else {
      if (has_media(fr)) {  // path.hlsli:474-479: the walk attenuates the contribution itself, in the path's own stream
        float dir_pdf = 1, nee_pdf = 1;
        trace_visibility_media(rng, ray_origin, c.ray_direction, ray_distance, medium, contrib, dir_pdf, nee_pdf, 64);
        if (nee_pdf <= 0) return;
        contrib = contrib / nee_pdf;
        if (all_le0(contrib)) return;
      } else if (occluded(ray_origin, c.ray_direction, ray_distance))
        return;
      if (dbg && fr.debug(STHIP_DEBUG_RESERVOIR_WEIGHT)) debug_add(V3(W));  // path.hlsli:482-483
      accumulate_contribution(beta * contrib, weight, 1);                  // :485
    }
  }

  // connect_light_reservoir at a vertex inside a medium (the same function upstream, BSDF = Medium): setup()'s medium branch
  // (path.hlsli:207-212) makes local_to_light the WORLD direction, so the target is luminance(Le) G |direction.z|; no ray
  // offset, no distance epsilon, no shading-normal term (:446). Spatial reuse as at a surface (:402-439); the geometry normal
  // upstream takes the jitter's tangent plane from, and stores with the reservoir, is the stale one of the last surface query
  // at a vertex inside a medium: pinned to the packed value 0, as the first-hit normals of such a vertex are.
  void connect_light_reservoir_medium(const Medium& m) {
    const bool presampled = fr.flag(STHIP_ePresampleLights);
    LightCandidate c;
    memset(&c, 0, sizeof(c));
    float total_weight = 0, r_target_pdf = 0;
    uint32_t M = 0;
    uint32_t ti = rng.next_uint();  // :378 (drawn in either mode)
    for (uint32_t i = 0; i < fr.pc.gReservoirM; i++) {
      if (presampled) ti = rng.next_uint();  // :385
      const LightCandidate c_i = light_candidate(presampled, ti);
      if (c_i.pdfA <= 0 || all_le0(c_i.Le)) continue;
      const float target_pdf_i = luminance(c_i.Le) * c_i.G * fabsf(c_i.ray_direction.z);
      const float w = target_pdf_i / c_i.pdfA;
      M++;
      total_weight += w;
      if (rng.next_float() * total_weight <= w) {
        r_target_pdf = target_pdf_i;
        c = c_i;
      }
    }
    const bool reuse = fr.flag(STHIP_eNEEReservoirReuse);
    v3 t = V3(0.0f), b = V3(0.0f);
    float cell_size = 0;
    auto jittered = [&]() {
      const float phi = rng.next_float() * 2 * DET_PI;
      if (!fr.flag(STHIP_eHashGridJitter)) return isect.sd.position;
      const float radius = cell_size * rng.next_float();
      float sn, cs;
      det_sincosf(phi, &sn, &cs);
      return isect.sd.position + (t * cs + b * sn) * radius;
    };
    if (reuse) {
      make_orthonormal(unpack_normal_octahedron(0), t, b);
      cell_size = hashgrid_cell_size(isect.sd.position);
      if (fr.prev_nee_grid && fr.pc.gReservoirSpatialM > 0) {
        const v3 at = jittered();
        const uint32_t bucket = fr.prev_nee_grid->table.find(at, cell_size);
        if (bucket != 0xFFFFFFFFu) {
          const uint32_t bucket_start = fr.prev_nee_grid->table.indices[bucket], bucket_size = fr.prev_nee_grid->table.counters[bucket];
          uint32_t Msum = M;
          for (uint32_t i = 0; i < fr.pc.gReservoirSpatialM; i++) {
            const NEEReservoir& prev = fr.prev_nee_grid->data[bucket_start + rng.next_uint() % bucket_size];
            const LightCandidate c_i = light_candidate_from(prev.y);
            if (c_i.pdfA <= 0 || all_le0(c_i.Le)) continue;
            Msum += prev.r.M;
            const float target_pdf_i = luminance(c_i.Le) * c_i.G * fabsf(c_i.ray_direction.z);
            const float w = target_pdf_i * prev.W * (float)prev.r.M;
            M++;
            total_weight += w;
            if (rng.next_float() * total_weight <= w) {
              r_target_pdf = target_pdf_i;
              c = c_i;
            }
          }
          M = Msum;
        }
      }
    }
    const float W = (r_target_pdf > 0 && M > 0) ? total_weight / ((float)M * r_target_pdf) : 0;
    if (W <= 1e-6f || W != W) return;
    if (reuse) {  // path.hlsli:434-439
      const v3 at = jittered();
      const size_t k = (size_t)path_index() * fr.pc.gMaxDiffuseVertices + (diffuse_vertices - 1);
      if (fr.nee_appends && diffuse_vertices >= 1 && diffuse_vertices <= fr.pc.gMaxDiffuseVertices) {
        HashGridOf<NEEReservoir>::Append& a = fr.nee_appends[k];
        a.pos = at;
        a.cell_size = cell_size;
        a.y.r.total_weight = total_weight;
        a.y.r.M = std::min(M, fr.pc.gReservoirMaxM);
        a.y.packed_geometry_normal = 0;
        a.y.W = W;
        a.y.y.position = c.position;
        a.y.y.packed_geometry_normal = c.packed_geometry_normal;
        a.y.y.Le = c.Le;
        a.y.y.pdfA = c.pdfA;
        fr.nee_append_valid[k] = 1;
      }
    }
    const float f = m.phase(local_dir_in, c.ray_direction);
    v3 contrib = c.Le * f * c.G * W;
    if (all_le0(contrib) || c.pdfA < 1e-6f) return;
    float weight = 1;
    if (fr.bdpt()) {  // path.hlsli:458-465, emission_pdfA = 0 at a medium vertex
      const float dL = connection_dVC(W, 0.0f, 1 / W, false);
      const float G_rev = prev_cos_out / len_sqr(origin - isect.sd.position);
      const float dE = connection_dVC(dVC, f * G_rev, bsdf_pdf * G, prev_specular);
      weight = 1 / (1 + dE * pow2(0.0f) + dL * pow2(f * c.G));
    } else if (fr.flag(STHIP_eSampleBSDFs))
      weight = 1 - 0.5f;
    if (fr.flag(STHIP_eDeferShadowRays)) {
      const v3 cc = beta * contrib * weight;
      if (diffuse_vertices >= 1 && diffuse_vertices <= max_shadow) {
        sthip_ShadowRayData& rd = shadow_rays[diffuse_vertices - 1];
        rd.contribution[0] = cc.x;
        rd.contribution[1] = cc.y;
        rd.contribution[2] = cc.z;
        rd.rng_offset = rng.v[3];
        rd.ray_origin[0] = isect.sd.position.x;
        rd.ray_origin[1] = isect.sd.position.y;
        rd.ray_origin[2] = isect.sd.position.z;
        rd.medium = medium;
        rd.ray_direction[0] = c.ray_direction.x;
        rd.ray_direction[1] = c.ray_direction.y;
        rd.ray_direction[2] = c.ray_direction.z;
        rd.ray_distance = c.ray_distance;
      }
    } else {
      float dir_pdf = 1, nee_pdf = 1;
      trace_visibility_media(rng, isect.sd.position, c.ray_direction, c.ray_distance, medium, contrib, dir_pdf, nee_pdf, 64);
      if (nee_pdf <= 0) return;
      contrib = contrib / nee_pdf;
      if (all_le0(contrib)) return;
      if (dbg && fr.debug(STHIP_DEBUG_RESERVOIR_WEIGHT)) debug_add(V3(W));
      accumulate_contribution(beta * contrib, weight, 1);
    }
  }

  // intersection.hlsli:192-239 without media
  bool occluded(v3 o, v3 d, float t_max) {
    if (!(t_max > 1e-6f)) return false;
    rays_total++;
    IntersectionVertex tmp;
    trace_ray(fr, o, d, t_max, tmp, true, counters);
    return tmp.instance_index() != STHIP_INVALID_INSTANCE;
  }

  // path.hlsli:898-952
  bool sample_direction(const DisneyMaterial& m) {
    const float r0 = rng.next_float(), r1 = rng.next_float(), r2 = rng.next_float();
    MaterialSampleRecord ms;
    path_contrib = path_contrib * m.sample(ms, V3(r0, r1, r2), local_dir_in, beta, trace_light);  // path.hlsli:901
    if (ms.pdf_fwd < 1e-6f) {
      beta = V3(0.0f);
      return false;
    }
    if (ms.eta != 0) eta_scale /= pow2(ms.eta);
    if (fr.flag(STHIP_eRayCones)) {  // path.hlsli:911-916, RayDifferential::reflect / refract :232-243
      float spec_spread = rd_spread + 2 * isect.sd.mean_curvature * rd_radius;
      if (ms.eta != 0) spec_spread = spec_spread / ms.eta;
      rd_spread = fmaxf(0.0f, lerpf(spec_spread, 0.2f, ms.roughness));
    }
    {  // MIS quantities, path.hlsli:919-925
      const float G_rev = prev_cos_out / len_sqr(origin - isect.sd.position);
      if (trace_light || path_length > 2) path_pdf_rev *= ms.pdf_rev * G_rev;
      dVC = connection_dVC(dVC, ms.pdf_rev * G_rev, bsdf_pdf * G, m.is_specular());
      prev_specular = m.is_specular();
    }
    bsdf_pdf = ms.pdf_fwd;
    const float ndotout = ms.dir_out.z;
    ms.dir_out = normalize(isect.sd.to_world(ms.dir_out));
    const v3 geometry_normal = isect.sd.geometry_normal();
    const float ngdotout = dot(geometry_normal, ms.dir_out);
    origin = ray_offset(isect.sd.position, ngdotout > 0 ? geometry_normal : -geometry_normal);
    beta = beta * shading_normal_correction(local_dir_in.z, ndotout, ngdotin, ngdotout, dot(geometry_normal, isect.sd.shading_normal()), fr.flag(STHIP_eShadingNormalShadowFix), trace_light);
    prev_cos_out = ngdotout;
    if (all_le0(beta)) return false;
    direction = ms.dir_out;
    if (dbg && fr.debug(STHIP_DEBUG_DIR_OUT)) debug_set(direction * .5f + V3(.5f));  // path.hlsli:950
    return true;
  }

  // path.hlsli:955-998
  bool next_vertex_m(const DisneyMaterial& m) {
    if (!trace_light && path_length > 2) eval_emission(m.Le());
    if (!m.can_eval() || path_length >= fr.pc.gMaxPathVertices) return false;
    if (!m.is_specular()) {
      diffuse_vertices++;
      if (diffuse_vertices > fr.pc.gMaxDiffuseVertices) return false;
      if (trace_light && fr.flag(STHIP_eConnectToLightPaths) && path_length + 2 <= fr.pc.gMaxPathVertices && diffuse_vertices < fr.pc.gMaxDiffuseVertices)
        store_light_vertex();
      if (trace_light) {
        if (fr.flag(STHIP_eConnectToViews)) connect_view(m);
      } else {
        if (path_length >= fr.pc.gMinPathVertices)
          if (!russian_roulette()) return false;
        if (fr.flag(STHIP_eNEE)) connect_light(m);
        if (aborted) return false;
        if (fr.flag(STHIP_eConnectToLightPaths)) {
          if (fr.flag(STHIP_eLVC))
            connect_lvc(m);
          else
            connect_light_subpath(m);
        }
        if (aborted) return false;
      }
    }
    if (fr.flag(STHIP_eSampleBSDFs) || trace_light) return sample_direction(m);
    return false;
  }

  // next_vertex(BSDF) for a Medium, path.hlsli:955-998: no emission; stop tests; RR; NEE with the phase function; a
  // phase-function bounce
  bool next_vertex_medium(const Medium& m) {
    if (!m.can_eval() || path_length >= fr.pc.gMaxPathVertices) return false;
    if (!m.is_specular()) {
      diffuse_vertices++;
      if (diffuse_vertices > fr.pc.gMaxDiffuseVertices) return false;
      if (trace_light && fr.flag(STHIP_eConnectToLightPaths) && path_length + 2 <= fr.pc.gMaxPathVertices && diffuse_vertices < fr.pc.gMaxDiffuseVertices)
        store_light_vertex();
      if (trace_light) {
        if (fr.flag(STHIP_eConnectToViews)) connect_view_medium(m);
      } else {
        if (path_length >= fr.pc.gMinPathVertices)
          if (!russian_roulette()) return false;
        if (fr.flag(STHIP_eNEE)) {
          if (fr.flag(STHIP_eNEEReservoirs)) connect_light_reservoir_medium(m);
          else connect_light_medium(m);
        }
        if (fr.flag(STHIP_eConnectToLightPaths)) {
          if (fr.flag(STHIP_eLVC)) connect_lvc_any(nullptr, &m);
          else connect_light_subpath_any(nullptr, &m);
        }
      }
    }
    if (!fr.flag(STHIP_eSampleBSDFs) && !trace_light) return false;
    // sample_direction, path.hlsli:898-952, medium branch
    const float r0 = rng.next_float(), r1 = rng.next_float(), r2 = rng.next_float();
    (void)r2;
    float pdf, roughness;
    const v3 dir_out = m.sample(r0, r1, local_dir_in, pdf, roughness);
    path_contrib = path_contrib * pdf;  // path.hlsli:901 (Medium::sample returns the phase value)
    if (pdf < 1e-6f) {
      beta = V3(0.0f);
      return false;
    }
    // eta = -1: eta_scale /= 1. Ray cones: RayDifferential::refract with eta = -1 and the (stale) mean curvature of the
    // last surface hit; pinned to mean_curvature = 0 for a medium vertex
    if (fr.flag(STHIP_eRayCones)) rd_spread = fmaxf(0.0f, lerpf((rd_spread + 2 * 0.0f * rd_radius) / -1.0f, 0.2f, roughness));
    {
      const float G_rev = prev_cos_out / len_sqr(origin - isect.sd.position);
      if (trace_light || path_length > 2) path_pdf_rev *= pdf * G_rev;
      dVC = connection_dVC(dVC, pdf * G_rev, bsdf_pdf * G, m.is_specular());
      prev_specular = m.is_specular();
    }
    bsdf_pdf = pdf;
    origin = isect.sd.position;
    prev_cos_out = 1;
    direction = dir_out;
    if (dbg && fr.debug(STHIP_DEBUG_DIR_OUT)) debug_set(direction * .5f + V3(.5f));  // path.hlsli:950
    return true;
  }

  // connect_light at a medium vertex, path.hlsli:311-366 with DirectLightSample::setup's medium branch (:207-212):
  // no ray offset, no distance epsilon, no shading-normal terms; the phase function is f and both pdfs
  void connect_light_medium(const Medium& m) {
    const bool presampled = fr.flag(STHIP_ePresampleLights);
    const LightCandidate cand = light_candidate(presampled, presampled ? rng.next_uint() : 0u);
    if (all_le0(cand.Le) && cand.pdfA < 1e-6f) return;
    const float f = m.phase(local_dir_in, cand.ray_direction);
    float pdfA_fwd = f * cand.G;
    if (pdfA_fwd < 1e-6f) return;
    v3 Le = cand.Le;
    float pdfA = cand.pdfA;
    const bool defer = fr.flag(STHIP_eDeferShadowRays);
    if (!defer) {  // path.hlsli:329-332 (a medium vertex: no ray offset, no distance epsilon, :207-212)
      trace_visibility_media(rng, isect.sd.position, cand.ray_direction, cand.ray_distance, medium, Le, pdfA_fwd, pdfA, 64);
      if (all_le0(Le)) return;
    }
    const v3 contrib = Le * f * cand.G / pdfA;
    if (all_le0(contrib)) return;
    float weight = 1;
    if (fr.bdpt()) {  // BDPT MIS, path.hlsli:341-351; setup()'s medium branch leaves emission_pdfA = 0 (:211), the phase function is pdf_rev
      if (fr.flag(STHIP_eMIS)) {
        const float dL = connection_dVC(1 / pdfA, 0.0f, pdfA, false);
        const float G_rev = prev_cos_out / len_sqr(origin - isect.sd.position);
        const float dE = connection_dVC(dVC, f * G_rev, bsdf_pdf * G, prev_specular);
        weight = 1 / (1 + dE * pow2(0.0f) + dL * pow2(pdfA_fwd));
      } else {
        weight = path_weight(path_length, 1);
      }
    } else if (fr.flag(STHIP_eSampleBSDFs))
      weight = mis2(fr, pdfA, pdfA_fwd);
    if (!defer) {
      accumulate_contribution(beta * contrib, weight, 1);  // path.hlsli:365
      return;
    }
    const v3 c = beta * contrib * weight;
    if (diffuse_vertices >= 1 && diffuse_vertices <= max_shadow) {
      sthip_ShadowRayData& rd = shadow_rays[diffuse_vertices - 1];
      rd.contribution[0] = c.x;
      rd.contribution[1] = c.y;
      rd.contribution[2] = c.z;
      rd.rng_offset = rng.v[3];
      rd.ray_origin[0] = isect.sd.position.x;
      rd.ray_origin[1] = isect.sd.position.y;
      rd.ray_origin[2] = isect.sd.position.z;
      rd.medium = medium;
      rd.ray_direction[0] = cand.ray_direction.x;
      rd.ray_direction[1] = cand.ray_direction.y;
      rd.ray_direction[2] = cand.ray_direction.z;
      rd.ray_distance = cand.ray_distance;
    }
  }

  // path_weight, path.hlsli:16-28
  float path_weight(uint32_t view_length, uint32_t light_length) const {
    const uint32_t n_vertices = view_length + light_length;
    if (n_vertices <= 2) return 1;
    uint32_t n = 1;
    if (fr.flag(STHIP_eNEE)) n++;
    if (fr.flag(STHIP_eConnectToViews) && n_vertices <= fr.pc.gMaxPathVertices + 1) n++;
    if (fr.flag(STHIP_eConnectToLightPaths)) n += std::min(fr.pc.gMaxPathVertices, n_vertices - 2);
    return 1.f / (float)n;
  }

  // vertex() / store_light_vertex(), path.hlsli:491-531 (no light vertex cache: slot = light_vertex_index, :64)
  void store_light_vertex() {
    uint32_t flags = 0;
    if (isect.instance_index() != STHIP_INVALID_INSTANCE) flags |= 2u;  // IS_BACKGROUND as upstream sets it
    if (prev_specular) flags |= 8u;
    const bool in_medium = has_media(fr) && isect.sd.shape_area == 0;
    if (in_medium) flags |= 4u;  // PATH_VERTEX_FLAG_IS_MEDIUM, path.hlsli:494
    PathVertex v;
    if (path_length > 1) {
      v.material_address = fr.pc.gEnvironmentMaterialAddress;
      if (isect.instance_index() != STHIP_INVALID_INSTANCE) v.material_address = fr.sc->instances[isect.instance_index()].material_address();
    } else
      v.material_address = 0xFFFFFFFFu;
    v.position[0] = isect.sd.position.x;
    v.position[1] = isect.sd.position.y;
    v.position[2] = isect.sd.position.z;
    v.packed_geometry_normal = isect.sd.packed_geometry_normal;
    v.packed_local_dir_in = pack_normal_octahedron(local_dir_in);
    v.packed_shading_normal = isect.sd.packed_shading_normal;
    v.packed_tangent = isect.sd.packed_tangent;
    v.uv[0] = isect.sd.u;
    v.uv[1] = isect.sd.v;
    if (in_medium) {  // (normals, tangent and uv are the stale ones of the last surface query upstream; nothing reads them of a medium vertex: pinned to 0)
      v.packed_geometry_normal = v.packed_shading_normal = v.packed_tangent = 0;
      v.uv[0] = v.uv[1] = 0;
    }
    v.pack_beta(fr.flag(STHIP_eLVCReservoirs) ? path_contrib : beta, path_length, diffuse_vertices, flags);  // path.hlsli:513
    v.prev_dVC = dVC;
    v.G_rev = prev_cos_out / len_sqr(origin - isect.sd.position);
    v.prev_pdfA_fwd = bsdf_pdf * G;
    v.path_pdf = path_pdf;
    if (fr.lvc()) {  // the cache: staged per (path, vertex), compacted in that order after the light pass (see Frame)
      fr.lvc_staging[(size_t)path_index() * (fr.pc.gMaxDiffuseVertices - 1) + (diffuse_vertices - 1)] = v;
      return;
    }
    const size_t idx = (size_t)fr.pc.gOutputExtent[0] * fr.pc.gOutputExtent[1] * (diffuse_vertices - 1) + path_index();
    if (idx < fr.light_vertex_count) fr.light_vertices[idx] = v;
  }

  // eval_bsdf(PathVertex, ...), path.hlsli:100-123: the BSDF of a stored light vertex towards dir_out. The material is
  // loaded again at the stored uv with a zero footprint, and a normal map perturbs the stored (already perturbed) frame
  // a second time, as upstream's inout arguments do.
  bool eval_bsdf_vertex(PathVertex v, v3 dir_out, MaterialEvalRecord& ev, float& ngdotout) const {
    if (has_media(fr) && v.is_medium()) {  // path.hlsli:103-108: the phase function with the stored (octahedron-packed) direction
      Medium mm;
      mm.load(*fr.sc, v.material_address);
      if (mm.is_specular()) return false;
      const float ph = mm.phase(unpack_normal_octahedron(v.packed_local_dir_in), dir_out);
      ev.f = V3(ph);
      ev.pdf_fwd = ev.pdf_rev = ph;
      ngdotout = 1;
      return true;
    }
    DisneyMaterial lm;
    lm.load(*fr.sc, v.material_address, v.uv[0], v.uv[1], 0.0f, v.packed_shading_normal, v.packed_tangent, fr.sampling_flags);
    if (lm.is_specular()) return false;
    const v3 n = unpack_normal_octahedron(v.packed_shading_normal), t = unpack_normal_octahedron(v.packed_tangent);
    const v3 b = cross(n, t) * (v.flip_bitangent() ? -1.0f : 1.0f);
    const v3 lv_dir_in = unpack_normal_octahedron(v.packed_local_dir_in);
    const v3 lv_dir_out = normalize(V3(dot(dir_out, t), dot(dir_out, b), dot(dir_out, n)));
    lm.eval(ev, lv_dir_in, lv_dir_out, true);
    if (ev.pdf_fwd < 1e-6f) return false;
    const v3 ng = unpack_normal_octahedron(v.packed_geometry_normal);
    ngdotout = dot(ng, dir_out);
    const v3 world_in = normalize(t * lv_dir_in.x + b * lv_dir_in.y + n * lv_dir_in.z);
    ev.f = ev.f * shading_normal_correction(lv_dir_in.z, lv_dir_out.z, dot(ng, world_in), ngdotout, dot(ng, n), fr.flag(STHIP_eShadingNormalShadowFix), true);
    return true;
  }

  // connect_light_vertex, path.hlsli:618-680 (surfaces only)
  v3 connect_light_vertex(const DisneyMaterial& m, const PathVertex& lv, float& weight, v3& ray_origin, v3& ray_direction, float& ray_distance) const {
    return connect_light_vertex_any(&m, nullptr, lv, weight, ray_origin, ray_direction, ray_distance);
  }
  // (`phase` != null: the view vertex lies inside a medium — no geometry, the direction itself as local_to_light, :649-650)
  v3 connect_light_vertex_any(const DisneyMaterial* mp, const Medium* phase, const PathVertex& lv, float& weight, v3& ray_origin, v3& ray_direction, float& ray_distance) const {
    v3 contrib = lv.beta();
    if (all_le0(contrib) || any_nan(contrib)) return V3(0.0f);
    ray_origin = isect.sd.position;
    ray_direction = V3(lv.position[0], lv.position[1], lv.position[2]) - isect.sd.position;
    ray_distance = length(ray_direction);
    const float rcp_dist = 1 / ray_distance;
    ray_direction = ray_direction * rcp_dist;
    const float rcp_dist2 = pow2(rcp_dist);
    contrib = contrib * rcp_dist2;
    float connection_G_fwd = rcp_dist2;
    if (!lv.is_medium()) ray_distance = ray_distance * 0.999f;  // visibility_distance_epsilon (:634-635)
    float cos_theta_light = 0;
    MaterialEvalRecord lv_eval;
    if (!eval_bsdf_vertex(lv, -ray_direction, lv_eval, cos_theta_light)) return V3(0.0f);  // f = 0 upstream
    contrib = contrib * lv_eval.f;
    connection_G_fwd *= fabsf(cos_theta_light);
    const float dL = connection_dVC(lv.prev_dVC, lv_eval.pdf_rev * lv.G_rev, lv.prev_pdfA_fwd, lv.is_prev_delta());
    float pdfA_rev = lv_eval.pdf_fwd * rcp_dist2;
    if (all_le0(contrib) || any_nan(contrib)) return V3(0.0f);
    MaterialEvalRecord ev;
    if (phase) {
      const float ph = phase->phase(local_dir_in, ray_direction);
      ev.f = V3(ph);
      ev.pdf_fwd = ev.pdf_rev = ph;
    } else {
      const v3 local_to_light = normalize(isect.sd.to_local(ray_direction));
      const v3 geometry_normal = isect.sd.geometry_normal();
      const float ngdotout = dot(geometry_normal, ray_direction);
      const float ngdotns = dot(geometry_normal, isect.sd.shading_normal());
      ray_origin = ray_offset(ray_origin, ngdotout > 0 ? geometry_normal : -geometry_normal);
      pdfA_rev *= fabsf(ngdotout);
      contrib = contrib * shading_normal_correction(local_dir_in.z, local_to_light.z, ngdotin, ngdotout, ngdotns, fr.flag(STHIP_eShadingNormalShadowFix), false);
      mp->eval(ev, local_dir_in, local_to_light, false);
    }
    if (ev.pdf_fwd < 1e-6f) return V3(0.0f);
    contrib = contrib * ev.f;
    if (all_le0(contrib)) return V3(0.0f);
    if (fr.flag(STHIP_eMIS)) {
      const float G_rev = prev_cos_out / len_sqr(origin - isect.sd.position);
      const float dE = connection_dVC(dVC, ev.pdf_rev * G_rev, bsdf_pdf * G, prev_specular);
      weight = 1 / (1 + dE * pow2(pdfA_rev) + dL * pow2(ev.pdf_fwd * connection_G_fwd));
    } else
      weight = path_weight(path_length, lv.subpath_length());
    return contrib;
  }

  // connect_light_subpath, path.hlsli:802-822: this view vertex to every stored vertex of the light subpath that shares
  // its path index; a slot beyond the buffer reads as a zero vertex (robust buffer access)
  void connect_light_subpath(const DisneyMaterial& m) { connect_light_subpath_any(&m, nullptr); }
  void connect_light_subpath_any(const DisneyMaterial* mp, const Medium* phase) {
    for (uint32_t i = 1; i < fr.pc.gMaxDiffuseVertices; i++) {
      const size_t idx = (size_t)fr.pc.gOutputExtent[0] * fr.pc.gOutputExtent[1] * (i - 1) + path_index();
      if (idx >= fr.light_vertex_count) break;
      const PathVertex lv = fr.light_vertices[idx];
      if (lv.subpath_length() + path_length > fr.pc.gMaxPathVertices || lv.diffuse_vertices() + diffuse_vertices > fr.pc.gMaxDiffuseVertices || all_le0(lv.beta())) break;
      v3 ray_origin, ray_direction;
      float ray_distance = 0, weight = 0;
      v3 contrib = beta * connect_light_vertex_any(mp, phase, lv, weight, ray_origin, ray_direction, ray_distance);
      if (all_le0(contrib) || weight <= 0) continue;
      if (has_media(fr)) {  // :814-818: the walk through the media, in this path's own stream
        float dir_pdf = 1, nee_pdf = 1;
        trace_visibility_media(rng, ray_origin, ray_direction, ray_distance, medium, contrib, dir_pdf, nee_pdf, 64);
        if (all_le0(contrib) || nee_pdf <= 0) continue;
        contrib = contrib / nee_pdf;
      } else if (occluded(ray_origin, ray_direction, ray_distance))
        continue;
      accumulate_contribution(contrib, weight, lv.subpath_length());  // path.hlsli:820
    }
  }

  // connect_lvc, path.hlsli:683-800 without the reservoir reuse through the hash grid: this view vertex to ONE vertex of the
  // light vertex cache, picked uniformly (li % n) or, with eLVCReservoirs, by resampled importance sampling over
  // gReservoirM uniform picks with target luminance(contribution). The cache order is the defined one (see Frame). An
  // empty cache (n = 0) is a division by zero upstream; here the random numbers are drawn and nothing connects. With
  // eDeferShadowRays the record goes to this vertex's gShadowRays slot — the slot connect_light has just written, as upstream.
  void connect_lvc(const DisneyMaterial& m) { connect_lvc_any(&m, nullptr); }
  // (`phase` != null: from a vertex inside a medium — connect_light_vertex's medium branch; the geometry normal the reuse takes its
  // jitter plane from and stores is the stale one of the last surface query there: pinned to the packed value 0)
  void connect_lvc_any(const DisneyMaterial* mp, const Medium* phase) {
    const uint32_t n = std::min<uint64_t>(fr.lvc_count, (uint64_t)fr.pc.gLightPathCount * fr.pc.gMaxDiffuseVertices);
    uint32_t li = rng.next_uint();
    const bool coherent = coherent_sampling();
    if (coherent && !coherent_index(1, li)) return;  // :688
    const PathVertex zero{};
    auto fits = [&](const PathVertex& v) {
      return !(v.subpath_length() + path_length > fr.pc.gMaxPathVertices || v.diffuse_vertices() + diffuse_vertices > fr.pc.gMaxDiffuseVertices || all_le0(v.beta()));
    };
    PathVertex lv = n ? fr.light_vertices[li % n] : zero;
    v3 contrib = V3(0.0f), ray_origin = V3(0.0f), ray_direction = V3(0.0f);
    float weight = 1, ray_distance = 0;
    if (fr.flag(STHIP_eLVCReservoirs)) {
      Reservoir r;
      r.init();
      float r_target_pdf = 0;  // (uninitialised upstream when no candidate is taken; W() of it then multiplies a zero contribution)
      for (uint32_t i = 0; i < fr.pc.gReservoirM; i++) {
        const uint32_t pick = coherent ? li + (1 + i) * 32u : rng.next_uint();  // :703
        const PathVertex lv_i = n ? fr.light_vertices[pick % n] : zero;
        if (!fits(lv_i)) continue;
        v3 ro_i = V3(0.0f), rd_i = V3(0.0f);
        float dist_i = 0, weight_i = 0;
        const v3 contrib_i = connect_light_vertex_any(mp, phase, lv_i, weight_i, ro_i, rd_i, dist_i);
        const float target_pdf_i = luminance(contrib_i);
        if (r.update(rng.next_float(), target_pdf_i / lv_i.path_pdf)) {
          contrib = contrib_i;
          weight = weight_i;
          ray_origin = ro_i;
          ray_direction = rd_i;
          ray_distance = dist_i;
          r_target_pdf = target_pdf_i;
          lv = lv_i;
        }
      }
      if (fr.flag(STHIP_eLVCReservoirReuse)) {  // path.hlsli:727-768
        v3 t, b;
        make_orthonormal(phase ? unpack_normal_octahedron(0) : isect.sd.geometry_normal(), t, b);
        const float cell_size = hashgrid_cell_size(isect.sd.position);
        auto jittered = [&]() {
          const float phi = rng.next_float() * 2 * DET_PI;
          if (!fr.flag(STHIP_eHashGridJitter)) return isect.sd.position;
          const float radius = cell_size * rng.next_float();
          float sn, cs;
          det_sincosf(phi, &sn, &cs);
          return isect.sd.position + (t * cs + b * sn) * radius;
        };
        if (fr.prev_lvc_grid && fr.pc.gReservoirSpatialM > 0) {
          const v3 at = jittered();
          const uint32_t bucket = fr.prev_lvc_grid->table.find(at, cell_size);
          if (bucket != 0xFFFFFFFFu) {
            const uint32_t bucket_start = fr.prev_lvc_grid->table.indices[bucket], bucket_size = fr.prev_lvc_grid->table.counters[bucket];
            uint32_t M = r.M;
            for (uint32_t i = 0; i < fr.pc.gReservoirSpatialM; i++) {
              const PathVertexReservoir& prev = fr.prev_lvc_grid->data[bucket_start + rng.next_uint() % bucket_size];
              const PathVertex lv_i = prev.y;
              if (!fits(lv_i)) continue;
              M += prev.r.M;
              v3 ro_i = V3(0.0f), rd_i = V3(0.0f);
              float dist_i = 0, weight_i = 0;
              const v3 contrib_i = connect_light_vertex_any(mp, phase, lv_i, weight_i, ro_i, rd_i, dist_i);
              const float target_pdf_i = luminance(contrib_i);
              if (r.update(rng.next_float(), target_pdf_i / lv_i.path_pdf)) {
                contrib = contrib_i;
                weight = weight_i;
                ray_origin = ro_i;
                ray_direction = rd_i;
                ray_distance = dist_i;
                r_target_pdf = target_pdf_i;
                lv = lv_i;
              }
            }
            r.M = M;
          }
        }
        const float W = r.W(r_target_pdf);
        const v3 at = jittered();
        r.M = std::min(r.M, fr.pc.gReservoirMaxM);
        const size_t k = (size_t)path_index() * fr.pc.gMaxDiffuseVertices + (diffuse_vertices - 1);
        if (fr.lvc_appends && diffuse_vertices >= 1 && diffuse_vertices <= fr.pc.gMaxDiffuseVertices) {
          HashGridOf<PathVertexReservoir>::Append& a = fr.lvc_appends[k];
          a.pos = at;
          a.cell_size = cell_size;
          a.y.r = r;
          a.y.packed_geometry_normal = phase ? 0u : isect.sd.packed_geometry_normal;
          a.y.W = W;
          a.y.y = lv;
          fr.lvc_append_valid[k] = 1;
        }
      }
      contrib = contrib * r.W(r_target_pdf);
    } else if (fits(lv)) {
      contrib = connect_light_vertex_any(mp, phase, lv, weight, ray_origin, ray_direction, ray_distance);
    }
    contrib = contrib * (float)(fr.pc.gMaxDiffuseVertices - 1);
    contrib = contrib * beta;
    if (fr.flag(STHIP_eDeferShadowRays)) {
      if (!(diffuse_vertices >= 1 && diffuse_vertices <= max_shadow)) return;
      sthip_ShadowRayData& rd = shadow_rays[diffuse_vertices - 1];
      const v3 c = contrib * weight;
      rd.contribution[0] = c.x;
      rd.contribution[1] = c.y;
      rd.contribution[2] = c.z;
      rd.rng_offset = rng.v[3];
      rd.ray_origin[0] = ray_origin.x;
      rd.ray_origin[1] = ray_origin.y;
      rd.ray_origin[2] = ray_origin.z;
      rd.medium = medium;
      rd.ray_direction[0] = ray_direction.x;
      rd.ray_direction[1] = ray_direction.y;
      rd.ray_direction[2] = ray_direction.z;
      rd.ray_distance = ray_distance;
    } else if (any_gt0(contrib) && weight > 0) {
      if (has_media(fr)) {  // :791-796: the walk through the media, in this path's own stream
        float dir_pdf = 1, nee_pdf = 1;
        trace_visibility_media(rng, ray_origin, ray_direction, ray_distance, medium, contrib, dir_pdf, nee_pdf, 64);
        if (all_le0(contrib) || nee_pdf <= 0) return;
        contrib = contrib / nee_pdf;
        accumulate_contribution(contrib, weight, lv.subpath_length());
      } else if (!occluded(ray_origin, ray_direction, ray_distance))
        accumulate_contribution(contrib, weight, lv.subpath_length());  // path.hlsli:797
    }
  }

  // connect_view, path.hlsli:533-613: the light-path vertex seen from the camera, splatted into gLightTraceSamples
  // connect_view, path.hlsli:533-613. At a vertex inside a medium (`phase` != null): no geometry — ngdotout = 1, the direction
  // itself as local_to_view, no ray offset, no shading-normal factor (:562-565) — and the phase function as f and both pdfs.
  void connect_view(const DisneyMaterial& m) { connect_view_any(&m, nullptr); }
  void connect_view_medium(const Medium& mm) { connect_view_any(nullptr, &mm); }
  void connect_view_any(const DisneyMaterial* mp, const Medium* phase) {
    uint32_t view_index = 0;
    if (fr.pc.gViewCount > 1) view_index = (uint32_t)fminf(rng.next_float() * (float)fr.pc.gViewCount, (float)(fr.pc.gViewCount - 1));
    const sthip_ViewData& view = fr.fd.gViews[view_index];
    float sp[4];
    project_point(view.projection, transform_point(fr.fd.gInverseViewTransforms[view_index], isect.sd.position), sp);
    sp[1] = -sp[1];
    sp[0] = sp[0] / sp[3];
    sp[1] = sp[1] / sp[3];
    sp[2] = sp[2] / sp[3];
    if (fabsf(sp[0]) >= 1 || fabsf(sp[1]) >= 1 || fabsf(sp[2]) >= 1 || sp[2] <= 0) return;
    const float u = sp[0] * .5f + .5f, v = sp[1] * .5f + .5f;
    const int ix = view.image_min[0] + (int)((float)(view.image_max[0] - view.image_min[0]) * u);
    const int iy = view.image_min[1] + (int)((float)(view.image_max[1] - view.image_min[1]) * v);
    const uint32_t output_index = (uint32_t)iy * fr.pc.gOutputExtent[0] + (uint32_t)ix;
    const sthip_TransformData& t = fr.fd.gViewTransforms[view_index];
    const v3 position = V3(t.m[0][3], t.m[1][3], t.m[2][3]);
    const v3 view_normal = normalize(transform_vector(t, V3(0, 0, 1)));
    v3 to_view = position - isect.sd.position;
    const float dist = length(to_view);
    to_view = to_view / dist;
    const float sensor_cos_theta = fabsf(dot(to_view, view_normal));
    const float sensor_importance = 1 / (view.projection.sensor_area * 1.0f * (pow2(sensor_cos_theta) * pow2(sensor_cos_theta)));
    // pdfAtoW(1 / lens_area, cos / dist^2) = 1 / (cos / dist^2)
    v3 contribution = beta * sensor_importance / (1.0f / (sensor_cos_theta / pow2(dist)));
    const float G_rev = fabsf(prev_cos_out) / len_sqr(origin - isect.sd.position);
    v3 ray_origin = isect.sd.position;
    MaterialEvalRecord ev;
    if (phase) {
      const float v = phase->phase(local_dir_in, to_view);
      ev.f = V3(v);
      ev.pdf_fwd = ev.pdf_rev = v;
    } else {
      const v3 geometry_normal = isect.sd.geometry_normal();
      const float ngdotout = dot(to_view, geometry_normal);
      ray_origin = ray_offset(isect.sd.position, ngdotout > 0 ? geometry_normal : -geometry_normal);
      const v3 local_to_view = normalize(isect.sd.to_local(to_view));
      contribution = contribution * shading_normal_correction(local_dir_in.z, local_to_view.z, ngdotin, ngdotout, dot(geometry_normal, isect.sd.shading_normal()), fr.flag(STHIP_eShadingNormalShadowFix), true);
      mp->eval(ev, local_dir_in, local_to_view, true);
    }
    if (ev.pdf_fwd < 1e-6f) return;
    contribution = contribution * ev.f;
    if (all_le0(contribution)) return;
    if (has_media(fr)) {  // trace_visibility_ray through the media, in this light path's own stream (:577-581)
      float nee_pdf = 1, dir_pdf = 1;
      trace_visibility_media(rng, ray_origin, to_view, dist, medium, contribution, dir_pdf, nee_pdf, 64);
      if (nee_pdf > 0) contribution = contribution / nee_pdf;
    } else if (occluded(ray_origin, to_view, dist))
      return;  // trace_visibility_ray over the full distance (:580)
    float weight;
    if (fr.flag(STHIP_eMIS)) {
      if (fr.flag(STHIP_eConnectToLightPaths)) {  // dL_1, path.hlsli:591-596
        const float dL_1 = connection_dVC(dVC, ev.pdf_rev * G_rev, bsdf_pdf * G, prev_specular);
        weight = 1 / (1 + dL_1 * pow2(1.0f));
      } else
        weight = prev_specular ? 1.0f : mis2(fr, path_pdf, 1.0f * path_pdf_rev * (ev.pdf_rev * G_rev));
    } else
      weight = path_weight(1, path_length);
    if (fr.debug(STHIP_DEBUG_LIGHT_TRACE_CONTRIBUTION)) weight = 1;  // path.hlsli:608-609
    if (fr.debug(STHIP_DEBUG_PATH_LENGTH_CONTRIBUTION) && fr.pc.gDebugViewPathLength == 1) {  // :611-613: only the asked light length, unweighted
      if (fr.pc.gDebugLightPathLength != path_length) return;
      weight = 1;
    }
    // accumulate_light_contribution, path.hlsli:47-60: quantised integer sums (order-independent) + overflow bits
    const v3 c = contribution * weight;
    const float q = (float)fr.light_trace_quantization;
    const float cf[3] = {fmaxf(0.0f, c.x) * q, fmaxf(0.0f, c.y) * q, fmaxf(0.0f, c.z) * q};
    uint32_t ci[3];
    for (int k = 0; k < 3; k++) ci[k] = cf[k] >= 4294967296.0f ? 0xFFFFFFFFu : (cf[k] == cf[k] ? (uint32_t)cf[k] : 0u);
    if (ci[0] == 0 && ci[1] == 0 && ci[2] == 0) return;
    uint32_t overflow_mask = 0;
    for (int k = 0; k < 3; k++) {
      const uint32_t prev = fr.light_trace[4 * (size_t)output_index + k].fetch_add(ci[k], std::memory_order_relaxed);
      if (ci[k] > 0xFFFFFFFFu - prev) overflow_mask |= 1u << k;
    }
    if (overflow_mask) fr.light_trace[4 * (size_t)output_index + 3].fetch_or(overflow_mask, std::memory_order_relaxed);
  }
  // path.hlsli:1048-1075
  void next_vertex() {
    if (isect.instance_index() == STHIP_INVALID_INSTANCE) {
      if (has_environment(fr)) {
        Environment env;
        env.load(*fr.sc, fr.pc.gEnvironmentMaterialAddress);
        eval_emission(env.eval(direction));
      }
      beta = V3(0.0f);
      return;
    }
    const uint32_t material_address = fr.sc->instances[isect.instance_index()].material_address();
    if (has_media(fr) && isect.sd.shape_area == 0) {  // path.hlsli:1062-1066
      Medium mm;
      mm.load(*fr.sc, material_address);
      local_dir_in = -direction;
      if (!next_vertex_medium(mm)) {
        beta = V3(0.0f);
        return;
      }
      trace();
      return;
    }
    DisneyMaterial m;
    m.load(*fr.sc, material_address, isect.sd, fr.sampling_flags);
    local_dir_in = normalize(isect.sd.to_local(-direction));
    if (!next_vertex_m(m)) {
      beta = V3(0.0f);
      return;
    }
    trace();
  }
};

inline v3 back_project(const sthip_ProjectionData& p, float cx, float cy) {  // transform.h:136-147
  v3 r;
  if (p.vertical_fov < 0) {
    r.x = (cx - p.offset[0]) / p.scale[0];
    r.y = (cy - p.offset[1]) / p.scale[1];
  } else {
    r.x = p.near_plane * (cx * sgn(p.near_plane) - p.offset[0]) / p.scale[0];
    r.y = p.near_plane * (cy * sgn(p.near_plane) - p.offset[1]) / p.scale[1];
  }
  r.z = p.near_plane;
  return r;
}
inline void project_point(const sthip_ProjectionData& p, v3 v, float r[4]) {  // transform.h:118-135
  if (p.vertical_fov < 0) {
    r[0] = v.x * p.scale[0] + p.offset[0];
    r[1] = v.y * p.scale[1] + p.offset[1];
    r[2] = (v.z - p.far_plane) / (p.near_plane - p.far_plane);
    r[3] = 1;
  } else {
    r[0] = v.x * p.scale[0] + v.z * p.offset[0];
    r[1] = v.y * p.scale[1] + v.z * p.offset[1];
    r[2] = fabsf(p.near_plane);
    r[3] = v.z * sgn(p.near_plane);
  }
}

struct PixelAOV {
  float albedo[4];
  sthip_VisibilityInfo vis;
  sthip_DepthInfo depth;
  float prev_uv[2];
};

inline int get_view_index(const Frame& fr, uint32_t x, uint32_t y) {  // scene.h:132-137
  for (uint32_t i = 0; i < fr.fd.view_count; i++) {
    const sthip_ViewData& v = fr.fd.gViews[i];
    if ((int)x >= v.image_min[0] && (int)y >= v.image_min[1] && (int)x < v.image_max[0] && (int)y < v.image_max[1]) return (int)i;
  }
  return -1;
}

inline v3 primary_dir(const sthip_ViewData& view, const sthip_TransformData& t, float fx, float fy, v3* local_out) {
  const float ex = (float)(view.image_max[0] - view.image_min[0]), ey = (float)(view.image_max[1] - view.image_min[1]);
  const float u = (fx + 0.5f - (float)view.image_min[0]) / ex;
  const float v = (fy + 0.5f - (float)view.image_min[1]) / ey;
  const float cx = 2 * u - 1;
  const float cy = -(2 * v - 1);
  const v3 local_dir = normalize(back_project(view.projection, cx, cy));
  if (local_out) *local_out = local_dir;
  return normalize(transform_vector(t, local_dir));
}

// bdpt.hlsl:149-300 (sample_visibility) + :302-326 (trace_shadows) for one pixel and one seed.
// Returns gRadiance[px] (rgb; alpha is 1) and, when aov != NULL, the AOVs of :222-296.
// rr: the group decisions of eCoherentRR (in) and what this pixel's path found missing (out); null = not part of a group
struct RRControl {
  const PathIntegrator::GroupRR* decisions = nullptr;
  int missing_length = -1;
  float missing_p = 0, missing_rnd = 0;
  // eCoherentSampling: the group's index values per site (in) and what the path found missing (out)
  const PathIntegrator::GroupValue* nee = nullptr;
  const PathIntegrator::GroupValue* lvc = nullptr;
  int cs_site = -1, cs_length = -1;
  uint32_t cs_value = 0;
};
// dbg: gDebugImage[pixel] (rgba, in / out: upstream's image persists from frame to frame and most modes only add to it or
// overwrite it where a path gets somewhere), or null
bool render_pixel(const Frame& fr, uint32_t x, uint32_t y, uint32_t seed, float out_rgb[3], PixelAOV* aov, uint64_t stats[4], RRControl* rr = nullptr, float* dbg = nullptr) {
  const int view_index = get_view_index(fr, x, y);
  if (fr.debug_mode == 0) dbg = nullptr;
  // add_light_trace's debug write (bdpt.hlsl:328-338) for a pixel whose sample_visibility returned before tracing anything — it
  // is a pass of its own over EVERY pixel of the image, dispatched whenever eConnectToViews is set (BDPT.cpp:753); the samples
  // it loads are this frame's (zero where no photon landed, or where none was sampled: pinned, see render_frame)
  auto light_trace_debug_only = [&]() {
    if (!dbg || !fr.light_trace || fr.debug(STHIP_DEBUG_VIEW_TRACE_CONTRIBUTION)) return;
    if (!(fr.debug(STHIP_DEBUG_LIGHT_TRACE_CONTRIBUTION) || (fr.debug(STHIP_DEBUG_PATH_LENGTH_CONTRIBUTION) && fr.pc.gDebugViewPathLength == 1))) return;
    const size_t idx = (size_t)y * fr.pc.gOutputExtent[0] + x;
    uint32_t v[4];
    for (int k = 0; k < 4; k++) v[k] = fr.light_trace[4 * idx + k].load(std::memory_order_relaxed);
    for (int k = 0; k < 3; k++)
      if (v[3] & (1u << k)) v[k] = 0xFFFFFFFFu;
    v3 lc = V3((float)v[0], (float)v[1], (float)v[2]) / (float)fr.light_trace_quantization;
    if (lc.x < 0 || lc.y < 0 || lc.z < 0 || any_nan(lc)) lc = V3(0.0f);
    dbg[0] = lc.x;
    dbg[1] = lc.y;
    dbg[2] = lc.z;
    dbg[3] = 1;
  };
  if (view_index < 0) {
    light_trace_debug_only();
    return false;
  }
  PathIntegrator path(fr, x, y, seed);
  path.dbg = dbg;
  PixelAOV aov_scratch;
  if (dbg && fr.debug(STHIP_DEBUG_PREV_UV) && !aov) {  // (the mode reads the previous-frame uv the G-buffer block computes)
    memset(&aov_scratch, 0, sizeof(aov_scratch));
    aov = &aov_scratch;
  }
  if (dbg && (fr.debug(STHIP_DEBUG_PATH_LENGTH_CONTRIBUTION) || fr.debug(STHIP_DEBUG_VIEW_TRACE_CONTRIBUTION))) {  // bdpt.hlsl:161-162
    dbg[0] = dbg[1] = dbg[2] = 0;
    dbg[3] = 1;
  }
  if (rr) {
    path.group_rr = rr->decisions;
    path.group_nee = rr->nee;
    path.group_lvc = rr->lvc;
    path.lane = (y & 3u) * 8u + (x & 7u);
  }
  out_rgb[0] = out_rgb[1] = out_rgb[2] = 0;
  if (fr.pc.gMaxPathVertices < 2) {
    light_trace_debug_only();
    return true;
  }
  // what follows the path for every pixel inside a view, also one whose sample_visibility returned early: trace_shadows, add_light_trace, the outputs
  auto finish = [&]() -> bool {
    // trace_shadows, bdpt.hlsl:302-326
    v3 c = V3(0.0f);
    for (uint32_t i = 1; i <= fr.pc.gMaxDiffuseVertices && i <= path.max_shadow; i++) {
      const sthip_ShadowRayData& rd = path.shadow_rays[i - 1];
      v3 contribution = V3(rd.contribution[0], rd.contribution[1], rd.contribution[2]);
      if (all_le0(contribution)) continue;
      if (has_media(fr)) {
        Rng r;  // rng_init(pixel_coord, rd.rng_offset), bdpt.hlsl:315
        r.v[0] = x;
        r.v[1] = y;
        r.v[2] = seed;
        r.v[3] = rd.rng_offset;
        float dir_pdf = 1, nee_pdf = 1;
        path.trace_visibility_media(r, V3(rd.ray_origin[0], rd.ray_origin[1], rd.ray_origin[2]), V3(rd.ray_direction[0], rd.ray_direction[1], rd.ray_direction[2]), rd.ray_distance, rd.medium,
                                    contribution, dir_pdf, nee_pdf);
        if (nee_pdf > 0) contribution = contribution / nee_pdf;
      } else if (path.occluded(V3(rd.ray_origin[0], rd.ray_origin[1], rd.ray_origin[2]), V3(rd.ray_direction[0], rd.ray_direction[1], rd.ray_direction[2]), rd.ray_distance))
        contribution = V3(0.0f);
      c = c + contribution;
    }
    v3 rad = path.radiance + c;
    if (fr.light_trace && !fr.debug(STHIP_DEBUG_VIEW_TRACE_CONTRIBUTION)) {  // add_light_trace, bdpt.hlsl:328-338 with load_light_sample, path.hlsli:38-46
      const size_t idx = (size_t)y * fr.pc.gOutputExtent[0] + x;
      uint32_t v[4];
      for (int k = 0; k < 4; k++) v[k] = fr.light_trace[4 * idx + k].load(std::memory_order_relaxed);
      for (int k = 0; k < 3; k++)
        if (v[3] & (1u << k)) v[k] = 0xFFFFFFFFu;
      v3 lc = V3((float)v[0], (float)v[1], (float)v[2]) / (float)fr.light_trace_quantization;
      if (lc.x < 0 || lc.y < 0 || lc.z < 0 || any_nan(lc)) lc = V3(0.0f);
      rad = rad + lc;
      if (dbg && (fr.debug(STHIP_DEBUG_LIGHT_TRACE_CONTRIBUTION) || (fr.debug(STHIP_DEBUG_PATH_LENGTH_CONTRIBUTION) && fr.pc.gDebugViewPathLength == 1))) path.debug_set(lc);  // bdpt.hlsl:335-336
    }
    out_rgb[0] = rad.x;
    out_rgb[1] = rad.y;
    out_rgb[2] = rad.z;
    stats[0] += path.rays_total;
    stats[1] += path.rays_path;
    stats[2] += path.counters[0];
    stats[3] += path.counters[1];
    if (rr) {
      rr->missing_length = path.rr_missing_length;
      rr->missing_p = path.rr_missing_p;
      rr->missing_rnd = path.rr_missing_rnd;
      rr->cs_site = path.cs_missing_site;
      rr->cs_length = path.cs_missing_length;
      rr->cs_value = path.cs_missing_value;
    }
    return true;
};

  const sthip_ViewData& view = fr.fd.gViews[view_index];
  const sthip_TransformData& t = fr.fd.gViewTransforms[view_index];
  const float ex = (float)(view.image_max[0] - view.image_min[0]), ey = (float)(view.image_max[1] - view.image_min[1]);
  const float uvx = ((float)x + 0.5f - (float)view.image_min[0]) / ex;
  const float uvy = ((float)y + 0.5f - (float)view.image_min[1]) / ey;
  v3 local_dir_out;
  path.direction = primary_dir(view, t, (float)x, (float)y, &local_dir_out);
  path.prev_cos_out = fabsf(local_dir_out.z);
  path.origin = V3(t.m[0][3], t.m[1][3], t.m[2][3]);
  if (fr.flag(STHIP_eRayCones)) {  // bdpt.hlsl:176-189
    const float cxx = 2 * (((float)x + 1.0f + 0.5f - (float)view.image_min[0]) / ex) - 1, cxy = -(2 * uvy - 1);
    const float cyx = 2 * uvx - 1, cyy = -(2 * (((float)y + 1.0f + 0.5f - (float)view.image_min[1]) / ey) - 1);
    const v3 dir_dx = back_project(view.projection, cxx, cxy);
    const v3 dir_dy = back_project(view.projection, cyx, cyy);
    const v3 l = local_dir_out / local_dir_out.z;
    path.rd_radius = 0;
    path.rd_spread = fminf(length(dir_dx / dir_dx.z - l), length(dir_dy / dir_dy.z - l));
  }
  if (dbg && fr.debug(STHIP_DEBUG_ENVIRONMENT_SAMPLE_TEST)) {  // bdpt.hlsl:190-199 (no ray is traced; without an environment the record at gEnvironmentMaterialAddress is not one: not restated)
    if (!has_environment(fr)) return finish();
    Environment env;
    env.load(*fr.sc, fr.pc.gEnvironmentMaterialAddress);
    for (uint32_t i = 0; i < 8; i++) {
      float pdf;
      v3 dir;
      const float r0 = path.rng.next_float(), r1 = path.rng.next_float();
      (void)env.sample(r0, r1, dir, pdf, fr.flag(STHIP_eSampleEnvironmentMapDirectly));
      const float v = 1024 * det_powf(fmaxf(0.0f, dot(dir, path.direction)), 1024.0f);
      path.debug_add(V3(v));
    }
    return finish();  // (light tracing's image is still added: add_light_trace is a pass of its own)
  }
  if (dbg && fr.debug(STHIP_DEBUG_ENVIRONMENT_SAMPLE_PDF)) {  // bdpt.hlsl:200-204 (.rgb = pdf: the alpha stays)
    if (!has_environment(fr)) return finish();
    Environment env;
    env.load(*fr.sc, fr.pc.gEnvironmentMaterialAddress);
    const float pdf = env.eval_pdf(path.direction, fr.flag(STHIP_eSampleEnvironmentMapDirectly));
    dbg[0] = dbg[1] = dbg[2] = pdf;
    return finish();
  }
  path.beta = V3(1.0f);
  path.medium = (has_media(fr) && fr.fd.gViewMediumInstances) ? fr.fd.gViewMediumInstances[view_index] : STHIP_INVALID_INSTANCE;  // bdpt.hlsl:208
  path.trace();
  path.path_pdf = 1;  // bdpt.hlsl:213-220
  path.path_pdf_rev = 1;
  path.bsdf_pdf = 1;
  path.G = 1;
  path.dVC = 1;

  // bdpt.hlsl:222-223 (on a miss upstream reads normals nobody wrote; pinned to the packed value 0, as the visibility output is)
  if (dbg && (fr.debug(STHIP_DEBUG_GEOMETRY_NORMAL) || fr.debug(STHIP_DEBUG_SHADING_NORMAL))) {
    const bool miss = path.isect.instance_index() == STHIP_INVALID_INSTANCE || (has_media(fr) && path.isect.sd.shape_area == 0);  // (a first vertex inside a medium: stale normals upstream, pinned likewise)
    const v3 n = fr.debug(STHIP_DEBUG_GEOMETRY_NORMAL) ? (miss ? unpack_normal_octahedron(0) : path.isect.sd.geometry_normal()) : (miss ? unpack_normal_octahedron(0) : path.isect.sd.shading_normal());
    path.debug_set(n * .5f + V3(.5f));
  }
  sthip_VisibilityInfo vis;
  vis.instance_primitive_index = path.isect.instance_primitive_index;
  vis.packed_normal = path.isect.sd.packed_shading_normal;
  if (path.isect.instance_index() == STHIP_INVALID_INSTANCE) {
    if (aov) {
      aov->albedo[0] = aov->albedo[1] = aov->albedo[2] = aov->albedo[3] = 1;
      vis.packed_normal = 0;  // uninitialised in the reference on a miss (sd is not written); pinned to 0
      aov->vis = vis;
      aov->depth.z = POS_INF;
      aov->depth.prev_z = POS_INF;
      aov->depth.dz_dxy[0] = aov->depth.dz_dxy[1] = 0;
      aov->prev_uv[0] = uvx;
      aov->prev_uv[1] = uvy;
    }
    if (has_environment(fr)) {  // bdpt.hlsl:236-240
      Environment env;
      env.load(*fr.sc, fr.pc.gEnvironmentMaterialAddress);
      path.eval_emission(env.eval(path.direction));
    }
  } else {
    if (!has_media(fr) || path.isect.sd.shape_area > 0) {  // bdpt.hlsl:245: a vertex inside a medium has no albedo / emission
      DisneyMaterial m;
      ShadingData tmp_sd = path.isect.sd;  // bdpt.hlsl:246-251: the first-hit lookup works on a copy
      m.load(*fr.sc, fr.sc->instances[path.isect.instance_index()].material_address(), tmp_sd, fr.sampling_flags);
      vis.packed_normal = tmp_sd.packed_shading_normal;
      path.eval_emission(m.Le());
      if (aov) {
        const v3 a = m.albedo();
        aov->albedo[0] = a.x;
        aov->albedo[1] = a.y;
        aov->albedo[2] = a.z;
        aov->albedo[3] = 1;
      }
      if (dbg) {  // bdpt.hlsl:257-260
        if (fr.debug(STHIP_DEBUG_ALBEDO)) path.debug_set(m.albedo());
        else if (fr.debug(STHIP_DEBUG_SPECULAR)) path.debug_set(V3(m.is_specular() ? 1.0f : 0.0f));
        else if (fr.debug(STHIP_DEBUG_EMISSION)) path.debug_set(m.Le());
        else if (fr.debug(STHIP_DEBUG_SHADING_NORMAL)) path.debug_set(tmp_sd.shading_normal() * .5f + V3(.5f));
      }
    }
    const bool medium_vertex = has_media(fr) && path.isect.sd.shape_area == 0;
    if (medium_vertex) vis.packed_normal = 0;  // upstream stores the stale normal of the last query here; pinned to 0
    if (aov) {
      aov->vis = vis;
      const sthip_TransformData& prev_inv_view = fr.fd.gPrevInverseViewTransforms ? fr.fd.gPrevInverseViewTransforms[view_index] : fr.fd.gInverseViewTransforms[view_index];
      const sthip_ViewData& prev_view = fr.fd.gPrevViews ? fr.fd.gPrevViews[view_index] : view;
      const v3 prev_cam_pos = transform_point(tmul(prev_inv_view, fr.sc->motion_xf[path.isect.instance_index()]), path.isect.sd.position);
      aov->depth.z = length(path.isect.sd.position - path.origin);
      aov->depth.prev_z = length(prev_cam_pos);
      const v3 gn = path.isect.sd.geometry_normal();
      const v3 dir_x = primary_dir(view, t, (float)(x + 1), (float)y, nullptr);
      aov->depth.dz_dxy[0] = ray_plane(path.origin - path.isect.sd.position, dir_x, gn) - aov->depth.z;
      const v3 dir_y = primary_dir(view, t, (float)x, (float)(y + 1), nullptr);
      aov->depth.dz_dxy[1] = ray_plane(path.origin - path.isect.sd.position, dir_y, gn) - aov->depth.z;
      if (medium_vertex) aov->depth.dz_dxy[0] = aov->depth.dz_dxy[1] = 0;  // ... and the stale geometry normal here; pinned to 0
      float pc[4];
      project_point(prev_view.projection, prev_cam_pos, pc);
      pc[1] = -pc[1];
      pc[0] = pc[0] / pc[3];
      pc[1] = pc[1] / pc[3];
      aov->prev_uv[0] = pc[0] * .5f + .5f;
      aov->prev_uv[1] = pc[1] * .5f + .5f;
      if (dbg && fr.debug(STHIP_DEBUG_PREV_UV))  // bdpt.hlsl:294-295
        path.debug_set(V3(fabsf(aov->prev_uv[0] - uvx) * (float)fr.pc.gOutputExtent[0], fabsf(aov->prev_uv[1] - uvy) * (float)fr.pc.gOutputExtent[1], 0.0f));
    }
    while (any_gt0(path.beta) && !any_nan(path.beta)) path.next_vertex();
  }

  return finish();
}

template <typename F>
void parallel_rows(uint32_t rows, int threads, F fn) {
  if (threads <= 1) {
    for (uint32_t r = 0; r < rows; r++) fn(r, 0);
    return;
  }
  std::atomic<uint32_t> next{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; t++)
    pool.emplace_back([&, t]() {
      for (;;) {
        const uint32_t r = next.fetch_add(1);
        if (r >= rows) break;
        fn(r, t);
      }
    });
  for (auto& th : pool) th.join();
}

}  // namespace

// =============================================================================================
// C entry points (used by tests/, smoke() and bench.py's cpu_baseline only)
// =============================================================================================
extern "C" {

orc_scene* orc_scene_create(const sthip_scene_desc* d) {
  if (!d || !d->gInstances || !d->gInstanceTransforms || !d->gInstanceInverseTransforms || !d->gMaterialData) return nullptr;
  if ((d->vertex_count && !d->gVertices) || (d->indices_bytes && !d->gIndices)) return nullptr;
  orc_scene* sc = new orc_scene();
  if (d->vertex_count) sc->vertices.assign(d->gVertices, d->gVertices + d->vertex_count);
  if (d->indices_bytes) sc->indices.assign((const uint8_t*)d->gIndices, (const uint8_t*)d->gIndices + d->indices_bytes);
  sc->indices.resize(sc->indices.size() + 8, 0);  // Load2 at the tail of a 16-bit index buffer
  sc->instances.resize(d->instance_count);
  for (uint32_t i = 0; i < d->instance_count; i++) sc->instances[i].d = d->gInstances[i];
  sc->xf.assign(d->gInstanceTransforms, d->gInstanceTransforms + d->instance_count);
  sc->inv_xf.assign(d->gInstanceInverseTransforms, d->gInstanceInverseTransforms + d->instance_count);
  if (d->gInstanceMotionTransforms)
    sc->motion_xf.assign(d->gInstanceMotionTransforms, d->gInstanceMotionTransforms + d->instance_count);
  else {
    sthip_TransformData I;
    memset(&I, 0, sizeof(I));
    I.m[0][0] = I.m[1][1] = I.m[2][2] = 1;
    sc->motion_xf.assign(d->instance_count, I);
  }
  sc->materials.assign((const uint8_t*)d->gMaterialData, (const uint8_t*)d->gMaterialData + d->material_bytes);
  if (d->gLightInstances) sc->lights.assign(d->gLightInstances, d->gLightInstances + d->light_count);
  if (d->gDistributions) sc->distributions.assign(d->gDistributions, d->gDistributions + d->distribution_count);
  for (uint32_t i = 0; i < d->image1_count && d->gImage1s; i++) {
    OrcImage1 im;
    im.w = d->gImage1s[i].width;
    im.h = d->gImage1s[i].height;
    im.px.assign(d->gImage1s[i].pixels, d->gImage1s[i].pixels + (size_t)im.w * im.h);
    sc->images1.push_back(std::move(im));
  }
  for (uint32_t i = 0; i < d->image_count && d->gImages; i++) {
    OrcImage im;
    uint32_t w = d->gImages[i].width, h = d->gImages[i].height;
    im.w.push_back(w);
    im.h.push_back(h);
    im.mip.emplace_back(d->gImages[i].pixels, d->gImages[i].pixels + (size_t)w * h * 4);
    while (w > 1 || h > 1) {
      const uint32_t nw = std::max(1u, w / 2), nh = std::max(1u, h / 2);
      std::vector<float> next((size_t)nw * nh * 4);
      const std::vector<float>& prev = im.mip.back();
      for (uint32_t y = 0; y < nh; y++)
        for (uint32_t x = 0; x < nw; x++) {
          const uint32_t x0 = std::min(2 * x, w - 1), x1 = std::min(2 * x + 1, w - 1), y0 = std::min(2 * y, h - 1), y1 = std::min(2 * y + 1, h - 1);
          for (int k = 0; k < 4; k++) {
            const float a = prev[4 * ((size_t)y0 * w + x0) + k], b = prev[4 * ((size_t)y0 * w + x1) + k];
            const float c = prev[4 * ((size_t)y1 * w + x0) + k], e = prev[4 * ((size_t)y1 * w + x1) + k];
            next[4 * ((size_t)y * nw + x) + k] = ((a + b) + (c + e)) * 0.25f;
          }
        }
      im.mip.push_back(std::move(next));
      w = nw;
      h = nh;
      im.w.push_back(w);
      im.h.push_back(h);
    }
    sc->images.push_back(std::move(im));
  }

  for (uint32_t i = 0; i < d->volume_count; i++) {  // gVolumes: NanoVDB float grids
    NvdbGrid g;
    const uint8_t* b = (const uint8_t*)d->gVolumes[i].data;
    g.bytes.assign(b, b + d->gVolumes[i].bytes);
    if (!g.init()) {
      delete sc;
      return nullptr;
    }
    sc->volumes.push_back(std::move(g));
  }

  // one BLAS per unique mesh range
  std::map<std::tuple<uint32_t, uint32_t, uint32_t, uint32_t>, uint32_t> mesh_of;
  sc->inst_mesh.assign(d->instance_count, 0);
  sc->inst_identity.assign(d->instance_count, 0);
  std::vector<Aabb> inst_boxes(d->instance_count);
  for (uint32_t i = 0; i < d->instance_count; i++) {
    const Inst& in = sc->instances[i];
    sc->inst_identity[i] = is_identity(sc->inv_xf[i]) ? 1 : 0;
    inst_boxes[i].reset();
    if (in.type() == STHIP_INSTANCE_TYPE_SPHERE) {  // the sphere's world box, generously padded (sphere_test decides)
      const float r = fabsf(in.radius());
      for (int a = 0; a < 3; a++) {
        const float c = sc->xf[i].m[a][3];
        const float pad = 1e-3f * r + 1e-5f * fabsf(c);
        inst_boxes[i].lo[a] = c - r - pad;
        inst_boxes[i].hi[a] = c + r + pad;
      }
      continue;
    }
    if (in.type() == STHIP_INSTANCE_TYPE_VOLUME) {  // the world box of the grid's root bounding box, padded (volume_test decides)
      if (in.volume_index() >= sc->volumes.size()) {
        delete sc;
        return nullptr;
      }
      const NvdbGrid& g = sc->volumes[in.volume_index()];
      for (int c = 0; c < 8; c++) {
        const v3 ip = V3((float)((c & 1) ? g.bbox_max[0] + 1 : g.bbox_min[0]), (float)((c & 2) ? g.bbox_max[1] + 1 : g.bbox_min[1]), (float)((c & 4) ? g.bbox_max[2] + 1 : g.bbox_min[2]));
        const v3 wp = transform_point(sc->xf[i], g.index_to_world(ip));
        const float w[3] = {wp.x, wp.y, wp.z};
        inst_boxes[i].grow(w);
      }
      for (int a = 0; a < 3; a++) {
        const float pad = 1e-3f * (inst_boxes[i].hi[a] - inst_boxes[i].lo[a]) + 1e-5f * std::max(fabsf(inst_boxes[i].lo[a]), fabsf(inst_boxes[i].hi[a]));
        inst_boxes[i].lo[a] -= pad;
        inst_boxes[i].hi[a] += pad;
      }
      continue;
    }
    if (in.type() != STHIP_INSTANCE_TYPE_TRIANGLES) continue;
    auto key = std::make_tuple(in.first_vertex(), in.indices_byte_offset(), in.prim_count(), in.index_stride());
    auto it = mesh_of.find(key);
    if (it == mesh_of.end()) {
      Mesh m;
      m.first_vertex = in.first_vertex();
      m.indices_byte_offset = in.indices_byte_offset();
      m.prim_count = in.prim_count();
      m.stride = in.index_stride();
      std::vector<Aabb> boxes(in.prim_count());
      for (uint32_t p = 0; p < in.prim_count(); p++) {
        uint32_t tri[3];
        sc->load_tri(in, p, tri);
        boxes[p].reset();
        for (int k = 0; k < 3; k++) boxes[p].grow(sc->vertices[tri[k]].position);
      }
      m.bvh.build(boxes, 4);
      it = mesh_of.emplace(key, (uint32_t)sc->meshes.size()).first;
      sc->meshes.push_back(std::move(m));
    }
    sc->inst_mesh[i] = it->second;
    // world box from transformed vertices of the mesh (exact hull of the triangles)
    for (uint32_t p = 0; p < in.prim_count(); p++) {
      uint32_t tri[3];
      sc->load_tri(in, p, tri);
      for (int k = 0; k < 3; k++) {
        const float* q = sc->vertices[tri[k]].position;
        const v3 w = transform_point(sc->xf[i], V3(q[0], q[1], q[2]));
        const float wp[3] = {w.x, w.y, w.z};
        inst_boxes[i].grow(wp);
      }
    }
    // the traversal ray lives in object space through Minv, the box in world space through M:
    // pad generously so that M*Minv != I rounding cannot cull a hit
    for (int a = 0; a < 3; a++) {
      const float m = std::max(std::max(fabsf(inst_boxes[i].lo[a]), fabsf(inst_boxes[i].hi[a])), inst_boxes[i].hi[a] - inst_boxes[i].lo[a]);
      inst_boxes[i].lo[a] -= 1e-4f * m;
      inst_boxes[i].hi[a] += 1e-4f * m;
    }
  }
  sc->tlas.build(inst_boxes, 1);
  return sc;
}

void orc_scene_destroy(orc_scene* sc) { delete sc; }

// sample_photons, bdpt.hlsl:101-147: one light subpath per thread of dispatch_over(W, ceil(gLightPathCount / W)) in 8x4
// groups (threads of the padding columns run too when their path index is below gLightPathCount, as upstream)
void trace_light_paths(const Frame& fr, uint32_t seed, int threads, uint64_t* tstats) {
  const uint32_t W = fr.pc.gOutputExtent[0];
  const uint32_t count = fr.pc.gLightPathCount;
  const uint32_t rows = (count + W - 1) / W;
  const uint32_t gw = (W + 7) / 8, gh = (rows + 3) / 4;
  parallel_rows(gh * 4, threads, [&](uint32_t y, int tid) {
    for (uint32_t x = 0; x < gw * 8; x++) {
      uint32_t path_index;
      if (fr.flag(STHIP_eRemapThreads))
        path_index = ((y / 4) * gw + (x / 8)) * 32 + (y % 4) * 8 + (x % 8);
      else
        path_index = y * W + x;
      if (path_index >= count) continue;
      PathIntegrator path(fr, x, y, seed, true);
      float rnd[4];
      for (float& r : rnd) r = path.rng.next_float();
      LightSampleRecord ls;
      sample_point_on_light(fr, ls, rnd, V3(0.0f));
      if (ls.pdf <= 0 || all_le0(ls.radiance)) continue;
      path.isect.instance_primitive_index = 0xFFFFFFFFu;
      path.isect.sd.position = ls.position;
      path.isect.sd.packed_geometry_normal = pack_normal_octahedron(ls.normal);
      path.isect.sd.shape_area = 1;
      path.path_contrib = ls.radiance;  // bdpt.hlsl:120
      path.beta = ls.radiance / ls.pdf;
      path.path_pdf = ls.pdf;
      path.path_pdf_rev = 1;
      path.dVC = 1 / ls.pdf;
      path.G = 1;
      path.prev_cos_out = 1;
      path.bsdf_pdf = ls.pdf;
      const float u1 = path.rng.next_float(), u2 = path.rng.next_float();
      const v3 local_dir_out = sample_cos_hemisphere(u1, u2);
      path.bsdf_pdf = cosine_hemisphere_pdfW(local_dir_out.z);
      path.beta = path.beta * (local_dir_out.z / path.bsdf_pdf);
      path.path_contrib = path.path_contrib * local_dir_out.z;  // bdpt.hlsl:135
      path.prev_cos_out = local_dir_out.z;
      v3 T, B;
      make_orthonormal(ls.normal, T, B);
      path.direction = T * local_dir_out.x + B * local_dir_out.y + ls.normal * local_dir_out.z;
      path.origin = ray_offset(ls.position, ls.normal);
      path.trace();
      while (any_gt0(path.beta) && !any_nan(path.beta)) path.next_vertex();
      uint64_t* st = tstats + (size_t)tid * 4;
      st[0] += path.rays_total;
      st[1] += path.rays_path;
      st[2] += path.counters[0];
      st[3] += path.counters[1];
    }
  });
}

// stats_out[4]: rays_total (gRayCount[0]), rays_path (gRayCount[1]), nodes visited, triangles tested
int orc_render_window(orc_scene* sc, const sthip_BDPTPushConstants* pc, uint32_t sampling_flags, uint32_t scene_flags, const sthip_frame_desc* frame, uint32_t seed_begin,
                      uint32_t seed_count, const sthip_outputs* out, int threads, uint64_t* stats_out, const uint32_t* window);
int orc_render(orc_scene* sc, const sthip_BDPTPushConstants* pc, uint32_t sampling_flags, uint32_t scene_flags, const sthip_frame_desc* frame, uint32_t seed_begin,
               uint32_t seed_count, const sthip_outputs* out, int threads, uint64_t* stats_out) {
  return orc_render_window(sc, pc, sampling_flags, scene_flags, frame, seed_begin, seed_count, out, threads, stats_out, nullptr);
}
// window = {x0, y0, x1, y1} (may be NULL: the whole frame): only the pixels of that rectangle of the W x H frame are
// rendered — exactly the pixels the full frame has there (same pixel coordinates, hence the same rays and RNG keys); the
// others are left as the caller passed them. For the parity tests of frames too large for the oracle to render whole.
int orc_render_window(orc_scene* sc, const sthip_BDPTPushConstants* pc, uint32_t sampling_flags, uint32_t scene_flags, const sthip_frame_desc* frame, uint32_t seed_begin,
                      uint32_t seed_count, const sthip_outputs* out, int threads, uint64_t* stats_out, const uint32_t* window) {
  if (!sc || !pc || !frame || !out || !out->gRadiance || !frame->gViews || !frame->gViewTransforms) return STHIP_ERR_INVALID_ARGUMENT;
  if (scene_flags & STHIP_BDPT_FLAG_TRACE_LIGHT) return STHIP_ERR_UNSUPPORTED;

  if ((scene_flags & STHIP_BDPT_FLAG_HAS_ENVIRONMENT) && (size_t)pc->gEnvironmentMaterialAddress + 16 > sc->materials.size()) return STHIP_ERR_INVALID_ARGUMENT;
  const uint32_t unsupported = (1u << STHIP_eSampleLightPower);
  if (sampling_flags & unsupported) return STHIP_ERR_UNSUPPORTED;
  Frame fr;
  fr.sc = sc;
  fr.pc = *pc;
  fr.sampling_flags = sampling_flags;
  fr.scene_flags = scene_flags;
  fr.fd = *frame;
  fr.debug_mode = out->gDebugImage ? out->debug_mode : 0u;
  if (fr.debug_mode >= STHIP_DEBUG_MODE_COUNT) return STHIP_ERR_INVALID_ARGUMENT;
  // BDPT.cpp:488-509
  if (pc->gLightCount == 0) fr.scene_flags &= ~STHIP_BDPT_FLAG_HAS_EMISSIVES;
  if (!has_environment(fr)) fr.pc.gEnvironmentSampleProbability = 0;
  if (!has_emissives(fr)) fr.pc.gEnvironmentSampleProbability = 1;
  if (!has_emissives(fr) && !has_environment(fr)) fr.sampling_flags &= ~(1u << STHIP_eNEE);

  if (fr.pc.gLightCount > sc->lights.size()) return STHIP_ERR_INVALID_ARGUMENT;
  if (!fr.flag(STHIP_eNEE)) fr.sampling_flags &= ~((1u << STHIP_ePresampleLights) | (1u << STHIP_eNEEReservoirs) | (1u << STHIP_eNEEReservoirReuse));  // BDPT.cpp:511-515
  if (!fr.flag(STHIP_eNEEReservoirs)) fr.sampling_flags &= ~(1u << STHIP_eNEEReservoirReuse);  // only connect_light_reservoir touches the grid
  if (fr.flag(STHIP_eNEEReservoirReuse) && has_environment(fr)) return STHIP_ERR_UNSUPPORTED;  // a stored environment sample is re-read as a surface point upstream (sample_Le leaves pdfA positive)
  if (!has_emissives(fr) && !has_environment(fr)) fr.sampling_flags &= ~((1u << STHIP_eConnectToViews) | (1u << STHIP_eConnectToLightPaths));  // BDPT.cpp:504-509
  if (!fr.flag(STHIP_eLVC)) fr.sampling_flags &= ~((1u << STHIP_eLVCReservoirs) | (1u << STHIP_eLVCReservoirReuse));  // BDPT.cpp:517-520
  if (!fr.flag(STHIP_eConnectToLightPaths)) fr.sampling_flags &= ~((1u << STHIP_eLVC) | (1u << STHIP_eLVCReservoirs) | (1u << STHIP_eLVCReservoirReuse));  // the cache is only read by connect_lvc
  if (!fr.flag(STHIP_eLVCReservoirs)) fr.sampling_flags &= ~(1u << STHIP_eLVCReservoirReuse);  // the reuse sits inside the reservoir branch of connect_lvc
  if (!fr.flag(STHIP_eNEE) && !fr.flag(STHIP_eLVC)) fr.sampling_flags &= ~(1u << STHIP_eDeferShadowRays);  // BDPT.cpp:522-523
  // eCoherentSampling only touches the index of a presampled light (path.hlsli:317,379) and connect_lvc's (:688,703)
  if (!fr.flag(STHIP_ePresampleLights) && !fr.flag(STHIP_eLVC)) fr.sampling_flags &= ~(1u << STHIP_eCoherentSampling);
  if (fr.flag(STHIP_eCoherentSampling) && (scene_flags & STHIP_BDPT_FLAG_HAS_MEDIA) && !sc->volumes.empty()) return STHIP_ERR_UNSUPPORTED;  // walks through volumes break the lockstep
  if (has_media(fr)) {
    // with media every visibility ray draws random numbers from the stream it is given: the path's own for an inline NEE ray
    // (path.hlsli:329-332, 474-479), the light path's for connect_view (:577-581), the view path's for the connections to light vertices (:791-796,814-818).
    if (sc->volumes.empty()) fr.scene_flags &= ~STHIP_BDPT_FLAG_HAS_MEDIA;
  }
  // presample_lights, bdpt.hlsl:84-99, once per seed (BDPT.cpp:644-651): rng_init(-1, index), reference point 0.
  // An environment sample leaves `position` unset upstream, so that combination is not restated.
  std::vector<std::vector<PresampledLightPoint>> presampled;
  if (fr.flag(STHIP_ePresampleLights) && has_environment(fr)) return STHIP_ERR_UNSUPPORTED;
  if (fr.flag(STHIP_ePresampleLights) && fr.pc.gMaxPathVertices > 2) {
    const size_t n = (size_t)fr.pc.gLightPresampleTileSize * fr.pc.gLightPresampleTileCount;
    if (n == 0 || n * seed_count > (1u << 26)) return STHIP_ERR_INVALID_ARGUMENT;
    presampled.resize(seed_count);
    for (uint32_t s = 0; s < seed_count; s++) {
      presampled[s].resize(n);
      for (size_t i = 0; i < n; i++) {
        Rng rng;
        rng.v[0] = rng.v[1] = 0xFFFFFFFFu;
        rng.v[2] = seed_begin + s;
        rng.v[3] = (uint32_t)i;
        float rnd[4];
        for (float& r : rnd) r = rng.next_float();
        LightSampleRecord ls;
        sample_point_on_light(fr, ls, rnd, V3(0.0f));
        PresampledLightPoint& l = presampled[s][i];
        l.position = ls.position;
        l.packed_geometry_normal = pack_normal_octahedron(ls.normal);
        l.Le = ls.radiance;
        l.pdfA = ls.is_environment ? -ls.pdf : ls.pdf;
      }
    }
    fr.presampled = presampled.data();
    fr.seed_begin = seed_begin;
  }
  const uint32_t W = pc->gOutputExtent[0], H = pc->gOutputExtent[1];
  if (threads <= 0) threads = (int)std::max(1u, std::thread::hardware_concurrency());
  std::vector<uint64_t> tstats((size_t)threads * 4, 0);
  // light tracing (eConnectToViews, BDPT.cpp:653-667,740-748): per seed, sample_photons fills gLightTraceSamples before
  // the view paths run, add_light_trace adds it to gRadiance afterwards
  if (!has_emissives(fr) && !has_environment(fr)) fr.sampling_flags &= ~((1u << STHIP_eConnectToViews) | (1u << STHIP_eConnectToLightPaths));  // BDPT.cpp:504-509
  const bool light_tracing = fr.bdpt() && fr.pc.gMaxPathVertices > 2;
  std::vector<std::vector<std::atomic<uint32_t>>> light_images;
  std::vector<std::vector<PathVertex>> light_vertices;
  std::vector<uint32_t> lvc_counts(seed_count, 0);
  if (fr.bdpt()) {
    if (has_environment(fr)) return STHIP_ERR_UNSUPPORTED;  // env light paths start from an unset position upstream
    if (!frame->gInverseViewTransforms) return STHIP_ERR_INVALID_ARGUMENT;
    if ((uint64_t)seed_count * W * H > (1ull << 26)) return STHIP_ERR_INVALID_ARGUMENT;
    // without eRemapThreads the padding columns of sample_photons alias the next row's vertex slots (a write race upstream)
    if (fr.flag(STHIP_eConnectToLightPaths) && !fr.flag(STHIP_eRemapThreads) && W % 8 != 0) return STHIP_ERR_UNSUPPORTED;
  }
  if (fr.bdpt()) {
    light_images = std::vector<std::vector<std::atomic<uint32_t>>>(seed_count);
    if (fr.flag(STHIP_eConnectToLightPaths)) light_vertices.resize(seed_count);
    for (uint32_t s = 0; s < seed_count; s++) {
      light_images[s] = std::vector<std::atomic<uint32_t>>((size_t)W * H * 4);
      for (auto& a : light_images[s]) a.store(0, std::memory_order_relaxed);
      Frame lf = fr;
      lf.light_trace = light_images[s].data();
      std::vector<PathVertex> staging;
      if (fr.flag(STHIP_eConnectToLightPaths)) {  // BDPT.cpp:569-572
        light_vertices[s].assign((size_t)fr.pc.gLightPathCount * fr.pc.gMaxDiffuseVertices, PathVertex{});
        lf.light_vertices = light_vertices[s].data();
        lf.light_vertex_count = light_vertices[s].size();
        if (fr.lvc()) {
          if (fr.pc.gMaxDiffuseVertices < 2 || (uint64_t)fr.pc.gLightPathCount * fr.pc.gMaxDiffuseVertices > (1ull << 28)) return STHIP_ERR_INVALID_ARGUMENT;
          staging.assign((size_t)fr.pc.gLightPathCount * (fr.pc.gMaxDiffuseVertices - 1), PathVertex{});
          lf.lvc_staging = staging.data();
        }
      }
      if (light_tracing) trace_light_paths(lf, seed_begin + s, threads, tstats.data());
      if (fr.lvc()) {  // the cache in its defined order: paths by index, a path's vertices as it stored them
        uint32_t n = 0;
        for (const PathVertex& v : staging)
          if (v.packed_beta[1] != 0) light_vertices[s][n++] = v;  // a stored vertex has subpath_length >= 2 in these bits
        lvc_counts[s] = n;
      }
    }
  }
  const uint32_t wx0 = window ? std::min(window[0], W) : 0u, wy0 = window ? std::min(window[1], H) : 0u;
  const uint32_t wx1 = window ? std::min(window[2], W) : W, wy1 = window ? std::min(window[3], H) : H;
  if (window && fr.bdpt()) return STHIP_ERR_UNSUPPORTED;  // light subpaths are a whole-frame pass
  // one sample of one pixel folded into that pixel's running mean (temporal_accumulation.hlsl:102-131: NaN/Inf samples
  // are dropped); returns false for a pixel outside every view
  auto seed_frame = [&](const Frame& base, uint32_t s) {
    Frame sf = base;
    if (fr.flag(STHIP_eConnectToViews)) sf.light_trace = light_images[s].data();
    if (fr.flag(STHIP_eConnectToLightPaths)) {
      sf.light_vertices = light_vertices[s].data();
      sf.light_vertex_count = light_vertices[s].size();
      sf.lvc_count = lvc_counts[s];
    }
    return sf;
  };
  auto fold = [&](uint32_t x, uint32_t y, uint32_t s, float acc[4], const float rgb[3], const PixelAOV& aov) {
    const size_t p = (size_t)y * W + x;
    float cur[4] = {rgb[0], rgb[1], rgb[2], 1};
    if (std::isinf(cur[0]) || std::isinf(cur[1]) || std::isinf(cur[2]) || cur[0] != cur[0] || cur[1] != cur[1] || cur[2] != cur[2]) cur[0] = cur[1] = cur[2] = cur[3] = 0;
    if (acc[3] > 0) {
      const float n = acc[3] + cur[3];
      const float alpha = fminf(fmaxf(cur[3] / n, 0.0f), 1.0f);
      for (int c = 0; c < 3; c++) acc[c] = lerpf(acc[c], cur[c], alpha);
      acc[3] = n;
    } else {
      for (int c = 0; c < 4; c++) acc[c] = cur[c];
    }
    if (s == 0) {
      if (out->gAlbedo) memcpy(out->gAlbedo + 4 * p, aov.albedo, 16);
      if (out->gVisibility) out->gVisibility[p] = aov.vis;
      if (out->gDepth) out->gDepth[p] = aov.depth;
      if (out->gPrevUVs) memcpy(out->gPrevUVs + 2 * p, aov.prev_uv, 8);
    }
  };
  const bool want_aovs = out->gAlbedo || out->gVisibility || out->gDepth || out->gPrevUVs;
  auto sample_pixel = [&](const Frame& base, uint32_t x, uint32_t y, uint32_t s, float acc[4], int tid) -> bool {
    float rgb[3];
    PixelAOV aov;
    memset(&aov, 0, sizeof(aov));  // fields a first vertex inside a medium leaves unwritten (albedo) read as zero
    const Frame sf = seed_frame(base, s);
    float* dbg = (out->gDebugImage && out->debug_mode) ? out->gDebugImage + 4 * ((size_t)y * W + x) : nullptr;  // (this thread owns the pixel, seed after seed: upstream's frame after frame)
    if (!render_pixel(sf, x, y, seed_begin + s, rgb, (s == 0 && want_aovs) ? &aov : nullptr, &tstats[(size_t)tid * 4], nullptr, dbg)) return false;
    fold(x, y, s, acc, rgb, aov);
    return true;
  };
  // eCoherentRR: one sample of the 32 pixels of the reference's 8x4 workgroup (gx, gy), which share their Russian-roulette
  // decisions (PathIntegrator::russian_roulette). The group is rendered with the decisions known so far; a path that
  // needs one more reports it and stops; the decision is worked out from the lanes that reported — p = the largest p,
  // the verdict of the first lane (lowest y * 8 + x in the group) — and the group is rendered again. With the usual
  // limits on diffuse scenes no path ever reaches the roulette and the first rendering is the final one. acc_of(x, y)
  // gives a pixel's running mean (or null: the pixel takes part in the decisions but is not an output of this call).
  const bool coherent_rr = (fr.flag(STHIP_eCoherentRR) || fr.flag(STHIP_eCoherentSampling)) && !has_media(fr);  // rendered group by group
  auto sample_group = [&](const Frame& base, uint32_t gx, uint32_t gy, uint32_t s, const std::function<float*(uint32_t, uint32_t)>& acc_of, int tid) {
    PathIntegrator::GroupRR decisions[256];
    PathIntegrator::GroupValue nee_values[256], lvc_values[256];  // eCoherentSampling: the first lane's index per site and path length
    const Frame sf = seed_frame(base, s);
    struct Lane {
      bool inside;
      float rgb[3];
      PixelAOV aov;
      uint64_t stats[4];
      float dbg[4];  // the pixel of gDebugImage as this rendering of the group leaves it (only the last rendering's is stored)
    } lanes[32];
    for (;;) {
      int missing = -1;
      RRControl ctl[32];
      for (uint32_t l = 0; l < 32; l++) {
        const uint32_t x = gx * 8 + (l & 7), y = gy * 4 + (l >> 3);
        lanes[l].inside = false;
        if (x >= W || y >= H) continue;
        memset(&lanes[l].aov, 0, sizeof(PixelAOV));
        memset(lanes[l].stats, 0, sizeof(lanes[l].stats));
        ctl[l].decisions = fr.flag(STHIP_eCoherentRR) ? decisions : nullptr;
        ctl[l].nee = fr.flag(STHIP_eCoherentSampling) ? nee_values : nullptr;
        ctl[l].lvc = fr.flag(STHIP_eCoherentSampling) ? lvc_values : nullptr;
        const bool debugging = out->gDebugImage && out->debug_mode;
        if (debugging) memcpy(lanes[l].dbg, out->gDebugImage + 4 * ((size_t)y * W + x), 16);
        lanes[l].inside = render_pixel(sf, x, y, seed_begin + s, lanes[l].rgb, (s == 0 && want_aovs) ? &lanes[l].aov : nullptr, lanes[l].stats, &ctl[l], debugging ? lanes[l].dbg : nullptr);
      }
      // the earliest statement some lane could not execute: by path length, then in the order of a vertex (roulette, NEE index,
      // connect_lvc's index). A lane reports at most one.
      int key = -1;
      for (uint32_t l = 0; l < 32; l++) {
        if (!lanes[l].inside) continue;
        int k = -1;
        if (ctl[l].missing_length >= 0) k = ctl[l].missing_length * 4;
        if (ctl[l].cs_length >= 0) k = ctl[l].cs_length * 4 + 1 + ctl[l].cs_site;
        if (k >= 0 && (key < 0 || k < key)) key = k;
      }
      missing = key < 0 ? -1 : key >> 2;
      if (missing < 0) break;
      const int what = key & 3;
      if (what == 0) {
        PathIntegrator::GroupRR d;
        d.valid = true;
        d.p_max = -1;
        int first = -1;
        for (uint32_t l = 0; l < 32; l++)
          if (lanes[l].inside && ctl[l].missing_length == missing) {
            if (first < 0) first = (int)l;
            d.p_max = std::max(d.p_max, ctl[l].missing_p);  // WaveActiveMax
          }
        d.kill = d.p_max < 1 && ctl[first].missing_rnd > d.p_max;  // WaveReadLaneFirst(rnd > p)
        decisions[missing & 255] = d;
      } else {
        PathIntegrator::GroupValue v;
        for (uint32_t l = 0; l < 32 && !v.valid; l++)
          if (lanes[l].inside && ctl[l].cs_length == missing && ctl[l].cs_site == what - 1) {
            v.valid = true;
            v.value = ctl[l].cs_value;  // WaveReadLaneFirst
          }
        (what == 1 ? nee_values : lvc_values)[missing & 255] = v;
      }
    }
    for (uint32_t l = 0; l < 32; l++) {
      const uint32_t x = gx * 8 + (l & 7), y = gy * 4 + (l >> 3);
      if (x >= W || y >= H) continue;
      float* acc = acc_of(x, y);
      if (!acc) continue;
      // (a pixel outside every view has no path, but add_light_trace may have written its debug pixel: render_pixel)
      if (out->gDebugImage && out->debug_mode) memcpy(out->gDebugImage + 4 * ((size_t)y * W + x), lanes[l].dbg, 16);
      if (!lanes[l].inside) continue;
      for (int k = 0; k < 4; k++) tstats[(size_t)tid * 4 + k] += lanes[l].stats[k];
      fold(x, y, s, acc, lanes[l].rgb, lanes[l].aov);
    }
  };
  if (fr.flag(STHIP_eNEEReservoirReuse) || fr.flag(STHIP_eLVCReservoirReuse)) {
    // Reservoir reuse couples the seeds of a call: seed s looks into the hash grid that seed s - 1 built (the reference's
    // frame and previous frame; the first seed of a call has no previous frame: gReservoirSpatialM = 0, BDPT.cpp:482-483).
    // So the frame is rendered seed by seed, and after each seed its appends — staged per (path, vertex) — are put into
    // the grid in that order (see HashGridTable).
    if (window) return STHIP_ERR_UNSUPPORTED;
    if (fr.pc.gHashGridBucketCount == 0 || fr.pc.gHashGridBucketCount > (1u << 28) || !(fr.pc.gHashGridMinBucketRadius > 0)) return STHIP_ERR_INVALID_ARGUMENT;
    const size_t path_slots = (size_t)((W + 7) / 8) * ((H + 3) / 4) * 32;  // covers both map_pixel_coord forms (bdpt_util.hlsli:76-83)
    const uint32_t D = std::max(1u, fr.pc.gMaxDiffuseVertices);
    const bool nee_reuse = fr.flag(STHIP_eNEEReservoirReuse), lvc_reuse = fr.flag(STHIP_eLVCReservoirReuse);
    std::vector<HashGridOf<NEEReservoir>::Append> staged(nee_reuse ? path_slots * D : 0);
    std::vector<HashGridOf<PathVertexReservoir>::Append> staged_lvc(lvc_reuse ? path_slots * D : 0);
    std::vector<uint8_t> valid(staged.size()), valid_lvc(staged_lvc.size());
    HashGridOf<NEEReservoir> grids[2];
    HashGridOf<PathVertexReservoir> grids_lvc[2];
    const HashGridOf<NEEReservoir>* prev = nullptr;
    const HashGridOf<PathVertexReservoir>* prev_lvc = nullptr;
    std::vector<float> accs((size_t)W * H * 4, 0.0f);
    for (uint32_t s = 0; s < seed_count; s++) {
      std::fill(valid.begin(), valid.end(), 0);
      std::fill(valid_lvc.begin(), valid_lvc.end(), 0);
      Frame base = fr;
      if (nee_reuse) {
        base.prev_nee_grid = prev;
        base.nee_appends = staged.data();
        base.nee_append_valid = valid.data();
      }
      if (lvc_reuse) {
        base.prev_lvc_grid = prev_lvc;
        base.lvc_appends = staged_lvc.data();
        base.lvc_append_valid = valid_lvc.data();
      }
      if (coherent_rr)
        parallel_rows((H + 3) / 4, threads, [&](uint32_t gy, int tid) {
          for (uint32_t gx = 0; gx < (W + 7) / 8; gx++) sample_group(base, gx, gy, s, [&](uint32_t x, uint32_t y) { return &accs[4 * ((size_t)y * W + x)]; }, tid);
        });
      else
        parallel_rows(H, threads, [&](uint32_t y, int tid) {
          for (uint32_t x = 0; x < W; x++) sample_pixel(base, x, y, s, &accs[4 * ((size_t)y * W + x)], tid);
        });
      if (nee_reuse) {
        std::vector<HashGridOf<NEEReservoir>::Append> appends;
        for (size_t k = 0; k < staged.size(); k++)
          if (valid[k]) appends.push_back(staged[k]);
        grids[s & 1].build(fr.pc.gHashGridBucketCount, appends);
        prev = &grids[s & 1];
      }
      if (lvc_reuse) {
        std::vector<HashGridOf<PathVertexReservoir>::Append> appends;
        for (size_t k = 0; k < staged_lvc.size(); k++)
          if (valid_lvc[k]) appends.push_back(staged_lvc[k]);
        grids_lvc[s & 1].build(fr.pc.gHashGridBucketCount, appends);
        prev_lvc = &grids_lvc[s & 1];
      }
    }
    memcpy(out->gRadiance, accs.data(), accs.size() * 4);
  } else if (coherent_rr) {
    // group by group (the groups that touch the window; their pixels outside it take part in the decisions only)
    const uint32_t gy0 = wy0 / 4, gy1 = (wy1 + 3) / 4, gx0 = wx0 / 8, gx1 = (wx1 + 7) / 8;
    parallel_rows(gy1 > gy0 ? gy1 - gy0 : 0u, threads, [&](uint32_t row, int tid) {
      const uint32_t gy = gy0 + row;
      for (uint32_t gx = gx0; gx < gx1; gx++) {
        float accs[32][4];
        memset(accs, 0, sizeof(accs));
        for (uint32_t s = 0; s < seed_count; s++)
          sample_group(fr, gx, gy, s, [&](uint32_t x, uint32_t y) -> float* { return (x >= wx0 && x < wx1 && y >= wy0 && y < wy1) ? accs[(y - gy * 4) * 8 + (x - gx * 8)] : nullptr; }, tid);
        for (uint32_t l = 0; l < 32; l++) {
          const uint32_t x = gx * 8 + (l & 7), y = gy * 4 + (l >> 3);
          if (x >= wx0 && x < wx1 && y >= wy0 && y < wy1) memcpy(out->gRadiance + 4 * ((size_t)y * W + x), accs[l], 16);
        }
      }
    });
  } else
  parallel_rows(wy1 > wy0 ? wy1 - wy0 : 0u, threads, [&](uint32_t row, int tid) {
    const uint32_t y = wy0 + row;
    for (uint32_t x = wx0; x < wx1; x++) {
      float acc[4] = {0, 0, 0, 0};
      for (uint32_t s = 0; s < seed_count; s++)
        if (!sample_pixel(fr, x, y, s, acc, tid)) break;
      memcpy(out->gRadiance + 4 * ((size_t)y * W + x), acc, 16);
    }
  });
  uint64_t tot[4] = {0, 0, 0, 0};
  for (int t = 0; t < threads; t++)
    for (int k = 0; k < 4; k++) tot[k] += tstats[(size_t)t * 4 + k];
  if (out->gRayCount) {
    out->gRayCount[0] = tot[0];
    out->gRayCount[1] = tot[1];
  }
  if (stats_out) memcpy(stats_out, tot, sizeof(tot));
  return STHIP_OK;
}

// mode bit0: any-hit; bit1: brute force (no acceleration structure at all)
int orc_trace_rays(orc_scene* sc, const sthip_ray* rays, uint32_t n, sthip_hit* hits, uint32_t mode, int threads, uint64_t* counters_out) {
  if (!sc || !rays || !hits) return STHIP_ERR_INVALID_ARGUMENT;
  if (threads <= 0) threads = (int)std::max(1u, std::thread::hardware_concurrency());
  const uint32_t chunk = 1024;
  const uint32_t chunks = (n + chunk - 1) / chunk;
  std::vector<uint64_t> tc((size_t)threads * 2, 0);
  parallel_rows(chunks, threads, [&](uint32_t c, int tid) {
    const uint32_t lo = c * chunk, hi = std::min(n, lo + chunk);
    for (uint32_t i = lo; i < hi; i++) {
      Ray r;
      r.o = V3(rays[i].origin[0], rays[i].origin[1], rays[i].origin[2]);
      r.d = V3(rays[i].direction[0], rays[i].direction[1], rays[i].direction[2]);
      r.tmin = rays[i].tmin;
      r.tmax = rays[i].tmax;
      r.alpha_test = (mode & 4) != 0;  // mode: 1 = any hit, 2 = brute force, 4 = alpha test, 8 = flipped triangle uvs
      r.flip_uvs = (mode & 8) != 0;
      const Hit h = trace(*sc, r, (mode & 1) != 0, (mode & 2) != 0, &tc[(size_t)tid * 2]);
      hits[i].t = h.t;
      hits[i].b1 = h.b1;
      hits[i].b2 = h.b2;
      hits[i].instance_primitive_index = h.ip;
    }
  });
  if (counters_out) {
    counters_out[0] = counters_out[1] = 0;
    for (int t = 0; t < threads; t++) {
      counters_out[0] += tc[(size_t)t * 2];
      counters_out[1] += tc[(size_t)t * 2 + 1];
    }
  }
  return STHIP_OK;
}

// ---- unit entry points: one per row of SURVEY.md §8a that has a closed form ----
void orc_pcg4d(uint32_t* v, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) pcg4d(v + 4 * (size_t)i);
}
uint32_t orc_pcg(uint32_t v) { return pcg(v); }
uint32_t orc_xxhash32(uint32_t v) { return xxhash32(v); }
// rng stream: state (x,y,seed,counter0) -> n floats
void orc_rng_floats(uint32_t x, uint32_t y, uint32_t seed, uint32_t counter0, float* out, uint32_t n) {
  Rng r;
  r.v[0] = x;
  r.v[1] = y;
  r.v[2] = seed;
  r.v[3] = counter0;
  for (uint32_t i = 0; i < n; i++) out[i] = r.next_float();
}
void orc_pack_normal(const float* v, uint32_t* out, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) out[i] = pack_normal_octahedron(V3(v[3 * i], v[3 * i + 1], v[3 * i + 2]));
}
void orc_unpack_normal(const uint32_t* p, float* out, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) {
    const v3 r = unpack_normal_octahedron(p[i]);
    out[3 * i] = r.x;
    out[3 * i + 1] = r.y;
    out[3 * i + 2] = r.z;
  }
}
void orc_ray_offset(const float* pos, const float* nrm, float* out, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) {
    const v3 r = ray_offset(V3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]), V3(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]));
    out[3 * i] = r.x;
    out[3 * i + 1] = r.y;
    out[3 * i + 2] = r.z;
  }
}
void orc_f32tof16(const float* v, uint32_t* out, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) out[i] = det_f32tof16(v[i]);
}
void orc_f16tof32(const uint32_t* v, float* out, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) out[i] = det_f16tof32(v[i]);
}
void orc_sincos(const float* x, float* s, float* c, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) det_sincosf(x[i], &s[i], &c[i]);
}
void orc_log(const float* x, float* out, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) out[i] = det_logf(x[i]);
}
// sample_point_on_light for n (rnd4, ref_pos): out = radiance(3) pdf to_light(3) dist position(3) normal(3) area_measure is_environment
void orc_sample_light(orc_scene* sc, const sthip_BDPTPushConstants* pc, uint32_t sampling_flags, uint32_t scene_flags, const float* rnd4, const float* ref_pos, float* out16, uint32_t n) {
  Frame fr;
  fr.sc = sc;
  fr.pc = *pc;
  fr.sampling_flags = sampling_flags;
  fr.scene_flags = scene_flags;
  for (uint32_t i = 0; i < n; i++) {
    LightSampleRecord ls;
    sample_point_on_light(fr, ls, rnd4 + 4 * i, V3(ref_pos[3 * i], ref_pos[3 * i + 1], ref_pos[3 * i + 2]));
    float* o = out16 + 16 * (size_t)i;
    o[0] = ls.radiance.x, o[1] = ls.radiance.y, o[2] = ls.radiance.z, o[3] = ls.pdf;
    o[4] = ls.to_light.x, o[5] = ls.to_light.y, o[6] = ls.to_light.z, o[7] = ls.dist;
    o[8] = ls.position.x, o[9] = ls.position.y, o[10] = ls.position.z;
    o[11] = ls.normal.x, o[12] = ls.normal.y, o[13] = ls.normal.z;
    o[14] = ls.pdf_area_measure ? 1.0f : 0.0f;
    o[15] = ls.is_environment ? 1.0f : 0.0f;
  }
}
// One delta_track call per sample (medium.hlsli:74-127) for the medium whose record lies at `medium_address` of the scene's
// material bytes: in8 = origin, direction (object space of the volume instance), t_max, beta (same for all channels);
// keys = the rng state (x, y, seed, counter) before the call; out16 = beta, dir_pdf, nee_pdf, scatter position, scattered,
// counter after (as a float), -, -.
void orc_delta_track(orc_scene* sc, uint32_t medium_address, const uint32_t* keys, const float* in8, uint32_t can_scatter, uint32_t max_null_collisions, float* out16, uint32_t n) {
  Medium m;
  m.load(*sc, medium_address);
  for (uint32_t i = 0; i < n; i++) {
    Rng rng;
    memcpy(rng.v, keys + 4 * (size_t)i, 16);
    const float* q = in8 + 8 * (size_t)i;
    v3 beta = V3(q[7]), dir_pdf = V3(1.0f), nee_pdf = V3(1.0f), p = V3(0.0f);
    const bool scattered = m.delta_track(*sc, rng, V3(q[0], q[1], q[2]), V3(q[3], q[4], q[5]), q[6], beta, dir_pdf, nee_pdf, can_scatter != 0, max_null_collisions, p);
    float* o = out16 + 16 * (size_t)i;
    o[0] = beta.x, o[1] = beta.y, o[2] = beta.z;
    o[3] = dir_pdf.x, o[4] = dir_pdf.y, o[5] = dir_pdf.z;
    o[6] = nee_pdf.x, o[7] = nee_pdf.y, o[8] = nee_pdf.z;
    o[9] = p.x, o[10] = p.y, o[11] = p.z;
    o[12] = scattered ? 1.0f : 0.0f;
    o[13] = (float)(rng.v[3] - keys[4 * (size_t)i + 3]);
    o[14] = o[15] = 0.0f;
  }
}
// the NanoVDB reader on its own: header[8] = bbox min, bbox max, grid ok, -; header_f[1] = root maximum; values at coords;
// maps[m][4][3] = world_to_indexf, world_to_index_dirf of points[m], index_to_worldf of the first, index_to_world_dirf of the second
int orc_nvdb_probe(const void* grid, uint64_t bytes, const int32_t* coords, uint32_t n, float* values, int32_t* header, float* root_max, const float* points, uint32_t m, float* maps) {
  NvdbGrid g;
  g.bytes.assign((const uint8_t*)grid, (const uint8_t*)grid + bytes);
  if (!g.init()) return STHIP_ERR_INVALID_ARGUMENT;
  for (int k = 0; k < 3; k++) {
    header[k] = g.bbox_min[k];
    header[3 + k] = g.bbox_max[k];
  }
  *root_max = g.root_max();
  for (uint32_t i = 0; i < n; i++) values[i] = g.value(coords[3 * i], coords[3 * i + 1], coords[3 * i + 2]);
  for (uint32_t i = 0; i < m; i++) {
    const v3 p = V3(points[3 * i], points[3 * i + 1], points[3 * i + 2]);
    const v3 a = g.world_to_index(p), b = g.world_to_index_dir(p), c = g.index_to_world(a), d = g.index_to_world_dir(b);
    const v3 r[4] = {a, b, c, d};
    for (int k = 0; k < 4; k++) {
      maps[12 * i + 3 * k] = r[k].x;
      maps[12 * i + 3 * k + 1] = r[k].y;
      maps[12 * i + 3 * k + 2] = r[k].z;
    }
  }
  return STHIP_OK;
}
void orc_atan2(const float* y, const float* x, float* out, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) out[i] = det_atan2f(y[i], x[i]);
}
void orc_acos(const float* x, float* out, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) out[i] = det_acosf(x[i]);
}
void orc_asin(const float* x, float* out, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) out[i] = det_asinf(x[i]);
}
void orc_pow(const float* a, const float* b, float* out, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) out[i] = det_powf(a[i], b[i]);
}
// material record (72 B) + local dirs -> f(3), pdf_fwd, pdf_rev
void orc_disney_eval(const sthip_MaterialRecord* rec, const float* dir_in, const float* dir_out, float* out5, uint32_t n) {
  DisneyMaterial m;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 4; j++) m.data[i][j] = rec->values[i].value[j];
  for (uint32_t i = 0; i < n; i++) {
    MaterialEvalRecord r;
    m.eval(r, V3(dir_in[3 * i], dir_in[3 * i + 1], dir_in[3 * i + 2]), V3(dir_out[3 * i], dir_out[3 * i + 1], dir_out[3 * i + 2]), false);
    out5[5 * i] = r.f.x;
    out5[5 * i + 1] = r.f.y;
    out5[5 * i + 2] = r.f.z;
    out5[5 * i + 3] = r.pdf_fwd;
    out5[5 * i + 4] = r.pdf_rev;
  }
}
// the same with the adjoint flag (light subpaths, connections)
void orc_disney_eval_adjoint(const sthip_MaterialRecord* rec, const float* dir_in, const float* dir_out, float* out5, uint32_t n, uint32_t adjoint) {
  DisneyMaterial m;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 4; j++) m.data[i][j] = rec->values[i].value[j];
  for (uint32_t i = 0; i < n; i++) {
    MaterialEvalRecord r;
    m.eval(r, V3(dir_in[3 * i], dir_in[3 * i + 1], dir_in[3 * i + 2]), V3(dir_out[3 * i], dir_out[3 * i + 1], dir_out[3 * i + 2]), adjoint != 0);
    out5[5 * i] = r.f.x;
    out5[5 * i + 1] = r.f.y;
    out5[5 * i + 2] = r.f.z;
    out5[5 * i + 3] = r.pdf_fwd;
    out5[5 * i + 4] = r.pdf_rev;
  }
}
// the scalar helpers of the integrator, for tests: in5 = (ndotin, ndotout, ngdotin, ngdotout, ngdotns) per row
void orc_shading_normal_correction(const float* in5, float* out, uint32_t n, uint32_t shadow_fix, uint32_t adjoint) {
  for (uint32_t i = 0; i < n; i++) out[i] = shading_normal_correction(in5[5 * i], in5[5 * i + 1], in5[5 * i + 2], in5[5 * i + 3], in5[5 * i + 4], shadow_fix != 0, adjoint != 0);
}
// in3 = (dVC, pdfA_rev, prev_pdfA_fwd) per row -> connection_dVC (path.hlsli:31-38); in2 = (a, b) -> mis(a, b) (:11-15)
void orc_connection_dvc(const float* in3, float* out, uint32_t n, uint32_t specular) {
  for (uint32_t i = 0; i < n; i++) out[i] = connection_dVC(in3[3 * i], in3[3 * i + 1], in3[3 * i + 2], specular != 0);
}
void orc_mis(const float* in2, float* out, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) {
    const float a2 = pow2(in2[2 * i]), b2 = pow2(in2[2 * i + 1]);
    out[i] = a2 / (a2 + b2);
  }
}
// material + dir_in + rnd(3) -> dir_out(3), pdf_fwd, pdf_rev, eta, roughness, f(3), beta(3) (beta starts at 1)
void orc_disney_sample(const sthip_MaterialRecord* rec, const float* dir_in, const float* rnd, float* out13, uint32_t n) {
  DisneyMaterial m;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 4; j++) m.data[i][j] = rec->values[i].value[j];
  for (uint32_t i = 0; i < n; i++) {
    MaterialSampleRecord r;
    v3 beta = V3(1.0f);
    const v3 f = m.sample(r, V3(rnd[3 * i], rnd[3 * i + 1], rnd[3 * i + 2]), V3(dir_in[3 * i], dir_in[3 * i + 1], dir_in[3 * i + 2]), beta, false);
    float* o = out13 + 13 * (size_t)i;
    o[0] = r.dir_out.x;
    o[1] = r.dir_out.y;
    o[2] = r.dir_out.z;
    o[3] = r.pdf_fwd;
    o[4] = r.pdf_rev;
    o[5] = r.eta;
    o[6] = r.roughness;
    o[7] = f.x;
    o[8] = f.y;
    o[9] = f.z;
    o[10] = beta.x;
    o[11] = beta.y;
    o[12] = beta.z;
  }
}
// sample_image (image_value.h:81-97) of gImages[index] at n (u, v, uv_screen_size) triples -> RGBA
void orc_sample_image(orc_scene* sc, uint32_t index, const float* uvs, uint32_t ray_cones, float* out, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) DisneyMaterial::sample_image(*sc, index, uvs[3 * i], uvs[3 * i + 1], uvs[3 * i + 2], ray_cones != 0, out + 4 * (size_t)i);
}
// shading data of (instance, primitive, barycentrics) -> sthip_ShadingData
void orc_shading_data(orc_scene* sc, const uint32_t* inst_prim, const float* bary, sthip_ShadingData* out, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) {
    ShadingData sd;
    make_triangle_shading_data(*sc, sd, inst_prim[i] & 0xFFFF, inst_prim[i] >> 16, bary[2 * i], bary[2 * i + 1]);
    out[i].position[0] = sd.position.x;
    out[i].position[1] = sd.position.y;
    out[i].position[2] = sd.position.z;
    out[i].flags = sd.flags;
    out[i].packed_geometry_normal = sd.packed_geometry_normal;
    out[i].packed_shading_normal = sd.packed_shading_normal;
    out[i].packed_tangent = sd.packed_tangent;
    out[i].shape_area = sd.shape_area;
    out[i].uv[0] = sd.u;
    out[i].uv[1] = sd.v;
    out[i].uv_screen_size = sd.uv_screen_size;
    out[i].mean_curvature = sd.mean_curvature;
  }
}

}  // extern "C"
