// media.h — participating media on the device: a reader for the NanoVDB float grids the reference binds as gVolumes[]
// (ByteAddressBuffer, bdpt.hlsl:35) and the Medium of materials/medium.hlsli (Henyey-Greenstein phase function, delta
// tracking against the density grid's root maximum).
//
// The reader covers the subset of PNanoVDB.h (NanoVDB 32.3, vendored by the reference under src/extern/nanovdb) that
// medium.hlsli:58-71,85-88 and intersection.hlsli:93-113 call: value at an index coordinate (root tile -> upper 32^3 ->
// lower 16^3 -> leaf 8^3), the root bounding box and maximum, the grid's affine map. Layout constants are those of the
// published format (PNanoVDB.h:702-711,761-777,904-916,964-1111, FLOAT row of pnanovdb_grid_type_constants); the header
// fields are parsed once on the host (api.hip) into DeviceVolume. Reads outside the buffer return zero.
#pragma once

#include "bvh.h"
#include "device_math.h"


struct NvdbView {
  const uint32_t* w;
  DeviceVolume v;
  DEV uint32_t rd(uint32_t byte) const { return byte + 4u <= v.bytes ? w[byte >> 2] : 0u; }
  DEV uint64_t rd64(uint32_t byte) const { return (uint64_t)rd(byte) | ((uint64_t)rd(byte + 4u) << 32); }
  DEV bool bit(uint32_t mask, uint32_t n) const { return (rd(mask + 4u * (n >> 5)) >> (n & 31u)) & 1u; }
  DEV float value(int32_t x, int32_t y, int32_t z) const {
    const uint64_t key = (uint64_t)((uint32_t)z >> 12) | ((uint64_t)((uint32_t)y >> 12) << 21) | ((uint64_t)((uint32_t)x >> 12) << 42);
    const uint32_t tiles = rd(v.root + 24u);
    for (uint32_t i = 0; i < tiles; i++) {
      const uint32_t tile = v.root + 64u + 32u * i;
      if (rd64(tile) != key) continue;
      const uint64_t child = rd64(tile + 8u);
      if (child == 0) return __uint_as_float(rd(tile + 20u));
      const uint32_t upper = v.root + (uint32_t)child;
      const uint32_t n = ((((uint32_t)x & 4095u) >> 7) << 10) + ((((uint32_t)y & 4095u) >> 7) << 5) + (((uint32_t)z & 4095u) >> 7);
      if (!bit(upper + 4128u, n)) return __uint_as_float(rd(upper + 8256u + 8u * n));
      const uint32_t lower = upper + (uint32_t)rd64(upper + 8256u + 8u * n);
      const uint32_t n2 = ((((uint32_t)x & 127u) >> 3) << 8) + ((((uint32_t)y & 127u) >> 3) << 4) + (((uint32_t)z & 127u) >> 3);
      if (!bit(lower + 544u, n2)) return __uint_as_float(rd(lower + 1088u + 8u * n2));
      const uint32_t leaf = lower + (uint32_t)rd64(lower + 1088u + 8u * n2);
      const uint32_t n3 = (((uint32_t)x & 7u) << 6) + (((uint32_t)y & 7u) << 3) + ((uint32_t)z & 7u);
      return __uint_as_float(rd(leaf + 96u + 4u * n3));
    }
    return __uint_as_float(rd(v.root + 28u));  // background
  }
};

// pnanovdb_map_apply / _inverse / _jacobi / _inverse_jacobi, PNanoVDB.h:1988-2034 (left-to-right sums)
DEV f3 nvdb_index_to_world_dir(const DeviceVolume& v, f3 s) {
  return F3(s.x * v.matf[0] + s.y * v.matf[1] + s.z * v.matf[2], s.x * v.matf[3] + s.y * v.matf[4] + s.z * v.matf[5], s.x * v.matf[6] + s.y * v.matf[7] + s.z * v.matf[8]);
}
DEV f3 nvdb_index_to_world(const DeviceVolume& v, f3 s) {
  return F3(s.x * v.matf[0] + s.y * v.matf[1] + s.z * v.matf[2] + v.vecf[0], s.x * v.matf[3] + s.y * v.matf[4] + s.z * v.matf[5] + v.vecf[1],
            s.x * v.matf[6] + s.y * v.matf[7] + s.z * v.matf[8] + v.vecf[2]);
}
DEV f3 nvdb_world_to_index_dir(const DeviceVolume& v, f3 s) {
  return F3(s.x * v.invmatf[0] + s.y * v.invmatf[1] + s.z * v.invmatf[2], s.x * v.invmatf[3] + s.y * v.invmatf[4] + s.z * v.invmatf[5],
            s.x * v.invmatf[6] + s.y * v.invmatf[7] + s.z * v.invmatf[8]);
}
DEV f3 nvdb_world_to_index(const DeviceVolume& v, f3 p) { return nvdb_world_to_index_dir(v, F3(p.x - v.vecf[0], p.y - v.vecf[1], p.z - v.vecf[2])); }

// Volume instances in the traversal contract (intersection.hlsli:93-113): o, d in the object space of the instance. The
// slabs of the root bounding box [bbox_min, bbox_max + 1] in index space; t = the entry if it lies beyond tmin, else the
// exit; a hit needs entry <= exit (the reference leaves that to the driver's candidate test) and tmin < t < tmax.
// `face` = (t == t1) - (t == t0) per axis: the index-space normal of the face that was hit.
DEV bool volume_test(const DeviceVolume& v, f3 o, f3 d, float tmin, float tmax, float& t, f3* face = nullptr) {
  const f3 io = nvdb_world_to_index(v, o), id = nvdb_world_to_index_dir(v, d);
  const f3 lo = F3((float)v.bbox_min[0], (float)v.bbox_min[1], (float)v.bbox_min[2]);
  const f3 hi = F3((float)(v.bbox_max[0] + 1), (float)(v.bbox_max[1] + 1), (float)(v.bbox_max[2] + 1));
  const f3 t0 = F3((lo.x - io.x) / id.x, (lo.y - io.y) / id.y, (lo.z - io.z) / id.z);
  const f3 t1 = F3((hi.x - io.x) / id.x, (hi.y - io.y) / id.y, (hi.z - io.z) / id.z);
  const float near = fmaxf(fmaxf(fminf(t0.x, t1.x), fminf(t0.y, t1.y)), fminf(t0.z, t1.z));
  const float far = fminf(fminf(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y)), fmaxf(t0.z, t1.z));
  if (!(near <= far)) return false;
  const float tt = near > tmin ? near : far;
  if (!(tt > tmin && tt < tmax)) return false;
  t = tt;
  if (face)
    *face = F3((tt == t1.x ? 1.0f : 0.0f) - (tt == t0.x ? 1.0f : 0.0f), (tt == t1.y ? 1.0f : 0.0f) - (tt == t0.y ? 1.0f : 0.0f), (tt == t1.z ? 1.0f : 0.0f) - (tt == t0.z ? 1.0f : 0.0f));
  return true;
}
