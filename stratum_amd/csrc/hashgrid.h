// hashgrid.h — the spatial hash grid of the reservoir reuse (src/Shaders/common/hashgrid.hlsli:4-89), device side:
// cell size, bucket index + checksum, lookup. Upstream BUILDS the grid with atomics (compare-exchange probing, per-bucket
// counters, a global append counter), so which bucket a cell ends up in when cells compete and the order of a bucket's
// records depend on thread scheduling. The order DEFINED here: records are appended in (path index, diffuse vertex)
// order, i.e. the grid a serial run of upstream's kernel builds. A view vertex stages its record at
// [path_index * gMaxDiffuseVertices + diffuse_vertices - 1]; after the seed the stage is compacted in order (lvc.hip), the
// (home bucket, checksum) keys go to the host, which probes them sequentially exactly as find_or_insert does
// (api.hip: build_hash_grid), and the records are scattered to their bucket ranges. Probing does not wrap upstream (it
// runs off the buffer); here the table has 32 slots more than gHashGridBucketCount.
#pragma once

#include "shading.h"

DEV uint32_t hg_pcg(uint32_t v) {  // rng.hlsli:17-21
  const uint32_t state = v * 747796405u + 2891336453u;
  const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}
DEV uint32_t hg_xxhash32(uint32_t p) {  // rng.hlsli:6-15
  const uint32_t PRIME32_2 = 2246822519u, PRIME32_3 = 3266489917u, PRIME32_4 = 668265263u, PRIME32_5 = 374761393u;
  uint32_t h32 = p + PRIME32_5;
  h32 = PRIME32_4 * ((h32 << 17) | (h32 >> (32 - 17)));
  h32 = PRIME32_2 * (h32 ^ (h32 >> 15));
  h32 = PRIME32_3 * (h32 ^ (h32 >> 13));
  return h32 ^ (h32 >> 16);
}
// float -> uint as the hardware converts: NaN and negatives to 0, too large to 0xFFFFFFFF
DEV uint32_t hg_f2u_sat(float f) {
  if (!(f > 0.0f)) return 0u;
  if (f >= 4294967296.0f) return 0xFFFFFFFFu;
  return (uint32_t)f;
}
DEV int32_t hg_f2i_sat(float f) {
  if (f >= 2147483648.0f) return 0x7FFFFFFF;
  if (f <= -2147483648.0f) return (int32_t)0x80000000;
  return f == f ? (int32_t)f : 0;
}

// hashgrid_cell_size, hashgrid.hlsli:4-14 (view 0's position and projection)
DEV float hashgrid_cell_size(const sthip_BDPTPushConstants& pc, const sthip_ViewData& view, f3 view_position, f3 pos) {
  if (pc.gHashGridBucketPixelRadius < 0) return pc.gHashGridMinBucketRadius;
  const float dist = length3(pos - view_position);
  const float ex = (float)(view.image_max[0] - view.image_min[0]), ey = (float)(view.image_max[1] - view.image_min[1]);
  const float step = dist * det_tanf(pc.gHashGridBucketPixelRadius * view.projection.vertical_fov * fmaxf(1 / ey, ey / pow2f(ex)));
  const uint32_t level = min(hg_f2u_sat(det_log2f(step / pc.gHashGridMinBucketRadius)), 31u);
  return pc.gHashGridMinBucketRadius * (float)(int32_t)(1u << level);
}
// hashgrid_bucket_index, hashgrid.hlsli:15-21
DEV uint32_t hashgrid_bucket_index(f3 pos, float cell_size, uint32_t bucket_count, uint32_t& checksum) {
  const uint32_t px = (uint32_t)hg_f2i_sat(floorf(pos.x / cell_size) + 0.5f), py = (uint32_t)hg_f2i_sat(floorf(pos.y / cell_size) + 0.5f),
                 pz = (uint32_t)hg_f2i_sat(floorf(pos.z / cell_size) + 0.5f);
  const uint32_t cs = hg_xxhash32(hg_f2u_sat(cell_size + (float)hg_xxhash32(pz + hg_xxhash32(py + hg_xxhash32(px)))));
  checksum = cs > 1u ? cs : 1u;
  return hg_pcg(hg_f2u_sat(cell_size + (float)hg_pcg(pz + hg_pcg(py + hg_pcg(px))))) % bucket_count;
}
// HashGrid::find, hashgrid.hlsli:33-42
DEV uint32_t hashgrid_find(const uint32_t* checksums, uint32_t bucket_count, f3 pos, float cell_size) {
  uint32_t checksum;
  uint32_t b = hashgrid_bucket_index(pos, cell_size, bucket_count, checksum);
  for (uint32_t i = 0; i < 32; i++, b++)
    if (checksums[b] == checksum) return b;
  return 0xFFFFFFFFu;
}
