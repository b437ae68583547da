// lvc.hip — the light vertex cache in its defined order (eLVC; path.hlsli:523-527 hands out cache slots with an atomic
// counter, so upstream's order depends on scheduling). k_shade_light stages every stored vertex at
// [seed][path_index][diffuse_vertices - 1]; here the stage is compacted — an order-preserving stream compaction: flags,
// exclusive scan (hipCUB), scatter — into gLightPathVertices, so that cache entry k is the k-th stored vertex in
// (path index, vertex) order, the order a serial run of upstream's sample_photons would produce. The number of
// entries per seed in flight goes to `counts` (gLightPathVertexCount[0]).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include <string>

namespace sthip {
namespace {

// a staged PathVertex is 4 x float4; bits 16.. of its last word hold subpath_length >= 2, so that word is never zero
// a record is `rec` float4 long; a staged one is told by the last word of its float4 number `flag_at` not being zero
__global__ void k_lvc_flags(const float4* staging, uint32_t total, uint32_t rec, uint32_t flag_at, uint32_t* flags) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) flags[i] = __float_as_uint(staging[rec * (size_t)i + flag_at].w) != 0u ? 1u : 0u;
}
__global__ void k_lvc_scatter(const float4* staging, const uint32_t* flags, const uint32_t* offsets, uint32_t slots_per_seed, uint32_t seeds, uint32_t vertices_per_seed, uint32_t rec,
                              float4* cache, uint32_t* counts) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t total = slots_per_seed * seeds;
  if (i >= total) return;
  const uint32_t seed = i / slots_per_seed;
  const uint32_t base = offsets[seed * slots_per_seed];  // entries of the seeds in front of this one
  if (flags[i]) {
    const uint32_t k = offsets[i] - base;
    if (k < vertices_per_seed) {
      float4* dst = cache + rec * ((size_t)seed * vertices_per_seed + k);
      const float4* src = staging + rec * (size_t)i;
      for (uint32_t q = 0; q < rec; q++) dst[q] = src[q];
    }
  }
  if (i - seed * slots_per_seed == slots_per_seed - 1) counts[seed] = offsets[i] + flags[i] - base;
}

}  // namespace

// rec / flag_at: the record layout (a PathVertex or a NEE append: 4 float4, told by float4 2; an LVC append: 6, told by float4 0).
// scratch: `flags` and `offsets` hold slots_per_seed * seeds uint32 each; `tmp` / `tmp_bytes` hipCUB's temporary storage
// (query with tmp == nullptr). Everything is enqueued on `stream`.
hipError_t lvc_compact(const float4* staging, uint32_t slots_per_seed, uint32_t seeds, uint32_t vertices_per_seed, float4* cache, uint32_t* counts, uint32_t* flags, uint32_t* offsets,
                       void* tmp, size_t& tmp_bytes, hipStream_t stream, uint32_t rec, uint32_t flag_at) {
  const uint32_t total = slots_per_seed * seeds;
  if (!tmp) return hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, flags, offsets, (int)total, stream);
  if (!total) return hipSuccess;
  const uint32_t grid = (total + 255) / 256;
  hipLaunchKernelGGL(k_lvc_flags, dim3(grid), dim3(256), 0, stream, staging, total, rec, flag_at, flags);
  hipError_t e = hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, flags, offsets, (int)total, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_lvc_scatter, dim3(grid), dim3(256), 0, stream, staging, flags, offsets, slots_per_seed, seeds, vertices_per_seed, rec, cache, counts);
  return hipGetLastError();
}

}  // namespace sthip
