// ceilings.h — measured memory-system ceilings for the roofline of the traversal kernel (sthip_measure_ceiling).
//
// SURVEY.md section 8(d) prices k_trace against the HBM peak, but the acceleration structure of the bench scene (26 MB of
// nodes + 38 MB of leaf triangles) lives in L2 + Infinity Cache, so HBM is not what a node fetch waits for. What binds a
// divergent traversal is the rate at which the vector-memory path delivers nodes to 64 lanes that each ask for a
// different one. That rate is a property of the chip and of where the table is served from, and it can be measured:
// the kernels below issue the traversal's own node fetch (3 x global_load_dwordx4 per lane from one 48-byte packed
// node, or 4 from one 64-byte wide node when k_trace walks those: bvh.h) at pseudo-random node indices with NO dependence between fetches (8 in flight per lane, full
// occupancy), i.e. the same bytes through the same units with the latency chain and the arithmetic taken away.
//   * over the whole resident node array        -> what L2 / Infinity Cache deliver to random node fetches
//   * over a 2 MB prefix of it (fits one XCD L2) -> the L2-hit rate of the same access
//   * over a 16 KB prefix (fits the CU's L1)     -> the rate of the vector-memory front end itself (address
//                                                    processing of divergent 16-byte lane loads; nothing is slower than it)
// plus the classic stream triad for the HBM figure the guide asks to use as the measured denominator.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define CEIL_UNROLL 8

__device__ __forceinline__ uint32_t ceil_pcg(uint32_t v) {  // Jarzynski & Olano's pcg hash: only used to scatter addresses
  const uint32_t state = v * 747796405u + 2891336453u;
  const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}

// nodes: the BVH node array (node_bytes-byte records); node_count: how many of them to spread the fetches over
// LOADS: 16-byte loads per node — 3 for the 48-byte binary node, 4 for the 64-byte wide node (whichever k_trace walks)
template <int LOADS>
__global__ void __launch_bounds__(256) k_ceiling_node_gather(const float4* __restrict__ nodes, uint32_t node_count, uint32_t node_bytes, uint32_t iterations, float* sink) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t state = ceil_pcg(tid * 2654435761u + 12345u);
  float acc = 0.0f;
  uint32_t acc_u = 0;
  const char* base = reinterpret_cast<const char*>(nodes);
  for (uint32_t it = 0; it < iterations; it++) {
    uint32_t idx[CEIL_UNROLL];
#pragma unroll
    for (int u = 0; u < CEIL_UNROLL; u++) {
      state = ceil_pcg(state + (uint32_t)u);
      idx[u] = (uint32_t)(((uint64_t)state * node_count) >> 32);
    }
    float4 a[CEIL_UNROLL][LOADS];
#pragma unroll
    for (int u = 0; u < CEIL_UNROLL; u++) {
      const float4* n = reinterpret_cast<const float4*>(base + (size_t)idx[u] * node_bytes);
#pragma unroll
      for (int l = 0; l < LOADS; l++) a[u][l] = n[l];
    }
#pragma unroll
    for (int u = 0; u < CEIL_UNROLL; u++) {
#pragma unroll
      for (int l = 0; l < LOADS; l++) {
        acc += a[u][l].x + a[u][l].w;
        acc_u ^= __float_as_uint(a[u][l].y) + __float_as_uint(a[u][l].z);
      }
    }
  }
  if (acc == 123.456f && acc_u == 0x12345u) sink[tid] = acc;  // keeps the loads alive; practically never true
}

__global__ void __launch_bounds__(256) k_ceiling_triad(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ c, float s, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float4 x = a[i], y = b[i];
    c[i] = make_float4(x.x + s * y.x, x.y + s * y.y, x.z + s * y.z, x.w + s * y.w);
  }
}
