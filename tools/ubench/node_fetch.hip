// Microbenchmark (GPU box): what one 64-byte node fetch per lane costs the vector-memory front end, by the shape of the loads.
//   own  : every lane reads its own node with four 16-byte loads (what k_trace's wide step does)
//   quad : the four lanes of a quad read ONE node together, 16 bytes each, four times (once per lane's node), then transpose the
//          4 x 4 blocks inside the quad with DPP so that every lane holds its own node
//   quadx: as quad without the transpose (each lane consumes what it loaded: the cost of the loads alone)
// Lanes follow a data-dependent chain through a table of N nodes (next index from the loaded words), `active` of 64 lanes per wave walk.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/node_fetch tools/ubench/node_fetch.hip ; run: node_fetch [nodes] [active lanes] [iterations]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      std::printf("%s: %s\n", #x, hipGetErrorString(e_));                         \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t a) {
  a ^= a >> 16;
  a *= 0x7feb352du;
  a ^= a >> 15;
  a *= 0x846ca68bu;
  a ^= a >> 16;
  return a;
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ uint4 dpp4(uint4 v) {
  return make_uint4(dpp<CTRL>(v.x), dpp<CTRL>(v.y), dpp<CTRL>(v.z), dpp<CTRL>(v.w));
}
__device__ __forceinline__ uint4 sel(bool c, uint4 a, uint4 b) { return make_uint4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w); }

template <int MODE>
__global__ void __launch_bounds__(256) k_fetch(const uint4* __restrict__ table, uint32_t nodes, uint32_t active, uint32_t iterations, uint32_t* out) {
  const uint32_t lane = threadIdx.x & 63u, gid = blockIdx.x * blockDim.x + threadIdx.x;
  // a fixed scattered subset of the lanes walks (as in k_trace, where ~42 % of a wave's lanes are in the node loop)
  const bool walks = (mix(lane * 0x9E3779B9u + 12345u) % 64u) < active || active >= 64u;
  uint32_t idx = mix(gid) % nodes, acc = 0;
  for (uint32_t it = 0; it < iterations; it++) {
    uint4 n0, n1, n2, n3;
    if (MODE == 0) {
      if (walks) {
        const uint4* p = table + 4 * (size_t)idx;
        n0 = p[0];
        n1 = p[1];
        n2 = p[2];
        n3 = p[3];
      }
    } else {
      const uint32_t q = lane & 3u;
      const uint32_t i0 = dpp<0x00>(idx), i1 = dpp<0x55>(idx), i2 = dpp<0xAA>(idx), i3 = dpp<0xFF>(idx);
      const uint32_t w = walks ? 1u : 0u;
      const bool w0 = dpp<0x00>(w), w1 = dpp<0x55>(w), w2 = dpp<0xAA>(w), w3 = dpp<0xFF>(w);
      uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0;
      if (w0) r0 = table[4 * (size_t)i0 + q];
      if (w1) r1 = table[4 * (size_t)i1 + q];
      if (w2) r2 = table[4 * (size_t)i2 + q];
      if (w3) r3 = table[4 * (size_t)i3 + q];
      if (MODE == 1) {
        // lane q holds in r_j chunk q of node j; it needs chunk c of node q in n_c: exchange across lane bit 0, then bit 1
        const bool b0 = (lane & 1u) != 0, b1 = (lane & 2u) != 0;
        // bit 0: pairs (r0, r1) and (r2, r3)
        const uint4 a0 = sel(b0, dpp4<0xB1>(r1), r0), a1 = sel(b0, r1, dpp4<0xB1>(r0));
        const uint4 a2 = sel(b0, dpp4<0xB1>(r3), r2), a3 = sel(b0, r3, dpp4<0xB1>(r2));
        // bit 1: pairs (a0, a2) and (a1, a3)
        n0 = sel(b1, dpp4<0x4E>(a2), a0);
        n2 = sel(b1, a2, dpp4<0x4E>(a0));
        n1 = sel(b1, dpp4<0x4E>(a3), a1);
        n3 = sel(b1, a3, dpp4<0x4E>(a1));
      } else {
        n0 = r0;
        n1 = r1;
        n2 = r2;
        n3 = r3;
      }
    }
    if (walks) {
      const uint32_t h = n0.x ^ n1.y ^ n2.z ^ n3.w;
      acc += h;
      idx = mix(h + it) % nodes;
    }
  }
  out[gid] = acc + idx;
}

// host check of the transpose: table word w of node i = i * 16 + w
__global__ void k_check(const uint4* table, uint32_t nodes, uint32_t* bad) {
  const uint32_t lane = threadIdx.x & 63u, q = lane & 3u;
  const uint32_t idx = mix(threadIdx.x * 77u + 5u) % nodes;
  const uint32_t i0 = dpp<0x00>(idx), i1 = dpp<0x55>(idx), i2 = dpp<0xAA>(idx), i3 = dpp<0xFF>(idx);
  const uint4 r0 = table[4 * (size_t)i0 + q], r1 = table[4 * (size_t)i1 + q], r2 = table[4 * (size_t)i2 + q], r3 = table[4 * (size_t)i3 + q];
  const bool b0 = (lane & 1u) != 0, b1 = (lane & 2u) != 0;
  const uint4 a0 = sel(b0, dpp4<0xB1>(r1), r0), a1 = sel(b0, r1, dpp4<0xB1>(r0));
  const uint4 a2 = sel(b0, dpp4<0xB1>(r3), r2), a3 = sel(b0, r3, dpp4<0xB1>(r2));
  const uint4 n0 = sel(b1, dpp4<0x4E>(a2), a0), n2 = sel(b1, a2, dpp4<0x4E>(a0)), n1 = sel(b1, dpp4<0x4E>(a3), a1), n3 = sel(b1, a3, dpp4<0x4E>(a1));
  const uint4* p = table + 4 * (size_t)idx;
  auto ne = [](uint4 a, uint4 b) { return a.x != b.x || a.y != b.y || a.z != b.z || a.w != b.w; };
  if (ne(n0, p[0]) || ne(n1, p[1]) || ne(n2, p[2]) || ne(n3, p[3])) atomicAdd(bad, 1u);
}

int main(int argc, char** argv) {
  const uint32_t nodes = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 350000u;
  const uint32_t active = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 27u;
  const uint32_t iterations = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 1000u;
  std::vector<uint32_t> h((size_t)nodes * 16);
  for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)i;
  uint4* table;
  uint32_t *out, *bad;
  const uint32_t blocks = 256 * 4, threads = 256;
  CHECK(hipMalloc(&table, h.size() * 4));
  CHECK(hipMalloc(&out, (size_t)blocks * threads * 4));
  CHECK(hipMalloc(&bad, 4));
  CHECK(hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemset(bad, 0, 4));
  hipLaunchKernelGGL(k_check, dim3(1), dim3(256), 0, 0, table, nodes, bad);
  uint32_t hb = 1;
  CHECK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
  std::printf("transpose check: %u lanes wrong\n", hb);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const char* names[3] = {"own  (4 loads of the lane's own node)", "quad (quad-cooperative loads + DPP transpose)", "quadx (quad-cooperative loads, no transpose)"};
  for (int rep = 0; rep < 2; rep++)
    for (int mode = 0; mode < 3; mode++) {
      CHECK(hipEventRecord(e0, 0));
      if (mode == 0) hipLaunchKernelGGL((k_fetch<0>), dim3(blocks), dim3(threads), 0, 0, table, nodes, active, iterations, out);
      if (mode == 1) hipLaunchKernelGGL((k_fetch<1>), dim3(blocks), dim3(threads), 0, 0, table, nodes, active, iterations, out);
      if (mode == 2) hipLaunchKernelGGL((k_fetch<2>), dim3(blocks), dim3(threads), 0, 0, table, nodes, active, iterations, out);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double visits = (double)blocks * threads / 64.0 * iterations;  // wave-level node visits
      if (rep) std::printf("%-48s %8.3f ms  %7.1f ns per wave visit per CU (16 waves / CU), %6.1f G lane-fetches/s\n", names[mode], ms, ms * 1e6 / (visits / 256.0), visits * (active >= 64 ? 64 : active) / ms / 1e6);
    }
  return 0;
}
