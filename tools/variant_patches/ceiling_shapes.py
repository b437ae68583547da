# applies to stratum_amd/csrc as of commit bc88c5a (before the ceiling kernel learnt the wide node's shape itself): check that
# commit out to repeat tools/gather_shape_experiment.py
import re
p='ceilings.h'
s=open(p).read()
s=s.replace("__global__ void __launch_bounds__(256) k_ceiling_node_gather(","template <int LOADS>\n__global__ void __launch_bounds__(256) k_ceiling_node_gather(")
old=s[s.index("    float4 a[CEIL_UNROLL], b[CEIL_UNROLL], c[CEIL_UNROLL];"):s.index("  if (acc == 123.456f")]
new='''    float4 a[CEIL_UNROLL][LOADS];
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
'''
s=s.replace(old,new)
s=s.replace("#define CEIL_UNROLL 8","#define CEIL_UNROLL 4")
open(p,'w').write(s)
p='api.hip'
s=open(p).read()
old="    if (kind == STHIP_CEILING_NODE_GATHER_L2) count = std::min<uint32_t>(count, (2u << 20) / BVH_NODE_BYTES);\n    if (kind == STHIP_CEILING_NODE_GATHER_L1) count = std::min<uint32_t>(count, (16u << 10) / BVH_NODE_BYTES);\n"
new='''    uint32_t nb = BVH_NODE_BYTES, loads = 3;
    if (const char* e = getenv("STHIP_CEIL_NODE_BYTES")) nb = (uint32_t)atoi(e);
    if (const char* e = getenv("STHIP_CEIL_LOADS")) loads = (uint32_t)atoi(e);
    count = (uint32_t)std::min<uint64_t>((uint64_t)ctx->bvh_nodes * BVH_NODE_BYTES / nb - 1, 0xFFFFFFFFull);
    if (kind == STHIP_CEILING_NODE_GATHER_L2) count = std::min<uint32_t>(count, (2u << 20) / nb);
    if (kind == STHIP_CEILING_NODE_GATHER_L1) count = std::min<uint32_t>(count, (16u << 10) / nb);
'''
assert old in s
s=s.replace(old,new)
s=s.replace("bytes = (double)sizeof(BvhNodePacked) * (double)blocks * 256.0 * iterations * CEIL_UNROLL;","bytes = 16.0 * loads * (double)blocks * 256.0 * iterations * CEIL_UNROLL;")
old="      hipLaunchKernelGGL(k_ceiling_node_gather, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const float4*>(ctx->nodes.p), count, BVH_NODE_BYTES, iterations, sink.p);"
new='''      const float4* np_ = reinterpret_cast<const float4*>(ctx->nodes.p);
      if (loads == 2) hipLaunchKernelGGL(k_ceiling_node_gather<2>, dim3(blocks), dim3(256), 0, st, np_, count, nb, iterations, sink.p);
      else if (loads == 3) hipLaunchKernelGGL(k_ceiling_node_gather<3>, dim3(blocks), dim3(256), 0, st, np_, count, nb, iterations, sink.p);
      else if (loads == 4) hipLaunchKernelGGL(k_ceiling_node_gather<4>, dim3(blocks), dim3(256), 0, st, np_, count, nb, iterations, sink.p);
      else if (loads == 5) hipLaunchKernelGGL(k_ceiling_node_gather<5>, dim3(blocks), dim3(256), 0, st, np_, count, nb, iterations, sink.p);
      else if (loads == 7) hipLaunchKernelGGL(k_ceiling_node_gather<7>, dim3(blocks), dim3(256), 0, st, np_, count, nb, iterations, sink.p);
      else hipLaunchKernelGGL(k_ceiling_node_gather<8>, dim3(blocks), dim3(256), 0, st, np_, count, nb, iterations, sink.p);'''
assert old in s
s=s.replace(old,new)
open(p,'w').write(s)
