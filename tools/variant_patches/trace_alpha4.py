# k_trace's ALPHA instantiations (alpha masks, volumes: 144-153 registers, 3 waves per SIMD) capped at 128 registers (4 waves)
p = "kernels.h"
s = open(p).read()
old = "__global__ void __launch_bounds__(STHIP_BLOCK) STHIP_TRACE_ATTR k_trace("
assert s.count(old) == 1
open(p, "w").write(s.replace(old, "__global__ void __launch_bounds__(STHIP_BLOCK, ALPHA ? 4 : 1) STHIP_TRACE_ATTR k_trace("))
