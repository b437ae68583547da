# only the light-subpath k_shade instantiations (LT) at 2 blocks per CU
p = "kernels.h"
s = open(p).read()
old = "__launch_bounds__(STHIP_BLOCK, SHADE_BLOCKS) k_shade("
assert s.count(old) == 1
open(p, "w").write(s.replace(old, "__launch_bounds__(STHIP_BLOCK, LT ? 2 : SHADE_BLOCKS) k_shade("))
