# the extended k_shade instantiations (EXT or LT) at 2 blocks per CU (256 registers) instead of 3: fewer spills, less occupancy
p = "kernels.h"
s = open(p).read()
old = "__launch_bounds__(STHIP_BLOCK, SHADE_BLOCKS) k_shade("
assert s.count(old) == 1
open(p, "w").write(s.replace(old, "__launch_bounds__(STHIP_BLOCK, (LT || EXT) ? 2 : SHADE_BLOCKS) k_shade("))
