# ... and the textured plain instantiation too
p = "kernels.h"
s = open(p).read()
old = "__launch_bounds__(STHIP_BLOCK, SHADE_BLOCKS) k_shade("
assert s.count(old) == 1
open(p, "w").write(s.replace(old, "__launch_bounds__(STHIP_BLOCK, ((LT || EXT || TEXTURED) && !PROBE) ? 2 : SHADE_BLOCKS) k_shade("))
