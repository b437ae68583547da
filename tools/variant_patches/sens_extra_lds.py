# sensitivity: one more LDS read per node step (the slot two levels down, clamped), result unused
p='traverse.h'
s=open(p).read()
old="      const uint32_t popped = slot[0];\n"
new="      const uint32_t popped = slot[0];\n      const uint32_t sens_lds = *(volatile uint32_t*)(stack + (top >= 2 * STRIDE ? top - 2 * STRIDE : 0u));\n"
assert old in s
s=s.replace(old,new)
old="      top = BOUNDED ? min(next_top, limit) : next_top;\n      }\n"
new="      top = BOUNDED ? min(next_top, limit) : next_top;\n      asm volatile(\"\" ::\"v\"(sens_lds));\n      }\n"
assert old in s
s=s.replace(old,new)
open(p,'w').write(s)
