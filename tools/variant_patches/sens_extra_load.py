# sensitivity: one more 4-byte lane request per node step (the word behind the node), result unused
p='traverse.h'
s=open(p).read()
old="        n0 = n[0];\n        n1 = n[1];\n        nz = n[2];\n      }\n      const uint2 cr"
new="        n0 = n[0];\n        n1 = n[1];\n        nz = n[2];\n        sens_extra = *reinterpret_cast<const float*>(base + offset + 48u);\n      }\n      const uint2 cr"
assert old in s
s=s.replace(old,new)
s=s.replace("      float4 n0, n1, nz;\n      if (TOP &&","      float4 n0, n1, nz;\n      float sens_extra = 0.0f;\n      if (TOP &&")
old="      top = BOUNDED ? min(next_top, limit) : next_top;\n      }\n"
new="      top = BOUNDED ? min(next_top, limit) : next_top;\n      asm volatile(\"\" ::\"v\"(sens_extra));\n      }\n"
assert old in s
s=s.replace(old,new)
open(p,'w').write(s)
