# sensitivity: twelve more dependent vector instructions per node step (an fma chain on a value of the node), result unused
p='traverse.h'
s=open(p).read()
old="      top = BOUNDED ? min(next_top, limit) : next_top;\n      }\n"
new="""      top = BOUNDED ? min(next_top, limit) : next_top;
      {
        float x = tn0;
#pragma unroll
        for (int k = 0; k < 12; k++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(tf0), "v"(tn1));
        asm volatile("" ::"v"(x));
      }
      }
"""
assert old in s
s=s.replace(old,new)
open(p,'w').write(s)
