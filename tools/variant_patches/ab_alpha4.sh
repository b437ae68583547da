#!/bin/bash
# usage (GPU box): the ALPHA instantiations of k_trace capped at 128 registers (variant alpha4) on scenes that use them
for cfg in "--scene foliage --bdpt-flag alphatest" "--scene foliage --bdpt-flag alphatest --width 3840 --height 2160" "--scene fog_box" "--scene fog_box --bdpt-flag ~defershadowrays"; do
  for round in 1 2; do
    for v in base alpha4; do
      line=$(STHIP_LIB=_variants/$v.so python3 bench.py --steps 10 --warmup 2 --reps 3 --no-cpu-baseline --no-ceilings --no-other-workloads --sustained-seconds 0 --no-last-ray-filter $cfg 2>/dev/null | grep '"metric"')
      python3 -c "
import json,sys
d=json.loads(sys.argv[1]); print('%-6s %-70s | %7.3f ms | %6.0f Mray/s' % (sys.argv[3], sys.argv[2], d['ms_per_step'], d['value']))" "$line" "$cfg" "$v" 2>/dev/null || echo "$v $cfg failed"
    done
  done
done
