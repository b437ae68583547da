#!/bin/bash
# usage (GPU box): tools/variant_patches/ab_shade_blocks.sh — the extended k_shade instantiations at 2 instead of 3 blocks per CU
# (variants ext2 / lt2, tools/variant_patches/shade_blocks_*.py) on the bench scene with the estimators that use them
for flags in "--bdpt-flag neereservoirs" "--bdpt-flag connecttoviews" "--bdpt-flag connecttolightpaths" "--bdpt-flag connecttolightpaths --bdpt-flag lightvertexcache"; do
  for round in 1 2; do
    for v in base ext2 lt2; do
      line=$(STHIP_LIB=_variants/$v.so python3 bench.py --steps 10 --warmup 2 --reps 3 --no-cpu-baseline --no-ceilings --no-other-workloads --sustained-seconds 0 --no-last-ray-filter $flags 2>/dev/null | grep '"metric"')
      python3 -c "
import json,sys
d=json.loads(sys.argv[1]); print('%-6s %-60s | %7.3f ms | %6.0f Mray/s' % (sys.argv[3], sys.argv[2], d['ms_per_step'], d['value']))" "$line" "$flags" "$v"
    done
  done
done
