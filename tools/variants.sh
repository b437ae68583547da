#!/bin/bash
# Runs bench.py once per library build under _variants/ (kernel experiments; see STHIP_LIB in stratum_amd/_lib.py).
# usage: tools/variants.sh [bench args]
for so in stratum_amd/libstratum_hip.so _variants/*.so; do
  line=$(STHIP_LIB=$PWD/$so python3 bench.py --no-cpu-baseline "$@" 2>&1 | tail -1)
  echo "$so $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"])')"
done
