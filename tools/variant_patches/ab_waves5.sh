for round in 1 2; do
  STHIP_LIB=_variants/base.so python3 tools/ab_options.py inner_min_lanes=24 2>&1 | grep Mray | sed "s/^/base            /"
  STHIP_LIB=_variants/w5.so python3 tools/ab_options.py inner_min_lanes=24 2>&1 | grep Mray | sed "s/^/w5 auto         /"
  STHIP_LIB=_variants/w5.so python3 tools/ab_options.py inner_min_lanes=24 lds_stack_levels=24 2>&1 | grep Mray | sed "s/^/w5 lds24        /"
  STHIP_LIB=_variants/w5.so python3 tools/ab_options.py trace_blocks_per_cu=4,5,6 lds_stack_levels=24 2>&1 | grep Mray | sed "s/^/w5 lds24 blocks /"
  STHIP_LIB=_variants/w5.so python3 tools/ab_options.py trace_blocks_per_cu=5,6 lds_stack_levels=20 2>&1 | grep Mray | sed "s/^/w5 lds20 blocks /"
done
