#!/bin/bash
# usage (GPU box): tools/ab_libs.sh <variant> [<variant> ...] — the bench workload (radiance only, tools/ab_options.py) over the
# library variants _variants/<name>.so (tools/build_variant.sh), interleaved twice so that box drift shows
for round in 1 2; do
  for v in "$@"; do
    STHIP_LIB=_variants/$v.so python3 tools/ab_options.py inner_min_lanes=24 2>&1 | grep Mray | sed "s/^/$v  /"
  done
done
