#!/bin/bash
# usage (GPU box): tools/estimator_times.sh > gpurun_out/estimators.txt — the non-default estimators on the bench scene
# (atrium 1080p, one sample per pixel per step): Mray/s and ms per step, for profiles/README.md's table
for flags in "" "--bdpt-flag connecttoviews" "--bdpt-flag connecttolightpaths" "--bdpt-flag connecttolightpaths --bdpt-flag lightvertexcache --light-path-count 2073600" "--bdpt-flag connecttolightpaths --bdpt-flag lightvertexcache --light-path-count 2073600 --bdpt-flag lvcreservoirs --bdpt-flag lvcreservoirreuse" "--bdpt-flag neereservoirs" "--bdpt-flag neereservoirs --bdpt-flag neereservoirreuse" "--bdpt-flag presamplelights --bdpt-flag coherentsampling" "--bdpt-flag ~coherentrr" "--max-diffuse-vertices 4"; do
  line=$(python3 bench.py --steps 10 --warmup 2 --reps 3 --no-cpu-baseline --no-ceilings $flags 2>/dev/null | grep '"metric"')
  python3 -c "
import json,sys
d=json.loads(sys.argv[1]); print('%-110s | %6.1f M rays/step | %7.3f ms | %6.0f Mray/s' % (sys.argv[2] or 'default', d['config']['rays_per_step']/1e6, d['ms_per_step'], d['value']))" "$line" "$flags"
done
