#!/bin/bash
# usage (GPU box, repo root): tools/profile_round.sh <tag>
# Everything profiles/<round>/ holds, from one build: kernel-trace stats of the bench command, the PMC passes, traffic.json.
set -e
tag=$1; shift   # further arguments go to bench.py (e.g. --option wide_bvh=3)
kernel=${KERNEL:-"k_trace<false, false, true, false, 1>"}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# the oracle (bench.py's checker) is built BEFORE the profiler starts anything: a profiled, GPU-initialised process must
# not spawn make / g++; the profiled command itself carries no CPU leg (it would mix seconds of host time into the trace)
python3 -c "import __graft_entry__ as g; g.build_product(); g.build_oracle()"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -- python3 bench.py --steps 10 --warmup 2 --reps 1 --no-cpu-baseline --no-last-ray-filter --sustained-seconds 0 --no-other-workloads "$@" > gpurun_out/prof_${tag}.log 2>&1
grep '"metric"' gpurun_out/prof_${tag}.log > gpurun_out/prof_${tag}_bench_line.json
cp gpurun_out/prof_${tag}/*/*_kernel_stats.csv gpurun_out/prof_${tag}_kernel_stats.csv
tools/pmc.sh ${tag} --sustained-seconds 0 --no-other-workloads "$@"
python3 tools/traffic.py gpurun_out/pmc_${tag}_fetch/*/*_counter_collection.csv gpurun_out/pmc_${tag}_write/*/*_counter_collection.csv "$kernel" gpurun_out/pmc_${tag}_traffic.json
python3 tools/counters.py gpurun_out/pmc_${tag}_sq/*/*_counter_collection.csv gpurun_out/pmc_${tag}_tcc/*/*_counter_collection.csv "$kernel" gpurun_out/pmc_${tag}_counters.json
echo done
