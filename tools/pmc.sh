#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc.sh <tag> [bench args]
# Collects two PMC passes of bench.py (no tracing domains mixed in) under gpurun_out/pmc_<tag>_{sq,tcc}
tag=$1; shift
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_${tag}_sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/pmc_${tag}_sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_${tag}_tcc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/pmc_${tag}_tcc.log 2>&1
for d in gpurun_out/pmc_${tag}_sq gpurun_out/pmc_${tag}_tcc; do python3 tools/pmc_summary.py $d/*/*_counter_collection.csv > $d.txt; done
