#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc.sh <tag> [bench args]
# Collects two PMC passes of bench.py (no tracing domains mixed in) under gpurun_out/pmc_<tag>_{sq,tcc}
tag=$1; shift
# one rank only: the profiler's preloaded library initialises the GPU in the process it starts, which must therefore never spawn ranks
for a in "$@"; do case "$a" in --gpus|--gpus=*) echo "pmc.sh: profile a single rank (no --gpus)" >&2; exit 2;; esac; done
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_${tag}_sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-last-ray-filter --no-ceilings --reps 1 "$@" > gpurun_out/pmc_${tag}_sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_${tag}_tcc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-last-ray-filter --no-ceilings --reps 1 "$@" > gpurun_out/pmc_${tag}_tcc.log 2>&1
for d in gpurun_out/pmc_${tag}_sq gpurun_out/pmc_${tag}_tcc; do python3 tools/pmc_summary.py $d/*/*_counter_collection.csv > $d.txt; done
# HBM-side traffic of the kernels: fabric read requests by size, and WRITE_SIZE (separate passes: TCC has 4 slots)
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d gpurun_out/pmc_${tag}_rd -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-last-ray-filter --no-ceilings --reps 1 "$@" > gpurun_out/pmc_${tag}_rd.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${tag}_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-last-ray-filter --no-ceilings --reps 1 "$@" > gpurun_out/pmc_${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${tag}_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-last-ray-filter --no-ceilings --reps 1 "$@" > gpurun_out/pmc_${tag}_write.log 2>&1
for d in gpurun_out/pmc_${tag}_rd gpurun_out/pmc_${tag}_fetch gpurun_out/pmc_${tag}_write; do python3 tools/pmc_summary.py $d/*/*_counter_collection.csv > $d.txt; done
