#!/bin/bash
# usage (GPU box, repo root): tools/pmc_l1.sh <tag>
# The vector-memory front end of the kernels (texture addresser TA, L1 TCP), one small --pmc pass per counter group of
# bench.py --steps 2 (no tracing domains mixed in; a single rank): is k_trace held by address processing / tag lookups rather than
# by VALU issue or HBM?  Summaries: gpurun_out/pmc_<tag>_l1_<n>.txt
# The data-return block (TD) is not in the list. Round 3's seventh pass asked for a group of TD counters; what happened (its log,
# gpurun_out/pmc_r03_l1_7.log) was not a GPU hang: rocprofiler refused the group at the first HIP call of the profiled process —
# "rocprofiler_create_counter_config ... error code 38: Request exceeds the capabilities of the hardware to collect" (more
# counters of one block than the block has counter registers) — logged it as fatal, the process got SIGABRT 1.5 s after its
# start, and rocprofv3's signal handler ("finalizing after signal 6...") never returned: the tool stalled on the host, no kernel
# had been launched. A TD pass therefore takes ONE counter per pass (append e.g. "TD_TD_BUSY_sum" as a group of its own).
tag=$1
export TMPDIR=/tmp
n=0
for group in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" \
  "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
  "TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum"; do
  n=$((n + 1))
  d=gpurun_out/pmc_${tag}_l1_$n
  rocprofv3 --pmc $group --output-format csv -d $d -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-last-ray-filter --no-ceilings --reps 1 > $d.log 2>&1 || { echo "pass $n failed"; tail -3 $d.log; continue; }
  python3 tools/pmc_summary.py $d/*/*_counter_collection.csv > $d.txt
done
