#!/usr/bin/env python3
"""Counter-derived figures of one kernel from the separate rocprofv3 --pmc passes tools/pmc.sh collects, written as
profiles/<round>/counters.json (bench.py quotes it, labelled with its source: PMC counters cannot be read from inside
the bench process). Per LIVE launch (dispatches with < 1 % of the busiest dispatch's wave cycles are the empty bounce
rounds of the end of a frame):
  lane_utilisation   = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)     active lanes per issued VALU instruction
  wait_fraction      = SQ_WAIT_ANY / SQ_WAVE_CYCLES                           wave cycles spent waiting on anything
  issue_wait_fraction= SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES                      ... on an instruction to issue
  l2_hit_rate        = TCC_HIT / (TCC_HIT + TCC_MISS)
  l2_bytes           = (TCC_HIT + TCC_MISS) * 128 B                           L2-side request bytes
  fabric_read_bytes  = TCC_EA0_RDREQ * 128 B (every request of these kernels is a 128-byte one: pmc_rdreq.txt)
usage: tools/counters.py <sq_counter_collection.csv> <tcc_counter_collection.csv> <kernel-substring> <out.json>"""
import csv
import json
import sys
from collections import defaultdict


def per_dispatch(path, kernel):
    out = defaultdict(lambda: defaultdict(float))
    with open(path) as f:
        for row in csv.DictReader(f):
            if kernel in row["Kernel_Name"]:
                out[row["Dispatch_Id"]][row["Counter_Name"]] += float(row["Counter_Value"])
    return out


def live(d, key):
    top = max((v[key] for v in d.values()), default=0.0)
    return [v for v in d.values() if v[key] > 0.01 * top]


def main(sq_csv, tcc_csv, kernel, out_json):
    sq = live(per_dispatch(sq_csv, kernel), "SQ_WAVE_CYCLES")
    tcc = live(per_dispatch(tcc_csv, kernel), "TCC_HIT_sum")
    s = lambda rows, k: sum(r[k] for r in rows)
    hit, miss = s(tcc, "TCC_HIT_sum"), s(tcc, "TCC_MISS_sum")
    res = {
        "kernel": kernel,
        "live_launches": len(sq),
        "lane_utilisation": s(sq, "SQ_THREAD_CYCLES_VALU") / max(64.0 * s(sq, "SQ_ACTIVE_INST_VALU"), 1.0),
        "wait_fraction": s(sq, "SQ_WAIT_ANY") / max(s(sq, "SQ_WAVE_CYCLES"), 1.0),
        "issue_wait_fraction": s(sq, "SQ_WAIT_INST_ANY") / max(s(sq, "SQ_WAVE_CYCLES"), 1.0),
        "valu_insts_per_launch": s(sq, "SQ_INSTS_VALU") / max(len(sq), 1),
        "waves_per_launch": s(sq, "SQ_WAVES") / max(len(sq), 1),
        "l2_hit_rate": hit / max(hit + miss, 1.0),
        "l2_bytes_per_launch": (hit + miss) * 128.0 / max(len(tcc), 1),
        "fabric_read_bytes_per_launch": s(tcc, "TCC_EA0_RDREQ_sum") * 128.0 / max(len(tcc), 1),
        "l2_peak_gbs": 34500.0,
        "infinity_cache_gather_gbs": 8600.0,
        "note": "separate --pmc passes of bench.py --steps 2 (tools/pmc.sh); peaks from MI355X_MICROARCH.md",
    }
    json.dump(res, open(out_json, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main(*sys.argv[1:5])
