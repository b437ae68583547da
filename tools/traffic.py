#!/usr/bin/env python3
"""HBM-side traffic of a kernel per LIVE launch from two separate rocprofv3 --pmc passes
(FETCH_SIZE, WRITE_SIZE; tools/pmc.sh) as MI355X_MICROARCH.md prescribes:
  read bytes  = FETCH_SIZE [KiB] * 1024 * 2   (on gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B while every request of
                                               these kernels is a 128-byte one: checked with TCC_EA0_RDREQ_128B)
  write bytes = WRITE_SIZE [KiB] * 1024
Launches that fetched less than 1 MiB are the empty bounce rounds and are not counted as launches.
usage: tools/traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel-substring> <out.json>"""
import csv
import json
import sys


def per_dispatch(path, counter, kernel):
    out = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter and kernel in row["Kernel_Name"]:
                out[row["Dispatch_Id"]] = out.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
    return out


def main(fetch_csv, write_csv, kernel, out_json):
    fetch = per_dispatch(fetch_csv, "FETCH_SIZE", kernel)
    write = per_dispatch(write_csv, "WRITE_SIZE", kernel)
    live_f = [v for v in fetch.values() if v * 1024 > 1 << 20]
    live_w = sorted(write.values(), reverse=True)[: len(live_f)]
    res = {
        "kernel": kernel,
        "live_launches": len(live_f),
        "read_bytes_per_launch": sum(live_f) * 1024 * 2 / max(len(live_f), 1),
        "write_bytes_per_launch": sum(live_w) * 1024 / max(len(live_w), 1),
        "note": "FETCH_SIZE doubled (gfx950: 128-byte requests tallied at 64 B); Infinity-Cache hits are included in the count",
    }
    res["bytes_per_launch"] = res["read_bytes_per_launch"] + res["write_bytes_per_launch"]
    json.dump(res, open(out_json, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main(*sys.argv[1:5])
