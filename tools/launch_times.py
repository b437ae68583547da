#!/usr/bin/env python3
"""Per-dispatch kernel durations of the last bench step from a rocprofv3 --kernel-trace CSV.
usage: launch_times.py <dir containing *_kernel_trace.csv> [dispatches_to_show]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if r["Kernel_Name"].startswith(("k_", "void k_"))]
for r in rows[-n:]:
    name = r["Kernel_Name"].split("(")[0]
    print("%-60s %9.1f us  grid %s wg %s" % (name[:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))))
