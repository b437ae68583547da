"""Per-dispatch kernel durations of one bench step from a rocprofv3 --kernel-trace CSV (GPU box):
python tools/dispatch_times.py <kernel_trace.csv> — prints the last step's dispatches in order."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_generate")][-3]
t0 = int(rows[last]["Start_Timestamp"])
prev_end = t0
for r in rows[last:]:
    if r["Kernel_Name"].startswith("k_generate") and r is not rows[last]:
        break
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-60s start %8.1f us  dur %8.1f us  gap %6.1f us  vgpr %s lds %s grid %s" % (r["Kernel_Name"][:60], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("VGPR_Count", "?"), r.get("LDS_Block_Size", "?"), r.get("Grid_Size", "?")))
    prev_end = e
