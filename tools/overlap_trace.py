"""Reads a rocprofv3 --kernel-trace CSV of tools/frames_in_flight.py and prints the dispatches of ~two frames as a timeline
with their queue ids: do kernels of the two streams overlap?"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if not r["Kernel_Name"].startswith("__amd")]
n = len(rows)
sel = rows[int(n * 0.8):int(n * 0.8) + 40]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("q%-3s %-28s %8.1f -> %8.1f us (%7.1f)" % (r.get("Queue_Id", "?"), r["Kernel_Name"][:28], s / 1e3, e / 1e3, (e - s) / 1e3))
