"""tools/last_iteration.py <kernel_trace.csv> <marker kernel substring>: the dispatches between the last two launches of the marker
kernel in a rocprofv3 --kernel-trace run (one iteration of a per-view loop), with start / end / duration in microseconds."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
sel = rows[idx[-2] + 1: idx[-1] + 1]
t0 = int(sel[0]["Start_Timestamp"])
busy = 0.0
for r in sel:
    a, b = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    busy += b - a
    print("%9.1f %9.1f %8.1f us  %s" % (a, b, b - a, r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]))
print("iteration %.1f us, kernels %.1f us" % ((int(sel[-1]["End_Timestamp"]) - t0) / 1e3, busy))
