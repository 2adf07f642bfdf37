"""tools/timeline.py <kernel_trace.csv> [n_last]: start / end (us, relative) of the dispatches of the last step(s) of a
`rocprofv3 --kernel-trace --output-format csv` run of bench.py -- which kernels overlap, and what each costs then."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last dispatch of the scorer's reduce kernel ends the last timed step
idx = [i for i, r in enumerate(rows) if "k_reduce_scores" in r["Kernel_Name"]]
end = idx[-1] if idx else len(rows) - 1
sel = rows[max(0, end - n + 1):end + 1]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    a, b = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
    print(f"{a:9.1f} {b:9.1f} {b - a:8.1f} us  q{r.get('Queue_Id', '?'):>3s}  {name}")
