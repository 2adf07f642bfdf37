"""Print a rocprofv3 kernel_stats.csv compactly: tools/show_stats.py <csv> [min_percent]"""
import csv, sys
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) >= thr:
        print(f'{r["Name"][:56]:56s} {r["Calls"]:>5s} {float(r["AverageNs"]) / 1e3:10.1f} us  {float(r["Percentage"]):6.2f} %')
