#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter_collection CSVs (one pass per counter set) into profiles/pmc_k_fisher_tile_v2.json.
usage: tools/pmc_summary.py <tag> <csv> [<csv> ...]      (bench defaults: 500k Gaussians, 64 views, 256^2, C=4)"""
import csv, collections, json, os, sys

tag, files = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        if "k_fisher_tile_v2<4, true, false>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
out = {"kernel": "k_fisher_tile_v2<4,true,false>", "gaussians": 500000, "views": 64, "size": 256, "columns": 4, "round": tag}
if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
    out["FETCH_SIZE_KiB"] = m["FETCH_SIZE"]; out["WRITE_SIZE_KiB"] = m["WRITE_SIZE"]
    out["hbm_bytes_per_launch"] = (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
    out["note"] = ("separate --pmc passes (profiles/%s_pmc_*.csv); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md HBM "
                   "section (FETCH_SIZE halves wide streaming reads on gfx950; this kernel's 8-32 B gathers are not a calibrated access shape)" % tag)
for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
          "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY"):
    if k in m: out[k] = m[k]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# loop statistics of the same launch (tools/loopstats.py with a -DFR_LOOPSTATS build) are kept across refreshes
try:
    old = json.load(open(os.path.join(root, "profiles", "pmc_k_fisher_tile_v2.json")))
    for k in ("contributing_pairs_per_launch", "pass1_iterations_per_launch", "walk_steps_per_launch", "loop_stats_note"):
        if k in old: out[k] = old[k]
except Exception:
    pass
json.dump(out, open(os.path.join(root, "profiles", "pmc_k_fisher_tile_v2.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
