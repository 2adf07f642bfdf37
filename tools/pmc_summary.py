#!/usr/bin/env python3
"""Fold the rocprofv3 --pmc passes of tools/pmc_collect.sh (one counter_collection.csv per counter group, merged back under
gpurun_out/pmc_<tag>_<i>/) into profiles/pmc_k_fisher_tile_v4.json and copy the per-group folds to profiles/<tag>_pmc_<i>.txt.
The record is stamped with `kernel_code_id`: the sha256 of the profiled kernel's gfx950 machine code, read out of the library the
counters were taken on (tools/codeobj.py); bench.py recomputes it from the library it has loaded and ignores a record whose id
differs.  Nothing here is edited by hand.  With view groups a step launches the kernel several times: the counters are summed
over the dispatches of one step (`dispatches_per_step`).
usage: tools/pmc_summary.py <tag> [contributing_pairs_per_launch walk_iterations_per_launch]   (bench defaults: 500k Gaussians, 64 views, 256^2, C=4)"""
import collections, csv, glob, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fisher-nerf-customized_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from fisher_rast import _lib   # noqa: E402
import codeobj                 # noqa: E402

tag = sys.argv[1]
KERNEL = "k_fisher_tile_v4"
acc = collections.defaultdict(list)
files = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_*", "*counter_collection.csv")))
assert files, "no counter_collection.csv under gpurun_out/pmc_%s_*" % tag
for f in files:
    for r in csv.DictReader(open(f)):
        if (KERNEL + "(") in r["Kernel_Name"] or r["Kernel_Name"].strip() == KERNEL:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
# the last 3 steps are the steady-state scorer launches of tools/pmc_target.py; a step = `disp` dispatches of the kernel (view groups)
# (the target runs the scorer 4 times: one sizing run + 3 launches)
disp = int(os.environ.get("FR_PMC_DISPATCHES", "0")) or max(1, min(len(v) for v in acc.values()) // 4)
m = {k: sum(v[-3 * disp:]) / 3.0 for k, v in acc.items()}
out = {"kernel": KERNEL, "gaussians": 500000, "views": 64, "size": 256, "columns": 4, "round": tag,
       "commit": subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip(),
       "source_hash": _lib.source_hash(), "dispatches_per_step": disp,
       "kernel_code_id": codeobj.kernel_code_id(_lib.SO_PATH, "k_fisher_tile_v4"),
       "command": "rocprofv3 --pmc <group> -d gpurun_out/pmc_<tag>_<i> -o pmc --output-format csv -- python3 tools/pmc_target.py 4   "
                  "(tools/pmc_collect.sh: one pass per counter group, no trace options beside --pmc)"}
if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
    out["FETCH_SIZE_KiB"] = m["FETCH_SIZE"]; out["WRITE_SIZE_KiB"] = m["WRITE_SIZE"]
    out["hbm_bytes_per_step"] = (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
    out["note"] = ("bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md, HBM section (FETCH_SIZE halves wide streaming "
                   "reads on gfx950; this kernel's 16-64 B gathers are not a calibrated access shape, so 2x is an upper bound)")
for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES",
          "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY",
          "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "TCC_HIT_sum", "TCC_MISS_sum", "GRBM_GUI_ACTIVE"):
    if k in m:
        out[k] = m[k]
if len(sys.argv) > 3:
    out["contributing_pairs_per_step"] = int(float(sys.argv[2]))
    out["walk_iterations_per_step"] = int(float(sys.argv[3]))
    out["loop_stats_note"] = ("(pixel, splat) pairs that pass every test of forward.cu:338-363 and wave-level walk iterations of one "
                              "64-view step, counted by a -DFR_LOOPSTATS build (tools/loopstats.py, FR_DEBUG_MODE 5 and 4)")
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_k_fisher_tile_v4.json"), "w"), indent=1)
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_pmc_*.txt"))):
    shutil.copy(f, os.path.join(ROOT, "profiles", os.path.basename(f)))
print(json.dumps(out, indent=1))
