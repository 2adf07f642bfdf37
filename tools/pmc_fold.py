"""Average the counters of a rocprofv3 counter_collection.csv per kernel (scorer kernels only)."""
import collections, csv, sys
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if not any(t in k for t in ("k_fisher", "k_sort", "k_preprocess", "k_scatter", "k_scan", "k_pack", "k_backward", "k_render")):
        continue
    acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    # the last launches are the steady-state scorer launches; the first ones include H_train (other kernel variants)
    print(f"{k[:48]:48s} {c:24s} n={len(v):3d} mean_last3={sum(v[-3:]) / len(v[-3:]):.6g}")
