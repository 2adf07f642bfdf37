for r in 1 2; do
  for so in "" tools/_build/ab_noqc.so; do
    if [ -z "$so" ]; then unset FISHER_RAST_SO; echo "== product"; else export FISHER_RAST_SO=$PWD/$so; echo "== $so"; fi
    python tools/outh_bench.py 500000 4 2>&1 | grep -E "16 views|64 views acc|  1 views"
    python tools/outh_bench.py 500000 11 2>&1 | grep -E "16 views|64 views acc"
  done
done
