#!/bin/bash
# A/B of bench.py flags on one box: tools/ab_flags.sh <rounds> "<flags A>" "<flags B>" ...
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for f in "$@"; do
    python bench.py --steps 10 --warmup 3 --cpu-views 0 $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$f]', round(d['value'],1), 'views/s', round(d['ms_per_step'],3), 'ms/step  kernel', round(d['roofline']['kernel_ms'],3))"
  done
done
