#!/bin/bash
# A/B of environment variants on one box: ab_env.sh <rounds> "VAR=val" ...
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for e in "$@"; do
    env $e python bench.py --steps 10 --warmup 3 --cpu-views 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e', round(d['value'],1), 'views/s', round(d['ms_per_step'],3), 'ms/step  kernel', round(d['roofline']['kernel_ms'],3))"
  done
done
