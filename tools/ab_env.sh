#!/bin/bash
# A/B of the rig's environment switches on one box: ab_env.sh <rounds> "VAR=val" ...  (the switches exist in -DFR_AB builds only:
# tools/build_variant.sh rig, then FISHER_RAST_SO=tools/_build/ab_rig.so is set here)
rounds=$1; shift
export FISHER_RAST_SO="${FISHER_RAST_SO:-$PWD/tools/_build/ab_rig.so}"
[ -f "$FISHER_RAST_SO" ] || bash tools/build_variant.sh rig
for r in $(seq 1 $rounds); do
  for e in "$@"; do
    env $e python bench.py --steps 10 --warmup 3 --cpu-views 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e', round(d['value'],1), 'views/s', round(d['ms_per_step'],3), 'ms/step  kernel', round(d['roofline']['kernel_ms'],3))"
  done
done
