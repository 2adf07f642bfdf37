#!/bin/bash
# A/B of builds of libfisher_rast.so on the same GPU box, interleaved: tools/ab.sh <rounds> <other.so> [<other2.so> ...]
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for so in "" "$@"; do
    if [ -z "$so" ]; then label=current; unset FISHER_RAST_SO; else label=$(basename $so); export FISHER_RAST_SO=$PWD/$so; fi
    python bench.py --steps 10 --warmup 3 --cpu-views 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', round(d['value'],1), 'views/s', round(d['ms_per_step'],3), 'ms/step  kernel', round(d['roofline']['kernel_ms'],3))"
  done
done
