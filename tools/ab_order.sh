for r in 1 2 3; do
  for f in "" "--no-spatial-order"; do
    python bench.py --steps 10 --warmup 3 --cpu-views 0 $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('order' if '$f'=='' else 'no-order', round(d['value'],1), 'views/s', round(d['ms_per_step'],3), 'ms/step  kernel', round(d['roofline']['kernel_ms'],3))"
  done
done
