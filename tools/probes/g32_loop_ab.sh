#!/bin/bash
# dense_ue8m0 against dense_ue8m0_g32 and the K-grouped lines, three rounds on one box (the shifted-scale loop's cost beside the op_sel loop)
for i in 1 2 3; do
  for w in dense_ue8m0 dense_ue8m0_g32; do
    python bench.py --workload $w --steps 200 --warmup 50 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$w', round(d['roofline']['kernel_us'], 2), 'us', d['config'].get('kernel'))"
  done
done
