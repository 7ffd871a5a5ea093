#!/bin/bash
# C4 (m-grouped contiguous, 8 groups x ~512 rows, 4096 x 7168): the two-phase table walk (DG_TAB_BALANCE=0) against the balanced walk, same box, alternating.
mkdir -p gpurun_out/c4b
for rnd in 1 2 3; do
  for bal in 0 ${BALANCES:-1796}; do
    echo "== DG_TAB_BALANCE=$bal round $rnd"
    DG_TAB_BALANCE=$bal timeout 300 python bench.py --workload ${WORKLOAD:-contiguous} --no-cpu-baseline --no-secondary --steps 400 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print(json.dumps({'ms_per_step': r['ms_per_step'], 'value': r['value'], 'kernel_us': r['roofline'].get('kernel_us'), 'frac': r['roofline']['frac']}))"
  done
done 2>&1 | tee gpurun_out/c4b/ab.log
