#!/bin/bash
# C4 (m-grouped contiguous, 8 groups x ~512 rows, 4096 x 7168): remainder phase of the table path -- K-split duo tiles + reduction kernel (DG_TAB_REM=split) against
# 64 x 128 stream tiles (default), same box, alternating.   WORKLOAD=contiguous|... bash tools/c4_balance_ab.sh
mkdir -p gpurun_out/c4b
for rnd in 1 2 3; do
  for rem in split stream; do
    echo "== DG_TAB_REM=$rem round $rnd"
    DG_TAB_REM=$rem timeout 300 python bench.py --workload ${WORKLOAD:-contiguous} --no-cpu-baseline --no-secondary --steps 400 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print(json.dumps({'ms_per_step': r['ms_per_step'], 'value': r['value'], 'kernel_us': r['roofline'].get('kernel_us'), 'frac': r['roofline']['frac']}))"
  done
done 2>&1 | tee gpurun_out/c4b/ab.log
