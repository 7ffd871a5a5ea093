#!/bin/bash
# C4 (contiguous, 8 groups x ~512 rows, N 4096, K 7168): which workgroups walk their K-split remainder pieces BEFORE their 256-row tiles
# (DG_TAB_REM_FIRST = 0 none / 1 every second workgroup of an XCD / 2 all), same box, alternating; bit-identity by the parity test.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/c4
for round in 1 2 3; do
  for v in 0 1 2; do
    DG_TAB_REM_FIRST=$v timeout 300 python bench.py --workload contiguous --steps 200 --warmup 30 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | \
      python -c "import sys,json; r=json.loads(sys.stdin.read()); print('rem_first=$v', r['config']['kernel'], round(r['roofline']['kernel_us'],2), 'us', r['calc_diff_vs_reference_expr'])" | tee -a gpurun_out/c4/rem_first_ab.log
  done
done
for v in 1 2; do
  DG_TAB_REM_FIRST=$v timeout 600 python -m pytest tests/test_full_output_parity_gpu.py -q -x -k "c4" -p no:cacheprovider 2>&1 | tail -2 | tee -a gpurun_out/c4/rem_first_ab.log
done
