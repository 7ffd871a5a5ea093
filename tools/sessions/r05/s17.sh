#!/bin/bash
# round 5, session 17: packed-scale contiguous tiling with its K threshold (parity), the bench self-test with the new secondary line, its bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s17
mkdir -p $OUT
export PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests/test_gemm_gpu.py -q -m gpu -p no:cacheprovider -k "packed_ue8m0 or bench" 2>&1 | tail -15 ) > $OUT/pytest_subset.log 2>&1
grep -E "passed|failed|error" $OUT/pytest_subset.log | tail -3
for WL in contiguous_ue8m0 contiguous; do
  timeout 200 python bench.py --workload $WL --steps 60 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$WL', r['roofline']['kernel'], round(r['roofline']['kernel_us'],2), round(r['roofline']['frac'],4), r['roofline'].get('frac_useful'))"
done 2>&1 | tee $OUT/bench_lines.log
