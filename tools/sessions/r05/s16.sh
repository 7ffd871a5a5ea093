#!/bin/bash
# round 5, session 16: the group-relative tiling for packed scales on the contiguous layout (parity + A/B against the fixed 128-row grid), the
# coalesced skinny forms as the default (parity of the selection), one- against two-subtile skinny tiles, bench lines of the moved workloads
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s16
mkdir -p $OUT
export PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests/test_gemm_gpu.py -q -m gpu -p no:cacheprovider -k "packed_ue8m0 or skinny or decode" 2>&1 | tail -15 ) > $OUT/pytest_subset.log 2>&1
grep -E "passed|failed|error" $OUT/pytest_subset.log | tail -3
timeout 600 python tools/r5b_probe.py packed_c4 > $OUT/packed_c4.jsonl 2> $OUT/packed_c4.err; cat $OUT/packed_c4.jsonl; tail -3 $OUT/packed_c4.err
timeout 600 python tools/r5b_probe.py skinny_w > $OUT/skinny_w.jsonl 2> $OUT/skinny_w.err; cat $OUT/skinny_w.jsonl; tail -3 $OUT/skinny_w.err
for WL in wgrad_ksplit decode_m1 decode_m1_long; do
  timeout 200 python bench.py --workload $WL --steps 60 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$WL', r['roofline']['kernel'], round(r['roofline']['kernel_us'],2), round(r['roofline']['frac'],4))"
done 2>&1 | tee $OUT/bench_lines.log
