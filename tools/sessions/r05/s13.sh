#!/bin/bash
# round 5, session 13: two-per-CU stream tiles adopted -- parity subset, bench lines (masked, packed masked, expert MLP with the fused L1 either
# way), the wide-rule probe
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s13
mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_mega_gpu.py tests/test_ep_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -4 > $OUT/pytest_mega.log; tail -2 $OUT/pytest_mega.log
timeout 900 python -m pytest tests/test_gemm_gpu.py -q -m gpu -p no:cacheprovider -x -k "masked or stream or e8 or decode or small" 2>&1 | tail -4 > $OUT/pytest_gemm.log; tail -2 $OUT/pytest_gemm.log
for i in 1 2; do
  for WL in masked masked_ue8m0 expert_mlp; do
    timeout 200 python bench.py --workload $WL --steps 60 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$WL', r['roofline']['kernel'], round(r['roofline']['kernel_us'],2), round(r['roofline']['frac'],4))"
  done
  DG_SWIGLU_ONE_PER_CU=1 timeout 200 python bench.py --workload expert_mlp --steps 60 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('expert_mlp one-per-CU L1', r['roofline']['kernel'], round(r['roofline']['kernel_us'],2), round(r['roofline']['frac'],4))"
done 2>&1 | tee $OUT/bench_lines.log
timeout 400 python tools/stream2_wide_probe.py > $OUT/stream2_wide.jsonl 2> $OUT/wide.err; cut -c1-160 $OUT/stream2_wide.jsonl; tail -2 $OUT/wide.err
