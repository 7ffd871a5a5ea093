#!/bin/bash
# round 5, session 19: where the fused GEMM1's epilogue time goes -- the expert-MLP bench line with timing ablations of the SwiGLU epilogue
# (sw1: no exponential / division, sw2: no partner wait, sw3: both; wrong values, timing only), alternating with the product library
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s19
mkdir -p $OUT
export PYTHONUNBUFFERED=1
for r in 1 2; do
  for V in "" sw1 sw2 sw3; do
    DG_VARIANT=$V timeout 200 python bench.py --workload expert_mlp --steps 60 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('expert_mlp variant=[$V]', r['roofline']['kernel'], round(r['roofline']['kernel_us'],2))"
  done
  timeout 200 python bench.py --workload masked --steps 60 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('masked (plain C5)', r['roofline']['kernel'], round(r['roofline']['kernel_us'],2))"
done 2>&1 | tee $OUT/swiglu_epilogue_ablation.log
