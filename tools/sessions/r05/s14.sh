#!/bin/bash
# round 5, session 14: gate on the tree with the two-per-CU stream tiles adopted (whole GPU suite + default bench) and the expert-MLP line with GEMM2
# forced onto the 128 x 256 duo tile (the round-4 pick) beside the new default
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s14
mkdir -p $OUT
export PYTHONUNBUFFERED=1
( time timeout 1500 python -m pytest tests/ -q -m gpu -p no:cacheprovider 2>&1 | tail -8 ) > $OUT/pytest_gpu.log 2>&1
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -1
timeout 600 python bench.py 2>$OUT/bench.err > $OUT/bench.out
tail -1 $OUT/bench.out | cut -c1-1900
for i in 1 2; do
  for CFG in auto duo_128x256; do
    timeout 200 python bench.py --workload expert_mlp --config $CFG --steps 60 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('expert_mlp GEMM2=$CFG', r['roofline']['kernel'], round(r['roofline']['kernel_us'],2), round(r['roofline']['frac'],4))"
  done
done 2>&1 | tee $OUT/expert_mlp_gemm2_ab.log
