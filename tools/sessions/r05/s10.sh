#!/bin/bash
# Round 5, GPU session 10: the register-resident quad schedule with the fragment reads of phase 2 in FRONT of its pieces (e8_quad_h3): parity,
# then same-box A/B against the default schedule and h2
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r5s10; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_reference_sweeps_gpu.py -q -m gpu -x -k "packed or ue8m0 or e8" -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/pytest_e8.log
for r in 1 2 3; do for cfg in e8_quad_256x256 e8_quad_h3_256x256 e8_quad_h2_256x256; do
  line=$(timeout 200 python bench.py --workload dense_ue8m0 --config $cfg --no-cpu-baseline --no-secondary --steps 400 --clock-warmup-s 0.5 2>/dev/null | tail -1)
  echo "$r $cfg $(echo "$line" | python -c "import json,sys; p=json.loads(sys.stdin.read()); print(round(p['roofline']['kernel_us'],2), p['roofline']['kernel'], p['calc_diff_vs_reference_expr'])")"
done; done 2>&1 | tee $OUT/ab_quad_h3.log
