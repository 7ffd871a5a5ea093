#!/bin/bash
# Round 5, GPU session 1: the register-resident quad schedule (e8_quad_h*) -- parity, then same-box A/B against e8_quad_256x256 -- and the
# default bench line in its new compact form.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r5s1; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_reference_sweeps_gpu.py -q -m gpu -x -k "packed or ue8m0 or e8" -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/pytest_e8.log
for r in 1 2 3; do for cfg in e8_quad_256x256 e8_quad_h_256x256 e8_quad_h2_256x256; do
  line=$(timeout 200 python bench.py --workload dense_ue8m0 --config $cfg --no-cpu-baseline --no-secondary --steps 300 --clock-warmup-s 0.5 2>/dev/null | tail -1)
  echo "$r $cfg $(echo "$line" | python -c "import json,sys; p=json.loads(sys.stdin.read()); print(round(p['roofline']['kernel_us'],2), round(p['ms_per_step']*1e3,2), p['roofline']['kernel'], round(p['roofline']['frac'],4), p['calc_diff_vs_reference_expr'])")"
done; done 2>&1 | tee $OUT/ab_quad_h.log
for cfg in e8_quad_256x256 e8_quad_h_256x256 e8_quad_h2_256x256; do timeout 120 python tools/cycles_e8.py $cfg 2>&1 | tail -1; done | tee $OUT/cycles_quad_h.log
timeout 400 python bench.py 2>$OUT/bench_default.err | tee $OUT/bench_default.out | tail -1 | cut -c1-2100
