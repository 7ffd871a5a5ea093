#!/bin/bash
# Round 4, GPU session R: C4's two GEMM launches as one (dg_fp8_gemm_duo_tab_fused_kernel): parity, then fused vs DG_TAB_UNFUSED=1 on one box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r4r; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_full_output_parity_gpu.py tests/test_reference_sweeps_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "contiguous or grouped or c4 or C4" 2>&1 | tail -8 ) > $OUT/pytest.log 2>&1
echo "pytest: $(tail -1 $OUT/pytest.log)"; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head
for r in 1 2 3; do for u in fused unfused; do
  if [ $u = unfused ]; then export DG_TAB_UNFUSED=1; else unset DG_TAB_UNFUSED; fi
  line=$(timeout 200 python bench.py --workload contiguous --no-cpu-baseline --no-secondary --steps 300 --clock-warmup-s 0.5 2>/dev/null | tail -1)
  echo "$r contiguous $u $(echo "$line" | python -c "import json,sys; p=json.loads(sys.stdin.read()); print(round(p['roofline']['kernel_us'],2), round(p['ms_per_step']*1e3,2), p['roofline']['kernel'], round(p['roofline']['frac'],4), p['calc_diff_vs_reference_expr'])")"
done; done 2>&1 | tee $OUT/ab_fused.log
unset DG_TAB_UNFUSED
timeout 200 python tools/c4_diag.py --configs auto 2>&1 | grep -v amdgpu.ids | tee $OUT/c4_diag.log
timeout 200 python tools/grouped_bench.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee $OUT/grouped_bench.log
