#!/bin/bash
# Round 4, GPU session Q: accumulator zeroing pinned under the flight of the first loads (duo: asm ties, base vs nozt variant; e8_quad:
# 512 -> 256 v_accvgpr_write, moved in front of the landing wait: base vs the round-3 tree) -- parity, then same-box A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r4q; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_full_output_parity_gpu.py tests/test_sf_cast_mode_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "not bench_runs" 2>&1 | tail -8 ) > $OUT/pytest.log 2>&1
echo "pytest: $(tail -1 $OUT/pytest.log)"; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head
VARIANTS="base nozt" WORKLOADS="dense c3_nt contiguous" ROUNDS=2 bash tools/gpu_ab_variants.sh 2>&1 | tee $OUT/ab_zero_tie.log
for r in 1 2; do for w in dense_ue8m0 masked_ue8m0; do
  for tree in . _r3tree; do
  line=$(cd $tree && timeout 200 python bench.py --workload $w --no-cpu-baseline --no-secondary --steps 300 --clock-warmup-s 0.5 2>/dev/null | tail -1)
  echo "$r $w $tree $(echo "$line" | python -c "import json,sys; p=json.loads(sys.stdin.read()); print(round(p['roofline']['kernel_us'],2), round(p['ms_per_step']*1e3,2), p['roofline']['kernel'], round(p['roofline']['frac'],4))")"
done; done; done 2>&1 | tee $OUT/e8_vs_r3.log
timeout 100 python tools/cycles_e8.py 2>&1 | grep -v amdgpu.ids | tail -3 | tee $OUT/cycles_e8.log
