#!/bin/bash
# Round 4, GPU session S: touch-ahead of C in the recipe-(1,1,128) kernel (wgrad / K-grouped): parity, then base / notouch / touch64 on one box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r4s; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_full_output_parity_gpu.py tests/test_reference_sweeps_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "k_grouped or kgrouped or wgrad or per_col or accumulate or 1d1d or recipe" 2>&1 | tail -8 ) > $OUT/pytest.log 2>&1
echo "pytest: $(tail -1 $OUT/pytest.log)"; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head
VARIANTS="base notouch touch64" WORKLOADS="wgrad kgrouped wgrad_ksplit" ROUNDS=2 STEPS=100 bash tools/gpu_ab_variants.sh 2>&1 | tee $OUT/ab_touch.log
for v in base notouch; do
  if [ "$v" = base ]; then unset DG_VARIANT; else export DG_VARIANT=$v; fi
  timeout 200 python tools/cycles.py --configs pipe_pc_256x256 --shape 4096x4096x7168 --per-col 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /" | tee -a $OUT/cycles.log
done
