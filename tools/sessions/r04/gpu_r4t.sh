#!/bin/bash
# Round 4, GPU session T: the tests added after the evidence run + the expert-MLP split.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r4t; mkdir -p $OUT
( timeout 600 python -m pytest tests/test_sf_cast_mode_gpu.py tests/test_gemm_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "sm100 or group_relative or skip_head_mid or masked" 2>&1 | tail -15 ) > $OUT/pytest.log 2>&1
echo "pytest: $(tail -1 $OUT/pytest.log)"; grep -E "^FAILED|^ERROR|Error" $OUT/pytest.log | head
timeout 200 python tools/mlp_bench.py > $OUT/expert_mlp.log 2>&1; echo "mlp exit $?"; tail -4 $OUT/expert_mlp.log
