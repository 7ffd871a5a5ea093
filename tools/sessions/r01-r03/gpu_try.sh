#!/bin/bash
# Scratch runner for one experiment: the tests that cover the touched path, then the benches that measure it.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/try
( timeout 600 python -m pytest tests/test_gemm_gpu.py -x -q -m gpu -k "${TRY_TESTS:-per_column or wgrad or k_grouped}" 2>&1 | tail -4 ) | tee gpurun_out/try/pytest.log
if [ -n "$TRY_BENCH" ]; then
  bash -c "$TRY_BENCH" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/try/bench.log
fi
