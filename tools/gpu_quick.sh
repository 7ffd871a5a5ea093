#!/bin/bash
# Quick GPU iteration: selected parity tests + a config sweep.  TESTS="-k expr" CONFIGS=a,b SHAPES=... bash tools/gpu_quick.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout ${PYTEST_TIMEOUT:-600} python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider ${TESTS} > gpurun_out/pytest_quick.log 2>&1
echo "pytest exit $?"; tail -15 gpurun_out/pytest_quick.log
timeout 600 python tools/sweep.py --out gpurun_out/sweep_quick.jsonl ${CONFIGS:+--configs $CONFIGS} ${SHAPES:+--shapes $SHAPES} ${SWEEP_ARGS} 2>&1 | grep -v amdgpu.ids
