#!/bin/bash
# round 5, session 23: two robustness tests of the packed-scale grouped paths (alignment 256 in place; hipGraph capture with / without a workspace)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s23
mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gemm_gpu.py -q -m gpu -p no:cacheprovider -k "alignment_256 or in_a_hip_graph" 2>&1 | tail -25 > $OUT/pytest_subset.log; tail -25 $OUT/pytest_subset.log
