#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s26
mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/ -q -m gpu -p no:cacheprovider --tb=short 2>&1 | tail -60 > $OUT/pytest.log; tail -40 $OUT/pytest.log
