#!/bin/bash
# round 5, session 29: the 32-row skinny kernel with coalesced activation loads and three K blocks per chunk (no spill): parity + A/B against skinny_32c
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s29
mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gemm_gpu.py -q -m gpu -p no:cacheprovider -k "skinny" 2>&1 | tail -3 > $OUT/pytest.log; tail -2 $OUT/pytest.log
timeout 600 python tools/r5b_probe.py skinny_32 > $OUT/skinny_32.jsonl 2> $OUT/err.log; cat $OUT/skinny_32.jsonl; tail -2 $OUT/err.log
