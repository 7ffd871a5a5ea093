#!/bin/bash
# round 5, session 27: coalesced activation loads in the skinny kernels (skinny_16ca / _32ca): parity, then the A/B against the 'c' forms
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s27
mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gemm_gpu.py -q -m gpu -p no:cacheprovider -k "skinny" 2>&1 | tail -5 > $OUT/pytest.log; tail -3 $OUT/pytest.log
timeout 600 python tools/r5b_probe.py skinny_a > $OUT/skinny_a.jsonl 2> $OUT/skinny_a.err; cat $OUT/skinny_a.jsonl; tail -2 $OUT/skinny_a.err
