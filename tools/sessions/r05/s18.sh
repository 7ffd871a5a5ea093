#!/bin/bash
# round 5, session 18: randomised parity runs of the second-half paths (tools/fuzz_round5b.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s18
mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 800 python tools/fuzz_round5b.py 5100 60 > $OUT/fuzz_round5b.log 2>&1
tail -3 $OUT/fuzz_round5b.log; grep -c " ok\|x" $OUT/fuzz_round5b.log; grep FAILED $OUT/fuzz_round5b.log | head -10
