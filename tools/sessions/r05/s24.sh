#!/bin/bash
# round 5, session 24: every randomised parity tool on the last tree of the round (new seeds)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s24
mkdir -p $OUT
export PYTHONUNBUFFERED=1
( echo "# Randomised parity runs on the LAST tree of round 5 (one MI355X box): tools/fuzz_dense.py 5500 40, fuzz_grouped.py 6500 30, fuzz_packed_mn.py 9500 30, fuzz_round3.py 8500 16, fuzz_round5b.py 5600 40 -- tails of the five logs:"
  timeout 500 python tools/fuzz_dense.py 5500 40 2>&1 | tail -2
  timeout 500 python tools/fuzz_grouped.py 6500 30 2>&1 | tail -2
  timeout 500 python tools/fuzz_packed_mn.py 9500 30 2>&1 | tail -2
  timeout 500 python tools/fuzz_round3.py 8500 16 2>&1 | tail -2
  timeout 500 python tools/fuzz_round5b.py 5600 40 2>&1 | tail -2 ) > $OUT/fuzz_last_tree.log 2>&1
cat $OUT/fuzz_last_tree.log
