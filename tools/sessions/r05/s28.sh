#!/bin/bash
# round 5, session 28: one N-subtile with coalesced activation loads (skinny_16ca) against two subtiles (skinny_16wc) where n / 16 is 1..2 rounds
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s28
mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 600 python tools/r5b_probe.py skinny_w > $OUT/skinny_w3.jsonl 2> $OUT/err.log; cat $OUT/skinny_w3.jsonl; tail -2 $OUT/err.log
