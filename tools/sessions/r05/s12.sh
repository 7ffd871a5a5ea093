#!/bin/bash
# round 5, session 12: dense side of the two-resident stream tile probe
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r05_s12
export PYTHONUNBUFFERED=1
timeout 600 python tools/stream2_dense_probe.py > gpurun_out/r05_s12/stream2_dense_probe.jsonl 2> gpurun_out/r05_s12/err.log
cut -c1-170 gpurun_out/r05_s12/stream2_dense_probe.jsonl
tail -3 gpurun_out/r05_s12/err.log
