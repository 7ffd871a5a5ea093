#!/bin/bash
# round 5, session 11: two-resident 64 x 128 stream tiles (tools/stream2_probe.py) + the expert-MLP bench line with GEMM2 forced either way
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r05_s11
export PYTHONUNBUFFERED=1
timeout 600 python tools/stream2_probe.py > gpurun_out/r05_s11/stream2_probe.jsonl 2> gpurun_out/r05_s11/stream2_probe.err
cat gpurun_out/r05_s11/stream2_probe.jsonl | cut -c1-200
tail -3 gpurun_out/r05_s11/stream2_probe.err
