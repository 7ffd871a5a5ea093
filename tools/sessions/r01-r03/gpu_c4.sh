#!/bin/bash
# C4 (BASELINE configs[3]) diagnosis: per-tile stamps of each forced configuration, then the kernel-trace stats of the automatic choice
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
mkdir -p gpurun_out/c4
timeout 300 python tools/c4_diag.py ${C4_ARGS} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c4/diag.jsonl
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c4/stats -o c4 -- python tools/grouped_bench.py --cases 8x512x4096x7168 --configs auto > gpurun_out/c4/prof_stdout.log 2>&1
find gpurun_out/c4/stats -name "*kernel_stats*" | head -1 | xargs head -8
