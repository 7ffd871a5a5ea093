#!/bin/bash
# Round 4, GPU session P: where does C4's remainder phase (64-96 K-split 128-row tiles + reduction) spend its 37 us?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
OUT=gpurun_out/r4p; mkdir -p $OUT
timeout 300 python tools/c4_diag.py --configs auto,duo_128x256,stream_64x128,pipe_64x256,pipe_128x128 2>&1 | grep -v amdgpu.ids | tee $OUT/c4_diag.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4_stats -o c4 -- python bench.py --workload contiguous --no-cpu-baseline --no-secondary --steps 200 > $OUT/c4_stats.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
head -8 $OUT/c4_stats/*/c4_kernel_stats.csv 2>/dev/null | cut -c1-200 || find $OUT/c4_stats -name "*stats*" | head
