#!/bin/bash
# round 5, session 22: per-kernel times of the packed-scale contiguous tiling (rocprofv3 kernel-trace stats of its bench line)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s22
mkdir -p $OUT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o c4p -- python bench.py --workload contiguous_ue8m0 --steps 80 --warmup 10 --no-cpu-baseline --no-secondary > $OUT/stats.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
cut -c1-200 $OUT/stats/c4p_kernel_stats.csv | head -8
