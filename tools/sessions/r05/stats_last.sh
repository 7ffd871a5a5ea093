#!/bin/bash
# round 5: rocprofv3 kernel-trace stats of the headline command and of the default bench command on the LAST tree of the round
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_stats_last
mkdir -p $OUT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_headline -o bench -- python bench.py --no-secondary > $OUT/bench_stats_headline.log 2>&1
echo "headline stats exit $?"
timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py > $OUT/bench_stats.log 2>&1
echo "default stats exit $?"
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -type f ! -name "*.csv" ! -name "*.log" -delete
head -3 $OUT/stats_headline/bench_kernel_stats.csv | cut -c1-220
tail -1 $OUT/bench_stats_headline.log | cut -c1-300
grep -E "skinny|swiglu|quad_e8_kernel<128|tab_reduce|pipe_pc_kernel<192" $OUT/stats/bench_kernel_stats.csv | cut -c1-200
