#!/bin/bash
# End-of-round-2 kernel-trace stats on the final tree (committed under profiles/r02_final_c): the headline alone and the default bench
# command with its secondary workloads.  (Counter passes: profiles/r02_final -- the K loops have not changed since.)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export OUT=r02_final_c PYTHONUNBUFFERED=1 TMPDIR=/tmp
mkdir -p gpurun_out/$OUT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$OUT/stats_headline -o bench -- python bench.py --no-secondary > gpurun_out/$OUT/bench_stats_headline.log 2>&1
echo "headline stats exit $?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$OUT/stats -o bench -- python bench.py > gpurun_out/$OUT/bench_stats.log 2>&1
echo "default stats exit $?"
find gpurun_out/$OUT -name "*kernel_trace.csv" -delete        # (tens of MB; the stats summaries are what is kept)
tail -1 gpurun_out/$OUT/bench_stats_headline.log | cut -c1-400
head -4 gpurun_out/$OUT/stats_headline/*kernel_stats.csv | cut -c1-200
