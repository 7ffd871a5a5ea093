#!/bin/bash
# Round-2 final profiling session (committed under profiles/r02_final): rocprofv3 kernel-trace stats of the default bench.py
# command and of the headline alone, then the PMC passes of tools/gpu_prof_r2.sh.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export OUT=r02_final PYTHONUNBUFFERED=1 TMPDIR=/tmp
mkdir -p gpurun_out/$OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$OUT/stats_headline -o bench -- python bench.py --no-secondary > gpurun_out/$OUT/bench_stats_headline.log 2>&1
echo "headline stats exit $?"
bash tools/gpu_prof_r2.sh
