#!/bin/bash
# rocprofv3 kernel trace of bench.py --workload contiguous with the environment given (e.g. DG_TAB_UNFUSED=1 DG_TAB_REM=split): per-kernel durations of the phases
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c4u
for tag in "$@"; do
  env $(echo $tag | tr ',' ' ') rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c4u/$tag -o p -- python bench.py --workload contiguous --no-cpu-baseline --no-secondary --steps 200 > gpurun_out/c4u/$tag.log 2>&1
  echo "== $tag"; find gpurun_out/c4u/$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -5 {} | cut -c1-220'
done
find gpurun_out/c4u -type f ! -name "*stats.csv" ! -name "*.log" -delete
