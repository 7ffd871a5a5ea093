cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c4u
DG_TAB_BALANCE=0 DG_TAB_UNFUSED=1 rocprofv3 --kernel-trace --stats -d gpurun_out/c4u/prof -o c4u -- python bench.py --workload contiguous --no-cpu-baseline --no-secondary --steps 200 > gpurun_out/c4u/run.log 2>&1
find gpurun_out/c4u -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -6 {} | cut -c1-200'
