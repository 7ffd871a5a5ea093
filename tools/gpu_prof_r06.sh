#!/bin/bash
# Round-6 profiling session (committed evidence: profiles/r06_final/): for each of the workloads named in $WORKLOADS (default: the headline C2,
# C3 nt, C4, packed C2) kernel-trace stats of a bench.py run of that workload, then PMC passes -- each in its OWN run, counters + kernel trace only:
#   pmc1: SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT   (MFMA-busy %, effective clock)
#   pmc2: FETCH_SIZE        pmc3: WRITE_SIZE        pmc4: TCC_HIT_sum TCC_MISS_sum
# and, for the default bench.py command, the kernel-trace stats the headline's `roofline` is checked against.  OUT=gpurun_out/<name>.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${OUT:-r06_final}
mkdir -p $OUT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
if [ -z "$SKIP_DEFAULT" ]; then
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -o bench -- python bench.py > $OUT/bench_default_stats.log 2>&1
  echo "default stats exit $?"
fi
for W in ${WORKLOADS:-dense c3_nt contiguous dense_ue8m0}; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${W}_stats -o bench -- python bench.py --workload $W --steps 200 --warmup 20 --no-cpu-baseline --no-secondary > $OUT/${W}_stats.log 2>&1
  echo "$W stats exit $?"
  i=0
  for PMC in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/${W}_pmc$i -o pmc -- python bench.py --workload $W --steps 40 --warmup 8 --no-cpu-baseline --no-secondary > $OUT/${W}_pmc$i.log 2>&1
    echo "$W pmc$i ($PMC) exit $?"
  done
done
find $OUT -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" ! -name "*.json" -delete
python tools/trim_profiles.py $OUT > /dev/null
find $OUT -type f -size +2M -delete
python tools/summarize_prof.py $OUT > $OUT/SUMMARY.txt 2>&1
tail -q -n 5 $OUT/*_stats.log | grep -v amdgpu | cut -c1-400
