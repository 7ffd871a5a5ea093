#!/bin/bash
# Profiling session for the round's committed evidence: kernel-trace stats of the default bench.py command, then PMC
# passes (each in its own run: counters + kernel trace only) of a short bench.py run.  OUT=profiles-style directory name.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${OUT:-prof}
mkdir -p $OUT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py ${BENCH_ARGS} > $OUT/bench_stats.log 2>&1
echo "stats exit $?"
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pmc$i -o pmc -- python bench.py --steps 12 --warmup 4 --no-cpu-baseline ${BENCH_ARGS} > $OUT/pmc$i.log 2>&1
  echo "pmc$i ($PMC) exit $?"
done
find $OUT -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" -delete
find $OUT -type f -size +2M -delete
python tools/summarize_prof.py $OUT > $OUT/SUMMARY.txt 2>&1
cat $OUT/SUMMARY.txt
