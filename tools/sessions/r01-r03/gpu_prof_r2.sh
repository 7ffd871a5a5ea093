#!/bin/bash
# Round-2 profiling session (committed evidence under profiles/<OUT>): rocprofv3 kernel-trace stats of the default bench.py
# command (headline + secondary workloads), then PMC passes -- each in its own run, counters + kernel trace only -- of short
# bench.py runs of the FP32-scale headline kernel and of the packed-UE8M0 kernel.  OUT=directory name under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${OUT:-r02_prof}
mkdir -p $OUT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py > $OUT/bench_stats.log 2>&1
echo "stats exit $?"
for WL in dense dense_ue8m0; do
  i=0
  for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
             "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/${WL}_pmc$i -o pmc -- \
        python bench.py --workload $WL --steps 12 --warmup 4 --clock-warmup-s 0.3 --no-cpu-baseline --no-secondary > $OUT/${WL}_pmc$i.log 2>&1
    echo "$WL pmc$i ($PMC) exit $?"
  done
done
find $OUT -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" -delete
python tools/summarize_prof.py $OUT > $OUT/SUMMARY.txt 2>&1
# keep the repository small: the per-dispatch means are in SUMMARY.txt, the CSVs keep their first 400 rows
for f in $(find $OUT -name "*counter_collection.csv" -o -name "*kernel_trace.csv"); do head -400 $f > $f.tmp && mv $f.tmp $f; done
cat $OUT/SUMMARY.txt
