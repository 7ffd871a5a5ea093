#!/bin/bash
# End-of-round-3 evidence (committed under profiles/r03_final): rocprofv3 kernel-trace stats of the headline alone and of the default
# bench.py command, counter passes (each in its own run: counters + kernel trace only) of the FP32-scale headline kernel and of the
# packed-UE8M0 kernel, the reference's perf sweeps with the automatic selection (tools/survey.py), the decode-size expert MLP.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${OUT:-r03_final}
mkdir -p $OUT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_headline -o bench -- python bench.py --no-secondary > $OUT/bench_stats_headline.log 2>&1
echo "headline stats exit $?"
timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py > $OUT/bench_stats.log 2>&1
echo "default stats exit $?"
find $OUT -name "*kernel_trace.csv" -delete        # (tens of MB; the stats summaries are what is kept)
for WL in dense dense_ue8m0; do
  i=0
  for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/${WL}_pmc$i -o pmc -- \
        python bench.py --workload $WL --steps 12 --warmup 4 --clock-warmup-s 0.3 --no-cpu-baseline --no-secondary > $OUT/${WL}_pmc$i.log 2>&1
    echo "$WL pmc$i ($PMC) exit $?"
  done
done
timeout 600 python tools/survey.py > $OUT/survey_reference_sweeps.jsonl 2> $OUT/survey.err
echo "survey exit $?"
timeout 200 python tools/mlp_bench.py > $OUT/expert_mlp.log 2>&1
echo "mlp exit $?"
find $OUT -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" ! -name "*.jsonl" ! -name "*.err" -delete
python tools/summarize_prof.py $OUT > $OUT/SUMMARY.txt 2>&1
for f in $(find $OUT -name "*counter_collection.csv" -o -name "*kernel_trace.csv"); do head -400 $f > $f.tmp && mv $f.tmp $f; done
cat $OUT/SUMMARY.txt | cut -c1-250 | head -80
tail -1 $OUT/bench_stats_headline.log | cut -c1-600
