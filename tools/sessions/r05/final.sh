#!/bin/bash
# End-of-round-5 evidence (committed under profiles/r05_final): the round-end gate without a profiler (whole GPU suite + default bench line),
# rocprofv3 kernel-trace stats of the headline alone and of the default bench.py command, counter passes (each its own run: counters +
# kernel trace only) of the FP32-scale headline kernel and of the packed-UE8M0 kernel, the traffic JSONs bench.py reads.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${OUT:-r05_final}
mkdir -p $OUT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/ -q -m gpu -p no:cacheprovider 2>&1 | tail -8 ) > $OUT/pytest_gpu.log 2>&1
echo "pytest: $(grep -E 'passed|failed' $OUT/pytest_gpu.log | tail -1)"
timeout 600 python bench.py 2>$OUT/bench_default_run.err > $OUT/bench_default_run.out
tail -1 $OUT/bench_default_run.out > $OUT/bench_default_run.json
echo "bench: $(cut -c1-300 $OUT/bench_default_run.json)"
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
find $OUT -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" ! -name "*.jsonl" ! -name "*.json" ! -name "*.err" ! -name "*.out" -delete
python tools/summarize_prof.py $OUT > $OUT/SUMMARY.txt 2>&1
python tools/make_traffic_json.py $OUT dense_pmc2 dense_pmc3 dg_fp8_gemm_duo_kernel duo_p_256x256
python tools/make_traffic_json.py $OUT dense_ue8m0_pmc2 dense_ue8m0_pmc3 dg_fp8_gemm_quad_e8_kernel e8_quad_256x256
python tools/trim_profiles.py $OUT > /dev/null
cat $OUT/SUMMARY.txt | cut -c1-250 | grep -v "Cijk_\|^void at::" | head -70
tail -1 $OUT/bench_stats_headline.log | cut -c1-400
