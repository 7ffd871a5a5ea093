#!/bin/bash
# Profiling session: kernel-trace stats of bench.py and PMC passes (each in its own run, no other tracing) of one config.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/prof
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
CFG=${CFG:-pipe_s2_256x256}
rocprofv3 -L > gpurun_out/prof/counters_list.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats -o bench -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/prof/bench_stats.log 2>&1
echo "stats exit $?"
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $PMC --output-format csv -d gpurun_out/prof/pmc$i -o pmc -- python tools/prof_one.py --config $CFG --iters 6 --sets 3 > gpurun_out/prof/pmc$i.log 2>&1
  echo "pmc$i ($PMC) exit $?"
done
find gpurun_out/prof -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" -delete
find gpurun_out/prof -type f -size +8M -delete
du -sh gpurun_out/prof
find gpurun_out/prof -name "*.csv" | head -30
python tools/summarize_prof.py gpurun_out/prof
