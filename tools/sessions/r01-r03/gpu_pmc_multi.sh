#!/bin/bash
# PMC passes (each its own run, counters only) for several kernel configurations: CFGS="a b c" bash tools/gpu_pmc_multi.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/pmc
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
for CFG in ${CFGS:-pipe_256x256}; do
  i=0
  for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
             "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d gpurun_out/pmc/${CFG}_$i -o pmc -- python tools/prof_one.py --config $CFG --iters 6 --sets 3 ${SHAPE:+--shape $SHAPE} > gpurun_out/pmc/${CFG}_$i.log 2>&1
    echo "$CFG pmc$i exit $?"
  done
done
find gpurun_out/pmc -type f ! -name "*.csv" ! -name "*.log" -delete
python tools/summarize_prof.py gpurun_out/pmc
