#!/bin/bash
# Round 5, GPU session 7: SQ counters of the quad kernel's three schedules (default, register-resident h / h2) in one session
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
OUT=gpurun_out/r5s7; mkdir -p $OUT
for cfg in e8_quad_256x256 e8_quad_h2_256x256 e8_quad_h_256x256; do
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $OUT/$cfg -o pmc -- \
      python bench.py --workload dense_ue8m0 --config $cfg --steps 12 --warmup 4 --clock-warmup-s 0.3 --no-cpu-baseline --no-secondary > $OUT/$cfg.log 2>&1
  echo "$cfg exit $?"
done
python tools/summarize_prof.py $OUT 2>&1 | grep "quad_e8" | cut -c1-200
