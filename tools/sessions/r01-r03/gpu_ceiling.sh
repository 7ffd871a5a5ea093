#!/bin/bash
# Round-3 ceiling evidence (profiles/r03_ceiling): (1) tools/ceiling.py -- this library vs hipBLASLt (torch._scaled_mm) on the same
# reference-quantised C2 operands, same burst method; (2) tools/ubench/mfma_rate ceiling -- register-resident MFMA streams on zeros /
# uniform-random / reference-quantised bytes with the shader clock read two ways; (3) counter passes of the same ubench and of the
# headline (GRBM_GUI_ACTIVE vs SQ_BUSY_CYCLES vs s_memtime); (4) kernel-trace stats of the comparator run (which kernels ran, how long).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${OUT:-r03_ceiling}
mkdir -p $OUT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 600 python tools/ceiling.py --dump-operands $OUT/operands.bin > $OUT/ceiling.log 2>&1
echo "ceiling.py exit $?"; grep -v amdgpu.ids $OUT/ceiling.log | cut -c1-330
timeout 300 tools/ubench/mfma_rate ceiling $OUT/operands.bin > $OUT/mfma_rate_ceiling.log 2>&1
echo "mfma_rate exit $?"; cat $OUT/mfma_rate_ceiling.log | cut -c1-300
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/ubench_pmc -o pmc -- \
    tools/ubench/mfma_rate ceiling $OUT/operands.bin > $OUT/mfma_rate_ceiling_under_pmc.log 2>&1
echo "ubench pmc exit $?"
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/headline_pmc -o pmc -- \
    python bench.py --steps 12 --warmup 4 --clock-warmup-s 0.3 --no-cpu-baseline --no-secondary > $OUT/headline_pmc.log 2>&1
echo "headline pmc exit $?"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ceiling_stats -o ceiling -- python tools/ceiling.py --rounds 1 > $OUT/ceiling_under_trace.log 2>&1
echo "ceiling stats exit $?"
find $OUT -name "*kernel_trace.csv" -path "*ceiling_stats*" -delete
head -12 $OUT/ceiling_stats/*kernel_stats.csv 2>/dev/null | cut -c1-220
rm -f $OUT/operands.bin
for f in $(find $OUT -name "*counter_collection.csv" -o -name "*kernel_trace.csv"); do head -600 $f > $f.tmp && mv $f.tmp $f; done
find $OUT -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" ! -name "*.json" -delete
