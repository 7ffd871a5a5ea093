#!/bin/bash
# Counter passes of this library's C2 kernels next to hipBLASLt's plain-FP8 kernel on the same operands (profiles/r03_ceiling/pmc_compare):
# cycles, instruction mix, LDS activity, waits.  ARMS=comma list of tools/ceiling.py arms.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${OUT:-r03_pmc_compare}
ARMS=${ARMS:-dg_fp32_scales,dg_ue8m0,hipblaslt_tensor}
mkdir -p $OUT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_VALU_MFMA_BUSY_CYCLES" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pmc$i -o pmc -- \
      python tools/ceiling.py --only $ARMS --rounds 1 --burst 30 > $OUT/pmc$i.log 2>&1
  echo "pmc$i ($PMC) exit $?"
done
find $OUT -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" -delete
python tools/summarize_prof.py $OUT > $OUT/SUMMARY.txt 2>&1
for f in $(find $OUT -name "*counter_collection.csv" -o -name "*kernel_trace.csv"); do head -600 $f > $f.tmp && mv $f.tmp $f; done
grep -v "elementwise\|distribution\|reduce_kernel\|copyBuffer\|fillBuffer\|Memcpy\|CatArray\|index" $OUT/SUMMARY.txt | cut -c1-200
