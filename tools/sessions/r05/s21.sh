#!/bin/bash
# round 5, session 21: the remainder walk of the packed-scale contiguous tiling cut along K (TABSK + dg_e8_tab_reduce_kernel): parity, the
# randomised runs of the packed paths, then the bench line with / without the split (DG_E8_TAB_UNSPLIT) alternating
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s21
mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gemm_gpu.py -q -m gpu -p no:cacheprovider -k "packed_ue8m0 or hip_graph" 2>&1 | tail -12 > $OUT/pytest_subset.log; tail -3 $OUT/pytest_subset.log
timeout 600 python tools/fuzz_round5b.py 5300 24 packedtab,groupednn > $OUT/fuzz.log 2>&1; tail -2 $OUT/fuzz.log; grep FAILED $OUT/fuzz.log | head -5
for r in 1 2 3; do
  for V in split unsplit; do
    if [ $V = unsplit ]; then export DG_E8_TAB_UNSPLIT=1; else unset DG_E8_TAB_UNSPLIT; fi
    timeout 200 python bench.py --workload contiguous_ue8m0 --steps 80 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('contiguous_ue8m0 $V', r['config']['kernel'], round(r['roofline']['kernel_us'],2), round(r['roofline']['frac'],4))"
  done
done 2>&1 | tee $OUT/packed_c4_remainder_ksplit_ab.log
