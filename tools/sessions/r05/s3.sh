#!/bin/bash
# Round 5, GPU session 3: packed-scale K tail with MN-major B in place (e8_duo_bmn_kt_256x256): parity + the dgrad line; mega / host / layout tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r5s3; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_mega_gpu.py tests/test_layout_gpu.py -q -m gpu -x -k "k_tail or mega or swiglu or exchange or layout or transpose or fork" -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/pytest.log
for w in dgrad_ktail_ue8m0 dgrad_ktail dense_sfa_rowmajor; do
  line=$(timeout 200 python bench.py --workload $w --no-cpu-baseline --no-secondary --steps 200 --clock-warmup-s 0.5 2>/dev/null | tail -1)
  echo "$w $(echo "$line" | python -c "import json,sys; p=json.loads(sys.stdin.read()); print(round(p['roofline']['kernel_us'],2), round(p['ms_per_step']*1e3,2), p['roofline']['kernel'], round(p['roofline']['frac'],4), p['calc_diff_vs_reference_expr'])")"
done 2>&1 | tee $OUT/bench.log
