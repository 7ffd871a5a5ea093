#!/bin/bash
# Round 5, GPU session 2: (i) what the scale operand pair of v_mfma_scale costs the quad kernel (libdeepgemm_amd_noscale.so: the plain MFMA in
# the same instruction stream, results garbage) -- is hipBLASLt's 77 us reachable with hardware scaling in the loop?  (ii) VERDICT 1(c)'s
# ubench: one wave per SIMD, MFMA + 4..7 VALU + reads + pieces per step.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r5s2; mkdir -p $OUT
timeout 120 tools/ubench/issue_rate 2>&1 | tee $OUT/issue_rate.log
for r in 1 2 3; do for v in base noscale; do for cfg in e8_quad_256x256 e8_quad_h2_256x256; do
  if [ "$v" = base ]; then unset DG_VARIANT; else export DG_VARIANT=$v; fi
  line=$(timeout 200 python bench.py --workload dense_ue8m0 --config $cfg --no-cpu-baseline --no-secondary --steps 300 --clock-warmup-s 0.5 2>/dev/null | tail -1)
  echo "$r $v $cfg $(echo "$line" | python -c "import json,sys; p=json.loads(sys.stdin.read()); print(round(p['roofline']['kernel_us'],2), round(p['ms_per_step']*1e3,2), p['roofline']['kernel'], round(p['roofline']['frac'],4))")"
done; done; done 2>&1 | tee $OUT/ab_noscale.log
unset DG_VARIANT
