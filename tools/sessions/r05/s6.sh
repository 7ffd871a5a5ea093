#!/bin/bash
# Round 5, GPU session 6: does the ORDER of the MFMAs matter on a power-limited part?  Boustrophedon order in the quad kernel (one operand of
# every MFMA equals its predecessor's) against the row-major order, same box, alternating; results bit-identical.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r5s6; mkdir -p $OUT
for r in 1 2 3 4; do for v in base serp; do
  if [ "$v" = base ]; then unset DG_VARIANT DG_VARIANT_FLAGS; else export DG_VARIANT=$v DG_VARIANT_FLAGS="-DDG_SERPENTINE"; fi
  line=$(timeout 200 python bench.py --workload dense_ue8m0 --no-cpu-baseline --no-secondary --steps 400 --clock-warmup-s 0.5 2>/dev/null | tail -1)
  echo "$r $v $(echo "$line" | python -c "import json,sys; p=json.loads(sys.stdin.read()); print(round(p['roofline']['kernel_us'],2), p['roofline']['kernel'], p['calc_diff_vs_reference_expr'])")"
done; done 2>&1 | tee $OUT/ab_serpentine.log
