#!/bin/bash
# Round 5, GPU session 8: boustrophedon step order in the matrix segments of the FP32-scale duo kernel (the headline), same box, alternating;
# parity of the variant first (bit-identical accumulator chains: the repeatability / every-element tests must pass with it)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r5s8; mkdir -p $OUT
DG_VARIANT=dserp DG_VARIANT_FLAGS="-DDG_DUO_SERPENTINE" timeout 900 python -m pytest tests/test_full_output_parity_gpu.py tests/test_gemm_gpu.py -q -m gpu -x -k "c2 or c3 or repeat or layout or dense or k_tail" -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/pytest_variant.log
for r in 1 2 3 4; do for v in base dserp; do
  if [ "$v" = base ]; then unset DG_VARIANT DG_VARIANT_FLAGS; else export DG_VARIANT=$v DG_VARIANT_FLAGS="-DDG_DUO_SERPENTINE"; fi
  for w in dense c3_nt; do
  line=$(timeout 200 python bench.py --workload $w --no-cpu-baseline --no-secondary --steps 400 --clock-warmup-s 0.5 2>/dev/null | tail -1)
  echo "$r $v $w $(echo "$line" | python -c "import json,sys; p=json.loads(sys.stdin.read()); print(round(p['roofline']['kernel_us'],2), p['roofline']['kernel'], p['calc_diff_vs_reference_expr'])")"
  done
done; done 2>&1 | tee $OUT/ab_duo_serpentine.log
