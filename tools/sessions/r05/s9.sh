#!/bin/bash
# Round 5, GPU session 9: the pieces of K blocks past the end as out-of-range no-ops (base) against re-reading the last block (DG_VARIANT=reread),
# same box, alternating; parity of the new default first
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r5s9; mkdir -p $OUT
timeout 900 python -m pytest tests/test_full_output_parity_gpu.py tests/test_gemm_gpu.py -q -m gpu -x -k "c2 or c3 or c4 or repeat or layout or dense or k_tail or packed or contiguous or split" -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/pytest.log
for r in 1 2 3 4; do for v in base reread; do
  if [ "$v" = base ]; then unset DG_VARIANT DG_VARIANT_FLAGS; else export DG_VARIANT=$v DG_VARIANT_FLAGS="-DDG_DEAD_PIECES_OOB=0"; fi
  for w in dense c3_nt dense_ue8m0; do
  line=$(timeout 200 python bench.py --workload $w --no-cpu-baseline --no-secondary --steps 400 --clock-warmup-s 0.5 2>/dev/null | tail -1)
  echo "$r $v $w $(echo "$line" | python -c "import json,sys; p=json.loads(sys.stdin.read()); print(round(p['roofline']['kernel_us'],2), p['roofline']['kernel'], p['calc_diff_vs_reference_expr'])")"
  done
done; done 2>&1 | tee $OUT/ab_dead_pieces.log
