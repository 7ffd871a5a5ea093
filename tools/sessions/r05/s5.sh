#!/bin/bash
# Round 5, GPU session 5: same-box A/B of the chained stream ring (base) against the drained one (DG_VARIANT=nochain: -DDG_STREAM_CHAIN=0)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r5s5; mkdir -p $OUT
for r in 1 2; do for v in base nochain; do
  if [ "$v" = base ]; then unset DG_VARIANT; else export DG_VARIANT=$v DG_VARIANT_FLAGS="-DDG_STREAM_CHAIN=0"; fi
  for w in masked dense_m128 expert_mlp; do
    line=$(timeout 200 python bench.py --workload $w --no-cpu-baseline --no-secondary --steps 200 --clock-warmup-s 0.5 2>/dev/null | tail -1)
    echo "$r $v $w $(echo "$line" | python -c "import json,sys; p=json.loads(sys.stdin.read()); print(round(p['roofline']['kernel_us'],2), p['roofline']['kernel'])")"
  done
  python tools/l2_shape_probe.py 2>&1 | grep -v amdgpu | grep "stream_nt_64x128\|auto" | sed "s/^/$r $v /"
  timeout 300 python tools/masked_bench.py auto 32x20 2>&1 | grep -v amdgpu | sed "s/^/$r $v /" | cut -c1-200
done; done 2>&1 | tee $OUT/ab_chain.log
