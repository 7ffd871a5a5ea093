#!/bin/bash
# Round 5, GPU session 4: the stream kernels with the ring chained across tiles -- parity of everything that runs them, then the shapes with
# several short tiles per CU (expert-MLP L2, masked decode sweep) and the one-tile-per-CU lines that must not move (C5, m = 128)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r5s4; mkdir -p $OUT
timeout 1200 python -m pytest tests -q -m gpu -x -k "masked or stream or decode or skinny or small or m128 or mega or swiglu or sweep or repeat or full_output" -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/pytest.log
python tools/l2_shape_probe.py 2>&1 | grep -v amdgpu | tee $OUT/l2_probe.log
for w in masked masked_ue8m0 expert_mlp dense_m128; do
  line=$(timeout 200 python bench.py --workload $w --no-cpu-baseline --no-secondary --steps 200 --clock-warmup-s 0.5 2>/dev/null | tail -1)
  echo "$w $(echo "$line" | python -c "import json,sys; p=json.loads(sys.stdin.read()); print(round(p['roofline']['kernel_us'],2), round(p['ms_per_step']*1e3,2), p['roofline']['kernel'], round(p['roofline']['frac'],4), p['calc_diff_vs_reference_expr'])")"
done 2>&1 | tee $OUT/bench.log
timeout 300 python tools/masked_bench.py auto 32x20 2>&1 | grep -v amdgpu | tee $OUT/masked_sweep_32x20.log
