#!/bin/bash
# Round 4, GPU session D: packed-scale K tail on the quad kernel, A_EARLY knob A/B, skinny_16 vs skinny_16w as graph replays.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r4d; mkdir -p $OUT
( timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "packed or k_tail or fork_safe or ue8m0" 2>&1 | tail -40 ) > $OUT/pytest.log 2>&1
echo "pytest: $(tail -1 $OUT/pytest.log)"; grep -E "^FAILED|^ERROR|Error|assert " $OUT/pytest.log | head -30
VARIANTS="base ae0 ae1 ae3 ae4" WORKLOADS="dense" ROUNDS=2 STEPS=300 bash tools/gpu_ab_variants.sh 2>&1 | tee $OUT/ab_a_early.log
for w in dgrad_ktail dgrad_ktail_ue8m0 dense_m128; do
  line=$(timeout 200 python bench.py --workload $w --no-cpu-baseline --no-secondary --steps 100 --clock-warmup-s 0.5 2>/dev/null | tail -1)
  echo "$w $(echo "$line" | python -c "import json,sys; p=json.loads(sys.stdin.read()); print(round(p['roofline']['kernel_us'],2), round(p['ms_per_step']*1e3,2), p['roofline']['kernel'], round(p['roofline']['frac'],4))")"
done 2>&1 | tee $OUT/ktail.log
for r in 1 2; do for cfg in skinny_16 skinny_16w; do
  line=$(timeout 200 python bench.py --workload decode_m1_long --config $cfg --no-cpu-baseline --no-secondary --steps 200 --clock-warmup-s 0.5 2>/dev/null | tail -1)
  echo "$r decode_m1_long $cfg $(echo "$line" | python -c "import json,sys; p=json.loads(sys.stdin.read()); print(round(p['roofline']['kernel_us'],2), round(p['ms_per_step']*1e3,2), p['roofline']['kernel'], round(p['roofline']['frac'],4), p.get('eager_call_us'))")"
done; done 2>&1 | tee $OUT/skinny_ab.log
