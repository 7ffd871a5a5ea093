#!/bin/bash
# Round 4, GPU session J: M0 sharing on the HBM-bound uses of the 128 x 256 duo tile (masked decode, expert MLP) -- same-box A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r4j; mkdir -p $OUT
( timeout 600 python -m pytest tests/test_gemm_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "c1_unit or dense_nt_vs_oracle or masked_vs_oracle" 2>&1 | tail -5 ) > $OUT/pytest.log 2>&1
echo "pytest: $(tail -1 $OUT/pytest.log)"
for r in 1 2 3; do
  for v in base nom0; do
    if [ "$v" = base ]; then unset DG_VARIANT; else export DG_VARIANT=$v; fi
    echo "== round $r variant $v"
    timeout 200 python tools/masked_bench.py auto 6x20 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('  masked', r.get('groups'), r.get('expected_m'), r.get('n'), r.get('k'), r.get('kernel'), r.get('us'))
"
    line=$(timeout 200 python bench.py --workload expert_mlp --no-cpu-baseline --no-secondary --steps 100 --clock-warmup-s 0.5 2>/dev/null | tail -1)
    echo "  expert_mlp $(echo "$line" | python -c "import json,sys; p=json.loads(sys.stdin.read()); print(round(p['roofline']['kernel_us'],2), round(p['ms_per_step']*1e3,2), p['roofline']['kernel'], p.get('eager_call_us'))")"
  done
done 2>&1 | tee $OUT/ab_m0_hbm.log
