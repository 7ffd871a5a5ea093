#!/bin/bash
# Round 4, GPU session V: packed-UE8M0 scales with an MN-major B read in place (e8_duo_bmn_256x256): parity, then nn with packed scales
# against the FP32-scale nn call and against the re-majoring route (forced quad kernel) on one box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r4v; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_sf_cast_mode_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "packed or ue8m0 or sm100" 2>&1 | tail -15 ) > $OUT/pytest.log 2>&1
echo "pytest: $(tail -1 $OUT/pytest.log)"; grep -E "^FAILED|^ERROR|Error|assert" $OUT/pytest.log | head
timeout 400 python tools/e8_mn_ab.py 2048x7168x2048 tt 2>&1 | grep -v amdgpu.ids | tee $OUT/e8_mn_ab.log
for r in 1; do for w in c3_nn_ue8m0; do for f in auto; do
  if [ $w = c3_nn ] && [ $f != auto ]; then continue; fi
  line=$(DG_FORCED=$f timeout 200 python - <<PY 2>/dev/null | tail -1
import os, sys, json, subprocess
sys.argv = ['bench.py', '--workload', '$w', '--no-cpu-baseline', '--no-secondary', '--steps', '300', '--clock-warmup-s', '0.5']
import deepgemm_amd as dg
if '$f' != 'auto': dg.set_forced_config('$f')
import runpy
runpy.run_path('bench.py', run_name='__main__')
PY
)
  echo "$r $w $f $(echo "$line" | python -c "import json,sys; p=json.loads(sys.stdin.read()); print(round(p['roofline']['kernel_us'],2), round(p['ms_per_step']*1e3,2), p['roofline']['kernel'], round(p['roofline']['frac'],4), p['calc_diff_vs_reference_expr'])")"
done; done; done 2>&1 | tee $OUT/c3_nn_packed.log
