#!/bin/bash
mkdir -p gpurun_out/r2r
timeout 120 python tools/variant_check.py duo_p_256x256,duo_256x256,duo_128x256 4096x4096x7168 pipe_256x256 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2r/bitcheck.log
timeout 200 python tools/cycles.py --configs duo_p_256x256,duo_128x256 --shape 4096x4096x7168 2>&1 | grep -v amdgpu.ids | cut -c1-420 | tee gpurun_out/r2r/cycles.log
timeout 200 python tools/c3_diag.py 2>&1 | grep -v amdgpu.ids | cut -c1-420 | tee gpurun_out/r2r/c3.log
timeout 300 python bench.py --steps 300 --warmup 30 2>&1 | tail -1 > gpurun_out/r2r/bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2r/bench.json').read())
print('C2', d['value'], d['roofline']['kernel_us'])
for s in d['secondary']: print(s['workload'][:60], s['roofline']['kernel'], round(s['roofline']['kernel_us'],1), round(s['roofline']['frac'],3))
PY
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q 2>&1 | tail -4
