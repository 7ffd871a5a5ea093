#!/bin/bash
mkdir -p gpurun_out/r2q
timeout 120 python tools/variant_check.py duo_p_256x256,duo_256x256 4096x4096x7168 pipe_256x256 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2q/bitcheck.log
timeout 120 python tools/variant_check.py duo_p_256x256,duo_256x256 1000x520x128 pipe_256x256 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2q/bitcheck.log
timeout 200 python tools/cycles.py --configs duo_p_256x256,duo_256x256 --shape 4096x4096x7168 2>&1 | grep -v amdgpu.ids | cut -c1-420 | tee gpurun_out/r2q/cycles.log
timeout 200 python tools/c3_diag.py --layouts nt,nn 2>&1 | grep -v amdgpu.ids | cut -c1-420 | tee gpurun_out/r2q/c3.log
timeout 300 python bench.py --no-secondary --steps 300 --warmup 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C2', d['value'], d['roofline']['kernel_us'])"
timeout 300 python bench.py --workload c3_nt --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3nt', d['value'], d['roofline']['kernel_us'])"
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -k "dense or contiguous or masked or repeat or accumulate or full_size" 2>&1 | tail -4
