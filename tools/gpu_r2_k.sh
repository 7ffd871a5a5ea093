#!/bin/bash
mkdir -p gpurun_out/r2k
timeout 200 python tools/c3_diag.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2k/c3_diag.log
timeout 200 python tools/c3_diag.py --shape 4096x4096x7168 --layouts nt,nn 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2k/c2_diag.log
timeout 300 python -m pytest tests/test_gemm_gpu.py -x -q -k "split_k" 2>&1 | tail -3
