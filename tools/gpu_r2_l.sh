#!/bin/bash
mkdir -p gpurun_out/r2l
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/ubench/permlane_swap_probe.hip -o /tmp/permlane_probe 2>/dev/null && /tmp/permlane_probe | tee gpurun_out/r2l/permlane_probe.log
timeout 600 python -m pytest tests/test_gemm_gpu.py -x -q -k "layouts or mn_major or native" 2>&1 | tail -8
timeout 200 python tools/c3_diag.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2l/c3_diag.log
