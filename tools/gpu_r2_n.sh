#!/bin/bash
mkdir -p gpurun_out/r2n
timeout 120 python tools/variant_check.py duo_m1_128x256,duo_m2_128x256,duo_m3_128x256 4096x4096x7168 duo_128x256 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2n/bitcheck.log
timeout 300 python tools/cycles.py --configs duo_128x256,duo_m_128x256,duo_m1_128x256,duo_m2_128x256,duo_m3_128x256 --shape 2048x4096x7168 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2n/cycles.log
timeout 300 python tools/grouped_bench.py --cases 8x512x4096x7168 --configs duo_128x256,duo_m_128x256,duo_m1_128x256,duo_m2_128x256,duo_m3_128x256 --iters 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2n/c4.log
