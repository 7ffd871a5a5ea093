#!/bin/bash
mkdir -p gpurun_out/r2m
for shp in 4096x4096x7168 4096x576x7168 640x2048x384 1000x520x1024; do
timeout 120 python tools/variant_check.py duo_m_128x256 $shp duo_128x256 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r2m/bitcheck.log
timeout 300 python tools/sweep.py --shapes 4096x576x7168,4096x4096x7168,640x4096x7168 --configs duo_128x256,duo_m_128x256 --rounds 3 --iters 10 --sets 2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2m/dense.log
timeout 300 python tools/grouped_bench.py --cases 8x512x4096x7168 --configs duo_128x256,duo_m_128x256,duo_sk_128x256,duo_sk_m_128x256 --iters 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2m/c4.log
timeout 300 python tools/grouped_bench.py --cases 8x512x4096x7168 --nn --configs duo_bmn_128x256,duo_bmn_m_128x256 --iters 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2m/c4nn.log
