#!/bin/bash
mkdir -p gpurun_out/r2j
A=512,512,512,512,512,512,512
for k in 7168 4096 2048; do
for ms in $A,512 $A,640 512,512,512,512,512,512,640,640 512,512,512,512,640,640,640,640 640,640,640,640,640,640,640,640; do
timeout 300 python tools/grouped_bench.py --cases 8x0x4096x$k --ms $ms --configs duo_128x256,duo_sk_128x256 --iters 20 2>&1 | grep -v amdgpu.ids
done; done | tee gpurun_out/r2j/sk3.log
timeout 600 python -m pytest tests/test_gemm_gpu.py -x -q -k "split_k" 2>&1 | tail -3
