#!/bin/bash
mkdir -p gpurun_out/r2c
timeout 300 python tools/quad_check.py > gpurun_out/r2c/quad_check.log 2>&1; tail -30 gpurun_out/r2c/quad_check.log
timeout 300 python tools/sustained.py duo_p_256x256,quad_128x256,quad_256x128 4096x4096x7168 200 3 > gpurun_out/r2c/sustained_c2.log 2>&1; cat gpurun_out/r2c/sustained_c2.log
timeout 300 python tools/sustained.py duo_p_256x256,quad_128x256,quad_256x128,stream_64x128 2048x7168x2048 300 3 > gpurun_out/r2c/sustained_c3.log 2>&1; cat gpurun_out/r2c/sustained_c3.log
timeout 300 python tools/cycles.py --configs duo_p_256x256,quad_128x256,quad_256x128 --shape 4096x4096x7168 > gpurun_out/r2c/cycles_c2.log 2>&1; cat gpurun_out/r2c/cycles_c2.log
timeout 300 python tools/grouped_bench.py --cases 8x512x4096x7168,4x8192x4096x7168 --configs auto,duo_128x256,quad_128x256,quad_256x128 --iters 20 > gpurun_out/r2c/grouped.log 2>&1; cat gpurun_out/r2c/grouped.log
timeout 300 python tools/e8_sweep.py e8_duo_256x256,e8_quad_256x256 4096x4096x7168 200 3 > gpurun_out/r2c/e8.log 2>&1; cat gpurun_out/r2c/e8.log
