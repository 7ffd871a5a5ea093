#!/bin/bash
mkdir -p gpurun_out/r2d
timeout 300 python tools/quad_check.py > gpurun_out/r2d/quad_check.log 2>&1; tail -3 gpurun_out/r2d/quad_check.log
timeout 300 python tools/cycles.py --configs duo_p_256x256,quad_128x256,quad_256x128 --shape 4096x4096x7168 > gpurun_out/r2d/cycles_quad2.log 2>&1; cat gpurun_out/r2d/cycles_quad2.log
timeout 300 python tools/grouped_bench.py --cases 8x512x4096x7168 --configs duo_128x256,quad_128x256,quad_256x128 --iters 20 > gpurun_out/r2d/grouped2.log 2>&1; cat gpurun_out/r2d/grouped2.log
timeout 300 python tools/sustained.py duo_p_256x256,quad_128x256,quad_256x128 2048x7168x2048 300 3 > gpurun_out/r2d/sustained_c3.log 2>&1; cat gpurun_out/r2d/sustained_c3.log
