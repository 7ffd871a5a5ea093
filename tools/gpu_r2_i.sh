#!/bin/bash
mkdir -p gpurun_out/r2i
timeout 300 python tools/grouped_bench.py --cases 1x512x4096x7168,2x256x4096x7168,8x448x4096x7168 --configs auto,duo_128x256,pipe_128x128,pipe_64x256,pipe_32x256,pipe_16x256,pipe_128x256 --iters 20 > gpurun_out/r2i/tail.log 2>&1; cat gpurun_out/r2i/tail.log
