#!/bin/bash
mkdir -p gpurun_out/r2b
python -c "import deepgemm_amd as dg; print(dg.list_configs())" > gpurun_out/r2b/configs.log 2>&1
timeout 600 python tools/e8_sweep.py e8_duo_256x256,e8_quad_256x256,e8_quad_v1,e8_quad_v2,e8_quad_v3 4096x4096x7168 200 3 > gpurun_out/r2b/e8_sweep.log 2>&1
cat gpurun_out/r2b/e8_sweep.log
