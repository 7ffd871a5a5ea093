#!/bin/bash
# round 2, session A: parity suite + first measurements of the quad e8 kernel
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
tail -5 gpurun_out/r2a/pytest.log
timeout 600 python tools/e8_sweep.py e8_duo_256x256,e8_quad_256x256,e8_quad_v4,e8_quad_v1,e8_quad_v2,e8_quad_v3,e8_quad_v5 4096x4096x7168 200 3 > gpurun_out/r2a/e8_sweep.log 2>&1
cat gpurun_out/r2a/e8_sweep.log
timeout 300 python tools/sustained.py duo_p_256x256 4096x4096x7168 200 2 > gpurun_out/r2a/sustained_fp32.log 2>&1
cat gpurun_out/r2a/sustained_fp32.log
