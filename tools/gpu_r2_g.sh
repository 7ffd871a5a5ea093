#!/bin/bash
mkdir -p gpurun_out/r2g
python -m pytest tests -m gpu -x -q > gpurun_out/r2g/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2g/pytest.log; tail -4 gpurun_out/r2g/pytest.log
OUT=r02_prof bash tools/gpu_prof_r2.sh > gpurun_out/r2g/prof.log 2>&1; tail -70 gpurun_out/r2g/prof.log
