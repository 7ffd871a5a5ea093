#!/bin/bash
mkdir -p gpurun_out/r2e
python -m pytest tests/test_gemm_gpu.py -m gpu -x -q -k "packed or ue8m0 or repeatab" > gpurun_out/r2e/pytest_e8.log 2>&1; echo "rc=$?" >> gpurun_out/r2e/pytest_e8.log; tail -15 gpurun_out/r2e/pytest_e8.log
