#!/bin/bash
# The round-end gate as the driver runs it: the whole GPU suite, then the default bench line.
mkdir -p gpurun_out/full
( time timeout 1500 python -m pytest tests/ -q -m gpu 2>&1 | tail -8 ) 2>&1 | tee gpurun_out/full/pytest.log
timeout 600 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee gpurun_out/full/bench.json
