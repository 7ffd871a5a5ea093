#!/bin/bash
# Round 4, GPU session U: randomised parity runs on the final tree (dense, grouped, the round-3 paths incl. the tabled contiguous walk --
# now one fused launch -- skinny, fused SwiGLU, cast mode), new seeds.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r4u; mkdir -p $OUT
timeout 400 python tools/fuzz_dense.py 4000 40 2>&1 | grep -v amdgpu.ids | tail -4 | tee $OUT/fuzz_dense.log
timeout 400 python tools/fuzz_grouped.py 5000 40 2>&1 | grep -v amdgpu.ids | tail -4 | tee $OUT/fuzz_grouped.log
timeout 600 python tools/fuzz_round3.py 6000 24 2>&1 | grep -v amdgpu.ids | tail -8 | tee $OUT/fuzz_round3.log
