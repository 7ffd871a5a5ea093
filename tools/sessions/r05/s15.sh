#!/bin/bash
# round 5, session 15 (second half of the round): parity of the three candidates -- 192-row tiles of the recipe-(1,1,128) kernel, the grouped nn
# form with packed scales read in place, coalesced weight loads in the skinny kernels -- then their same-box A/B (tools/r5b_probe.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s15
mkdir -p $OUT
export PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests/test_gemm_gpu.py -q -m gpu -p no:cacheprovider -k "per_column or packed_ue8m0_m_grouped or skinny or k_grouped or repeatability or hip_graph" 2>&1 | tail -15 ) > $OUT/pytest_subset.log 2>&1
grep -E "passed|failed|error" $OUT/pytest_subset.log | tail -3
timeout 600 python tools/r5b_probe.py pc192 > $OUT/pc192.jsonl 2> $OUT/pc192.err; cat $OUT/pc192.jsonl; tail -3 $OUT/pc192.err
timeout 600 python tools/r5b_probe.py skinny > $OUT/skinny.jsonl 2> $OUT/skinny.err; cat $OUT/skinny.jsonl; tail -3 $OUT/skinny.err
timeout 600 python tools/r5b_probe.py grouped_nn > $OUT/grouped_nn.jsonl 2> $OUT/grouped_nn.err; cat $OUT/grouped_nn.jsonl; tail -3 $OUT/grouped_nn.err
