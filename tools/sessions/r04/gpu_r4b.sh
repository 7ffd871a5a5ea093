#!/bin/bash
# Round 4, GPU session B: parity of the tree (M0-sharing pieces, mega MoE entry, bounded spin) + same-box A/B of the M0 sharing.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r4b; mkdir -p $OUT
( timeout 1200 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -40 ) > $OUT/pytest.log 2>&1
echo "pytest: $(tail -1 $OUT/pytest.log)"; grep -E "FAILED|Error|assert" $OUT/pytest.log | head -20
VARIANTS="base nom0" WORKLOADS="dense dense_ue8m0" ROUNDS=3 STEPS=300 bash tools/gpu_ab_variants.sh 2>&1 | tee $OUT/ab_m0.log
VARIANTS="base nom0" WORKLOADS="contiguous c3_nt dgrad_ktail dense_sm100" ROUNDS=2 STEPS=100 bash tools/gpu_ab_variants.sh 2>&1 | tee $OUT/ab_m0_other.log
