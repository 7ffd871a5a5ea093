#!/bin/bash
# round 5, session 20: a table tile's group id from the tile mask's own load (no second dependent global load in front of the first piece):
# parity of the tabled paths, then the C4 bench lines new / base alternating (tools/gpu_ab.sh)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s20
mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_full_output_parity_gpu.py -q -m gpu -p no:cacheprovider -k "contiguous or group_relative or c4" 2>&1 | tail -4 > $OUT/pytest_subset.log; tail -2 $OUT/pytest_subset.log
export AB_ROUNDS=3
export AB_CMD='for WL in contiguous contiguous_ue8m0; do timeout 200 python bench.py --workload $WL --steps 80 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r[\"config\"][\"kernel\"], round(r[\"roofline\"][\"kernel_us\"],2))"; done'
bash tools/gpu_ab.sh
cp gpurun_out/ab/log.txt $OUT/c4_group_from_mask_ab.log
