#!/bin/bash
# round 5, gate 4: whole GPU suite + default bench line on the tree with the 192-row recipe tiles, the coalesced skinny loads, the packed-scale
# contiguous tiling and the grouped nn form in place
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${OUT:-r05_gate4}
mkdir -p $OUT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/ -q -m gpu -p no:cacheprovider 2>&1 | tail -8 ) > $OUT/pytest_gpu.log 2>&1
echo "pytest: $(grep -E 'passed|failed' $OUT/pytest_gpu.log | tail -1)"
timeout 600 python bench.py 2>$OUT/bench_default.err > $OUT/bench_default.out
tail -1 $OUT/bench_default.out > $OUT/bench_default.json
grep "^secondary_detail" $OUT/bench_default.out > $OUT/bench_secondary_detail.txt
echo "bench ($(wc -c < $OUT/bench_default.json) chars): $(cut -c1-1990 $OUT/bench_default.json)"
