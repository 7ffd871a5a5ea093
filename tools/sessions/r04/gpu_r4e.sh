#!/bin/bash
# Round 4, GPU session E: full parity of the tree + row-major SFA A/B + headline.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r4e; mkdir -p $OUT
( timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -60 ) > $OUT/pytest.log 2>&1
echo "pytest: $(tail -1 $OUT/pytest.log)"; grep -E "^FAILED|^ERROR|Error|assert " $OUT/pytest.log | head -30
for r in 1 2; do
  for w in dense dense_sfa_rowmajor; do
    line=$(timeout 200 python bench.py --workload $w --no-cpu-baseline --no-secondary --steps 300 --clock-warmup-s 0.5 2>/dev/null | tail -1)
    echo "$r $w $(echo "$line" | python -c "import json,sys; p=json.loads(sys.stdin.read()); print(round(p['roofline']['kernel_us'],2), round(p['ms_per_step']*1e3,2), p['roofline']['kernel'], round(p['roofline']['frac'],4))")"
  done
done 2>&1 | tee $OUT/sfa_rm.log
