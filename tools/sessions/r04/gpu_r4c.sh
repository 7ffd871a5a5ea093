#!/bin/bash
# Round 4, GPU session C: parity of the tree (duo_pc_128x256, mega MoE entry, bounded spin) + wgrad A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r4c; mkdir -p $OUT
( timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -60 ) > $OUT/pytest.log 2>&1
echo "pytest: $(tail -1 $OUT/pytest.log)"; grep -E "^FAILED|^ERROR|Error|assert " $OUT/pytest.log | head -30
for r in 1 2; do
  for cfg in auto pipe_pc_256x256; do
    line=$(timeout 200 python bench.py --workload wgrad --config $cfg --no-cpu-baseline --no-secondary --steps 100 --clock-warmup-s 0.5 2>/dev/null | tail -1)
    echo "$r wgrad $cfg $(echo "$line" | python -c "import json,sys; p=json.loads(sys.stdin.read()); print(round(p['roofline']['kernel_us'],2), round(p['ms_per_step']*1e3,2), p['roofline']['kernel'], p['roofline']['frac'])")"
  done
done 2>&1 | tee $OUT/wgrad_ab.log
