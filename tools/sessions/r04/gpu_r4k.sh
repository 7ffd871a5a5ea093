#!/bin/bash
# Round 4, GPU session K: the round-3 tree (git archive 0ad7ac8, built in _r3tree/) against the current tree on ONE box: every bench.py workload.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r4k; mkdir -p $OUT
run() {  # $1 = tree dir, $2 = tag
  ( cd $1 && timeout 600 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | tail -1 ) > $OUT/bench_$2.json
  python - "$OUT/bench_$2.json" "$2" <<'PY'
import json, sys
p = json.loads(open(sys.argv[1]).read())
print(sys.argv[2], 'HEADLINE', round(p['roofline']['kernel_us'], 2), round(p['ms_per_step'] * 1e3, 2), round(p['roofline']['frac'], 4))
for s in p.get('secondary', []):
    if 'error' in s: print(sys.argv[2], 'ERR', s); continue
    r = s['roofline']
    print(sys.argv[2], s['workload'][:58].ljust(58), r['kernel'].ljust(22), round(r['kernel_us'], 2), round(r['frac'], 4), s.get('eager_call_us') and round(s['eager_call_us'], 1))
PY
}
run . cur1
run _r3tree r3
run . cur2
for r in 1 2; do for v in base nom0; do
  if [ "$v" = base ]; then unset DG_VARIANT; else export DG_VARIANT=$v; fi
  line=$(timeout 200 python bench.py --workload expert_mlp --no-cpu-baseline --no-secondary --steps 100 --clock-warmup-s 0.5 2>/dev/null | tail -1)
  echo "$r expert_mlp $v $(echo "$line" | python -c "import json,sys; p=json.loads(sys.stdin.read()); print(round(p['roofline']['kernel_us'],2), round(p['ms_per_step']*1e3,2), p['roofline']['kernel'])")"
done; done
unset DG_VARIANT
