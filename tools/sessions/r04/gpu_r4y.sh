#!/bin/bash
# Round 4, GPU session Y: two cheap knobs on one box -- M tiles per L2 group of the tile walk (DG_GROUP_M) and the non-temporal output policy
# (nont variant) -- on the headline and C3.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r4y; mkdir -p $OUT
for r in 1 2; do for w in dense c3_nt; do for gm in default 2 8 16; do
  if [ $gm = default ]; then unset DG_GROUP_M; else export DG_GROUP_M=$gm; fi
  line=$(timeout 200 python bench.py --workload $w --no-cpu-baseline --no-secondary --steps 300 --clock-warmup-s 0.5 2>/dev/null | tail -1)
  echo "$r $w group_m=$gm $(echo "$line" | python -c "import json,sys; p=json.loads(sys.stdin.read()); print(round(p['roofline']['kernel_us'],2), round(p['ms_per_step']*1e3,2), p['roofline']['kernel'], round(p['roofline']['frac'],4))")"
done; done; done 2>&1 | tee $OUT/group_m.log
unset DG_GROUP_M
VARIANTS="base nont" WORKLOADS="c3_nt dense" ROUNDS=2 bash tools/gpu_ab_variants.sh 2>&1 | tee $OUT/ab_nt.log
