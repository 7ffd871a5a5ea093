#!/bin/bash
# A/B of library builds (deepgemm_amd/build.py DG_VARIANT) in one GPU session, alternating: VARIANTS="base nt" WORKLOADS="dense dense_ue8m0" ROUNDS=3
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
for r in $(seq 1 ${ROUNDS:-3}); do
  for w in ${WORKLOADS:-dense dense_ue8m0}; do
    for v in ${VARIANTS:-base nt}; do
      if [ "$v" = base ]; then unset DG_VARIANT; else export DG_VARIANT=$v; fi
      line=$(timeout 200 python bench.py --workload $w --no-cpu-baseline --no-secondary --steps ${STEPS:-300} --clock-warmup-s 0.5 2>/dev/null | tail -1)
      echo "$r $w $v $(echo "$line" | python -c "import json,sys; p=json.loads(sys.stdin.read()); print(round(p['roofline']['kernel_us'],2), round(p['ms_per_step']*1e3,2), p['roofline']['kernel'], p['calc_diff_vs_reference_expr'])")"
    done
  done
done
