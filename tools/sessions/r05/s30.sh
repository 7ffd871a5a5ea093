#!/bin/bash
# round 5, session 30: K pieces of the narrow-layer wgrad on the 192-row tiles (DG_KS_PIECES), two rounds
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s30
mkdir -p $OUT
export PYTHONUNBUFFERED=1
for r in 1 2; do
  for P in 0 3 4 5; do
    DG_KS_PIECES=$P timeout 120 python bench.py --workload wgrad_ksplit --steps 80 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$r pieces=$P', round(r['roofline']['kernel_us'],2), r['roofline']['kernel'])"
  done
done 2>&1 | tee $OUT/wgrad_ksplit_pieces_192.log
