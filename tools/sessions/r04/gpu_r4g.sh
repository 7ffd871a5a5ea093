#!/bin/bash
# Round 4, GPU session G: does the row pitch (K = 7168 = 28 x 256 bytes) cost the stream tiles / the 256-row tiles L2 or HBM channel conflicts?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r4g; mkdir -p $OUT
for pads in "0 0" "256 0" "0 256" "256 256" "128 128"; do
  set -- $pads
  echo "== pad_a $1 pad_b $2"
  timeout 300 python tools/sweep.py --rounds 3 --iters 20 --pad-a $1 --pad-b $2 --configs auto \
    --shapes 128x4096x7168,64x4096x7168,256x4096x7168,4096x4096x7168,2048x7168x2048,1x4096x7168 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print(r['shape'], r['config'], r.get('us_median'), r.get('us_min'), r.get('ok', r.get('error')))
"
done 2>&1 | tee $OUT/pitch.log
