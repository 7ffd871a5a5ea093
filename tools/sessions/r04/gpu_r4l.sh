#!/bin/bash
# Round 4, GPU session L: the whole-stage software pipeline of the 64 x 32 stream tile (four K blocks: reads, MFMAs, promotions) -- parity + mid-M.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r4l; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_reference_sweeps_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "c1_unit or dense_nt_vs_oracle or masked or skinny or sweep or golden" 2>&1 | tail -8 ) > $OUT/pytest.log 2>&1
echo "pytest: $(tail -1 $OUT/pytest.log)"; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head
timeout 400 python tools/sweep.py --rounds 5 --iters 20 --configs auto,stream_64x32,stream_l8_64x32 \
  --shapes 128x4096x7168,64x4096x7168,256x4096x7168,128x7168x2048,128x2112x7168,128x576x7168,33x4096x7168,64x7168x16384 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print(r['shape'], r['config'], r.get('us_median'), r.get('us_min'), r.get('ok', r.get('error')))
" | tee $OUT/sweep.log
