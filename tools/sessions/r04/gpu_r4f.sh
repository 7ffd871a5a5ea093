#!/bin/bash
# Round 4, GPU session F: split-ring stream kernel (stream2): parity on every dense shape + mid-M / decode-M sweeps + C5.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r4f; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gemm_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "c1_unit or dense_nt_vs_oracle or masked" 2>&1 | tail -40 ) > $OUT/pytest.log 2>&1
echo "pytest: $(tail -1 $OUT/pytest.log)"; grep -E "^FAILED|^ERROR|Error|assert " $OUT/pytest.log | head -30
timeout 600 python tools/sweep.py --out $OUT/sweep_stream2.jsonl --rounds 5 --iters 20 \
  --configs stream_l8_64x32,stream2_64x32,stream2b_64x32,stream_64x128,stream2_64x128,auto \
  --shapes 128x4096x7168,128x2112x7168,128x576x7168,128x7168x2048,64x4096x7168,256x4096x7168,33x4096x7168,64x7168x16384,128x24576x1536 2>&1 | grep -v amdgpu.ids > $OUT/sweep_stream2.log
cat $OUT/sweep_stream2.log | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print(r['shape'], r['config'], r.get('us_median'), r.get('us_min'), r.get('ok', r.get('error')))
"
for cfg in stream_nt_64x128 stream2_nt_64x128 stream2_64x128; do
  timeout 200 python bench.py --workload masked --config $cfg --no-secondary --no-cpu-baseline --steps 100 --warmup 20 2>&1 | grep -v amdgpu.ids | tail -1 > $OUT/c5_$cfg.json
  python -c "
import json,sys
r=json.loads(open('$OUT/c5_$cfg.json').read()); print('C5', '$cfg', r['ms_per_step']*1e3, r['roofline'].get('kernel_us'), r['roofline'].get('frac'))
"
done
