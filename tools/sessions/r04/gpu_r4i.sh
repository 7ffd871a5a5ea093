#!/bin/bash
# Round 4, GPU session I: waterfall loops out of the stream / pipe kernels -- full parity + the shapes those kernels serve.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r4i; mkdir -p $OUT
( timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -40 ) > $OUT/pytest.log 2>&1
echo "pytest: $(tail -1 $OUT/pytest.log)"; grep -E "^FAILED|^ERROR|Error|assert " $OUT/pytest.log | head -30
timeout 400 python tools/sweep.py --rounds 5 --iters 20 --configs auto,stream_64x32,stream_64x128,pipe_128x128 \
  --shapes 128x4096x7168,64x4096x7168,256x4096x7168,128x7168x2048,128x24576x1536,128x32768x512,512x4096x7168,1x24576x1536 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print(r['shape'], r['config'], r.get('us_median'), r.get('us_min'), r.get('ok', r.get('error')))
" | tee $OUT/sweep.log
for w in masked masked_ue8m0 expert_mlp dense_m128; do
  line=$(timeout 200 python bench.py --workload $w --no-cpu-baseline --no-secondary --steps 100 --clock-warmup-s 0.5 2>/dev/null | tail -1)
  echo "$w $(echo "$line" | python -c "import json,sys; p=json.loads(sys.stdin.read()); print(round(p['roofline']['kernel_us'],2), round(p['ms_per_step']*1e3,2), p['roofline']['kernel'], round(p['roofline']['frac'],4), p.get('fused_us'), p.get('unfused_us'))")"
done 2>&1 | tee $OUT/bench.log
timeout 300 python tools/masked_bench.py auto 2>&1 | grep -v amdgpu.ids | tee $OUT/masked_sweep.jsonl | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print(r.get('groups'), r.get('expected_m'), r.get('n'), r.get('k'), r.get('kernel'), r.get('us'), r.get('gbs'), r.get('tflops'))
"
