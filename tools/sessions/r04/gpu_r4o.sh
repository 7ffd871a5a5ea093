#!/bin/bash
# Round 4, GPU session O: store-tail probe (cache policies / per-CU vs chip limit), float-reciprocal tile mapping: parity + bench,
# headline sweep over the tile shapes, in-kernel cycle stamps.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r4o; mkdir -p $OUT
timeout 120 tools/ubench/store_rate 2>&1 | tee $OUT/store_rate.log
( timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_full_output_parity_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "not bench_runs" 2>&1 | tail -8 ) > $OUT/pytest.log 2>&1
echo "pytest: $(tail -1 $OUT/pytest.log)"; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head
timeout 200 python tools/cycles.py --configs duo_p_256x256,duo_256x256,duo_128x256 --shape 4096x4096x7168 2>&1 | grep -v amdgpu.ids | tee $OUT/cycles.log
timeout 200 python tools/cycles.py --configs duo_p_256x256 --shape 2048x7168x2048 2>&1 | grep -v amdgpu.ids | tee -a $OUT/cycles.log
timeout 300 python tools/sweep.py --shapes 4096x4096x7168,2048x7168x2048 --configs duo_p_256x256,duo_256x256,duo_128x256,duo_prio_128x256,duo_pprio_256x256 --rounds 5 --iters 20 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep.log
for r in 1 2; do for w in dense c3_nt contiguous; do
  line=$(timeout 200 python bench.py --workload $w --no-cpu-baseline --no-secondary --steps 300 --clock-warmup-s 0.5 2>/dev/null | tail -1)
  echo "$r $w $(echo "$line" | python -c "import json,sys; p=json.loads(sys.stdin.read()); print(round(p['roofline']['kernel_us'],2), round(p['ms_per_step']*1e3,2), p['roofline']['kernel'], round(p['roofline']['frac'],4))")"
done; done 2>&1 | tee $OUT/bench.log
( cd _r3tree && for w in dense c3_nt contiguous; do
  line=$(timeout 200 python bench.py --workload $w --no-cpu-baseline --no-secondary --steps 300 --clock-warmup-s 0.5 2>/dev/null | tail -1)
  echo "r3 $w $(echo "$line" | python -c "import json,sys; p=json.loads(sys.stdin.read()); print(round(p['roofline']['kernel_us'],2), round(p['ms_per_step']*1e3,2), p['roofline']['kernel'], round(p['roofline']['frac'],4))")"
done ) 2>&1 | tee -a $OUT/bench.log
