#!/bin/bash
# Round 4, GPU session N: block 0 of a workgroup's first tile issued before the accumulators are zeroed -- parity, stamps, headline / C3 / C4.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
OUT=gpurun_out/r4n; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_full_output_parity_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "not bench_runs" 2>&1 | tail -8 ) > $OUT/pytest.log 2>&1
echo "pytest: $(tail -1 $OUT/pytest.log)"; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head
DG_VARIANT=stamp python tools/prologue_stamps.py 4096x4096x7168 2>&1 | grep -v amdgpu.ids | tee $OUT/stamps.log
for r in 1 2; do for w in dense c3_nt contiguous; do
  line=$(timeout 200 python bench.py --workload $w --no-cpu-baseline --no-secondary --steps 300 --clock-warmup-s 0.5 2>/dev/null | tail -1)
  echo "$r $w $(echo "$line" | python -c "import json,sys; p=json.loads(sys.stdin.read()); print(round(p['roofline']['kernel_us'],2), round(p['ms_per_step']*1e3,2), p['roofline']['kernel'], round(p['roofline']['frac'],4))")"
done; done 2>&1 | tee $OUT/bench.log
( cd _r3tree && for w in dense c3_nt contiguous; do
  line=$(timeout 200 python bench.py --workload $w --no-cpu-baseline --no-secondary --steps 300 --clock-warmup-s 0.5 2>/dev/null | tail -1)
  echo "r3 $w $(echo "$line" | python -c "import json,sys; p=json.loads(sys.stdin.read()); print(round(p['roofline']['kernel_us'],2), round(p['ms_per_step']*1e3,2), p['roofline']['kernel'], round(p['roofline']['frac'],4))")"
done ) 2>&1 | tee -a $OUT/bench.log
