#!/bin/bash
# round 5, session 25: the SwiGLU epilogue's arithmetic on all four waves -- parity of the fused expert MLP and fp8_mega_moe, then the expert-MLP
# bench line new / base alternating
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s25
mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_mega_gpu.py tests/test_quant_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -6 > $OUT/pytest_mega.log; tail -3 $OUT/pytest_mega.log
timeout 400 python tools/fuzz_round3.py 8700 16 swiglu 2>&1 | tail -2
export AB_ROUNDS=3
export AB_CMD='timeout 200 python bench.py --workload expert_mlp --steps 80 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(\"expert_mlp\", round(r[\"roofline\"][\"kernel_us\"],2))"'
bash tools/gpu_ab.sh
cp gpurun_out/ab/log.txt $OUT/swiglu_four_wave_epilogue_ab.log
