#!/bin/bash
# One GPU-box session: smoke, full parity suite, the three bench workloads.  Outputs land in gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
python -c "import torch; print(torch.cuda.get_device_name(0), torch.cuda.device_count())" > gpurun_out/session.log 2>&1
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/session.log 2>&1
echo "smoke exit $?" >> gpurun_out/session.log
timeout ${PYTEST_TIMEOUT:-1200} python -m pytest tests -m gpu -q --maxfail=12 --tb=short -p no:cacheprovider > gpurun_out/pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/session.log
tail -15 gpurun_out/pytest.log >> gpurun_out/session.log
for W in dense contiguous masked; do
  timeout 600 python bench.py --workload $W --steps 100 --warmup 20 $([ $W != dense ] && echo --no-cpu-baseline) > gpurun_out/bench_$W.json 2> gpurun_out/bench_$W.err
  echo "bench $W exit $?" >> gpurun_out/session.log
done
grep -v amdgpu.ids gpurun_out/session.log
cat gpurun_out/bench_dense.json gpurun_out/bench_contiguous.json gpurun_out/bench_masked.json
