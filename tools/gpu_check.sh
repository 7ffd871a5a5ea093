#!/bin/bash
# One GPU-box session: parity tests, smoke, config sweep, headline bench.  Outputs land in gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
echo "== device" > gpurun_out/session.log
python -c "import torch; print(torch.cuda.get_device_name(0), torch.cuda.device_count(), torch.cuda.get_device_capability())" >> gpurun_out/session.log 2>&1
echo "== smoke" >> gpurun_out/session.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/session.log 2>&1
echo "smoke exit $?" >> gpurun_out/session.log
timeout ${PYTEST_TIMEOUT:-1500} python -m pytest tests -m gpu -q --maxfail=12 --tb=short -p no:cacheprovider > gpurun_out/pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/session.log
tail -5 gpurun_out/pytest.log >> gpurun_out/session.log
timeout 600 python tools/sweep.py --out gpurun_out/sweep_c2.jsonl ${SWEEP_ARGS} > gpurun_out/sweep.log 2>&1
echo "sweep exit $?" >> gpurun_out/session.log
timeout 600 python bench.py --steps 100 --warmup 20 > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench exit $?" >> gpurun_out/session.log
cat gpurun_out/session.log
cat gpurun_out/sweep.log
cat gpurun_out/bench.json
