#!/bin/bash
mkdir -p gpurun_out/r2f
python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r2f/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2f/pytest.log; tail -30 gpurun_out/r2f/pytest.log
python bench.py > gpurun_out/r2f/bench.json 2> gpurun_out/r2f/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
line=json.loads([l for l in open('gpurun_out/r2f/bench.json') if l.startswith('{')][-1])
print({k: line[k] for k in ('value','ms_per_step','pct_of_mfma_peak')}, line['roofline']['kernel'], round(line['roofline']['kernel_us'],1), round(line['roofline']['frac'],3))
for s in line.get('secondary', []):
    if 'error' in s: print('ERR', s)
    else: print(s['workload'][:70], '|', s['roofline']['kernel'], round(s['roofline']['kernel_us'],1), 'us', s['roofline']['bound'], round(s['roofline']['frac'],3), 'diff', s['calc_diff_vs_reference_expr'])
print(line.get('cpu_baseline'))
PY
