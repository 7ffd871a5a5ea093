#!/usr/bin/env python3
"""Sustained (back-to-back, no gaps) time per launch of kernel configurations: bursts of N launches each, round-robin over
the configurations for several rounds, after a warm-up.  The chip is power-limited on this workload: the ranking under
sustained load differs from the ranking of isolated launches or cycle counts (tools/ab.py, tools/cycles.py).
    python tools/sustained.py cfg_a,cfg_b,... [MxNxK] [launches per burst] [rounds]"""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
from deepgemm_amd.testing import generators as gen                      # noqa: E402

cfgs = sys.argv[1].split(',')
m, n, k = (int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else '4096x4096x7168').split('x'))
burst = int(sys.argv[3]) if len(sys.argv) > 3 else 300
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
cases = []
for i in range(4):
    gen.reset_seed(i)
    c = gen.generate_normal(m, n, k)
    c.a = (c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1]))
    cases.append(c)
t_end = time.time() + 2.0
while time.time() < t_end:
    for c in cases:
        dg.fp8_gemm_nt(c.a, c.b, c.d)
    torch.cuda.synchronize()
times = {cfg: [] for cfg in cfgs}
for r in range(rounds):
    for cfg in cfgs:
        dg.set_forced_config(cfg)
        for it in range(20):
            dg.fp8_gemm_nt(cases[it % 4].a, cases[it % 4].b, cases[it % 4].d)
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for it in range(burst):
            c = cases[it % 4]
            dg.fp8_gemm_nt(c.a, c.b, c.d)
        end.record()
        torch.cuda.synchronize()
        times[cfg].append(start.elapsed_time(end) / burst * 1e3)
for cfg in cfgs:
    us = statistics.median(times[cfg])
    print(json.dumps({'config': cfg, 'shape': f'{m}x{n}x{k}', 'burst': burst, 'us_per_launch': [round(t, 1) for t in times[cfg]],
                      'us_median': round(us, 1), 'tflops': round(2.0 * m * n * k / us / 1e6, 1)}), flush=True)
dg.set_forced_config('auto')
