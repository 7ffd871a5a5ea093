#!/usr/bin/env python3
"""A/B timing of kernel configurations under identical thermal conditions: the configurations alternate launch by launch
(every launch timed with its own pair of HIP events), after a warm-up that brings the chip to its sustained clock.
The chip is power-limited on this workload, so cycle counts and back-to-back blocks of one configuration both mislead.
    python tools/ab.py cfg_a,cfg_b[,cfg_c] [MxNxK] [launches per config]"""
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
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 300
cases = []
for i in range(4):
    gen.reset_seed(i)
    c = gen.generate_normal(m, n, k)
    c.a = (c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1]))
    cases.append(c)
t_end = time.time() + 1.0
while time.time() < t_end:                      # warm-up: sustained clock
    for cfg in cfgs:
        dg.set_forced_config(cfg)
        for c in cases:
            dg.fp8_gemm_nt(c.a, c.b, c.d)
    torch.cuda.synchronize()
events = {cfg: [] for cfg in cfgs}
for it in range(launches):
    for j, cfg in enumerate(cfgs):
        dg.set_forced_config(cfg)
        c = cases[(it + j) % len(cases)]
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        dg.fp8_gemm_nt(c.a, c.b, c.d)
        end.record()
        events[cfg].append((start, end))
torch.cuda.synchronize()
for cfg in cfgs:
    us = sorted(s.elapsed_time(e) * 1e3 for s, e in events[cfg])
    print(json.dumps({'config': cfg, 'shape': f'{m}x{n}x{k}', 'launches': len(us), 'us_median': round(statistics.median(us), 2),
                      'us_mean': round(statistics.fmean(us), 2), 'us_p10': round(us[len(us) // 10], 2),
                      'us_p90': round(us[len(us) * 9 // 10], 2),
                      'tflops_median': round(2.0 * m * n * k / statistics.median(us) / 1e6, 1)}), flush=True)
dg.set_forced_config('auto')
