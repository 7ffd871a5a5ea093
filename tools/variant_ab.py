#!/usr/bin/env python3
"""Bit-compare and time kernel configurations against a reference configuration on one dense shape (FP32 scales):
    python tools/variant_ab.py ref_cfg cfg_a[,cfg_b...] [MxNxK] [burst] [rounds]"""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
from deepgemm_amd.testing import calc_diff, generators as gen           # noqa: E402

ref_cfg, cfgs = sys.argv[1], sys.argv[2].split(',')
m, n, k = (int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else '4096x4096x7168').split('x'))
burst = int(sys.argv[4]) if len(sys.argv) > 4 else 200
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 3
cases = []
for i in range(4):
    gen.reset_seed(i)
    c = gen.generate_normal(m, n, k)
    c.a = (c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1]))
    cases.append(c)
dg.set_forced_config(ref_cfg)
refs = []
for c in cases:
    d = torch.empty_like(c.d)
    dg.fp8_gemm_nt(c.a, c.b, d)
    refs.append(d)
for cfg in cfgs:
    dg.set_forced_config(cfg)
    ok, worst = True, 0.0
    for rep in range(3):
        for c, r in zip(cases, refs):
            c.d.fill_(float('nan'))
            dg.fp8_gemm_nt(c.a, c.b, c.d)
            same = torch.equal(c.d, r)
            ok = ok and same
            if not same:
                worst = max(worst, calc_diff(c.d, r))
    print(json.dumps({'config': cfg, 'bit_equal_to': ref_cfg, 'equal': ok, 'worst_calc_diff': worst, 'vs_reference_expr': calc_diff(cases[0].d, cases[0].ref_d)}), flush=True)
t_end = time.time() + 1.5
while time.time() < t_end:
    for c in cases:
        dg.fp8_gemm_nt(c.a, c.b, c.d)
    torch.cuda.synchronize()
allc = [ref_cfg] + cfgs
times = {cfg: [] for cfg in allc}
for r in range(rounds):
    for cfg in allc:
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
for cfg in allc:
    us = statistics.median(times[cfg])
    print(json.dumps({'config': cfg, 'shape': f'{m}x{n}x{k}', 'us_per_launch': [round(t, 1) for t in times[cfg]], 'us_median': round(us, 1),
                      'tflops': round(2.0 * m * n * k / us / 1e6, 1), 'frac': round(2.0 * m * n * k / us / 1e6 / 5000, 4)}), flush=True)
dg.set_forced_config('auto')
