#!/usr/bin/env python3
"""The L2 GEMM of the decode-size expert MLP (masked, 8 experts x <= 64 rows, N = 7168, K = 2048: 448 tiles of 64 x 128 = 1.75 rounds) on each
stream tile: does a tile count that divides the CU count (64 x 32: 1792 tiles = 7 rounds exactly) beat the quantisation loss?"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg
from deepgemm_amd.testing import generators as gen
shapes = [(8, 64, 48, 7168, 2048), (8, 64, 48, 4096, 7168)]
for groups, max_m, expected, n, k in shapes:
    cases = []
    for i in range(3):
        gen.reset_seed(i)
        c = gen.generate_m_grouped_masked(groups, max_m, expected, n, k)
        cases.append(((c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1])), c))
    for cfg in ['auto', 'stream_64x128', 'stream_nt_64x128', 'stream_64x32', 'stream_l8_64x32']:
        dg.set_forced_config(cfg)
        try:
            for a, c in cases:
                dg.m_grouped_fp8_gemm_nt_masked(a, c.b, c.d, c.masked_m, expected)
            torch.cuda.synchronize()
        except RuntimeError as e:
            print(json.dumps({'n': n, 'k': k, 'config': cfg, 'error': str(e)[:120]})); continue
        best = []
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(60):
                a, c = cases[i % 3]
                dg.m_grouped_fp8_gemm_nt_masked(a, c.b, c.d, c.masked_m, expected)
            e.record(); torch.cuda.synchronize()
            best.append(s.elapsed_time(e) / 60 * 1e3)
        best.sort()
        print(json.dumps({'n': n, 'k': k, 'config': cfg, 'kernel': dg.last_config(), 'us_median': round(best[2], 2), 'us_min': round(best[0], 2),
                          'TBps': round(groups * n * k / best[2] / 1e6, 2)}))
dg.set_forced_config('auto')
