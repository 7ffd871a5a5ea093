#!/usr/bin/env python3
"""m_grouped_fp8_gemm_nt_masked over the reference's masked sweep (tests/generators.py:177-187: (groups, expected M) x
DeepSeek-V3 expert shapes, max M 4096): one JSON line per case with the kernel picked, TFLOPS over the valid rows and
GB/s over the bytes the reference counts (tests/test_fp8_fp4.py:182-183)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
from deepgemm_amd.testing import generators as gen                      # noqa: E402

configs = sys.argv[1].split(',') if len(sys.argv) > 1 else ['auto']
only = sys.argv[2] if len(sys.argv) > 2 else None
for groups, max_m, expected, n, k in gen.enumerate_m_grouped_masked():
    if only and only != f'{groups}x{expected}':
        continue
    gen.reset_seed(0)
    case = gen.generate_m_grouped_masked(groups, max_m, expected, n, k)
    a = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1]))
    valid = int(case.masked_m.sum())
    for cfg in configs:
        dg.set_forced_config(cfg)
        try:
            t_end = time.time() + 0.2
            while time.time() < t_end:
                for _ in range(4):
                    dg.m_grouped_fp8_gemm_nt_masked(a, case.b, case.d, case.masked_m, expected)
                torch.cuda.synchronize()
        except RuntimeError as e:
            print(json.dumps({'groups': groups, 'expected_m': expected, 'n': n, 'k': k, 'config': cfg, 'error': str(e)[:100]}))
            continue
        bursts = []
        for _ in range(5):
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
            for _ in range(10):
                dg.m_grouped_fp8_gemm_nt_masked(a, case.b, case.d, case.masked_m, expected)
            end.record()
            torch.cuda.synchronize()
            bursts.append(start.elapsed_time(end) / 10 * 1e3)
        us = sorted(bursts)[2]
        nbytes = valid * k + groups * n * k + valid * n * 2
        print(json.dumps({'groups': groups, 'expected_m': expected, 'n': n, 'k': k, 'valid_rows': valid, 'kernel': dg.last_config(),
                          'us': round(us, 1), 'tflops': round(2.0 * valid * n * k / us / 1e6, 1),
                          'gbs': round(nbytes / us / 1e3, 1)}), flush=True)
dg.set_forced_config('auto')
