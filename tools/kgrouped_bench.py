#!/usr/bin/env python3
"""k_grouped_fp8_gemm_{nt,tn}_contiguous on the reference's sweep sizes (tests/generators.py:190-208: (groups, m, n,
expected k per group)): one JSON line per case and operand form."""
import json
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
from deepgemm_amd.testing import calc_diff, generators as gen          # noqa: E402

cases = sys.argv[1] if len(sys.argv) > 1 else '4x4096x7168x8192,8x4096x7168x4096,16x7168x2048x2048'
for case_s in cases.split(','):
    g, m, n, ek = (int(x) for x in case_s.split('x'))
    random.seed(0)
    ks = [max(128, int(ek * random.uniform(0.7, 1.3)) // 128 * 128) for _ in range(g)]
    for k_major in (True, False):
        gen.reset_seed(0)
        case = gen.generate_k_grouped_contiguous(g, m, n, ks, k_major)
        fn = dg.k_grouped_fp8_gemm_nt_contiguous if k_major else dg.k_grouped_fp8_gemm_tn_contiguous
        d = case.c.clone()
        fn(case.a, case.b, d, ks, case.grouped_layout, c=case.c)
        torch.cuda.synchronize()
        diff = calc_diff(d, case.ref_d)
        t_end = time.time() + 0.3
        while time.time() < t_end:
            fn(case.a, case.b, d, ks, case.grouped_layout, c=d)
            torch.cuda.synchronize()
        bursts = []
        for _ in range(5):
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
            for _ in range(3):
                fn(case.a, case.b, d, ks, case.grouped_layout, c=d)
            end.record()
            torch.cuda.synchronize()
            bursts.append(start.elapsed_time(end) / 3 * 1e3)
        us = sorted(bursts)[2]
        print(json.dumps({'case': case_s, 'form': 'nt (K-major blocks)' if k_major else 'tn (MN-major)', 'sum_k': sum(ks),
                          'kernel': dg.last_config(), 'us': round(us, 1),
                          'tflops': round(2.0 * m * n * sum(ks) / us / 1e6, 1), 'calc_diff': diff}), flush=True)
        del case, d
        torch.cuda.empty_cache()
