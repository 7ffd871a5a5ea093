#!/usr/bin/env python3
"""Times `fp8_gemm_nt` with the automatic tile selection over the reference's dense sweep (tests/generators.py:119-121:
DeepSeek-V3 (n, k) pairs x m in {1, 128, 4096}) and checks the reference gate.  One JSON line per shape."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
from deepgemm_amd.testing import calc_diff, count_bytes, generators as gen   # noqa: E402

configs = sys.argv[1].split(',') if len(sys.argv) > 1 else ['auto']
for n, k in gen.DENSE_NK:
    for m in gen.DENSE_M_FWD:
        gen.reset_seed(0)
        case = gen.generate_normal(m, n, k)
        a = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1]))
        for cfg in configs:
            dg.set_forced_config(cfg)
            for _ in range(3):
                dg.fp8_gemm_nt(a, case.b, case.d)
            torch.cuda.synchronize()
            diff = calc_diff(case.d, case.ref_d)
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            iters = 20
            start.record()
            for _ in range(iters):
                dg.fp8_gemm_nt(a, case.b, case.d)
            end.record()
            torch.cuda.synchronize()
            us = start.elapsed_time(end) / iters * 1e3
            print(json.dumps({'m': m, 'n': n, 'k': k, 'kernel': dg.last_config(), 'us': round(us, 1),
                              'tflops': round(2.0 * m * n * k / us / 1e6, 1),
                              'gbs': round(count_bytes(case.a, case.b, case.d) / us / 1e3, 1), 'ok': bool(diff < 1e-3)}), flush=True)
dg.set_forced_config('auto')
