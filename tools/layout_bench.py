#!/usr/bin/env python3
"""BASELINE configs[2]: fp8_gemm_{nt,nn,tn,tt} at M=2048 N=7168 K=2048 -- whole-call time (re-majoring pass included
for MN-major operands) with HIP events, parity vs the reference expression.  One JSON line per layout."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
from deepgemm_amd.testing import calc_diff, generators as gen          # noqa: E402

m, n, k = 2048, 7168, 2048
for layout in ('nt', 'nn', 'tn', 'tt'):
    gen.reset_seed(0)
    case = gen.generate_normal(m, n, k, layout[0] == 'n', layout[1] == 't')
    a = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1]))
    for _ in range(5):
        dg.fp8_gemm_nt(a, case.b, case.d)
    torch.cuda.synchronize()
    diff = calc_diff(case.d, case.ref_d)
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 30
    start.record()
    for _ in range(iters):
        dg.fp8_gemm_nt(a, case.b, case.d)
    end.record()
    torch.cuda.synchronize()
    us = start.elapsed_time(end) / iters * 1e3
    print(json.dumps({'layout': layout, 'shape': [m, n, k], 'us_per_call': round(us, 2), 'tflops': round(2.0 * m * n * k / us / 1e6, 1),
                      'kernel': dg.last_config(), 'calc_diff': diff}), flush=True)
