#!/usr/bin/env python3
"""Recipe (1, 1, 128) (per-row SFA x per-row SFB, FP32 accumulate into D): the LDS-DMA kernel next to the layout-agnostic
kernel that served this recipe before.  One JSON line per shape and kernel."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
from deepgemm_amd.testing import calc_diff                             # noqa: E402
from deepgemm_amd.utils.math import per_token_cast_to_fp8             # noqa: E402

shapes = sys.argv[1] if len(sys.argv) > 1 else '4096x4096x7168,7168x4096x4096'
kernels = sys.argv[2].split(',') if len(sys.argv) > 2 else ['auto', 'generic_128x128']
for shape in shapes.split(','):
    m, n, k = (int(x) for x in shape.split('x'))
    sets = []
    for i in range(4):
        torch.manual_seed(i)
        a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16)
        b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
        a_q, sfa = per_token_cast_to_fp8(a, use_ue8m0=False)
        b_q, sfb = per_token_cast_to_fp8(b, use_ue8m0=False)
        ref = (a.float() @ b.float().t()) if i == 0 else None
        sets.append(dict(a=(a_q, dg.get_mn_major_tma_aligned_tensor(sfa)), b=(b_q, dg.get_mn_major_tma_aligned_tensor(sfb)),
                         d=torch.zeros((m, n), device='cuda', dtype=torch.float), ref=ref))
    for kernel in kernels:
        dg.set_forced_config(kernel)

        def call(s):
            dg.fp8_gemm_nt(s['a'], s['b'], s['d'], c=s['d'], recipe=(1, 1, 128))
        sets[0]['d'].zero_()
        for s in sets:
            call(s)
        torch.cuda.synchronize()
        diff = calc_diff(sets[0]['d'], sets[0]['ref'])
        times = []
        for _ in range(5):
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
            for it in range(8):
                call(sets[it % 4])
            end.record()
            torch.cuda.synchronize()
            times.append(start.elapsed_time(end) / 8 * 1e3)
        times.sort()
        print(json.dumps({'shape': shape, 'recipe': '1,1,128', 'kernel': dg.last_config(), 'us_median': round(times[2], 2),
                          'us_min': round(times[0], 2), 'tflops_median': round(2.0 * m * n * k / times[2] / 1e6, 1),
                          'calc_diff_vs_fp32_ref': diff}), flush=True)
    dg.set_forced_config('auto')
