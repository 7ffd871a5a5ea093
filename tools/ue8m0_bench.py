#!/usr/bin/env python3
"""fp8_gemm_nt with power-of-two scales in the packed UE8M0 format (hardware-scaled MFMA path) next to the same problem
with the same scales as FP32 tensors (promotion path).  One JSON line per shape and path."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
from deepgemm_amd.testing import calc_diff                             # noqa: E402
from deepgemm_amd.utils.math import pack_ue8m0_to_int, per_block_cast_to_fp8, per_token_cast_to_fp8   # noqa: E402

shapes = sys.argv[1] if len(sys.argv) > 1 else '4096x4096x7168'
for shape in shapes.split(','):
    m, n, k = (int(x) for x in shape.split('x'))
    sets = []
    for i in range(4):
        torch.manual_seed(i)
        a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16)
        b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
        a_q, sfa = per_token_cast_to_fp8(a, use_ue8m0=True)
        b_q, sfb = per_block_cast_to_fp8(b, use_ue8m0=True)
        sfb_rows = sfb.repeat_interleave(128, dim=0)[:n].contiguous()
        ref = (a.float() @ b.float().t()).to(torch.bfloat16) if i == 0 else None
        packed_a = dg.get_mn_major_tma_aligned_tensor(pack_ue8m0_to_int(sfa).view(torch.float)).view(torch.int)
        packed_b = dg.get_mn_major_tma_aligned_tensor(pack_ue8m0_to_int(sfb_rows).view(torch.float)).view(torch.int)
        sets.append(dict(a=a_q, pa=packed_a, b=b_q, pb=packed_b, fa=dg.get_mn_major_tma_aligned_tensor(sfa), fb=sfb,
                         d=torch.empty((m, n), device='cuda', dtype=torch.bfloat16), ref=ref))
    for path in ('packed_ue8m0', 'fp32_scales'):
        def call(s):
            if path == 'packed_ue8m0':
                dg.fp8_gemm_nt((s['a'], s['pa']), (s['b'], s['pb']), s['d'])
            else:
                dg.fp8_gemm_nt((s['a'], s['fa']), (s['b'], s['fb']), s['d'])
        for s in sets:
            call(s)
        torch.cuda.synchronize()
        diff = calc_diff(sets[0]['d'], sets[0]['ref'])
        best = 1e30
        times = []
        for _ in range(5):
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
            for it in range(20):
                call(sets[it % 4])
            end.record()
            torch.cuda.synchronize()
            times.append(start.elapsed_time(end) / 20 * 1e3)
        times.sort()
        print(json.dumps({'shape': shape, 'path': path, 'kernel': dg.last_config(), 'us_median': round(times[2], 2),
                          'us_min': round(times[0], 2), 'tflops_median': round(2.0 * m * n * k / times[2] / 1e6, 1),
                          'tflops_best': round(2.0 * m * n * k / times[0] / 1e6, 1), 'calc_diff_vs_bf16_ref': diff}), flush=True)
