#!/usr/bin/env python3
"""Fused per-token quantiser: time and HBM rate (2 B read + 1 B written per element + scales) next to the torch expression."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
from deepgemm_amd.utils import per_block_cast_to_fp8, per_channel_cast_to_fp8, per_token_cast_to_fp8   # noqa: E402

for shape in (sys.argv[1] if len(sys.argv) > 1 else '4096x7168,16384x7168,128x7168').split(','):
    m, n = (int(v) for v in shape.split('x'))
    xs = [torch.randn((m, n), device='cuda', dtype=torch.bfloat16) for _ in range(4)]
    for name, fn in (('fused_hip', lambda x: dg.fused_per_token_cast_to_fp8(x, sf_mn_major=True)),
                     ('torch_expr', lambda x: per_token_cast_to_fp8(x, use_ue8m0=False)),
                     ('fused_hip_per_block', lambda x: dg.fused_per_block_cast_to_fp8(x)),
                     ('torch_expr_per_block', lambda x: per_block_cast_to_fp8(x, use_ue8m0=False)),
                     ('fused_hip_per_channel', lambda x: dg.fused_per_channel_cast_to_fp8(x)),
                     ('torch_expr_per_channel', lambda x: per_channel_cast_to_fp8(x, use_ue8m0=False))):
        for x in xs:
            fn(x)
        torch.cuda.synchronize()
        times = []
        for _ in range(5):
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
            for it in range(20):
                fn(xs[it % 4])
            end.record()
            torch.cuda.synchronize()
            times.append(start.elapsed_time(end) / 20 * 1e3)
        times.sort()
        nbytes = m * n * 3 + m * ((n + 127) // 128) * 4
        print(json.dumps({'shape': shape, 'impl': name, 'us_median': round(times[2], 2),
                          'GBps': round(nbytes / times[2] / 1e3, 1)}), flush=True)
