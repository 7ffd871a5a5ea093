#!/usr/bin/env python3
"""Cost of the reduce-add epilogue (D = C + A B^T with C = D): the reference's forward sweep runs every BF16-output shape with
accumulation too (tests/generators.py:138-140).  One JSON line per shape and output form.
    python tools/acc_bench.py [MxNxK,...]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
from deepgemm_amd.testing import calc_diff, generators as gen          # noqa: E402

shapes = sys.argv[1] if len(sys.argv) > 1 else '4096x4096x7168,4096x7168x2048,4096x2112x7168'
for shape in shapes.split(','):
    m, n, k = (int(x) for x in shape.split('x'))
    for out_dtype, acc in ((torch.bfloat16, False), (torch.bfloat16, True), (torch.float, False), (torch.float, True)):
        gen.reset_seed(0)
        case = gen.generate_normal(m, n, k, accumulate=acc, out_dtype=out_dtype)
        a = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1]))
        dg.fp8_gemm_nt(a, case.b, case.d, c=case.c)
        torch.cuda.synchronize()
        diff = calc_diff(case.d, case.ref_d)
        t_end = time.time() + 0.25
        while time.time() < t_end:
            for _ in range(4):
                dg.fp8_gemm_nt(a, case.b, case.d, c=case.c)
            torch.cuda.synchronize()
        bursts = []
        for _ in range(5):
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
            for _ in range(10):
                dg.fp8_gemm_nt(a, case.b, case.d, c=case.c)
            end.record()
            torch.cuda.synchronize()
            bursts.append(start.elapsed_time(end) / 10 * 1e3)
        print(json.dumps({'shape': shape, 'out': str(out_dtype).split('.')[-1], 'accumulate': acc, 'kernel': dg.last_config(),
                          'us': round(sorted(bursts)[2], 1), 'calc_diff_first_call': float(diff)}), flush=True)
