#!/usr/bin/env python3
"""256 x 224 tiles against 256 x 256 tiles, same process, alternating, rotating input sets (round 6, VERDICT item 1: C3).
    python tools/n224_ab.py [MxNxK ...]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
from deepgemm_amd.testing import generators as gen                      # noqa: E402

shapes = [tuple(int(x) for x in s.split('x')) for s in sys.argv[1:]] or [(2048, 7168, 2048), (4096, 7168, 2048), (2048, 7168, 7168), (4096, 7168, 7168)]
for m, n, k in shapes:
    cases = []
    for i in range(6):
        gen.reset_seed(i)
        c = gen.generate_normal(m, n, k)
        cases.append(((c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1])), c.b, c.d))
        c.a_bf16 = c.b_bf16 = None
    out = {}
    ref = None
    for rnd in range(3):
        for cfg in ('duo_p_256x256', 'duo_p_256x224'):
            dg.set_forced_config(cfg)
            for it in range(60):
                a, b, d = cases[it % 6]
                dg.fp8_gemm_nt(a, b, d)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for it in range(200):
                a, b, d = cases[it % 6]
                dg.fp8_gemm_nt(a, b, d)
            e.record()
            torch.cuda.synchronize()
            out.setdefault(cfg, []).append(round(s.elapsed_time(e) / 200 * 1e3, 2))
            a, b, d = cases[0]
            dg.fp8_gemm_nt(a, b, d)
            if ref is None:
                ref = d.clone()
            else:
                assert torch.equal(d.view(torch.int16), ref.view(torch.int16)), 'the two tilings must agree bit for bit'
    dg.set_forced_config('auto')
    print(json.dumps({'shape': f'{m}x{n}x{k}', 'us_per_call': out, 'tflops_224': round(2.0 * m * n * k / min(out['duo_p_256x224']) / 1e6, 1),
                      'tflops_256': round(2.0 * m * n * k / min(out['duo_p_256x256']) / 1e6, 1)}), flush=True)
