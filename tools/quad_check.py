#!/usr/bin/env python3
"""Bit-comparison of the FP32-scale quad kernels against the 8-wave duo kernels (same promotion order => identical bits) on
dense and grouped-contiguous problems, including ragged edges.   python tools/quad_check.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
from deepgemm_amd.testing import calc_diff, generators as gen          # noqa: E402

ok = True
for m, n, k in ((4096, 4096, 7168), (2048, 7168, 2048), (300, 520, 896), (1024, 768, 128), (130, 4096, 1536)):
    gen.reset_seed(m + n)
    case = gen.generate_normal(m, n, k)
    case.a = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1]))
    outs = {}
    for cfg in ('duo_256x256', 'quad_128x256', 'quad_256x128'):
        dg.set_forced_config(cfg)
        d = torch.full((m, n), float('nan'), device='cuda', dtype=torch.bfloat16)
        dg.fp8_gemm_nt(case.a, case.b, d)
        outs[cfg] = d
    torch.cuda.synchronize()
    for cfg in ('quad_128x256', 'quad_256x128'):
        same = torch.equal(outs[cfg], outs['duo_256x256'])
        ok = ok and same
        print(json.dumps({'dense': f'{m}x{n}x{k}', 'config': cfg, 'bit_equal_to_duo': same,
                          'calc_diff_vs_ref': calc_diff(outs[cfg], case.ref_d)}), flush=True)
    # FP32 output with accumulation through the shared epilogue
    c32 = torch.randn((m, n), device='cuda', dtype=torch.float)
    outs = {}
    for cfg in ('duo_256x256', 'quad_128x256', 'quad_256x128'):
        dg.set_forced_config(cfg)
        d = c32.clone()
        dg.fp8_gemm_nt(case.a, case.b, d, c=d)
        outs[cfg] = d
    for cfg in ('quad_128x256', 'quad_256x128'):
        same = torch.equal(outs[cfg], outs['duo_256x256'])
        ok = ok and same
        print(json.dumps({'dense_fp32_acc': f'{m}x{n}x{k}', 'config': cfg, 'bit_equal_to_duo': same}), flush=True)
for g, em, n, k, psum in ((8, 512, 4096, 7168, False), (4, 200, 520, 896, False), (3, 300, 768, 512, True)):
    gen.reset_seed(g)
    case = gen.generate_m_grouped_contiguous(g, em, n, k, use_psum_layout=psum)
    case.a = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1]))
    outs = {}
    for cfg in ('duo_128x256', 'quad_128x256', 'quad_256x128'):
        dg.set_forced_config(cfg)
        d = torch.full_like(case.d, float('nan'))
        try:
            dg.m_grouped_fp8_gemm_nt_contiguous(case.a, case.b, d, case.grouped_layout, use_psum_layout=psum)
        except RuntimeError as e:
            print(json.dumps({'contiguous': f'{g}x{em}x{n}x{k}', 'psum': psum, 'config': cfg, 'error': str(e)[:100]}), flush=True)
            continue
        outs[cfg] = d
    torch.cuda.synchronize()
    for cfg in ('quad_128x256', 'quad_256x128'):
        if cfg in outs:
            a, b = torch.nan_to_num(outs[cfg].float(), nan=-7.0), torch.nan_to_num(outs['duo_128x256'].float(), nan=-7.0)
            same = torch.equal(a, b)
            ok = ok and same
            print(json.dumps({'contiguous': f'{g}x{em}x{n}x{k}', 'psum': psum, 'm': case.m, 'config': cfg, 'bit_equal_to_duo_128': same}), flush=True)
dg.set_forced_config('auto')
print('QUAD_CHECK', 'OK' if ok else 'MISMATCH')
