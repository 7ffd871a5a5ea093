#!/usr/bin/env python3
"""The reference's performance sweeps (tests/test_fp8_fp4.py:57-68,120-125,176-189 print TFLOPS / GB/s for each) with the
automatic kernel selection: dense forward / dgrad / wgrad shapes, M-grouped contiguous, M-grouped masked.  One JSON line per
case: kernel picked, microseconds (median of 5 warm bursts), TFLOPS, GB/s, and the larger of the two roofline fractions
(5 PF dense FP8 / 8 TB/s).   python tools/survey.py [dense] [contiguous] [masked]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
from deepgemm_amd.testing import generators as gen                      # noqa: E402

which = set(sys.argv[1:]) or {'dense', 'contiguous', 'masked'}


def timed(fn, iters=10):
    t_end = time.time() + 0.25
    while time.time() < t_end:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
    bursts = []
    for _ in range(5):
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(iters):
            fn()
        end.record()
        torch.cuda.synchronize()
        bursts.append(start.elapsed_time(end) / iters * 1e3)
    return sorted(bursts)[2]


def emit(kind, label, us, flops, nbytes):
    tf, gbs = flops / us / 1e6, nbytes / us / 1e3
    print(json.dumps({'kind': kind, 'case': label, 'kernel': dg.last_config(), 'us': round(us, 1), 'tflops': round(tf, 1),
                      'gbs': round(gbs, 1), 'frac_mfma': round(tf / 5000, 3), 'frac_hbm': round(gbs / 8000, 3)}), flush=True)


if 'dense' in which:
    for m, n, k, a_k, b_k, acc, out_dtype, per_token_b in gen.enumerate_normal():
        if acc and not per_token_b:
            continue                        # same kernels as the plain forward shape
        gen.reset_seed(0)
        case = gen.generate_normal(m, n, k, a_k, b_k, acc, out_dtype, per_token_b)
        a = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1]))
        b = (case.b[0], dg.get_mn_major_tma_aligned_tensor(case.b[1])) if per_token_b else case.b
        recipe = (1, 1, 128) if per_token_b else None
        fn = lambda: dg.fp8_gemm_nt(a, b, case.d, c=case.c, recipe=recipe)      # noqa: E731
        us = timed(fn)
        elem = 4 if out_dtype == torch.float else 2
        nbytes = m * k + n * k + m * n * elem * (2 if acc else 1)
        form = ('nt' if a_k and b_k else ('nn' if a_k else ('tt' if b_k else 'tn'))) + (' wgrad fp32 acc' if per_token_b and acc else (' wgrad' if per_token_b else ''))
        emit('dense', f'{form} m={m} n={n} k={k}', us, 2.0 * m * n * k, nbytes)
        del case, a, b

if 'contiguous' in which:
    for groups, expected, n, k, b_k, psum in gen.enumerate_m_grouped_contiguous():
        if psum:
            continue
        gen.reset_seed(0)
        case = gen.generate_m_grouped_contiguous(groups, expected, n, k, b_k_major=b_k)
        a = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1]))
        fn = lambda: dg.m_grouped_fp8_gemm_nt_contiguous(a, case.b, case.d, case.grouped_layout)   # noqa: E731
        us = timed(fn, iters=5)
        valid = sum(case.actual_ms)
        emit('contiguous', f'g={groups} m~{expected} (M={case.m}) n={n} k={k} b_k_major={b_k}', us, 2.0 * valid * n * k,
             case.m * k + groups * n * k + case.m * n * 2)
        del case, a

if 'masked' in which:
    for groups, max_m, expected, n, k in gen.enumerate_m_grouped_masked():
        gen.reset_seed(0)
        case = gen.generate_m_grouped_masked(groups, max_m, expected, n, k)
        a = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1]))
        fn = lambda: dg.m_grouped_fp8_gemm_nt_masked(a, case.b, case.d, case.masked_m, expected)   # noqa: E731
        us = timed(fn)
        valid = int(case.masked_m.sum())
        emit('masked', f'g={groups} m~{expected} n={n} k={k}', us, 2.0 * valid * n * k, valid * k + groups * n * k + valid * n * 2)
        del case, a
dg.set_forced_config('auto')
