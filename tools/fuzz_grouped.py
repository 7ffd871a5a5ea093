#!/usr/bin/env python3
"""Random M-grouped problems through the automatic selection, each against the oracle: contiguous (plain / psum layouts, K-major and
MN-major B, ragged and empty groups, padding rows must come out zero) and masked (rows >= masked_m must stay untouched).
    python tools/fuzz_grouped.py [first_seed] [count]"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import deepgemm_amd as dg                                              # noqa: E402
import oracle                                                          # noqa: E402
from deepgemm_amd.testing import generators as gen                      # noqa: E402
from gpu_helpers import assert_close_to_oracle, cpu_pair               # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bad = 0
for seed in range(first, first + count):
    rng = random.Random(seed)
    gen.reset_seed(seed)
    try:
        groups = rng.choice([1, 3, 4, 8])
        n, k = rng.choice([256, 520, 1024, 2048]), rng.choice([128, 384, 1024, 4096])
        ms = [rng.choice([0, 1, 77, 128, 129, 300, 512, 640]) for _ in range(groups)]
        if sum(ms) == 0:
            ms[0] = 100
        psum, b_k = rng.random() < 0.5, (rng.random() < 0.6 or n % 16 != 0)
        case = gen.generate_m_grouped_contiguous(groups, 0, n, k, b_k, psum, actual_ms=ms)
        want = torch.full(case.d.shape, float('nan'), dtype=torch.bfloat16)
        oracle.m_grouped_fp8_gemm_nt_contiguous(*cpu_pair(case.a), *cpu_pair(case.b), want, case.grouped_layout.cpu(), psum)
        case.d.fill_(float('nan'))
        if b_k:
            dg.m_grouped_fp8_gemm_nt_contiguous(case.a, case.b, case.d, case.grouped_layout, use_psum_layout=psum)
        else:
            dg.m_grouped_fp8_gemm_nn_contiguous(case.a, (case.b[0].mT, case.b[1].mT), case.d, case.grouped_layout, use_psum_layout=psum)
        label = f'contiguous seed={seed} ms={ms} n={n} k={k} psum={psum} b_k={b_k} [{dg.last_config()}]'
        start = 0
        for actual, aligned in zip(case.actual_ms, case.aligned_ms):
            if actual:
                assert_close_to_oracle(case.d[start:start + actual], want[start:start + actual], label)
            assert bool((case.d[start + actual:start + aligned] == 0).all()), label + ': padding rows must be zeros'
            start += aligned
        masked_ms = [rng.choice([0, 1, 20, 64, 100, 200, 256]) for _ in range(groups)]
        max_m = rng.choice([256, 512])
        mk = gen.generate_m_grouped_masked(groups, max_m, 0, n, k, masked_ms=masked_ms)
        mwant = torch.full(mk.d.shape, float('nan'), dtype=torch.bfloat16)
        oracle.m_grouped_fp8_gemm_nt_masked(*cpu_pair(mk.a), *cpu_pair(mk.b), mwant, mk.masked_m.cpu())
        mk.d.fill_(float('nan'))
        dg.m_grouped_fp8_gemm_nt_masked(mk.a, mk.b, mk.d, mk.masked_m, max(1, sum(masked_ms) // groups))
        label = f'masked seed={seed} masked_ms={masked_ms} max_m={max_m} n={n} k={k} [{dg.last_config()}]'
        for g, rows in enumerate(masked_ms):
            if rows:
                assert_close_to_oracle(mk.d[g, :rows], mwant[g, :rows], label)
            assert bool(torch.isnan(mk.d[g, rows:]).all()), label + ': rows >= masked_m must not be written'
    except (AssertionError, RuntimeError) as e:
        bad += 1
        print('FAIL', str(e)[:400], flush=True)
print('done, failures:', bad)
