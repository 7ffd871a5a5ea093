#!/usr/bin/env python3
"""Masked decode entries of the reference sweep (6 experts x ~20 rows) with 50-110 MB of weights per launch: default against non-temporal weight
policy of the two-per-CU stream tile over a COLD rotation (> 320 MB of weights between two uses of a set).
python tools/probes/masked_nt_policy_cold.py"""
import sys
sys.path.insert(0, '.')
import torch, deepgemm_amd as dg
from deepgemm_amd.testing import generators as gen


def time_us(fn, n=40):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for groups, max_m, expected, n, k in ((6, 256, 20, 4096, 2048), (6, 256, 20, 4096, 4096), (6, 256, 20, 7168, 3072), (2, 256, 20, 4096, 2048)):
    sets = int(320e6 // (groups * n * k)) + 2
    cases = []
    for i in range(sets):
        gen.reset_seed(i)
        c = gen.generate_m_grouped_masked(groups, max_m, expected, n, k)
        cases.append(((c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1])), c.b, c.d, c.masked_m))
    out = []
    for rep in range(2):
        for cfg in ('auto', 'stream2_64x128', 'stream_nt2_64x128'):
            dg.set_forced_config(cfg)
            try:
                it = [0]
                def call():
                    a, b, d, mm = cases[it[0] % sets]; it[0] += 1
                    dg.m_grouped_fp8_gemm_nt_masked(a, b, d, mm, expected)
                t = time_us(call)
                out.append(f'{cfg}{"=" + dg.last_config() if cfg == "auto" else ""} {t:.1f}')
            finally:
                dg.set_forced_config('auto')
    print(f'masked g={groups} m~{expected} n={n} k={k} ({groups * n * k / 1e6:.0f} MB, {sets} sets): ' + ' | '.join(out), flush=True)
    del cases
