#!/usr/bin/env python3
"""Randomised parity runs for the paths of the second half of round 5, every result against the oracle's arithmetic on the device (FP64 block
products: tests/test_full_output_parity_gpu.py) or against the form it must equal bit for bit:
  pc192       recipe (1, 1, 128), FP32 accumulate: random M (192- and 256-row tiles, K split or not) against the device oracle and the layout-agnostic kernel
  skinnyc     dense M <= 32: the coalesced-load skinny forms against the register-direct ones (same bits) and the oracle
  packedtab   contiguous layout with packed UE8M0 scales, K >= 4096: the group-relative tiling against 128-row tiles on the fixed grid (same bits)
  groupednn   m_grouped_fp8_gemm_nn_contiguous with packed scales: MN-major weights in place (forced) against the K-major call (same bits)
python tools/fuzz_round5b.py [first_seed] [count] [which,...]"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import deepgemm_amd as dg                                               # noqa: E402
from deepgemm_amd.testing import generators as gen                      # noqa: E402
from gpu_helpers import assert_close_fp32, assert_close_to_oracle       # noqa: E402
import oracle                                                            # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 8
which = set((sys.argv[3] if len(sys.argv) > 3 else 'pc192,skinnyc,packedtab,groupednn').split(','))


def pc192(seed):
    rng = random.Random(seed)
    m = rng.choice([65, 100, 190, 192, 193, 300, 384, 385, 576, 577, 640, 1000, 1152, 2112])
    n = rng.choice([256, 520, 1024, 2304, 4096])
    k = 128 * rng.choice([2, 7, 16, 24, 32, 56])
    gen.reset_seed(seed)
    case = gen.generate_normal(m, n, k, accumulate=True, out_dtype=torch.float, per_token_b=True)
    c0 = case.c.clone()
    dg.fp8_gemm_nt(case.a, case.b, case.d, c=case.c, recipe=(1, 1, 128))
    cfg = dg.last_config()
    want = oracle.fp8_gemm_nt_blockwise_torch(case.a[0], case.a[1], case.b[0], case.b[1], gran_n=1, out_dtype=torch.float, c=c0)
    assert_close_fp32(case.d, want, f'pc192 seed {seed} {m}x{n}x{k} {cfg}')
    if '_ks_' not in cfg:                  # one launch: the K blocks in order, as the layout-agnostic kernel sums them
        dg.set_forced_config('generic_128x128')
        d2 = c0.clone()
        dg.fp8_gemm_nt(case.a, case.b, d2, c=d2, recipe=(1, 1, 128))
        dg.set_forced_config('auto')
        assert torch.equal(d2, case.d), f'pc192 seed {seed}: {cfg} differs from generic_128x128'
    return f'{m}x{n}x{k} {cfg}'


def skinnyc(seed):
    rng = random.Random(seed)
    m = rng.choice([1, 2, 7, 15, 16, 17, 24, 31, 32])
    n = rng.choice([16 * rng.randint(1, 512), 4 * rng.randint(5, 2000)])
    k = 128 * rng.randint(1, 80)
    gen.reset_seed(seed)
    case = gen.generate_normal(m, n, k)
    want = oracle.fp8_gemm_nt_blockwise_torch(case.a[0], case.a[1], case.b[0], case.b[1])
    pairs = ([('skinny_16', 'skinny_16c'), ('skinny_16', 'skinny_16ca'), ('skinny_16w', 'skinny_16wc')] if m <= 16 else []) + [('skinny_32', 'skinny_32c')]
    for plain, coal in pairs:
        outs = []
        for cfg in (plain, coal):
            dg.set_forced_config(cfg)
            d = torch.full_like(case.d, float('nan'))
            dg.fp8_gemm_nt(case.a, case.b, d)
            outs.append(d)
        dg.set_forced_config('auto')
        assert torch.equal(outs[0], outs[1]), f'skinnyc seed {seed} {m}x{n}x{k}: {coal} differs from {plain}'
        assert_close_to_oracle(outs[1], want, f'skinnyc seed {seed} {m}x{n}x{k} {coal}')
    return f'{m}x{n}x{k}'


def _packed_case(seed, k_choices):
    rng = random.Random(seed)
    groups = rng.randint(2, 10)
    ms = [rng.choice([0, 1, 127, 128, 129, 255, 256, 300, 511, 512, 640, 900]) for _ in range(groups)]
    while sum(-(-x // 128) for x in ms) > 64:
        ms.pop()
    if sum(ms) == 0:
        ms[0] = 200
    n = 256 * rng.randint(8, 16)
    k = rng.choice(k_choices)
    blocks = lambda: sum(-(-x // 128) for x in ms)                 # noqa: E731
    while blocks() * (n // 256) < 256:       # enough 128-row blocks for the tilings under test to be picked, at most 64 of them
        ms.append(rng.choice([300, 512, 640]) if blocks() <= 58 else 100)
    gen.reset_seed(seed)
    case = gen.generate_m_grouped_contiguous(len(ms), 0, n, k, True, False, actual_ms=ms, use_ue8m0=True)
    return case, gen.packed_ue8m0_operand(*case.a), gen.packed_ue8m0_operand(*case.b, mn_rows=n), ms, n, k


def _check_rows(case, got, other, tag):
    start = 0
    for g, (actual, aligned) in enumerate(zip(case.actual_ms, case.aligned_ms)):
        rows = slice(start, start + actual)
        assert torch.equal(got[rows], other[rows]), f'{tag}: group {g} differs'
        if actual:
            want = oracle.fp8_gemm_nt_blockwise_torch(case.a[0][rows], case.a[1][rows], case.b[0][g], case.b[1][g])
            assert_close_to_oracle(got[rows], want, f'{tag} group {g}')
        assert bool((got[start + actual:start + aligned] == 0).all()), f'{tag}: group {g}: padding rows must be zeros'
        start += aligned


def packedtab(seed):
    case, a, b, ms, n, k = _packed_case(seed, [4096, 4608, 7168])
    d = torch.full_like(case.d, float('nan'))
    dg.m_grouped_fp8_gemm_nt_contiguous(a, b, d, case.grouped_layout)
    cfg = dg.last_config()
    assert cfg == 'e8_quad_tab_256x256', cfg
    dg.set_forced_config('e8_quad_128x256')
    fixed = torch.full_like(case.d, float('nan'))
    dg.m_grouped_fp8_gemm_nt_contiguous(a, b, fixed, case.grouped_layout)
    dg.set_forced_config('auto')
    _check_rows(case, d, fixed, f'packedtab seed {seed} ms={ms} n={n} k={k}')
    return f'ms={ms} n={n} k={k} {cfg}'


def groupednn(seed):
    case, a, b, ms, n, k = _packed_case(seed, [512, 1024, 2048, 4096])
    ref = torch.full_like(case.d, float('nan'))
    dg.m_grouped_fp8_gemm_nt_contiguous(a, b, ref, case.grouped_layout)
    b_nn = b[0].mT.contiguous()
    dg.set_forced_config('e8_duo_bmn_256x256')
    d = torch.full_like(case.d, float('nan'))
    dg.m_grouped_fp8_gemm_nn_contiguous(a, (b_nn, b[1].mT), d, case.grouped_layout)
    cfg = dg.last_config()
    dg.set_forced_config('auto')
    assert cfg == 'e8_duo_bmn_256x256', cfg
    _check_rows(case, d, ref, f'groupednn seed {seed} ms={ms} n={n} k={k}')
    return f'ms={ms} n={n} k={k} {cfg}'


bad = 0
for name, fn in (('pc192', pc192), ('skinnyc', skinnyc), ('packedtab', packedtab), ('groupednn', groupednn)):
    if name not in which:
        continue
    for seed in range(first, first + count):
        try:
            print(name, seed, fn(seed), flush=True)
        except AssertionError as exc:
            bad += 1
            dg.set_forced_config('auto')
            print(name, seed, 'FAILED:', str(exc)[:300], flush=True)
print('done, failures:', bad)
sys.exit(1 if bad else 0)
