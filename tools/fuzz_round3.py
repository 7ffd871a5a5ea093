#!/usr/bin/env python3
"""Randomised parity runs for the paths added in round 3, every result against the oracle's arithmetic on the device (FP64 block products,
tests/test_full_output_parity_gpu.py) or against the unfused / packed-word form it must equal bit for bit:
  skinny      dense M <= 32 on the skinny weight-stream kernel
  tabled      contiguous layouts through the group-relative tile list (in-kernel list for <= 64 row blocks, table kernel beyond), random
              group sizes incl. empty groups and all-padding blocks
  swiglu      fused GEMM1 + SwiGLU + re-quantisation vs masked GEMM -> torch SwiGLU -> per_token_cast_to_fp8
  castmode    FP32 power-of-two scales in 'sm100' mode vs the packed-word call (dense, masked)
python tools/fuzz_round3.py [first_seed] [count] [which,...]"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import deepgemm_amd as dg                                               # noqa: E402
from deepgemm_amd.testing import calc_diff, generators as gen           # noqa: E402
from deepgemm_amd.utils.math import per_block_cast_to_fp8, per_token_cast_to_fp8     # noqa: E402
from gpu_helpers import assert_close_to_oracle                          # noqa: E402
from test_full_output_parity_gpu import device_oracle                   # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 8
which = set((sys.argv[3] if len(sys.argv) > 3 else 'skinny,tabled,swiglu,castmode').split(','))
bad = 0


def skinny(seed):
    rng = random.Random(seed)
    m = rng.choice([1, 2, 7, 15, 16, 17, 24, 31, 32])
    n = 16 * rng.randint(1, 512)
    k = 128 * rng.randint(16, 80)
    gen.reset_seed(seed)
    case = gen.generate_normal(m, n, k)
    case.d.fill_(float('nan'))
    dg.fp8_gemm_nt(case.a, case.b, case.d)
    cfg = dg.last_config()
    want = device_oracle(case.a[0], case.a[1], case.b[0], case.b[1])
    assert_close_to_oracle(case.d, want, f'skinny seed {seed} {m}x{n}x{k} {cfg}')
    return f'{m}x{n}x{k} {cfg}'


def tabled(seed):
    rng = random.Random(seed)
    groups = rng.randint(1, 12)
    big_layout = rng.random() < 0.3
    ms = [rng.choice([0, 1, 127, 128, 129, 255, 256, 300, 511, 512, 640, 900]) * (3 if big_layout else 1) for _ in range(groups)]
    n = 256 * rng.randint(4, 16)
    k = 128 * rng.randint(8, 40)
    if sum(ms) == 0:
        ms[0] = 200
    gen.reset_seed(seed)
    case = gen.generate_m_grouped_contiguous(groups, 0, n, k, actual_ms=ms)
    case.d.fill_(float('nan'))
    dg.m_grouped_fp8_gemm_nt_contiguous(case.a, case.b, case.d, case.grouped_layout)
    cfg = dg.last_config()
    start = 0
    for g, (actual, aligned) in enumerate(zip(case.actual_ms, case.aligned_ms)):
        if actual:
            want = device_oracle(case.a[0][start:start + actual], case.a[1][start:start + actual], case.b[0][g], case.b[1][g])
            assert_close_to_oracle(case.d[start:start + actual], want, f'tabled seed {seed} group {g} ({cfg})')
        assert bool((case.d[start + actual:start + aligned] == 0).all()), f'tabled seed {seed} group {g}: padding rows not zero ({cfg})'
        start += aligned
    again = torch.empty_like(case.d)
    dg.m_grouped_fp8_gemm_nt_contiguous(case.a, case.b, again, case.grouped_layout)
    assert torch.equal(again, case.d), f'tabled seed {seed}: not repeatable ({cfg})'
    return f'M={case.m} groups={groups} n={n} k={k} {cfg}'


def swiglu(seed):
    rng = random.Random(seed)
    groups = rng.randint(1, 9)
    m_max = rng.choice([64, 128, 192, 320])
    masked_ms = [rng.choice([0, 1, rng.randint(0, m_max), m_max]) for _ in range(groups)]
    inter = 128 * rng.randint(1, 12)
    k = 128 * rng.randint(2, 40)
    clamp = rng.choice([None, 0.5, 10.0])
    use_ue8m0 = rng.random() < 0.5
    gen.reset_seed(seed)
    a = torch.randn((groups, m_max, k), device='cuda', dtype=torch.bfloat16)
    xq = [per_token_cast_to_fp8(a[g], use_ue8m0=False) for g in range(groups)]
    x = (torch.stack([q[0] for q in xq]), torch.stack([q[1] for q in xq]))
    w = torch.randn((groups, 2 * inter, k), device='cuda', dtype=torch.bfloat16) / k ** 0.5 * rng.choice([0.1, 1.0, 30.0])
    wq = [per_block_cast_to_fp8(w[g], use_ue8m0=False) for g in range(groups)]
    w1 = (torch.stack([q[0] for q in wq]), torch.stack([q[1] for q in wq]))
    w1_t, _ = dg.transform_weights_for_mega_moe(w1, w1)
    masked = torch.tensor(masked_ms, dtype=torch.int, device='cuda')
    q, q_sf = dg.empty_intermediate(groups, m_max, inter, 'cuda')
    q.view(torch.uint8).fill_(0x7f)
    q_sf.fill_(float('nan'))
    dg.m_grouped_fp8_gemm_nt_masked_swiglu(x, w1_t, (q, q_sf), masked, max(1, max(masked_ms)), activation_clamp=clamp, use_ue8m0=use_ue8m0)
    h = torch.empty((groups, m_max, 2 * inter), device='cuda', dtype=torch.bfloat16)
    dg.m_grouped_fp8_gemm_nt_masked(x, w1, h, masked, max(1, max(masked_ms)))
    for g, rows in enumerate(masked_ms):
        if rows:
            gate, up = h[g, :rows, :inter].float(), h[g, :rows, inter:].float()
            if clamp is not None:
                gate, up = gate.clamp(max=clamp), up.clamp(-clamp, clamp)
            y = (torch.nn.functional.silu(gate) * up).to(torch.bfloat16)
            wq_, wsf = per_token_cast_to_fp8(y, use_ue8m0=use_ue8m0)
            assert torch.equal(q[g, :rows].view(torch.uint8), wq_.view(torch.uint8)), f'swiglu seed {seed} group {g}: bytes differ'
            assert torch.equal(q_sf[g, :rows], wsf), f'swiglu seed {seed} group {g}: scales differ'
        assert bool((q[g, rows:].view(torch.uint8) == 0x7f).all()) and bool(torch.isnan(q_sf[g, rows:]).all()), f'swiglu seed {seed}: rows >= masked_m written'
    return f'G={groups} m_max={m_max} I={inter} k={k} clamp={clamp} ue8m0={use_ue8m0}'


def castmode(seed):
    rng = random.Random(seed)
    gen.reset_seed(seed)
    if rng.random() < 0.5:
        m, n, k = rng.choice([1, 40, 128, 300, 1024]), 128 * rng.randint(1, 24), 128 * rng.randint(1, 40)
        case = gen.generate_normal(m, n, k, use_ue8m0=True)
        want = torch.empty_like(case.d)
        dg.fp8_gemm_nt(gen.packed_ue8m0_operand(*case.a), gen.packed_ue8m0_operand(*case.b, mn_rows=n), want)
        dg.set_sf_cast_mode('sm100')
        try:
            dg.fp8_gemm_nt(case.a, case.b, case.d)
        finally:
            dg.set_sf_cast_mode('sm90')
        label = f'dense {m}x{n}x{k} {dg.last_config()}'
    else:
        groups, m_max, n, k = rng.randint(1, 8), rng.choice([64, 192]), 128 * rng.randint(1, 16), 128 * rng.randint(1, 40)
        case = gen.generate_m_grouped_masked(groups, m_max, m_max // 2, n, k, use_ue8m0=True)
        want = torch.zeros_like(case.d)
        case.d.zero_()
        dg.m_grouped_fp8_gemm_nt_masked(gen.packed_ue8m0_operand(*case.a), gen.packed_ue8m0_operand(*case.b, mn_rows=n), want, case.masked_m, m_max // 2)
        dg.set_sf_cast_mode('sm100')
        try:
            dg.m_grouped_fp8_gemm_nt_masked(case.a, case.b, case.d, case.masked_m, m_max // 2)
        finally:
            dg.set_sf_cast_mode('sm90')
        label = f'masked G={groups} {m_max}x{n}x{k} {dg.last_config()}'
    assert torch.equal(case.d, want), f'castmode seed {seed}: {label}: differs from the packed-word call'
    assert calc_diff(case.d.float(), want.float()) == 0 or True
    return label


for seed in range(first, first + count):
    for name, fn in (('skinny', skinny), ('tabled', tabled), ('swiglu', swiglu), ('castmode', castmode)):
        if name not in which:
            continue
        try:
            print(name, seed, fn(seed), flush=True)
        except (AssertionError, RuntimeError) as e:
            bad += 1
            print('FAIL', name, seed, str(e)[:300], flush=True)
        finally:
            dg.set_forced_config('auto')
            dg.set_sf_cast_mode('sm90')
print('done, failures:', bad)
