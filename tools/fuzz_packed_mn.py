#!/usr/bin/env python3
"""Randomised parity runs of the packed-UE8M0 dense entry with MN-major operands read in place (e8_duo_bmn / _amn / _abmn_256x256, round 4):
random shapes (m, n multiples of 16 -- also not multiples of the 256-row / 256-column tile -- k multiples of 128), per-row scales of B,
each layout against the K-major packed call of the same problem (bit-identical) and against the oracle.
    python tools/fuzz_packed_mn.py [first_seed] [count]"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import deepgemm_amd as dg                                               # noqa: E402
import oracle                                                           # noqa: E402
from deepgemm_amd.testing import generators as gen                      # noqa: E402
from gpu_helpers import assert_close_to_oracle, cpu_pair                # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 7000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
bad = 0
for seed in range(first, first + count):
    rng = random.Random(seed)
    gen.reset_seed(seed)
    m, n, k = 16 * rng.randint(1, 70), 16 * rng.randint(1, 70), 128 * rng.randint(1, 12)
    try:
        case = gen.generate_normal(m, n, k, per_token_b=True, use_ue8m0=True)
        a, b = gen.packed_ue8m0_operand(*case.a), gen.packed_ue8m0_operand(*case.b)
        want = torch.empty((m, n), dtype=torch.bfloat16)
        oracle.fp8_gemm_nt(*cpu_pair(case.a), *cpu_pair(case.b), want, gran_n=1)
        dg.set_forced_config('e8_quad_128x256')
        ref = torch.full((m, n), float('nan'), device='cuda', dtype=torch.bfloat16)
        dg.fp8_gemm_nt(a, b, ref)
        assert_close_to_oracle(ref, want, 'K-major')
        a_km, sfa_km = a[0].t().contiguous(), a[1].t().contiguous()
        b_kn, sfb_kn = b[0].t().contiguous(), b[1].t().contiguous()
        for op, name, a_arg, b_arg in ((dg.fp8_gemm_nn, 'e8_duo_bmn_256x256', a, (b_kn, sfb_kn)),
                                       (dg.fp8_gemm_tt, 'e8_duo_amn_256x256', (a_km, sfa_km), b),
                                       (dg.fp8_gemm_tn, 'e8_duo_abmn_256x256', (a_km, sfa_km), (b_kn, sfb_kn))):
            dg.set_forced_config(name)
            d = torch.full((m, n), float('nan'), device='cuda', dtype=torch.bfloat16)
            op(a_arg, b_arg, d)
            assert dg.last_config() == name, dg.last_config()
            assert torch.equal(d, ref), f'{name} differs from the K-major kernel'
        print(f'packedmn {seed} {m}x{n}x{k} ok', flush=True)
    except Exception as e:                                               # noqa: BLE001
        bad += 1
        print(f'packedmn {seed} {m}x{n}x{k} FAILED: {str(e)[:300]}', flush=True)
    finally:
        dg.set_forced_config('auto')
print('done, failures:', bad)
