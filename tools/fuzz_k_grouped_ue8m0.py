#!/usr/bin/env python3
"""Randomised parity runs of the K-grouped GEMM with UE8M0 scales (e8_quad_kg_*): random group counts and extents (zeros included), M / N that are
not tile multiples, both granularities, K alignments 32 .. 256, host extents or the psum layout (ks_cpu given or missing), FP32 scale tensors or
packed words.  Checker: a float64 product of the dequantised operands of every group (torch on the GPU; the layout logic -- group starts, scale rows,
masked tails -- is what is fuzzed; tolerance: the FP32-output bound of tests/gpu_helpers.py, rel-Frobenius 5e-5; the arithmetic is pinned to the oracle by tests/test_k_grouped_ue8m0_gpu.py).
    python tools/fuzz_k_grouped_ue8m0.py [seeds] [first_seed] [aligned]        (aligned: m > 128, m and n multiples of 16 -- every case reads its
    MN-major operands in place, e8_quad_kg_mn_*)"""
import random
import sys
sys.path.insert(0, '.')
import torch
import deepgemm_amd as dg
from deepgemm_amd.testing import generators as gen

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
aligned = len(sys.argv) > 3 and sys.argv[3] == 'aligned'
fails = 0
for seed in range(first, first + seeds):
    rng = random.Random(seed)
    gran_k = rng.choice((32, 128))
    k_alignment = rng.choice((32, 64, 96, 128, 160, 192, 224, 256))
    use_psum = rng.random() < 0.6
    groups = rng.randint(1, 70 if use_psum and rng.random() < 0.15 else 9)
    m = 4 * rng.randint(1, 150)
    n = 4 * rng.randint(1, 180)
    if aligned:
        m, n = 16 * rng.randint(9, 60), 16 * rng.randint(1, 70)
    if use_psum:
        real_ks = [0 if rng.random() < 0.15 else rng.randint(1, 900) for _ in range(groups)]
    else:
        real_ks = [k_alignment * rng.randint(0, max(1, 900 // k_alignment)) for _ in range(groups)]
    if sum(real_ks) == 0:
        real_ks[0] = k_alignment
    packed_words = rng.random() < 0.5
    ks_mode = rng.choice(('given', 'none')) if use_psum else 'given'
    gen.reset_seed(seed)
    dg.set_mk_alignment_for_contiguous_layout(k_alignment)
    mode = dg.get_sf_cast_mode()
    try:
        case = gen.generate_k_grouped_contiguous_ue8m0(groups, m, n, real_ks, gran_k, k_alignment, use_psum_layout=use_psum)
        a, b = case.a, case.b
        if packed_words:
            a = (a[0], gen.pack_k_grouped_ue8m0(a[1], real_ks, gran_k)); b = (b[0], gen.pack_k_grouped_ue8m0(b[1], real_ks, gran_k))
        elif gran_k == 128:
            dg.set_sf_cast_mode('sm100')
        d = case.c.clone()
        dg.k_grouped_fp8_gemm_tn_contiguous(a, b, d, case.ks if ks_mode == 'given' else None, case.grouped_layout, c=d, recipe=(1, 1, gran_k),
                                            use_psum_layout=use_psum)
        cfg = dg.last_config()
        worst = 0.0
        for g_, k in enumerate(real_ks):
            if k == 0:
                ok = torch.equal(d[g_], case.c[g_])
                worst = max(worst, 0.0 if ok else 1.0)
                continue
            (a_g, sfa_g), (b_g, sfb_g) = case.a_groups[g_], case.b_groups[g_]
            ad = a_g.double() * sfa_g.double().repeat_interleave(gran_k, dim=1)[:, :a_g.size(1)]
            bd = b_g.double() * sfb_g.double().repeat_interleave(gran_k, dim=1)[:, :b_g.size(1)]
            want = case.c[g_].double() + ad @ bd.t()
            rel = ((d[g_].double() - want).norm() / want.norm()).item()
            worst = max(worst, rel)
        ok = worst <= 5e-5 and cfg.startswith('e8_quad_kg_mn_' if aligned else 'e8_quad_kg_')
        fails += 0 if ok else 1
        print(f'seed {seed}: gran {gran_k} align {k_alignment} psum {int(use_psum)} ks {ks_mode} packed {int(packed_words)} G {groups} m {m} n {n} '
              f'sum_k {sum(real_ks)} {cfg} worst rel {worst:.2e} {"ok" if ok else "FAIL"}')
    finally:
        dg.set_mk_alignment_for_contiguous_layout(128)
        dg.set_sf_cast_mode(mode)
print(f'{seeds} cases, {fails} failures')
sys.exit(1 if fails else 0)
