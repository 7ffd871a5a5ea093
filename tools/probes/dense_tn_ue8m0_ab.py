#!/usr/bin/env python3
"""Dense packed-UE8M0 GEMMs whose operands are BOTH MN-major (the tn layout: the weight-gradient form of the reference's sweep) on one box: the
four-wave kernel reading them in place (e8_quad_mn_256x256, automatic) against the 8-wave in-place kernel (e8_duo_abmn_256x256, forced) and the
K-major quad kernel on the same values (nt layout, no re-majoring counted).   python tools/probes/dense_tn_ue8m0_ab.py"""
import sys
sys.path.insert(0, '.')
import torch, deepgemm_amd as dg
from deepgemm_amd.testing import generators as gen
from deepgemm_amd.testing.numeric import calc_diff

def time_us(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (m, n, k, acc, dt) in ((4096, 4096, 7168, False, torch.bfloat16), (4096, 4096, 7168, True, torch.float), (7168, 4096, 4096, True, torch.float), (2048, 7168, 2048, False, torch.bfloat16)):
    gen.reset_seed(1)
    c = gen.generate_normal(m, n, k, accumulate=acc, out_dtype=dt, per_token_b=True, use_ue8m0=True)
    a, b = gen.packed_ue8m0_operand(*c.a), gen.packed_ue8m0_operand(*c.b)                   # K-major [M, K], [N, K]; words [mn, K / 512]
    a_t = (a[0].t().contiguous().t(), a[1]); b_t = (b[0].t().contiguous().t(), b[1])        # the same values stored [K, M] / [K, N] (MN-major views)
    kw = dict(c=c.d if acc else None, recipe=(1, 1, 128))
    d0 = c.d.clone()
    dg.fp8_gemm_nt(a, b, c.d, **kw); ref_cfg = dg.last_config(); want = c.d.clone()
    out = {}
    for forced in ('auto', 'e8_duo_abmn_256x256'):
        dg.set_forced_config(forced)
        c.d.copy_(d0)
        dg.fp8_gemm_nt(a_t, b_t, c.d, **kw)
        cfg = dg.last_config()
        same = torch.equal(c.d, want)
        out[forced] = (time_us(lambda: dg.fp8_gemm_nt(a_t, b_t, c.d, **kw)), cfg, same)
    dg.set_forced_config('auto')
    t_nt = time_us(lambda: dg.fp8_gemm_nt(a, b, c.d, **kw))
    print(f'{m}x{n}x{k} acc={int(acc)} {str(dt)[6:]}: tn auto {out["auto"][0]:.1f} us {out["auto"][1]} same-bits {out["auto"][2]} | tn 8-wave {out["e8_duo_abmn_256x256"][0]:.1f} us '
          f'{out["e8_duo_abmn_256x256"][1]} same-bits {out["e8_duo_abmn_256x256"][2]} | nt {t_nt:.1f} us {ref_cfg}')
