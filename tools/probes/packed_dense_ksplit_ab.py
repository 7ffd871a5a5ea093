#!/usr/bin/env python3
"""Under-filled packed-scale dense shapes: the K-split of the hardware-scaled path (e8_quad_ks_*: K pieces as the groups of one launch + summing
kernel) against one launch (forced 128-row kernel) and against the FP32-scale path on the same problem.   python tools/probes/packed_dense_ksplit_ab.py"""
import sys
sys.path.insert(0, '.')
import torch, deepgemm_amd as dg
from deepgemm_amd.testing import generators as gen
from deepgemm_amd.testing.numeric import calc_diff

def time_us(fn, n=20, sets=1):
    for i in range(4): fn(i % sets)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i % sets)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (m, n, k, acc, dt) in ((4096, 512, 32768, False, torch.bfloat16), (576, 4096, 7168, True, torch.float), (4096, 576, 7168, False, torch.bfloat16),
                           (2112, 4096, 7168, True, torch.float), (1024, 1024, 16384, False, torch.bfloat16)):
    sets = 3
    cs = []
    for i in range(sets):
        gen.reset_seed(i)
        c = gen.generate_normal(m, n, k, accumulate=acc, out_dtype=dt, per_token_b=True, use_ue8m0=True)
        a, b = gen.packed_ue8m0_operand(*c.a), gen.packed_ue8m0_operand(*c.b)
        af = (c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1])); bf = (c.b[0], dg.get_mn_major_tma_aligned_tensor(c.b[1]))
        c.a_bf16 = c.b_bf16 = None
        cs.append((c, a, b, af, bf))
    kw = lambda c: dict(c=c.d if acc else None, recipe=(1, 1, 128))
    t_auto = time_us(lambda i: dg.fp8_gemm_nt(cs[i][1], cs[i][2], cs[i][0].d, **kw(cs[i][0])), sets=sets); cfg_auto = dg.last_config()
    diff = calc_diff(cs[0][0].d, cs[0][0].ref_d) if not acc else float('nan')
    dg.set_forced_config('e8_quad_128x256')
    t_one = time_us(lambda i: dg.fp8_gemm_nt(cs[i][1], cs[i][2], cs[i][0].d, **kw(cs[i][0])), sets=sets); cfg_one = dg.last_config()
    dg.set_forced_config('auto')
    t_f32 = time_us(lambda i: dg.fp8_gemm_nt(cs[i][3], cs[i][4], cs[i][0].d, **kw(cs[i][0])), sets=sets); cfg_f32 = dg.last_config()
    print(f'{m}x{n}x{k} acc={int(acc)} {str(dt)[6:]}: packed auto {t_auto:.1f} us {cfg_auto} (diff {diff:.2e}) | packed one launch {t_one:.1f} us {cfg_one} | FP32 scales {t_f32:.1f} us {cfg_f32}')
    del cs
