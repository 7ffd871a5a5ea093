#!/usr/bin/env python3
"""What the hardware-scaled dense kernel does with the weight-gradient form today: packed UE8M0 scales per row of BOTH operands
(recipe (1, 1, 128)), FP32 D accumulated in place.   python tools/probes/wgrad_ue8m0_probe.py"""
import sys
sys.path.insert(0, '.')
import torch, deepgemm_amd as dg
from deepgemm_amd.testing import generators as gen
from deepgemm_amd.testing.numeric import calc_diff

def time_us(fn, n=30, sets=1):
    for _ in range(5): fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i % sets)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for m, n, k in ((4096, 4096, 7168), (576, 4096, 7168), (7168, 4096, 4096)):
    sets = 3
    cs = []
    for i in range(sets):
        gen.reset_seed(i)
        c = gen.generate_normal(m, n, k, accumulate=True, out_dtype=torch.float, per_token_b=True, use_ue8m0=True)
        a, b = gen.packed_ue8m0_operand(*c.a), gen.packed_ue8m0_operand(*c.b)
        af = (c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1])); bf = (c.b[0], dg.get_mn_major_tma_aligned_tensor(c.b[1]))
        cs.append((c, a, b, af, bf))
    c, a, b, af, bf = cs[0]
    d0 = c.d.clone()
    dg.fp8_gemm_nt(a, b, c.d, c=c.d, recipe=(1, 1, 128))
    print(m, n, k, 'packed diff', calc_diff(c.d, c.ref_d), dg.get_last_config() if hasattr(dg, 'get_last_config') else '')
    d1 = d0.clone()
    dg.fp8_gemm_nt(af, bf, d1, c=d1, recipe=(1, 1, 128))
    print('   fp32-scale diff', calc_diff(d1, c.ref_d), 'packed vs fp32-scale', calc_diff(c.d, d1))
    t_p = time_us(lambda i: dg.fp8_gemm_nt(cs[i][1], cs[i][2], cs[i][0].d, c=cs[i][0].d, recipe=(1, 1, 128)), sets=sets)
    t_f = time_us(lambda i: dg.fp8_gemm_nt(cs[i][3], cs[i][4], cs[i][0].d, c=cs[i][0].d, recipe=(1, 1, 128)), sets=sets)
    print(f'   packed {t_p:.1f} us   fp32 scales {t_f:.1f} us   ({2.0 * m * n * k / t_p / 1e6:.0f} / {2.0 * m * n * k / t_f / 1e6:.0f} TFLOPS)')
