#!/usr/bin/env python3
"""K-grouped GEMM of the reference's sweep (8 groups, m 4096, n 7168, k ~ 4096 each) in three forms on one box: FP32 scales (pipe_pc, K-major flat /
MN-major in place) and packed UE8M0 words at granularity 128 / 32 (e8_quad_kg_mn: operands in place; e8_quad_kg: re-majoring pass + hardware-scaled kernel), with the re-majoring
pass timed alone.   python tools/probes/kgrouped_ue8m0_probe.py"""
import sys, random
sys.path.insert(0, '.')
import torch, deepgemm_amd as dg
from deepgemm_amd.testing import generators as gen
from deepgemm_amd.testing.numeric import calc_diff
from deepgemm_amd.gemm import _remajor

def time_us(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

g, m, n, ek = 8, 4096, 7168, 4096
random.seed(0)
ks = [max(128, int(ek * random.uniform(0.7, 1.3)) // 128 * 128) for _ in range(g)]
flops = 2.0 * m * n * sum(ks)
print('ks', ks, 'sum', sum(ks))
gen.reset_seed(0)
c32 = gen.generate_k_grouped_contiguous(g, m, n, ks, True)
t = time_us(lambda: dg.k_grouped_fp8_gemm_nt_contiguous(c32.a, c32.b, c32.d, c32.ks, c32.grouped_layout, c=c32.d))
print(f'FP32 scales, K-major flat (nt): {t:.0f} us  {flops / t / 1e6:.0f} TFLOPS  {dg.last_config()}')
del c32
c32 = gen.generate_k_grouped_contiguous(g, m, n, ks, False)
t = time_us(lambda: dg.k_grouped_fp8_gemm_tn_contiguous(c32.a, c32.b, c32.d, c32.ks, c32.grouped_layout, c=c32.d))
print(f'FP32 scales, MN-major in place (tn): {t:.0f} us  {flops / t / 1e6:.0f} TFLOPS  {dg.last_config()}')
del c32
for gran_k in (128, 32):
    gen.reset_seed(0)
    case = gen.generate_k_grouped_contiguous_ue8m0(g, m, n, ks, gran_k)
    a = (case.a[0], gen.pack_k_grouped_ue8m0(case.a[1], ks, gran_k)); b = (case.b[0], gen.pack_k_grouped_ue8m0(case.b[1], ks, gran_k))
    d = case.c.clone()
    dg.k_grouped_fp8_gemm_tn_contiguous(a, b, d, ks, case.grouped_layout, c=d, recipe=(1, 1, gran_k))
    print('gran', gran_k, 'diff', calc_diff(d, case.ref_d), dg.last_config())
    t = time_us(lambda: dg.k_grouped_fp8_gemm_tn_contiguous(a, b, d, ks, case.grouped_layout, c=d, recipe=(1, 1, gran_k)))
    tr = time_us(lambda: (_remajor(a[0].transpose(0, 1)), _remajor(b[0].transpose(0, 1))))
    print(f'packed UE8M0 gran {gran_k}, MN-major operands in place: call {t:.0f} us ({flops / t / 1e6:.0f} TFLOPS) {dg.last_config()}')
    # the same operands behind a base that is off 16 bytes: the library declines the in-place form, the host re-majors (K-major kernel)
    a_off = torch.empty((a[0].numel() + 1,), dtype=torch.uint8, device='cuda')[1:].view(torch.float8_e4m3fn).view(a[0].shape)
    a_off.copy_(a[0])
    t2 = time_us(lambda: dg.k_grouped_fp8_gemm_tn_contiguous((a_off, a[1]), b, d, ks, case.grouped_layout, c=d, recipe=(1, 1, gran_k)))
    print(f'   re-majored (operands off alignment): call {t2:.0f} us ({flops / t2 / 1e6:.0f} TFLOPS) {dg.last_config()}; the re-majoring pass alone {tr:.0f} us')
    del a_off
    del case, a, b, d
