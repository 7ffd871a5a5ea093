#!/usr/bin/env python3
"""Per-tile fixed cost against per-K-block cost of the K-grouped hardware-scaled kernel: the same 8 x (4096 x 7168) outputs with 1024 / 2048 / 4096 /
8192 K per group, operands pre-re-majored, C entry called directly (no host work in the timed region).   python tools/probes/kgrouped_fit_probe.py"""
import sys, ctypes
sys.path.insert(0, '.')
import os
import torch, deepgemm_amd as dg
LAYOUT = 2 if os.environ.get('KG_MN') else 1          # KG_MN=1: the operands as [sum_k, mn] (read in place)
from deepgemm_amd._lib import lib, check, current_stream_ptr
from deepgemm_amd.testing import generators as gen
from deepgemm_amd.gemm import _remajor

def time_us(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

g, m, n = 8, 4096, 7168
rows = []
for k in (1024, 2048, 4096, 8192):
    ks = [k] * g
    gen.reset_seed(0)
    sum_k = sum(ks)
    a = torch.randn((sum_k, m) if LAYOUT == 2 else (m, sum_k), device='cuda').to(torch.float8_e4m3fn); b = torch.randn((sum_k, n) if LAYOUT == 2 else (n, sum_k), device='cuda').to(torch.float8_e4m3fn)
    sfa = torch.full((sum_k // 512, m), 0x7f7f7f7f, dtype=torch.int32, device='cuda'); sfb = torch.full((sum_k // 512, n), 0x7f7f7f7f, dtype=torch.int32, device='cuda')
    d = torch.zeros((g, m, n), device='cuda')
    ks_arr = (ctypes.c_int32 * g)(*ks)
    def call():
        check(lib.dg_k_grouped_fp8_gemm_ue8m0(a.data_ptr(), sfa.data_ptr(), b.data_ptr(), sfb.data_ptr(), d.data_ptr(), m, n, sum_k,
                                              ctypes.cast(ks_arr, ctypes.c_void_p), None, g, 128, 128, LAYOUT, a.stride(0), b.stride(0), sfa.stride(0), sfb.stride(0),
                                              current_stream_ptr()))
    t = time_us(call)
    rounds = g * 16 * 28 / 256
    rows.append((k, t))
    print(f'k per group {k}: {t:.0f} us, per round of 256 tiles {t / rounds:.1f} us, {2.0 * m * n * sum_k / t / 1e6:.0f} TFLOPS  {dg.last_config()}')
    del a, b, d
(k0, t0), (k1, t1) = rows[0], rows[-1]
per_kb = (t1 - t0) / ((k1 - k0) / 128) / 14
print(f'fit: {per_kb:.3f} us per K block and round, fixed {t0 / 14 - per_kb * k0 / 128:.1f} us per round')
