#!/usr/bin/env python3
"""Dense FP32-scale problems of 129 .. 256 rows with few tiles and a long K loop: the stream tiles cut along K inside the kernel against the 8-wave
K split (duo_sk_128x256 + summing kernel) -- eager calls over cold operand sets.   python tools/probes/ks_vs_duo_sk_ab.py"""
import sys
sys.path.insert(0, '.')
import torch, deepgemm_amd as dg
from deepgemm_amd.testing import calc_diff, generators as gen


def time_us(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for m, n, k in ((200, 1024, 16384), (256, 1024, 16384), (130, 1024, 16384), (192, 1536, 16384), (200, 1024, 8192), (256, 1024, 7168), (256, 512, 16384), (192, 2048, 16384)):
    sets = max(4, min(32, int(320e6 // (n * k)) + 1))
    ops = []
    for i in range(sets):
        gen.reset_seed(i)
        c = gen.generate_normal(m, n, k)
        c.a_bf16 = c.b_bf16 = None
        ops.append(((c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1])), c.b, c.d))
    out, ref = [], None
    for cfg in ('auto', 'duo_sk_128x256', 'stream_ks_64x32', 'stream_ks_64x128', 'stream_l8_64x32'):
        try:
            dg.set_forced_config(cfg)
            dg.fp8_gemm_nt(*ops[0])
            name = dg.last_config()
            res = ops[0][2].float().clone()
            if ref is None: ref = res
            it = [0]
            def call():
                o = ops[it[0] % sets]; it[0] += 1
                dg.fp8_gemm_nt(*o)
            t = time_us(call)
            out.append(f'{cfg}{"=" + name if cfg == "auto" else ""} {t:.1f} ({calc_diff(res, ref):.1e})')
        except RuntimeError as e:
            out.append(f'{cfg}: {str(e)[:50]}')
        finally:
            dg.set_forced_config('auto')
    print(f'{m} x {n} x {k} ({sets} sets): ' + ' | '.join(out), flush=True)
    del ops
