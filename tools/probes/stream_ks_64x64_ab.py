#!/usr/bin/env python3
"""A 64 x 64 stream tile cut along K inside the kernel (stream_ks_64x64: two K blocks per stage, four stages) at 65 .. 128 rows where neither the
64 x 32 tiles nor the 64 x 128 K split fills the chip well (m = 128, 4096 x 7168: the dense_m128 line) -- eager, cold sets, every candidate by name.
python tools/probes/stream_ks_64x64_ab.py"""
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


for m, n, k in ((128, 4096, 7168), (128, 2112, 7168), (128, 3072, 7168), (96, 4096, 7168), (128, 4096, 4096), (128, 5120, 7168), (128, 1536, 7168), (64, 4096, 7168), (256, 2112, 7168)):
    sets = max(4, min(32, int(320e6 // (n * k)) + 1))
    ops = []
    for i in range(sets):
        gen.reset_seed(i)
        c = gen.generate_normal(m, n, k)
        c.a_bf16 = c.b_bf16 = None
        ops.append(((c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1])), c.b, c.d))
    out, ref = [], None
    for cfg in ('auto', 'stream_ks_64x64', 'stream_ks_64x128', 'stream_ks_64x32', 'stream_l8_64x32'):
        try:
            dg.set_forced_config(cfg)
            ops[0][2].fill_(float('nan'))
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
