#!/usr/bin/env python3
"""64 .. 128 rows, wide layers, long K loops (m = 128, 7168 x 16384 is in the reference's sweep): the automatic pick against the stream tiles cut
along K inside the kernel, both scale formats -- eager calls over cold operand sets.   python tools/probes/m128_long_k_ab.py"""
import sys
sys.path.insert(0, '.')
import torch, deepgemm_amd as dg
from deepgemm_amd.testing import calc_diff, generators as gen
from deepgemm_amd.utils.math import pack_ue8m0_to_int, per_token_cast_to_fp8


def time_us(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def packed(x, mn, k, gran=128):
    q = per_token_cast_to_fp8(x, True, gran)
    return q[0], dg.transform_sf_into_required_layout(pack_ue8m0_to_int(q[1]), mn, k, (1, gran))


for fmt in ('fp32', 'packed'):
    e = '' if fmt == 'fp32' else 'e8_'
    for m, n, k in ((128, 7168, 16384), (64, 7168, 16384), (96, 7168, 16384), (128, 4096, 16384), (128, 7168, 8192), (128, 4096, 10240), (128, 6144, 7168)):
        sets = max(3, min(32, int(320e6 // (n * k)) + 1))
        ops = []
        for i in range(sets):
            if fmt == 'fp32':
                gen.reset_seed(i)
                c = gen.generate_normal(m, n, k)
                c.a_bf16 = c.b_bf16 = None
                ops.append(((c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1])), c.b, c.d))
            else:
                torch.manual_seed(i)
                a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16); b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
                ops.append((packed(a, m, k), packed(b, n, k), torch.empty((m, n), device='cuda', dtype=torch.bfloat16)))
        out, ref = [], None
        for cfg in ('auto', e + 'stream_ks_64x128', e + 'stream_ks_64x32', e + 'stream2_64x128', e + 'stream_nt2_64x128'):
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
            except Exception as ex:
                out.append(f'{cfg}: {str(ex)[:50]}')
            finally:
                dg.set_forced_config('auto')
        print(f'{fmt} {m} x {n} x {k} ({sets} sets): ' + ' | '.join(out), flush=True)
        del ops
