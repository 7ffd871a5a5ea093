#!/usr/bin/env python3
"""Packed-scale stream tiles cut along K inside the kernel (e8_stream_ks_64x32 / _64x128 and their granularity-32 forms; end of round 6) against
the unsplit tiles they replace in the automatic selection: eager calls on the current stream (the K split needs the stream's scratch buffer), cold
operand sets, calc_diff against the first result of the row.
python tools/probes/e8_stream_ks_ab.py"""
import sys
sys.path.insert(0, '.')
import torch, deepgemm_amd as dg
from deepgemm_amd.testing import calc_diff
from deepgemm_amd._lib import lib
from deepgemm_amd.utils.math import pack_ue8m0_to_int, per_token_cast_to_fp8


def time_us(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def packed(x, mn, k, gran):
    q = per_token_cast_to_fp8(x, True, gran)
    return q[0], dg.transform_sf_into_required_layout(pack_ue8m0_to_int(q[1]), mn, k, (1, gran))


SHAPES = ((128, 576, 7168), (64, 576, 7168), (33, 4096, 7168), (64, 4096, 7168), (64, 2112, 7168), (128, 576, 16384), (256, 576, 7168), (128, 512, 4096),
          (192, 4096, 7168), (256, 4096, 7168), (256, 2112, 7168))
for gran in (128, 32):
    g = '_g32' if gran == 32 else ''
    for m, n, k in SHAPES:
        sets = max(4, min(32, int(320e6 // (n * k)) + 1))
        ops = []
        for i in range(sets):
            torch.manual_seed(i)
            a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16); b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
            ops.append((packed(a, m, k, gran), packed(b, n, k, gran), torch.empty((m, n), device='cuda', dtype=torch.bfloat16)))
        del a, b
        wide = m > 128
        cfgs = ('auto', 'auto_nows', f'e8_stream2{g}_64x128' if wide else f'e8_stream_l8{g}_64x32', f'e8_stream_ks{g}_64x128' if wide else f'e8_stream_ks{g}_64x32')
        out, ref = [], None
        for cfg in cfgs:
            try:
                if cfg == 'auto_nows':      # (the selection before this change: the rule as it is without a workspace)
                    if gran != 128:
                        continue
                    dg.set_forced_config(lib.dg_select_config(0, m, n, k, 1, 0, 0, 0, 128, 0, 0, 1).decode())
                else:
                    dg.set_forced_config(cfg)
                dg.fp8_gemm_nt(ops[0][0], ops[0][1], ops[0][2], recipe=(1, 1, gran))
                name = dg.last_config()
                res = ops[0][2].float().clone()
                if ref is None: ref = res
                it = [0]
                def call():
                    o = ops[it[0] % sets]; it[0] += 1
                    dg.fp8_gemm_nt(o[0], o[1], o[2], recipe=(1, 1, gran))
                t = time_us(call)
                out.append(f'{cfg}{"=" + name if cfg.startswith("auto") else ""} {t:.1f} ({calc_diff(res, ref):.1e})')
            except Exception as e:
                out.append(f'{cfg}: {str(e)[:60]}')
            finally:
                dg.set_forced_config('auto')
        print(f'gran {gran}: {m} x {n} x {k} ({sets} sets): ' + ' | '.join(out), flush=True)
        del ops
