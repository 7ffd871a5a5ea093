#!/usr/bin/env python3
"""17 .. 32 rows: the skinny weight-stream kernels (the automatic pick) against the 64 x 32 stream tile cut along K inside the kernel, both scale
formats -- hipGraph replays of one call per cold operand set (the capture stream warmed first: it owns the K split's scratch buffer).
python tools/probes/m17_32_skinny_vs_ks.py"""
import os, sys
sys.path.insert(0, '.')
GRAN = int(os.environ.get('GRAN', '128'))       # (GRAN=32: the packed rows at granularity 32, forced names with _g32)
G = '_g32' if GRAN == 32 else ''
import torch, deepgemm_amd as dg
from deepgemm_amd.testing import calc_diff, generators as gen
from deepgemm_amd.utils.math import pack_ue8m0_to_int, per_token_cast_to_fp8


def time_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def packed(x, mn, k, gran=GRAN):
    q = per_token_cast_to_fp8(x, True, gran)
    return q[0], dg.transform_sf_into_required_layout(pack_ue8m0_to_int(q[1]), mn, k, (1, gran))


for fmt in (('packed',) if GRAN == 32 else ('fp32', 'packed')):
    for (n, k) in (((576, 7168), (1536, 7168), (2112, 7168), (4096, 4096)) if GRAN == 32 else ((4096, 7168), (2112, 7168), (576, 7168), (1536, 7168), (4096, 4096), (7168, 16384 // 2))):
        for m in (16, 17, 24, 32, 33):
            sets = max(4, min(32, int(320e6 // (n * k)) + 1))
            ops = []
            for i in range(sets):
                if fmt == 'fp32':
                    gen.reset_seed(i)
                    c = gen.generate_normal(m, n, k)
                    ops.append(((c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1])), c.b, c.d))
                else:
                    torch.manual_seed(i)
                    a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16); b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
                    ops.append((packed(a, m, k), packed(b, n, k), torch.empty((m, n), device='cuda', dtype=torch.bfloat16)))
            out, ref = [], None
            KW = dict(recipe=(1, 1, GRAN)) if fmt == 'packed' else {}
            for cfg in ('auto', 'stream_ks_64x32' if fmt == 'fp32' else f'e8_stream_ks{G}_64x32', 'stream_l8_64x32' if fmt == 'fp32' else f'e8_stream_l8{G}_64x32') + ((f'e8_skinny{G}_32',) if fmt == 'packed' and 16 < m <= 32 else ()):
                try:
                    dg.set_forced_config(cfg)
                    side = torch.cuda.Stream()
                    with torch.cuda.stream(side):
                        dg.fp8_gemm_nt(*ops[0], **KW)
                    side.synchronize()
                    name = dg.last_config()
                    res = ops[0][2].float().clone()
                    if ref is None: ref = res
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, stream=side):
                        for o in ops:
                            dg.fp8_gemm_nt(*o, **KW)
                    t = time_us(graph.replay) / sets
                    out.append(f'{cfg}{"=" + name if cfg == "auto" else ""} {t:.1f} ({calc_diff(res, ref):.1e})')
                except Exception as e:
                    out.append(f'{cfg}: {str(e)[:60]}')
                finally:
                    dg.set_forced_config('auto')
            print(f'{fmt} {m} x {n} x {k} ({sets} sets): ' + ' | '.join(out), flush=True)
            del ops
