#!/usr/bin/env python3
"""Granularity-32 packed words in the group ring of the stream tiles (stream_kernel_body, GSG; end of round 6): dense decode-sized shapes per
configuration (hipGraph replay over cold operand sets) and the masked C5 shape.  Run once per library build (DG_VARIANT).
python tools/probes/g32_stream_group_words_ab.py"""
import sys
sys.path.insert(0, '.')
import torch, deepgemm_amd as dg
from deepgemm_amd.utils.math import pack_ue8m0_to_int, per_token_cast_to_fp8


def time_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def packed(x, mn, k, gran=32):
    q = per_token_cast_to_fp8(x, True, gran)
    return q[0], dg.transform_sf_into_required_layout(pack_ue8m0_to_int(q[1]), mn, k, (1, gran))


for (m, n, k) in ((40, 4096, 7168), (64, 4096, 7168), (128, 4096, 7168), (128, 7168, 2048), (256, 4096, 7168)):
    sets = max(2, int(320e6 // (n * k)) + 1)
    ops = []
    for i in range(sets):
        torch.manual_seed(i)
        a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16); b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
        ops.append((packed(a, m, k), packed(b, n, k)))
    d = torch.empty((m, n), device='cuda', dtype=torch.bfloat16)
    out, ref = [], None
    for cfg in ('auto', 'e8_stream_g32_64x32', 'e8_stream_l8_g32_64x32', 'e8_stream2_g32_64x128', 'e8_quad_g32_128x256'):
        try:
            dg.set_forced_config(cfg)
            dg.fp8_gemm_nt(ops[0][0], ops[0][1], d, recipe=(1, 1, 32))
            name = dg.last_config()
            bits = d.view(torch.int16).clone()
            if ref is None: ref = bits
            same = bool(torch.equal(bits, ref))
            side = torch.cuda.Stream(); graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, stream=side):
                for i in range(sets):
                    dg.fp8_gemm_nt(ops[i][0], ops[i][1], d, recipe=(1, 1, 32))
            t = time_us(graph.replay) / sets
            out.append(f'{cfg}{"=" + name if cfg == "auto" else ""} {t:.1f} us{"" if same else " BITS DIFFER"}')
        except RuntimeError as e:
            out.append(f'{cfg}: {str(e)[:50]}')
        finally:
            dg.set_forced_config('auto')
    print(f'dense g32 {m} x {n} x {k}: ' + ' | '.join(out), flush=True)
    del ops

groups, max_m, n, k = 8, 64, 4096, 7168
masked = torch.tensor([48, 33, 64, 12, 50, 64, 40, 57], device='cuda', dtype=torch.int32)
cases = []
for i in range(3):
    torch.manual_seed(i)
    a = torch.randn((groups * max_m, k), device='cuda', dtype=torch.bfloat16); b = torch.randn((groups * n, k), device='cuda', dtype=torch.bfloat16)
    qa, qb = per_token_cast_to_fp8(a, True, 32), per_token_cast_to_fp8(b, True, 32)
    sfa = dg.transform_sf_into_required_layout(pack_ue8m0_to_int(qa[1]).view(groups, max_m, -1), max_m, k, (1, 32), groups)
    sfb = dg.transform_sf_into_required_layout(pack_ue8m0_to_int(qb[1]).view(groups, n, -1), n, k, (1, 32), groups)
    cases.append(((qa[0].view(groups, max_m, k), sfa), (qb[0].view(groups, n, k), sfb)))
d = torch.empty((groups, max_m, n), device='cuda', dtype=torch.bfloat16)
it = [0]
def call():
    c = cases[it[0] % 3]; it[0] += 1
    dg.m_grouped_fp8_gemm_nt_masked(c[0], c[1], d, masked, 48, recipe=(1, 1, 32))
for rep in range(2):
    print(f'masked g32 8 x <=64 x {n} x {k}: {time_us(call, n=60):.1f} us {dg.last_config()}', flush=True)
