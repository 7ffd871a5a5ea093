#!/usr/bin/env python3
"""Decode-sized problems with packed UE8M0 scales at granularity 128 against granularity 32 (the MX recipe): masked grouped C5 shape and dense small M.
python tools/probes/g32_decode_probe.py"""
import sys
sys.path.insert(0, '.')
import torch, deepgemm_amd as dg
from deepgemm_amd.testing import generators as gen
from deepgemm_amd.utils.math import pack_ue8m0_to_int, per_token_cast_to_fp8

def time_us(fn, n=30, sets=1):
    for i in range(5): fn(i % sets)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i % sets)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def packed(x, mn, k, gran):
    q = per_token_cast_to_fp8(x, True, gran)
    return q[0], dg.transform_sf_into_required_layout(pack_ue8m0_to_int(q[1]), mn, k, (1, gran))

for m in (1, 16, 32, 64, 128, 256):
    n, k = 4096, 7168
    sets = 12
    ops = {128: [], 32: []}
    for i in range(sets):
        torch.manual_seed(i)
        a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16); b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
        for g in (128, 32):
            ops[g].append((packed(a, m, k, g), packed(b, n, k, g)))
    d = torch.empty((m, n), device='cuda', dtype=torch.bfloat16)
    res = {}
    for g in (128, 32):
        t = time_us(lambda i: dg.fp8_gemm_nt(ops[g][i][0], ops[g][i][1], d, recipe=(1, 1, g)), sets=sets)
        cfg = dg.last_config()
        # the same calls as a hipGraph replay: kernel + kernel boundary, no host path
        side = torch.cuda.Stream(); graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            for i in range(sets):
                dg.fp8_gemm_nt(ops[g][i][0], ops[g][i][1], d, recipe=(1, 1, g))
        tg = time_us(lambda i: graph.replay(), n=10) / sets
        res[g] = (t, cfg, tg)
    print(f'dense m={m} {n}x{k}: gran 128 {res[128][0]:.1f} us eager / {res[128][2]:.1f} us graph {res[128][1]} | gran 32 {res[32][0]:.1f} / {res[32][2]:.1f} us {res[32][1]}')
    del ops
# masked C5: 8 experts x 64 rows max, 4096 x 7168
groups, max_m, n, k = 8, 64, 4096, 7168
masked = torch.tensor([48, 33, 64, 12, 50, 64, 40, 57], device='cuda', dtype=torch.int32)
for g in (128, 32):
    cases = []
    for i in range(3):
        torch.manual_seed(i)
        a = torch.randn((groups * max_m, k), device='cuda', dtype=torch.bfloat16); b = torch.randn((groups * n, k), device='cuda', dtype=torch.bfloat16)
        qa, qb = per_token_cast_to_fp8(a, True, g), per_token_cast_to_fp8(b, True, g)
        sfa = dg.transform_sf_into_required_layout(pack_ue8m0_to_int(qa[1]).view(groups, max_m, -1), max_m, k, (1, g), groups)
        sfb = dg.transform_sf_into_required_layout(pack_ue8m0_to_int(qb[1]).view(groups, n, -1), n, k, (1, g), groups)
        cases.append(((qa[0].view(groups, max_m, k), sfa), (qb[0].view(groups, n, k), sfb)))
    d = torch.empty((groups, max_m, n), device='cuda', dtype=torch.bfloat16)
    t = time_us(lambda i: dg.m_grouped_fp8_gemm_nt_masked(cases[i][0], cases[i][1], d, masked, 48, recipe=(1, 1, g)), sets=3)
    print(f'masked 8 x <=64 x {n} x {k} gran {g}: {t:.1f} us {dg.last_config()}')
    del cases
