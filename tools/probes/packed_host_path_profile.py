#!/usr/bin/env python3
"""Host cost of a packed-UE8M0 dense call at decode size (granularity 128 and 32): enqueue time per call and a cProfile of the Python side.
python tools/probes/packed_host_path_profile.py [m]"""
import sys, time, cProfile, pstats, io
sys.path.insert(0, '.')
import torch, deepgemm_amd as dg
from deepgemm_amd.utils.math import pack_ue8m0_to_int, per_token_cast_to_fp8
m = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n, k = 4096, 7168
def packed(x, mn, g):
    q = per_token_cast_to_fp8(x, True, g)
    return q[0], dg.transform_sf_into_required_layout(pack_ue8m0_to_int(q[1]), mn, k, (1, g))
a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16); b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
d = torch.empty((m, n), device='cuda', dtype=torch.bfloat16)
for g in (128, 32):
    pa, pb = packed(a, m, g), packed(b, n, g)
    print('gran', g, 'sfa', tuple(pa[1].shape), pa[1].stride(), pa[1].dtype, 'sfb', tuple(pb[1].shape), pb[1].stride())
    for _ in range(50): dg.fp8_gemm_nt(pa, pb, d, recipe=(1, 1, g))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): dg.fp8_gemm_nt(pa, pb, d, recipe=(1, 1, g))
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f'gran {g}: enqueue {1e6 * (t1 - t0) / 200:.1f} us per call ({dg.last_config()})')
    pr = cProfile.Profile(); pr.enable()
    for _ in range(500): dg.fp8_gemm_nt(pa, pb, d, recipe=(1, 1, g))
    pr.disable(); torch.cuda.synchronize()
    out = io.StringIO(); pstats.Stats(pr, stream=out).sort_stats('cumulative').print_stats(14); print('\n'.join(out.getvalue().splitlines()[4:26]))
