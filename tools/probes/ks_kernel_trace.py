#!/usr/bin/env python3
"""Kernel-trace companion of the K-split probes: ONE shape per run, 300 calls of the automatic pick and 300 of the unsplit tile it replaced over cold
operand sets, so that `rocprofv3 --kernel-trace --stats` shows the two kernels' average durations side by side (no host path in them).
rocprofv3 --kernel-trace --stats --output-format csv -d OUT -o ks -- python tools/probes/ks_kernel_trace.py M N K [packed]"""
import sys
sys.path.insert(0, '.')
import torch, deepgemm_amd as dg
from deepgemm_amd.testing import generators as gen
from deepgemm_amd.utils.math import pack_ue8m0_to_int, per_token_cast_to_fp8

m, n, k = (int(x) for x in sys.argv[1:4])
is_packed = len(sys.argv) > 4


def packed(x, mn):
    q = per_token_cast_to_fp8(x, True, 128)
    return q[0], dg.transform_sf_into_required_layout(pack_ue8m0_to_int(q[1]), mn, k, (1, 128))


sets = max(4, min(32, int(320e6 // (n * k)) + 1))
ops = []
for i in range(sets):
    if is_packed:
        torch.manual_seed(i)
        a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16); b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
        ops.append((packed(a, m), packed(b, n), torch.empty((m, n), device='cuda', dtype=torch.bfloat16)))
    else:
        gen.reset_seed(i)
        c = gen.generate_normal(m, n, k)
        ops.append(((c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1])), c.b, c.d))
for cfg in ('auto', 'e8_stream_l8_64x32' if is_packed else 'stream_l8_64x32'):
    dg.set_forced_config(cfg)
    for i in range(300):
        dg.fp8_gemm_nt(*ops[i % sets])
    torch.cuda.synchronize()
    print(m, n, k, 'packed' if is_packed else 'fp32', cfg, '->', dg.last_config(), flush=True)
dg.set_forced_config('auto')
