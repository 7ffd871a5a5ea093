#!/usr/bin/env python3
"""Which 32-K group of a 128-K block does byte j of a granularity-32 scale word reach in the G32 kernels?  (round 6 bring-up probe)
1. words with four equal bytes must reproduce the gran-128 kernel bit for bit (data path, loads);
2. a factor 2 on ONE K group of A (resp. B): which group of the FP64 dequantised reference matches?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deepgemm_amd as dg
from deepgemm_amd.utils.math import pack_ue8m0_to_int, per_token_cast_to_fp8

torch.manual_seed(0)
m, n, k = 256, 256, 512
a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16)
b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
a_q, sfa = per_token_cast_to_fp8(a, use_ue8m0=True)
b_q, sfb = per_token_cast_to_fp8(b, use_ue8m0=True)
d128 = torch.empty((m, n), device='cuda', dtype=torch.float)
dg.fp8_gemm_nt((a_q, pack_ue8m0_to_int(sfa)), (b_q, pack_ue8m0_to_int(sfb)), d128)
print('gran128 kernel', dg.last_config())
sfa32, sfb32 = sfa.repeat_interleave(4, dim=1).contiguous(), sfb.repeat_interleave(4, dim=1).contiguous()


def run32(sa, sb):
    d = torch.empty((m, n), device='cuda', dtype=torch.float)
    dg.fp8_gemm_nt((a_q, pack_ue8m0_to_int(sa)), (b_q, pack_ue8m0_to_int(sb)), d, recipe=(1, 1, 32))
    return d


def ref(sa, sb):
    ad = (a_q.double().view(m, k // 32, 32) * sa.double().unsqueeze(-1)).view(m, k)
    bd = (b_q.double().view(n, k // 32, 32) * sb.double().unsqueeze(-1)).view(n, k)
    return (ad @ bd.t()).float()


d = run32(sfa32, sfb32)
print('g32 kernel', dg.last_config(), 'equal bytes: max |diff| vs gran128 kernel', float((d - d128).abs().max()), 'vs ref', float((d - ref(sfa32, sfb32)).abs().max()))
for side in 'ab':
    for g in range(4):
        sa, sb = sfa32.clone(), sfb32.clone()
        (sa if side == 'a' else sb)[:, g::4] *= 4.0
        d = run32(sa, sb)
        errs = []
        for g2 in range(4):
            sa2, sb2 = sfa32.clone(), sfb32.clone()
            (sa2 if side == 'a' else sb2)[:, g2::4] *= 4.0
            errs.append(float((d - ref(sa2, sb2)).abs().max()))
        print(side, 'factor on byte', g, '-> error against the reference with the factor on K group 0..3:', ['%.3g' % e for e in errs])
