#!/usr/bin/env python3
"""Phases of a tile (debug stamps: entry, K loop begin, K loop end, after the stores) of the dense packed-UE8M0 GEMM in the tn layout (e8_quad_mn_256x256)
beside the nt layout (e8_quad_256x256).   python tools/probes/dense_tn_phase_stamps.py [MxNxK] [bf16|fp32acc]"""
import sys
sys.path.insert(0, '.')
import torch, deepgemm_amd as dg
from deepgemm_amd._lib import lib
from deepgemm_amd.testing import generators as gen
m, n, k = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '4096x4096x7168').split('x'))
acc = len(sys.argv) > 2 and sys.argv[2] == 'fp32acc'
gen.reset_seed(1)
c = gen.generate_normal(m, n, k, accumulate=acc, out_dtype=torch.float if acc else torch.bfloat16, per_token_b=True, use_ue8m0=True)
a, b = gen.packed_ue8m0_operand(*c.a), gen.packed_ue8m0_operand(*c.b)
a_t = (a[0].t().contiguous().t(), a[1]); b_t = (b[0].t().contiguous().t(), b[1])
kw = dict(c=c.d if acc else None, recipe=(1, 1, 128))
blocks = -(-m // 256) * -(-n // 256)
dbg = torch.zeros(blocks * 4 * 4, dtype=torch.int64, device='cuda')
for name, ops in (('tn', (a_t, b_t)), ('nt', (a, b))):
    for _ in range(20): dg.fp8_gemm_nt(*ops, c.d, **kw)
    torch.cuda.synchronize()
    dbg.zero_()
    lib.dg_set_debug_buffer(dbg.data_ptr())
    for _ in range(3): dg.fp8_gemm_nt(*ops, c.d, **kw)
    torch.cuda.synchronize()
    lib.dg_set_debug_buffer(None)
    t = dbg.view(-1, 4).cpu().double(); t = t[t[:, 0] > 0]
    pro, loop, epi = (t[:, 1] - t[:, 0]).median().item(), (t[:, 2] - t[:, 1]).median().item(), (t[:, 3] - t[:, 2]).median().item()
    print(f'{name} {dg.last_config()}: waves {t.shape[0]}  ticks: prologue {pro:.0f}  K loop {loop:.0f} ({loop / (k / 128):.0f} per block)  epilogue {epi:.0f}')
