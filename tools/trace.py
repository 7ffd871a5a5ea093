#!/usr/bin/env python3
"""Per-step s_memtime trace of the ring kernel's trace build (rabl5): prints, for a few waves of one workgroup, the
cycle count between consecutive MFMA steps of K blocks 30 and 31."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
from deepgemm_amd._lib import lib                                       # noqa: E402
from deepgemm_amd.testing import generators as gen                      # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'rabl5_256x256'     # needs a DG_EXPERIMENTS=1 build
m, n, k = 4096, 4096, 7168
gen.reset_seed(0)
c = gen.generate_normal(m, n, k)
c.a = (c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1]))
dbg = torch.zeros(4096 * 8 * 4, dtype=torch.int64, device='cuda')
dg.set_forced_config(cfg)
lib.dg_set_debug_buffer(dbg.data_ptr())
for _ in range(6):
    dg.fp8_gemm_nt(c.a, c.b, c.d)
torch.cuda.synchronize()
lib.dg_set_debug_buffer(None)
tr = dbg[8192:].view(torch.int32)[:256 * 8 * 64].view(256, 8, 64).cpu().long()
for blk in (0, 100):
    base = tr[blk, :, 0].min().item()
    print(f'== block {blk}: per wave, K block 30 then 31: start offset, then deltas between steps (cycles)')
    for w in range(8):
        row = tr[blk, w]
        a = row[0:30]
        b = row[32:62]
        da = (a[1:] - a[:-1]).tolist()
        db = (b[1:] - b[:-1]).tolist()
        print(f'wave {w}: t0={row[0].item() - base:6d} blk30 total={a[-1].item() - a[0].item():5d} ' + ' '.join(f'{x:3d}' for x in da))
        print(f'         gap={b[0].item() - a[-1].item():5d}        blk31 total={b[-1].item() - b[0].item():5d} ' + ' '.join(f'{x:3d}' for x in db))
