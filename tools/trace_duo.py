#!/usr/bin/env python3
"""Segment-level s_memtime trace of the duo kernel's trace build (dabl3): 8 stamps per K block (before / after each of
the four segment barriers) for K blocks 28..35, per wave of one workgroup."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
from deepgemm_amd._lib import lib                                       # noqa: E402
from deepgemm_amd.testing import generators as gen                      # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'dabl3_256x256'     # needs a DG_EXPERIMENTS=1 build
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
names = ['bar', 'L_a', 'bar', 'M_a', 'bar', 'L_b', 'bar', 'M_b']
for blk in (0, 77):
    base = tr[blk, :, 0].min().item()
    print(f'== workgroup {blk}: per wave, deltas between stamps over K blocks 28..31 ({" ".join(names)} per block; "bar" = wait at the barrier in front of the segment)')
    for w in range(8):
        row = tr[blk, w, :32]
        d = (row[1:] - row[:-1]).tolist()
        print(f'wave {w}: t0={row[0].item() - base:6d} | ' + ' | '.join(' '.join(f'{x:4d}' for x in d[8 * b:8 * b + 8]) for b in range(4)))
        inner = tr[blk, w, 32:48]
        la = []
        for b in range(4):
            s1 = tr[blk, w, 8 * b + 1].item()     # stamp after the L_a barrier
            e = tr[blk, w, 8 * b + 2].item()      # stamp at the end of L_a
            q = inner[4 * b:4 * b + 3].tolist()
            la.append(f'reads-issued {q[0] - s1:4d} mul {q[1] - q[0]:4d} vmem-issued {q[2] - q[1]:4d} lds-wait {e - q[2]:4d}')
        print('         L_a inner: ' + ' | '.join(la))
