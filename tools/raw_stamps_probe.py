#!/usr/bin/env python3
"""Per-wave debug stamps of one launch (entry, K loop begin, K loop end, after the stores; s_memtime ticks = shader cycles, comparable only inside a
wave): where a tile's time goes.   python tools/raw_stamps_probe.py [config] [MxNxK]"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg
from deepgemm_amd._lib import lib
from deepgemm_amd.testing import generators as gen
cfg = sys.argv[1] if len(sys.argv) > 1 else 'duo_p_256x256'
m, n, k = (int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else '4096x4096x7168').split('x'))
gen.reset_seed(0)
c = gen.generate_normal(m, n, k)
a = (c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1]))
dbg = torch.zeros(4096 * 8 * 4, dtype=torch.int64, device='cuda')
dg.set_forced_config(cfg)
for _ in range(50):
    dg.fp8_gemm_nt(a, c.b, c.d)
torch.cuda.synchronize()
lib.dg_set_debug_buffer(dbg.data_ptr()); dbg.zero_()
dg.fp8_gemm_nt(a, c.b, c.d); torch.cuda.synchronize()
lib.dg_set_debug_buffer(None)
t = dbg[:256 * 8 * 4].view(256, 8, 4).cpu().double()
t = t[t[:, 0, 0] > 0]
pro, loop, epi = t[:, :, 1] - t[:, :, 0], t[:, :, 2] - t[:, :, 1], t[:, :, 3] - t[:, :, 2]
q = lambda v: [round(v.min().item()), round(v.median().item()), round(v.max().item())]
print(json.dumps({'config': cfg, 'shape': f'{m}x{n}x{k}', 'blocks': int(t.shape[0]), 'prologue_ticks_min_med_max': q(pro), 'loop_ticks': q(loop),
                  'ticks_per_k_block_median': round(loop.median().item() / (k // 128), 1), 'epilogue_ticks': q(epi)}))
