#!/usr/bin/env python3
"""In-kernel timing of one configuration on one dense shape: s_memtime stamps per wave (entry, K loop begin/end, after
stores) next to the HIP-event wall time -> shader clock, cycles per K block, fixed overhead.  Prints one JSON line per
configuration.   python tools/cycles.py --configs a,b --shape 4096x4096x7168"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
from deepgemm_amd._lib import lib                                       # noqa: E402
from deepgemm_amd.testing import generators as gen                      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--configs', default='ring_256x256')
ap.add_argument('--shape', default='4096x4096x7168')
ap.add_argument('--iters', type=int, default=20)
ap.add_argument('--sets', type=int, default=4)
ap.add_argument('--per-col', action='store_true', help='recipe (1, 1, 128): per-row SFB, FP32 accumulate into D')
args = ap.parse_args()
m, n, k = (int(x) for x in args.shape.split('x'))
cases = []
for i in range(args.sets):
    gen.reset_seed(i)
    c = gen.generate_normal(m, n, k, accumulate=args.per_col, out_dtype=torch.float if args.per_col else torch.bfloat16,
                            per_token_b=args.per_col)
    c.a = (c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1]))
    if args.per_col:
        c.b = (c.b[0], dg.get_mn_major_tma_aligned_tensor(c.b[1]))
    cases.append(c)
dbg = torch.zeros(4096 * 8 * 4, dtype=torch.int64, device='cuda')
for cfg in args.configs.split(','):
    dg.set_forced_config(cfg)
    lib.dg_set_debug_buffer(dbg.data_ptr())
    def call(cc):
        if args.per_col:
            dg.fp8_gemm_nt(cc.a, cc.b, cc.d, c=cc.d, recipe=(1, 1, 128))
        else:
            dg.fp8_gemm_nt(cc.a, cc.b, cc.d)
    for it in range(5):
        call(cases[it % len(cases)])
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    start.record()
    for it in range(args.iters):
        call(cases[it % len(cases)])
    end.record()
    torch.cuda.synchronize()
    wall_us = start.elapsed_time(end) / args.iters * 1e3
    lib.dg_set_debug_buffer(None)
    nwaves = 256 * 8
    t = dbg[:nwaves * 4].view(nwaves, 4).cpu().double()
    t0 = t[:, 0].min()
    total = (t[:, 3].max() - t0).item()
    loop = (t[:, 2] - t[:, 1])
    pro = (t[:, 1] - t[:, 0])
    epi = (t[:, 3] - t[:, 2])
    entry_skew = (t[:, 0].max() - t0).item()
    num_kb = k // 128
    print(json.dumps({'config': cfg, 'shape': args.shape, 'wall_us': round(wall_us, 2), 'ticks_total': total,
                      'MHz_if_ticks_are_cycles': round(total / wall_us, 1),
                      'loop_ticks_mean': round(loop.mean().item()), 'loop_ticks_max': loop.max().item(),
                      'ticks_per_kblock': round(loop.mean().item() / num_kb, 1),
                      'prologue_ticks_mean': round(pro.mean().item()), 'epilogue_ticks_mean': round(epi.mean().item()),
                      'epilogue_ticks_max': epi.max().item(), 'entry_skew_ticks': entry_skew,
                      'loop_frac': round(loop.mean().item() / total, 3)}), flush=True)
dg.set_forced_config('auto')
