#!/usr/bin/env python3
"""Where a C2 launch spends its cycles, from the per-wave s_memtime stamps (entry, K loop begin, K loop end, after the stores): launch
skew, prologue, loop, epilogue, exit skew -- per XCD (block b runs on XCD b % 8; the tick counters of different XCDs are not compared).
    python tools/timeline.py [--configs duo_p_256x256,e8_quad_256x256] [--shape 4096x4096x7168]"""
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
ap.add_argument('--configs', default='duo_p_256x256,e8_quad_256x256')
ap.add_argument('--shape', default='4096x4096x7168')
ap.add_argument('--iters', type=int, default=20)
args = ap.parse_args()
m, n, k = (int(x) for x in args.shape.split('x'))
cases, cases_e8 = [], []
for i in range(4):
    gen.reset_seed(i)
    c = gen.generate_normal(m, n, k)
    c.a = (c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1]))
    cases.append(c)
    gen.reset_seed(i)
    e = gen.generate_normal(m, n, k, use_ue8m0=True)
    cases_e8.append((gen.packed_ue8m0_operand(*e.a), gen.packed_ue8m0_operand(*e.b, mn_rows=n), e.d))
dbg = torch.zeros(4096 * 8 * 4, dtype=torch.int64, device='cuda')
for cfg in args.configs.split(','):
    e8 = cfg.startswith('e8_')
    dg.set_forced_config(cfg)
    lib.dg_set_debug_buffer(dbg.data_ptr())

    def call(i):
        if e8:
            dg.fp8_gemm_nt(cases_e8[i % 4][0], cases_e8[i % 4][1], cases_e8[i % 4][2])
        else:
            dg.fp8_gemm_nt(cases[i % 4].a, cases[i % 4].b, cases[i % 4].d)
    for it in range(40):
        call(it)
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    start.record()
    for it in range(args.iters):
        call(it)
    end.record()
    torch.cuda.synchronize()
    wall_us = start.elapsed_time(end) / args.iters * 1e3
    lib.dg_set_debug_buffer(None)
    waves_per_block = 4 if 'quad' in cfg else 8
    blocks = 256
    t = dbg[:blocks * waves_per_block * 4].view(blocks, waves_per_block, 4).cpu().double()
    rec = {'config': cfg, 'wall_us': round(wall_us, 2)}
    spans, entry_skew, exit_skew, firsts = [], [], [], []
    for x in range(8):
        tx = t[x::8]                                            # blocks of this XCD
        t0 = tx[:, :, 0].min()
        spans.append((tx[:, :, 3].max() - t0).item())
        entry_skew.append((tx[:, :, 0].max() - t0).item())
        exit_skew.append((tx[:, :, 3].max() - tx[:, :, 3].min()).item())
        firsts.append((tx[:, :, 1].min() - t0).item())
    pro, loop, epi = t[:, :, 1] - t[:, :, 0], t[:, :, 2] - t[:, :, 1], t[:, :, 3] - t[:, :, 2]
    q = lambda v: [round(v.min().item()), round(v.mean().item()), round(v.max().item())]          # noqa: E731
    rec.update({'span_ticks_per_xcd_mean': round(sum(spans) / 8), 'span_ticks_max': max(spans), 'clock_MHz_from_span': round(max(spans) / wall_us, 1),
                'entry_skew_per_xcd': [round(v) for v in entry_skew], 'exit_skew_per_xcd': [round(v) for v in exit_skew],
                'prologue_min_mean_max': q(pro), 'loop_min_mean_max': q(loop), 'epilogue_min_mean_max': q(epi),
                'ticks_per_kblock_mean': round(loop.mean().item() / (k // 128), 1),
                'block_total_min_mean_max': q(t[:, :, 3].amax(dim=1) - t[:, :, 0].amin(dim=1))})
    print(json.dumps(rec), flush=True)
dg.set_forced_config('auto')
