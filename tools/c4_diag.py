#!/usr/bin/env python3
"""Where BASELINE configs[3] (M-grouped contiguous, 8 groups, N 4096, K 7168) spends its time: whole-call HIP-event time of each
forced configuration next to the s_memtime stamps (100 MHz) of every workgroup's first tile -- entry, K loop begin / end, after the
stores -- so that the per-tile cost and the launch span can be set against rounds x tile time.  One JSON line per configuration.
    python tools/c4_diag.py [--case 8x512x4096x7168] [--configs auto,duo_128x256,...]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
from deepgemm_amd._lib import lib                                       # noqa: E402
from deepgemm_amd.testing import calc_diff, generators as gen          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--case', default='8x512x4096x7168')
ap.add_argument('--configs', default='auto,duo_128x256,pipe_128x256,duo_256x256')
ap.add_argument('--iters', type=int, default=10)
args = ap.parse_args()
g, em, n, k = (int(x) for x in args.case.split('x'))
gen.reset_seed(0)
case = gen.generate_m_grouped_contiguous(g, em, n, k)
a = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1]))
dbg = torch.zeros(4096 * 8 * 4, dtype=torch.int64, device='cuda')


def call():
    dg.m_grouped_fp8_gemm_nt_contiguous(a, case.b, case.d, case.grouped_layout)


for cfg in args.configs.split(','):
    dg.set_forced_config(cfg)
    try:
        call()
        torch.cuda.synchronize()
    except RuntimeError as e:
        print(json.dumps({'config': cfg, 'error': str(e)[:120]}), flush=True)
        continue
    diff = calc_diff(torch.nan_to_num(case.d), torch.nan_to_num(case.ref_d))
    t_end = time.time() + 0.3
    while time.time() < t_end:
        for _ in range(4):
            call()
        torch.cuda.synchronize()
    bursts = []
    for _ in range(5):
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(args.iters):
            call()
        end.record()
        torch.cuda.synchronize()
        bursts.append(start.elapsed_time(end) / args.iters * 1e3)
    us = sorted(bursts)[2]
    dbg.zero_()
    lib.dg_set_debug_buffer(dbg.data_ptr())
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    lib.dg_set_debug_buffer(None)
    t = dbg.view(-1, 4).cpu().double()
    t = t[t[:, 0] > 0]
    out = {'case': args.case, 'm_total': case.m, 'valid_rows': int(sum(case.actual_ms)), 'config': cfg, 'kernel': dg.last_config(),
           'us_per_call': round(us, 1), 'tflops_valid_rows': round(2.0 * sum(case.actual_ms) * n * k / us / 1e6, 1), 'calc_diff': float(diff),
           'waves_stamped': int(t.shape[0])}
    if t.shape[0] > 0:
        t0 = t[:, 0].min()
        loop = t[:, 2] - t[:, 1]
        out.update({'span_ticks': (t[:, 3].max() - t0).item(), 'entry_skew_ticks': (t[:, 0].max() - t0).item(),
                    'prologue_ticks_mean': round((t[:, 1] - t[:, 0]).mean().item()),
                    'loop_ticks_mean': round(loop.mean().item()), 'loop_ticks_min': loop.min().item(), 'loop_ticks_max': loop.max().item(),
                    'ticks_per_kblock_mean': round(loop.mean().item() / (k // 128), 2),
                    'epilogue_ticks_mean': round((t[:, 3] - t[:, 2]).mean().item()), 'epilogue_ticks_max': (t[:, 3] - t[:, 2]).max().item(),
                    'first_round_done_ticks_mean': round((t[:, 3] - t0).mean().item())})
    print(json.dumps(out), flush=True)
dg.set_forced_config('auto')
