#!/usr/bin/env python3
"""Where the layout variants of BASELINE configs[2] (2048 x 7168 x 2048) spend their time: s_memtime stamps per wave (entry,
K loop begin / end, after stores) of the GEMM kernel each layout selects, next to the whole-call HIP-event time.
One JSON line per layout.   python tools/c3_diag.py [--shape MxNxK] [--layouts nt,nn,tn,tt] [--configs auto,...]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
from deepgemm_amd._lib import lib                                       # noqa: E402
from deepgemm_amd.testing import calc_diff, generators as gen          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--shape', default='2048x7168x2048')
ap.add_argument('--layouts', default='nt,nn,tn,tt')
ap.add_argument('--configs', default='auto')
ap.add_argument('--iters', type=int, default=30)
ap.add_argument('--wgrad', action='store_true', help='recipe (1, 1, 128), FP32 accumulation into D (the wgrad form)')
args = ap.parse_args()
m, n, k = (int(x) for x in args.shape.split('x'))
dbg = torch.zeros(4096 * 8 * 4, dtype=torch.int64, device='cuda')
for layout in args.layouts.split(','):
    gen.reset_seed(0)
    if args.wgrad:
        case = gen.generate_normal(m, n, k, layout[0] == 'n', layout[1] == 't', accumulate=True, out_dtype=torch.float, per_token_b=True)
        case.b = (case.b[0], dg.get_mn_major_tma_aligned_tensor(case.b[1]))
    else:
        case = gen.generate_normal(m, n, k, layout[0] == 'n', layout[1] == 't')
    a = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1]))
    kwargs = {'c': case.d, 'recipe': (1, 1, 128)} if args.wgrad else {}
    for cfg in args.configs.split(','):
        dg.set_forced_config(cfg)
        try:
            for _ in range(5):
                dg.fp8_gemm_nt(a, case.b, case.d, **kwargs)
        except RuntimeError as e:
            print(json.dumps({'layout': layout, 'config': cfg, 'error': str(e)[:120]}), flush=True)
            continue
        torch.cuda.synchronize()
        diff = calc_diff(case.d, case.ref_d)
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(args.iters):
            dg.fp8_gemm_nt(a, case.b, case.d, **kwargs)
        end.record()
        torch.cuda.synchronize()
        us = start.elapsed_time(end) / args.iters * 1e3
        dbg.zero_()
        lib.dg_set_debug_buffer(dbg.data_ptr())
        for _ in range(3):
            dg.fp8_gemm_nt(a, case.b, case.d, **kwargs)
        torch.cuda.synchronize()
        lib.dg_set_debug_buffer(None)
        t = dbg.view(-1, 4).cpu().double()
        t = t[t[:, 0] > 0]
        out = {'layout': layout, 'config': cfg, 'kernel': dg.last_config(), 'us_per_call': round(us, 2),
               'tflops': round(2.0 * m * n * k / us / 1e6, 1), 'calc_diff': float(diff), 'waves_stamped': int(t.shape[0])}
        if t.shape[0] > 0:
            t0 = t[:, 0].min()
            out.update({'ticks_total': (t[:, 3].max() - t0).item(),
                        'prologue_ticks_mean': round((t[:, 1] - t[:, 0]).mean().item()),
                        'loop_ticks_mean': round((t[:, 2] - t[:, 1]).mean().item()),
                        'ticks_per_kblock': round((t[:, 2] - t[:, 1]).mean().item() / (k // 128), 1),
                        'epilogue_ticks_mean': round((t[:, 3] - t[:, 2]).mean().item()),
                        'epilogue_ticks_max': (t[:, 3] - t[:, 2]).max().item(),
                        'entry_skew_ticks': (t[:, 0].max() - t0).item()})
        print(json.dumps(out), flush=True)
dg.set_forced_config('auto')
