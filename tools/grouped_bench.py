#!/usr/bin/env python3
"""M-grouped contiguous GEMM timing over (groups, expected M per group, N, K) for a list of kernel configurations.
    python tools/grouped_bench.py --cases 4x8192x4096x7168,8x4096x4096x7168 --configs auto,pipe_128x128,duo_256x256"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
from deepgemm_amd.testing import calc_diff, generators as gen          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--cases', default='8x512x4096x7168')
ap.add_argument('--configs', default='auto')
ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--ms', default='', help='explicit rows per group, e.g. 512,512,640 (the case then reads GxIGNOREDxNxK)')
ap.add_argument('--nn', action='store_true', help='B stored MN-major ([G, K, N]): the nn form')
args = ap.parse_args()
for case_s in args.cases.split(','):
    g, em, n, k = (int(x) for x in case_s.split('x'))
    gen.reset_seed(0)
    case = gen.generate_m_grouped_contiguous(g, em, n, k, b_k_major=not args.nn,
                                             actual_ms=[int(x) for x in args.ms.split(',')] if args.ms else None)
    case.a = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1]))
    for cfg in args.configs.split(','):
        dg.set_forced_config(cfg)
        try:
            for _ in range(2):
                dg.m_grouped_fp8_gemm_nt_contiguous(case.a, case.b, case.d, case.grouped_layout)
            torch.cuda.synchronize()
        except RuntimeError as e:
            print(json.dumps({'case': case_s, 'config': cfg, 'error': str(e)[:120]}), flush=True)
            continue
        diff = calc_diff(torch.nan_to_num(case.d), torch.nan_to_num(case.ref_d))
        # warm the clocks (the first launches after an idle gap run at a ramping clock), then the median of 5 timed bursts
        import time
        t_end = time.time() + 0.3
        while time.time() < t_end:
            for _ in range(4):
                dg.m_grouped_fp8_gemm_nt_contiguous(case.a, case.b, case.d, case.grouped_layout)
            torch.cuda.synchronize()
        bursts = []
        for _ in range(5):
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
            for _ in range(args.iters):
                dg.m_grouped_fp8_gemm_nt_contiguous(case.a, case.b, case.d, case.grouped_layout)
            end.record()
            torch.cuda.synchronize()
            bursts.append(start.elapsed_time(end) / args.iters * 1e3)
        us = sorted(bursts)[2]
        print(json.dumps({'case': case_s, 'm_total': case.m, 'config': cfg, 'kernel': dg.last_config(), 'us': round(us, 1),
                          'tflops': round(2.0 * case.m * n * k / us / 1e6, 1), 'calc_diff': diff}), flush=True)
dg.set_forced_config('auto')
