#!/usr/bin/env python3
"""Tuning sweep: times every kernel configuration on a list of dense shapes (HIP events, interleaved rounds so that
within-run A/B deltas are meaningful) and checks each against the reference gate.  Writes JSON lines.

    python tools/sweep.py [--shapes 4096x4096x7168,...] [--configs a,b,...] [--rounds 5] [--iters 20] [--out file]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
from deepgemm_amd.testing import calc_diff, generators as gen          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shapes', default='4096x4096x7168')
    ap.add_argument('--configs', default='')
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--out', default='')
    ap.add_argument('--sets', type=int, default=4, help='rotating input sets (1 = operands stay in the Infinity Cache)')
    ap.add_argument('--pad-a', type=int, default=0, help='extra bytes per row of A (row pitch k + pad: a sub-view of a wider tensor)')
    ap.add_argument('--pad-b', type=int, default=0, help='extra bytes per row of B')
    args = ap.parse_args()
    configs = args.configs.split(',') if args.configs else [c for c in dg.list_configs() if not c.startswith('generic')]
    out = open(args.out, 'w') if args.out else None
    flush = torch.empty(int(512e6) // 4, dtype=torch.int, device='cuda')
    for shape in args.shapes.split(','):
        m, n, k = (int(x) for x in shape.split('x'))
        cases = []
        for i in range(args.sets):
            gen.reset_seed(i)
            c_ = gen.generate_normal(m, n, k)
            c_.a = (c_.a[0], dg.get_mn_major_tma_aligned_tensor(c_.a[1]))
            if args.pad_a:      # the same bytes at a different row pitch (L2 / HBM channel mapping experiments)
                wide = torch.zeros((m, k + args.pad_a), dtype=torch.uint8, device='cuda')
                wide[:, :k] = c_.a[0].view(torch.uint8)
                c_.a = (wide[:, :k].view(torch.float8_e4m3fn), c_.a[1])
            if args.pad_b:
                wide = torch.zeros((n, k + args.pad_b), dtype=torch.uint8, device='cuda')
                wide[:, :k] = c_.b[0].view(torch.uint8)
                c_.b = (wide[:, :k].view(torch.float8_e4m3fn), c_.b[1])
            cases.append(c_)
        case = cases[0]
        times = {c: [] for c in configs}
        diffs = {}
        for c in configs:
            dg.set_forced_config(c)
            try:
                case.d.zero_()
                dg.fp8_gemm_nt(case.a, case.b, case.d)
                torch.cuda.synchronize()
                diffs[c] = calc_diff(case.d, case.ref_d)
            except RuntimeError as e:
                diffs[c] = f'error: {e}'
        for _ in range(args.rounds):
            for c in configs:
                if isinstance(diffs[c], str):
                    continue
                dg.set_forced_config(c)
                flush.zero_()
                start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                start.record()
                for it in range(args.iters):
                    cc = cases[it % len(cases)]
                    dg.fp8_gemm_nt(cc.a, cc.b, cc.d)
                end.record()
                torch.cuda.synchronize()
                times[c].append(start.elapsed_time(end) / args.iters * 1e3)
        for c in configs:
            if isinstance(diffs[c], str):
                rec = {'shape': shape, 'config': c, 'error': diffs[c]}
            else:
                ts = sorted(times[c])
                med, best = ts[len(ts) // 2], ts[0]
                rec = {'shape': shape, 'config': c, 'us_median': round(med, 2), 'us_min': round(best, 2),
                       'tflops_median': round(2.0 * m * n * k / med / 1e6, 1), 'tflops_best': round(2.0 * m * n * k / best / 1e6, 1),
                       'calc_diff': diffs[c], 'ok': bool(diffs[c] < 1e-3)}
            line = json.dumps(rec)
            print(line, flush=True)
            if out:
                out.write(line + '\n')
    dg.set_forced_config('auto')


if __name__ == '__main__':
    main()
