#!/usr/bin/env python3
"""Decode-batch dense GEMMs (M <= 32) of the reference sweep: the skinny weight-stream kernel against the stream tiles it replaces.
One JSON line per (m, n, k): oracle parity of the automatic pick, microseconds of the whole operator call per configuration."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
import oracle                                                           # noqa: E402
from deepgemm_amd.testing import calc_diff, generators as gen           # noqa: E402


def timed(fn, iters=20):
    t_end = time.time() + 0.2
    while time.time() < t_end:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
    bursts = []
    for _ in range(5):
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(iters):
            fn()
        end.record()
        torch.cuda.synchronize()
        bursts.append(start.elapsed_time(end) / iters * 1e3)
    return sorted(bursts)[2]


ms = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '1,16,32').split(',')]
for m in ms:
    for n, k in gen.DENSE_NK:
        gen.reset_seed(0)
        case = gen.generate_normal(m, n, k)
        a = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1]))
        rec = {'m': m, 'n': n, 'k': k}
        dg.set_forced_config('auto')
        case.d.fill_(float('nan'))
        dg.fp8_gemm_nt(a, case.b, case.d)
        rec['auto'] = dg.last_config()
        want = torch.empty((m, n), dtype=torch.bfloat16)
        oracle.fp8_gemm_nt(case.a[0].cpu(), case.a[1].cpu(), case.b[0].cpu(), case.b[1].cpu(), want) if m <= 32 else want.copy_(case.ref_d)
        rec['calc_diff_vs_oracle'] = calc_diff(case.d.cpu(), want)
        rec['max_abs_err_over_rms'] = float((case.d.cpu().float() - want.float()).abs().max() / want.float().pow(2).mean().sqrt())
        for cfg in (sys.argv[2].split(',') if len(sys.argv) > 2 else ('auto', 'stream_64x32', 'stream_64x128')):
            dg.set_forced_config(cfg)
            try:
                rec[f'us_{cfg}'] = round(timed(lambda: dg.fp8_gemm_nt(a, case.b, case.d)), 1)
            except RuntimeError as e:
                rec[f'us_{cfg}'] = str(e)[:40]
        nbytes = m * k + n * k + 2 * m * n
        rec['frac_hbm_auto'] = round(nbytes / rec['us_auto'] / 1e3 / 8000, 3)
        print(json.dumps(rec), flush=True)
dg.set_forced_config('auto')
