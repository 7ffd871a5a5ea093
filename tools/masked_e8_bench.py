#!/usr/bin/env python3
"""m_grouped_fp8_gemm_nt_masked with packed UE8M0 scales next to the FP32-scale call on the same decode-sized problems
(BASELINE configs[4]'s per-rank shape and the reference's small-M masked sweep entries): kernel picked, microseconds, GB/s.
    python tools/masked_e8_bench.py [configs for the packed call, comma separated]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
from deepgemm_amd.testing import generators as gen                      # noqa: E402

configs = sys.argv[1].split(',') if len(sys.argv) > 1 else ['auto']


def timed(fn):
    t_end = time.time() + 0.2
    while time.time() < t_end:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
    bursts = []
    for _ in range(5):
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(10):
            fn()
        end.record()
        torch.cuda.synchronize()
        bursts.append(start.elapsed_time(end) / 10 * 1e3)
    return sorted(bursts)[2]


for groups, max_m, expected, n, k in ((8, 64, 48, 4096, 7168), (32, 4096, 20, 4096, 4096), (32, 4096, 20, 6144, 7168), (6, 4096, 20, 7168, 3072)):
    gen.reset_seed(0)
    case = gen.generate_m_grouped_masked(groups, max_m, expected, n, k, use_ue8m0=True)
    valid = int(case.masked_m.sum())
    nbytes = valid * k + groups * n * k + valid * n * 2
    a32 = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1]))
    dg.set_forced_config('auto')
    us = timed(lambda: dg.m_grouped_fp8_gemm_nt_masked(a32, case.b, case.d, case.masked_m, expected))
    print(json.dumps({'groups': groups, 'expected_m': expected, 'n': n, 'k': k, 'scales': 'fp32', 'kernel': dg.last_config(),
                      'us': round(us, 1), 'gbs': round(nbytes / us / 1e3, 1)}), flush=True)
    a8, b8 = gen.packed_ue8m0_operand(*case.a), gen.packed_ue8m0_operand(*case.b, mn_rows=n)
    for cfg in configs:
        dg.set_forced_config(cfg)
        try:
            us = timed(lambda: dg.m_grouped_fp8_gemm_nt_masked(a8, b8, case.d, case.masked_m, expected))
        except RuntimeError as e:
            print(json.dumps({'config': cfg, 'error': str(e)[:100]}))
            continue
        print(json.dumps({'groups': groups, 'expected_m': expected, 'n': n, 'k': k, 'scales': 'packed ue8m0', 'kernel': dg.last_config(),
                          'us': round(us, 1), 'gbs': round(nbytes / us / 1e3, 1)}), flush=True)
    del case, a32, a8, b8
dg.set_forced_config('auto')
