#!/usr/bin/env python3
"""Round 5 probe, dense side of tools/stream2_probe.py: fp8_gemm_nt at small / mid M on the 64 x 128 stream tile, one workgroup per CU
(6-stage ring) against two (3-stage ring).  Median of 7 bursts of 21 calls over 3 rotating input sets; bits compared with the first config."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
from deepgemm_amd.testing import generators as gen                      # noqa: E402

SHAPES = [(64, 7168, 2048), (64, 4096, 7168), (64, 24576, 1536), (64, 32768, 512), (64, 7168, 16384), (48, 2112, 7168), (256, 7168, 2048),
          (256, 4096, 7168), (512, 4096, 7168), (192, 7168, 7168), (128, 24576, 1536), (128, 4096, 7168)]
CONFIGS = sys.argv[1].split(',') if len(sys.argv) > 1 else ['auto', 'stream_64x128', 'stream_nt_64x128', 'stream2_64x128', 'stream_nt2_64x128']
for m, n, k in SHAPES:
    cases = []
    for i in range(3):
        gen.reset_seed(i)
        c = gen.generate_normal(m, n, k)
        cases.append((c, (c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1]))))
    want = None

    def burst(count):
        for i in range(count):
            c, aa = cases[i % 3]
            dg.fp8_gemm_nt(aa, c.b, c.d)
    for rep in range(2):
        for cfg in CONFIGS:
            dg.set_forced_config(cfg)
            try:
                burst(30)
                torch.cuda.synchronize()
            except RuntimeError as e:
                print(json.dumps({'shape': [m, n, k], 'config': cfg, 'error': str(e)[:100]}), flush=True)
                continue
            out = cases[2][0].d.clone()
            if want is None:
                want = out
            bursts = []
            for _ in range(7):
                start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                start.record()
                burst(21)
                end.record()
                torch.cuda.synchronize()
                bursts.append(start.elapsed_time(end) / 21 * 1e3)
            if rep == 1:
                us = sorted(bursts)[3]
                print(json.dumps({'shape': [m, n, k], 'config': cfg, 'kernel': dg.last_config(), 'us': round(us, 2),
                                  'tbs': round(n * k / us / 1e6, 2), 'same_bits': bool(torch.equal(out, want))}), flush=True)
dg.set_forced_config('auto')
