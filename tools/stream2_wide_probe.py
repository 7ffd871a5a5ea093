#!/usr/bin/env python3
"""Round 5 probe: should mid-M dense shapes with 257 .. 512 tiles of 64 x 128 take the two-per-CU stream tile (DG_STREAM2_WIDE rule)?
auto (rule off) against stream2 / stream_nt2 forced; rotating input sets; with and without the split-K workspace."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
from deepgemm_amd.testing import generators as gen                      # noqa: E402

SHAPES = [(128, 24576, 1536), (256, 8192, 7168), (256, 12288, 2048), (192, 16384, 4096), (128, 32768, 512), (128, 16384, 7168), (256, 16384, 1536),
          (96, 24576, 1536), (256, 7168, 7168), (80, 32768, 2048)]
for m, n, k in SHAPES:
    cases = []
    for i in range(3):
        gen.reset_seed(i)
        c = gen.generate_normal(m, n, k)
        cases.append((c, (c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1]))))

    def burst(count):
        for i in range(count):
            c, aa = cases[i % 3]
            dg.fp8_gemm_nt(aa, c.b, c.d)
    for rep in range(2):
        for cfg in ['auto', 'stream2_64x128', 'stream_nt2_64x128', 'duo_128x256']:
            dg.set_forced_config(cfg)
            burst(30)
            torch.cuda.synchronize()
            bursts = []
            for _ in range(7):
                start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                start.record()
                burst(21)
                end.record()
                torch.cuda.synchronize()
                bursts.append(start.elapsed_time(end) / 21 * 1e3)
            if rep == 1:
                print(json.dumps({'shape': [m, n, k], 'tiles128': -(-m // 64) * (n // 128), 'config': cfg, 'kernel': dg.last_config(),
                                  'us': round(sorted(bursts)[3], 2)}), flush=True)
dg.set_forced_config('auto')
