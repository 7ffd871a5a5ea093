#!/usr/bin/env python3
"""Per-workgroup END-time distribution of a C2 launch (one 256 x 256 tile per CU: the launch ends at the slowest CU) from the per-wave
s_memtime stamps -- VERDICT round 5, item 3 (i): if p100 - p50 > 3 us, K-range stealing inside an XCD would pay.  Per XCD (block b runs on
the shader-clock counter of the ordinary stamps is per CU and unsynchronised -- profiles/r06_probe/stamp_counters_are_per_cu.log -- so this needs the
-DDG_STAMP_REALTIME build, whose stamps are the chip-wide 100 MHz counter): the end of a workgroup = the last wave's stamp behind its output stores,
relative to the launch's first entry.
    DG_VARIANT=rt DG_VARIANT_FLAGS=-DDG_STAMP_REALTIME python tools/c2_end_time_histogram.py [--config duo_p_256x256] [--shape 4096x4096x7168]"""
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
ap.add_argument('--config', default='duo_p_256x256')
ap.add_argument('--shape', default='4096x4096x7168')
ap.add_argument('--reps', type=int, default=6)
ap.add_argument('--wgrad', action='store_true', help='recipe (1, 1, 128) with FP32 accumulation into D (pipe_pc kernels); --config auto')
args = ap.parse_args()
m, n, k = (int(x) for x in args.shape.split('x'))
e8 = args.config.startswith('e8_')
cases = []
for i in range(4):
    gen.reset_seed(i)
    if args.wgrad:
        c = gen.generate_normal(m, n, k, accumulate=True, out_dtype=torch.float, per_token_b=True)
        cases.append((((c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1])), (c.b[0], dg.get_mn_major_tma_aligned_tensor(c.b[1]))), c.d))
        continue
    c = gen.generate_normal(m, n, k, use_ue8m0=e8)
    cases.append(((gen.packed_ue8m0_operand(*c.a), gen.packed_ue8m0_operand(*c.b, mn_rows=n)) if e8 else
                  ((c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1])), c.b), c.d))
dbg = torch.zeros(4096 * 8 * 4, dtype=torch.int64, device='cuda')
dg.set_forced_config(args.config)
call = (lambda i: dg.fp8_gemm_nt(cases[i % 4][0][0], cases[i % 4][0][1], cases[i % 4][1], c=cases[i % 4][1], recipe=(1, 1, 128))) if args.wgrad else \
       (lambda i: dg.fp8_gemm_nt(cases[i % 4][0][0], cases[i % 4][0][1], cases[i % 4][1]))       # noqa: E731
for it in range(60):                                                    # clocks up
    call(it)
torch.cuda.synchronize()
waves = 4 if 'quad' in args.config else 8
blocks = min(256, -(-m // 256) * -(-n // 256))
for rep in range(args.reps):
    for it in range(8):
        call(it)
    lib.dg_set_debug_buffer(dbg.data_ptr())
    dbg.zero_()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    call(rep)
    e.record()
    torch.cuda.synchronize()
    lib.dg_set_debug_buffer(None)
    wall_us = s.elapsed_time(e) * 1e3
    t = dbg[:blocks * waves * 4].view(blocks, waves, 4).cpu().double()
    assert os.environ.get('DG_VARIANT'), 'needs the -DDG_STAMP_REALTIME build (see the docstring)'
    t0 = t[:, :, 0].min()
    ends_us = ((t[:, :, 3].amax(dim=1) - t0) / 100.0).tolist()                   # 100 MHz
    loop_end_us = ((t[:, :, 2].amax(dim=1) - t0) / 100.0).tolist()
    epi_us = ((t[:, :, 3].amax(dim=1) - t[:, :, 2].amax(dim=1)) / 100.0).tolist()
    entry_us = ((t[:, :, 0].amin(dim=1) - t0) / 100.0).tolist()
    q = lambda v, p: round(sorted(v)[min(len(v) - 1, int(p * len(v)))], 2)          # noqa: E731
    print(json.dumps({'config': args.config, 'shape': args.shape, 'rep': rep, 'event_us_single_launch': round(wall_us, 1),
                      'end_us_p0_p50_p90_p100': [q(ends_us, 0), q(ends_us, .5), q(ends_us, .9), q(ends_us, 1.0)],
                      'p100_minus_p50_us': round(q(ends_us, 1.0) - q(ends_us, .5), 2),
                      'entry_us_p50_p100': [q(entry_us, .5), q(entry_us, 1.0)],
                      'k_loop_end_us_p0_p50_p100': [q(loop_end_us, 0), q(loop_end_us, .5), q(loop_end_us, 1.0)],
                      'epilogue_us_p0_p50_p100': [q(epi_us, 0), q(epi_us, .5), q(epi_us, 1.0)]}), flush=True)
dg.set_forced_config('auto')
