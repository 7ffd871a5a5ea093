#!/usr/bin/env python3
"""Packed-UE8M0 scales with MN-major operands (nn / tt / tn layouts): read in place (e8_duo_bmn / _amn / _abmn_256x256, the automatic choice
where it pays) against the re-majoring pass(es) + K-major quad kernel (forced e8_quad_*), and against the FP32-scale call of the same shape
and layout.  One line per shape, layout and arm.
    python tools/e8_mn_ab.py [MxNxK,...] [nn,tt,tn]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
from deepgemm_amd.testing import generators as gen                      # noqa: E402

shapes = sys.argv[1] if len(sys.argv) > 1 else '2048x7168x2048,4096x4096x7168,4096x7168x4096,1024x4096x7168'
layouts = sys.argv[2].split(',') if len(sys.argv) > 2 else ['nn', 'tt', 'tn']
for shape, layout in [(s_, l_) for s_ in shapes.split(',') for l_ in layouts]:
    m, n, k = (int(x) for x in shape.split('x'))
    sets = []
    for i in range(3):
        gen.reset_seed(i)
        case = gen.generate_normal(m, n, k, layout[0] == 'n', layout[1] == 't', use_ue8m0=True)
        sets.append((gen.packed_ue8m0_operand(*case.a), gen.packed_ue8m0_operand(*case.b, mn_rows=n),
                     (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1])), case.b, case.d))
    arms = [('packed, automatic', 'auto', 0), ('packed, re-majored + quad', 'e8_quad_256x256' if k % 512 == 0 and m > 128 else 'e8_quad_128x256', 0),
            ('FP32 scales', 'auto', 2)]
    for label, forced, which in arms:
        dg.set_forced_config(forced)
        def call(s):
            dg.fp8_gemm_nt(s[which], s[which + 1], s[4], disable_ue8m0_cast=True)
        for s in sets:
            call(s)
        torch.cuda.synchronize()
        best = []
        for _ in range(3):
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
            for it in range(30):
                call(sets[it % 3])
            end.record()
            torch.cuda.synchronize()
            best.append(start.elapsed_time(end) / 30 * 1e3)
        print(json.dumps({'shape': shape, 'layout': layout, 'arm': label, 'kernel': dg.last_config(), 'us_per_call': round(sorted(best)[1], 2)}), flush=True)
    dg.set_forced_config('auto')
