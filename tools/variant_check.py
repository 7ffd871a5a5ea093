#!/usr/bin/env python3
"""Bit-compares the output of kernel variants with the production configuration on one dense shape (same arithmetic
order => identical bits), several launches each.   python tools/variant_check.py cfg_a,cfg_b [MxNxK] [baseline]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
from deepgemm_amd.testing import generators as gen                      # noqa: E402

cfgs = sys.argv[1].split(',')
m, n, k = (int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else '4096x4096x7168').split('x'))
baseline = sys.argv[3] if len(sys.argv) > 3 else 'duo_256x256'
gen.reset_seed(0)
case = gen.generate_normal(m, n, k)
dg.set_forced_config(baseline)
dg.fp8_gemm_nt(case.a, case.b, case.d)
want = case.d.clone()
for cfg in cfgs:
    dg.set_forced_config(cfg)
    bad = 0
    for _ in range(6):
        d = torch.empty_like(want)
        dg.fp8_gemm_nt(case.a, case.b, d)
        bad += int(not torch.equal(d, want))
    print(cfg, 'identical to', baseline, 'in', 6 - bad, 'of 6 launches')
dg.set_forced_config('auto')
