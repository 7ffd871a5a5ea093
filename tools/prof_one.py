#!/usr/bin/env python3
"""Runs one kernel configuration on one dense shape a few times (target process for rocprofv3)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                       # noqa: E402
from deepgemm_amd.testing import generators as gen              # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--config', default='auto')
ap.add_argument('--shape', default='4096x4096x7168')
ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--sets', type=int, default=1)
args = ap.parse_args()
m, n, k = (int(x) for x in args.shape.split('x'))
cases = []
for i in range(args.sets):
    gen.reset_seed(i)
    cases.append(gen.generate_normal(m, n, k))
dg.set_forced_config(args.config)
for i in range(args.iters):
    c = cases[i % len(cases)]
    dg.fp8_gemm_nt(c.a, c.b, c.d)
torch.cuda.synchronize()
print('done', dg.last_config())
