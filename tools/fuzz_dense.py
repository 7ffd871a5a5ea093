#!/usr/bin/env python3
"""More seeds of tests/test_gemm_gpu.py::test_dense_random_shapes_and_layouts_vs_oracle (random dense problems through the automatic
selection, each against the oracle).   python tools/fuzz_dense.py [first_seed] [count]"""
import sys, random, torch
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import tests.test_gemm_gpu as t
import deepgemm_amd as dg
bad = 0
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for seed in range(first, first + count):
    try:
        t.test_dense_random_shapes_and_layouts_vs_oracle(seed)
    except AssertionError as e:
        bad += 1
        print('FAIL seed', seed, str(e)[:300])
print('done, failures:', bad)
