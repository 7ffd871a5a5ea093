#!/usr/bin/env python3
"""Host cost of a cached dense call (plan cache hit, ctypes, hipLaunchKernel): enqueue time per call and a cProfile of the Python side.
Round 6 on the GPU box: 7.7 us per call, 6.6 of them inside the one ctypes call.   python tools/host_overhead_profile.py [MxNxK]"""
import sys, time, cProfile, pstats, io
sys.path.insert(0, '.')
import torch, deepgemm_amd as dg
from deepgemm_amd.testing import generators as gen
gen.reset_seed(0)
shape = tuple(int(v) for v in sys.argv[1].split("x")) if len(sys.argv) > 1 else (1, 576, 7168)
c = gen.generate_normal(*shape)
a = (c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1]))
for _ in range(100): dg.fp8_gemm_nt(a, c.b, c.d)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): dg.fp8_gemm_nt(a, c.b, c.d)                    # (few enough that the launch queue never fills: pure host time)
t1 = time.perf_counter()
torch.cuda.synchronize()
print('enqueue us per call (300 calls into an empty queue)', (t1 - t0) / 300 * 1e6)
t0 = time.perf_counter()
for _ in range(5000): dg.fp8_gemm_nt(a, c.b, c.d)
t1 = time.perf_counter()
torch.cuda.synchronize()
print('enqueue us per call (5000 calls: the larger of host time and kernel time)', (t1 - t0) / 5000 * 1e6)
pr = cProfile.Profile(); pr.enable()
for _ in range(5000): dg.fp8_gemm_nt(a, c.b, c.d)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14); print(s.getvalue()[:3000])
