import sys, time, torch
sys.path.insert(0, '/root/repo')
import deepgemm_amd as dg
from deepgemm_amd.testing import generators as gen
gen.reset_seed(0)
cases = []
for i in range(4):
    c = gen.generate_m_grouped_masked(8, 64, 48, 4096, 7168)
    c.a = (c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1]))
    cases.append(c)
for rnd in range(3):
    for cfg in ('stream_64x128', 'stream_nt_64x128'):
        dg.set_forced_config(cfg)
        for i in range(40):
            c = cases[i % 4]; dg.m_grouped_fp8_gemm_nt_masked(c.a, c.b, c.d, c.masked_m, 48)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(100):
            c = cases[i % 4]; dg.m_grouped_fp8_gemm_nt_masked(c.a, c.b, c.d, c.masked_m, 48)
        e.record(); torch.cuda.synchronize()
        print(cfg, round(s.elapsed_time(e) * 10, 2), 'us')
