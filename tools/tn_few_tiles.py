import sys, time, torch
sys.path.insert(0, '/root/repo')
import deepgemm_amd as dg
from deepgemm_amd.testing import generators as gen, calc_diff
for (m, n, k) in ((2048, 4096, 7168), (1024, 4096, 2048), (2048, 2048, 2048)):
    for layout in ('tn', 'tt'):
        gen.reset_seed(0)
        c = gen.generate_normal(m, n, k, False, layout[1] == 't')
        a = (c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1]))
        for _ in range(10): dg.fp8_gemm_nt(a, c.b, c.d)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50): dg.fp8_gemm_nt(a, c.b, c.d)
        e.record(); torch.cuda.synchronize()
        print(layout, m, n, k, dg.last_config(), round(s.elapsed_time(e) * 20, 1), 'us', float(calc_diff(c.d, c.ref_d)) < 1e-3)
