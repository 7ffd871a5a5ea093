import sys, torch
sys.path.insert(0, '/root/repo')
import deepgemm_amd as dg
from deepgemm_amd.testing import generators as gen
gen.reset_seed(0)
case = gen.generate_normal(4096, 4096, 7168)
outs = {}
for cfg in ['duo_256x256', 'duo_256x256', 'duo_p_256x256', 'pipe_256x256', 'duo_256x256']:
    dg.set_forced_config(cfg)
    d = torch.empty_like(case.d)
    dg.fp8_gemm_nt(case.a, case.b, d)
    torch.cuda.synchronize()
    if 'ref' not in outs:
        outs['ref'] = d
        continue
    neq = (d != outs['ref'])
    idx = neq.nonzero()
    print(cfg, 'mismatches:', int(neq.sum()), 'rows', sorted(set((idx[:, 0] // 256).tolist()))[:8], 'cols', sorted(set((idx[:, 1] // 256).tolist()))[:8],
          'first', idx[:3].tolist(), 'maxabs', float((d.float() - outs['ref'].float()).abs().max()))
