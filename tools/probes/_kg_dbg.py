import sys
sys.path.insert(0, '.')
import torch, deepgemm_amd as dg, oracle
from deepgemm_amd.testing import generators as gen
for (G, m, n, ks) in ((2, 200, 264, [128, 384]), (1, 256, 256, [256]), (1, 256, 256, [128])):
    gen.reset_seed(1)
    gran_k = 128
    case = gen.generate_k_grouped_contiguous_ue8m0(G, m, n, ks, gran_k)
    a = (case.a[0], gen.pack_k_grouped_ue8m0(case.a[1], ks, gran_k)); b = (case.b[0], gen.pack_k_grouped_ue8m0(case.b[1], ks, gran_k))
    d = case.c.clone()
    dg.k_grouped_fp8_gemm_tn_contiguous(a, b, d, ks, case.grouped_layout, c=d, recipe=(1, 1, gran_k))
    print(dg.last_config())
    for g in range(G):
        (a_g, sfa_g), (b_g, sfb_g) = case.a_groups[g], case.b_groups[g]
        want = torch.empty((m, n), dtype=torch.float)
        oracle.fp8_gemm_nt(a_g.cpu(), sfa_g.cpu(), b_g.cpu(), sfb_g.cpu(), want, c=case.c[g].cpu(), gran_n=1, gran_k=gran_k)
        err = (d[g].cpu() - want).abs()
        bad = err > 1e-2
        print('group', g, 'bad', int(bad.sum()), 'of', bad.numel(), 'max err', float(err.max()))
        if bad.any():
            rows = bad.any(dim=1).nonzero().flatten().tolist(); cols = bad.any(dim=0).nonzero().flatten().tolist()
            print('  rows', rows[:40], len(rows)); print('  cols', cols[:40], len(cols))
            r, c = bad.nonzero()[0].tolist()
            print('  first', r, c, float(d[g][r, c]), float(want[r, c]), float(case.c[g][r, c]))
c = gen.generate_normal(4096, 4096, 7168, accumulate=True, out_dtype=torch.float, per_token_b=True, use_ue8m0=True)
a, b = gen.packed_ue8m0_operand(*c.a), gen.packed_ue8m0_operand(*c.b)
dg.fp8_gemm_nt(a, b, c.d, c=c.d, recipe=(1, 1, 128)); print('dense wgrad config', dg.last_config())
