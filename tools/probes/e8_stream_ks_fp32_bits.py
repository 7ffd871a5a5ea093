import sys; sys.path.insert(0, '.')
import torch, deepgemm_amd as dg
from deepgemm_amd.utils.math import pack_ue8m0_to_int, per_token_cast_to_fp8
for gran in (128, 32):
    for (m, n, k) in ((128, 576, 7168), (192, 4096, 7168)):
        torch.manual_seed(1)
        a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16); b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
        qa, qb = per_token_cast_to_fp8(a, True, gran), per_token_cast_to_fp8(b, True, gran)
        pa = dg.transform_sf_into_required_layout(pack_ue8m0_to_int(qa[1]), m, k, (1, gran)); pb = dg.transform_sf_into_required_layout(pack_ue8m0_to_int(qb[1]), n, k, (1, gran))
        ad = (qa[0].float().view(m, k // gran, gran) * qa[1].unsqueeze(-1)).view(m, k).double()
        bd = (qb[0].float().view(n, k // gran, gran) * qb[1].unsqueeze(-1)).view(n, k).double()
        exact = ad @ bd.t()
        res = {}
        g = '_g32' if gran == 32 else ''
        for cfg in ('auto', f'e8_stream{g}_64x32', f'e8_quad{g}_128x256'):
            dg.set_forced_config(cfg)
            d = torch.full((m, n), float('nan'), device='cuda', dtype=torch.float)
            dg.fp8_gemm_nt((qa[0], pa), (qb[0], pb), d, recipe=(1, 1, gran))
            res[cfg] = (dg.last_config(), d)
            dg.set_forced_config('auto')
        base = res['auto'][1]
        for cfg, (name, d) in res.items():
            print(gran, (m, n, k), name, 'differs from auto in', int((d.view(torch.int32) != base.view(torch.int32)).sum()), 'of', m * n,
                  'max |d - exact| / |exact|max', float((d.double() - exact).abs().max() / exact.abs().max()), flush=True)
