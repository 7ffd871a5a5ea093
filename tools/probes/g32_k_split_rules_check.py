import sys; sys.path.insert(0, '.')
import torch, deepgemm_amd as dg
from deepgemm_amd.testing import calc_diff
from deepgemm_amd.utils.math import pack_ue8m0_to_int, per_token_cast_to_fp8
def time_us(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def packed(x, mn, k, gran):
    q = per_token_cast_to_fp8(x, True, gran)
    return q[0], dg.transform_sf_into_required_layout(pack_ue8m0_to_int(q[1]), mn, k, (1, gran))
for m, n, k in ((128, 6144, 7168), (128, 7168, 16384), (96, 7168, 8192), (24, 1536, 7168), (32, 576, 7168), (128, 5120, 7168), (192, 1536, 16384), (192, 2112, 7168)):
    sets = max(3, min(32, int(320e6 // (n * k)) + 1))
    ops = []
    for i in range(sets):
        torch.manual_seed(i)
        a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16); b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
        ops.append((packed(a, m, k, 32), packed(b, n, k, 32), torch.empty((m, n), device='cuda', dtype=torch.bfloat16)))
    out, ref = [], None
    for cfg in ('auto', 'e8_stream_l8_g32_64x32', 'e8_skinny_g32_32' if m <= 32 else 'e8_stream_nt2_g32_64x128'):
        try:
            dg.set_forced_config(cfg)
            dg.fp8_gemm_nt(ops[0][0], ops[0][1], ops[0][2], recipe=(1, 1, 32))
            name = dg.last_config(); res = ops[0][2].float().clone()
            if ref is None: ref = res
            it = [0]
            def call():
                o = ops[it[0] % sets]; it[0] += 1
                dg.fp8_gemm_nt(o[0], o[1], o[2], recipe=(1, 1, 32))
            out.append(f'{cfg}{"=" + name if cfg == "auto" else ""} {time_us(call):.1f} ({calc_diff(res, ref):.1e})')
        except Exception as e:
            out.append(f'{cfg}: {str(e)[:50]}')
        finally:
            dg.set_forced_config('auto')
    print(f'gran 32: {m} x {n} x {k} ({sets} sets): ' + ' | '.join(out), flush=True)
