#!/usr/bin/env python3
"""Random dense problems in and around the rules of the stream tiles cut along K inside the kernel (17 .. 256 rows, K = 4096 .. 16384, narrow and
wide layers; FP32 scales and packed UE8M0 words of both granularities) through the plain entry: every result against the FP64 statement of the
exactly scaled operands on the device (calc_diff: a lost or doubled K piece shows as >= 1e-3; BF16 rounding alone is ~6e-7), twice (bit-repeatable,
dirty workspace), a third time with the unsplit tile forced by name.  Prints the histogram of automatic picks.
python tools/fuzz_k_split.py [first_seed] [count]"""
import sys, os, random, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, deepgemm_amd as dg
from deepgemm_amd.testing import calc_diff, generators as gen
from deepgemm_amd.utils.math import pack_ue8m0_to_int, per_token_cast_to_fp8

first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 90
picks, bad = collections.Counter(), 0
for seed in range(first, first + count):
    rng = random.Random(seed)
    fmt = rng.choice(['fp32', 'packed128', 'packed32'])
    m = rng.choice([17, 24, 32, 33, 48, 64, 65, 96, 100, 128, 129, 160, 192, 200, 256])
    n = rng.choice([72, 576, 1024, 1536, 2112, 2560, 3072, 4096, 5120, 6144, 7168]) + rng.choice([0, 0, 8, 16])
    k = rng.choice([4096, 4608, 5120, 7168, 8192, 10240, 12288, 16384])
    label = f'seed {seed}: {fmt} {m} x {n} x {k}'
    try:
        if fmt == 'fp32':
            gen.reset_seed(seed)
            c = gen.generate_normal(m, n, k)
            a, b = c.a, c.b
            ad = (a[0].float().view(m, k // 128, 128) * a[1].unsqueeze(-1)).view(m, k).double()
            sfb = b[1].repeat_interleave(128, 0)[:n]
            bd = (b[0].float().view(n, k // 128, 128) * sfb.unsqueeze(-1)).view(n, k).double()
            kw, unsplit = {}, 'stream_64x32'
        else:
            gran = 128 if fmt == 'packed128' else 32
            torch.manual_seed(seed)
            x = torch.randn((m, k), device='cuda', dtype=torch.bfloat16); y = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
            qa, qb = per_token_cast_to_fp8(x, True, gran), per_token_cast_to_fp8(y, True, gran)
            a = (qa[0], dg.transform_sf_into_required_layout(pack_ue8m0_to_int(qa[1]), m, k, (1, gran)))
            b = (qb[0], dg.transform_sf_into_required_layout(pack_ue8m0_to_int(qb[1]), n, k, (1, gran)))
            ad = (qa[0].float().view(m, k // gran, gran) * qa[1].unsqueeze(-1)).view(m, k).double()
            bd = (qb[0].float().view(n, k // gran, gran) * qb[1].unsqueeze(-1)).view(n, k).double()
            kw, unsplit = dict(recipe=(1, 1, gran)), ('e8_stream_64x32' if gran == 128 else 'e8_stream_g32_64x32')
        exact = (ad @ bd.t()).float()
        outs = []
        for _ in range(2):
            d = torch.full((m, n), float('nan'), device='cuda', dtype=torch.bfloat16)
            dg.fp8_gemm_nt(a, b, d, **kw)
            outs.append(d)
        pick = dg.last_config()
        picks[pick] += 1
        dg.set_forced_config(unsplit)
        try:
            d3 = torch.full((m, n), float('nan'), device='cuda', dtype=torch.bfloat16)
            dg.fp8_gemm_nt(a, b, d3, **kw)
        finally:
            dg.set_forced_config('auto')
        e1, e3 = calc_diff(outs[0].float(), exact), calc_diff(d3.float(), exact)
        assert not bool(torch.isnan(outs[0]).any()), 'NaN left in the output'
        assert e1 < 3e-6 and e3 < 3e-6, f'{pick} {e1:.2e} / unsplit {e3:.2e}'
        assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)), f'{pick}: not repeatable'
        assert calc_diff(outs[0].float(), d3.float()) < 2e-6, f'{pick} against {unsplit}'
    except AssertionError as e:
        bad += 1
        print('FAIL', label, str(e)[:200], flush=True)
print('picks:', dict(picks))
print('done, cases:', count, 'failures:', bad)
