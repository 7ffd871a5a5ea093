#!/usr/bin/env python3
"""Mid-M dense shapes (64 < m <= 256) COLD (rotation larger than the Infinity Cache): the automatic pick against the K-split stream tile
(stream_ks_64x128, round 6).    python tools/m128_ks_ab.py [MxNxK ...]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg
from deepgemm_amd import _lib
from deepgemm_amd.testing import generators as gen

shapes = [tuple(int(x) for x in s.split('x')) for s in sys.argv[1:]] or [(128, 4096, 7168), (128, 7168, 2048), (128, 2112, 7168), (256, 4096, 7168), (128, 24576, 1536), (128, 7168, 16384),
                                                                           (192, 4096, 7168), (96, 4096, 7168), (128, 576, 7168), (128, 32768, 512)]
for m, n, k in shapes:
    sets = max(2, int(-(-320e6 // (n * k))))
    cases = []
    for i in range(sets):
        gen.reset_seed(i)
        c = gen.generate_normal(m, n, k)
        c.a_bf16 = c.b_bf16 = None
        cases.append(((c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1])), c.b, c.d))
    out, names = {}, {}
    for rnd in range(3):
        for cfg in ('auto', 'stream_ks_64x128', 'stream_ks_64x128@2'):
            os.environ['DG_STREAM_KS_PIECES'] = cfg.split('@')[1] if '@' in cfg else '8'
            _lib.lib.dg_reload_env()
            dg.set_forced_config(cfg.split('@')[0])
            for it in range(3 * sets):
                a, b, d = cases[it % sets]
                dg.fp8_gemm_nt(a, b, d)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10 * sets
            s.record()
            for it in range(reps):
                a, b, d = cases[it % sets]
                dg.fp8_gemm_nt(a, b, d)
            e.record()
            torch.cuda.synchronize()
            out.setdefault(cfg, []).append(round(s.elapsed_time(e) / reps * 1e3, 2))
            names[cfg] = dg.last_config()
    dg.set_forced_config('auto')
    print(json.dumps({'shape': f'{m}x{n}x{k}', 'sets': sets, 'auto_kernel': names['auto'], 'us_per_call': out}), flush=True)
