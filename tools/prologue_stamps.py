#!/usr/bin/env python3
"""Where does a 256-row duo tile spend the time before its first MFMA?  Needs a -DDG_STAMP_ISSUE build (slot 2 of the debug stamps = the
moment before the first load of the first tile is issued):  DG_VARIANT=stamp python tools/prologue_stamps.py [MxNxK] [config]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
from deepgemm_amd._lib import lib                                       # noqa: E402
from deepgemm_amd.testing import generators as gen                      # noqa: E402

m, n, k = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '4096x4096x7168').split('x'))
cfg = sys.argv[2] if len(sys.argv) > 2 else 'duo_p_256x256'
cases = []
for i in range(4):
    gen.reset_seed(i)
    c = gen.generate_normal(m, n, k)
    c.a = (c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1]))
    cases.append(c)
dbg = torch.zeros(4096 * 8 * 4, dtype=torch.int64, device='cuda')
dg.set_forced_config(cfg)
for it in range(8):
    dg.fp8_gemm_nt(cases[it % 4].a, cases[it % 4].b, cases[it % 4].d)
torch.cuda.synchronize()
lib.dg_set_debug_buffer(dbg.data_ptr())
for rep in range(3):
    dbg.zero_()
    cc = cases[rep % 4]
    dg.fp8_gemm_nt(cc.a, cc.b, cc.d)
    torch.cuda.synchronize()
    grid = min(256, -(-m // 256) * -(-n // 256))
    t = dbg[:grid * 8 * 4].view(grid * 8, 4).cpu().double()
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    print(json.dumps({'shape': f'{m}x{n}x{k}', 'config': cfg, 'waves': int(t.shape[0]),
                      'entry_spread_cycles': round((t[:, 0].max() - t0).item()),
                      'entry_to_first_issue_median': round((t[:, 2] - t[:, 0]).median().item()), 'entry_to_first_issue_max': round((t[:, 2] - t[:, 0]).max().item()),
                      'first_issue_to_loop_median': round((t[:, 1] - t[:, 2]).median().item()),
                      'entry_to_loop_median': round((t[:, 1] - t[:, 0]).median().item()),
                      'total_max': round((t[:, 3].max() - t0).item())}))
lib.dg_set_debug_buffer(None)
