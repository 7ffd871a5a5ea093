import os, sys, json, torch
sys.path.insert(0, '/root/repo')
import deepgemm_amd as dg
from deepgemm_amd._lib import lib
from deepgemm_amd.testing import generators as gen
m, n, k = 4096, 4096, 7168
gen.reset_seed(0)
c = gen.generate_normal(m, n, k)
a = (c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1]))
dbg = torch.zeros(4096 * 8 * 4, dtype=torch.int64, device='cuda')
dg.set_forced_config('duo_p_256x256')
for _ in range(50): dg.fp8_gemm_nt(a, c.b, c.d)
torch.cuda.synchronize()
lib.dg_set_debug_buffer(dbg.data_ptr()); dbg.zero_()
dg.fp8_gemm_nt(a, c.b, c.d); torch.cuda.synchronize()
lib.dg_set_debug_buffer(None)
t = dbg[:256 * 8 * 4].view(256, 8, 4).cpu()
print('zeros', int((t == 0).sum()))
for x in range(2):
    tx = t[x::8]
    t0 = int(tx[:, :, 0].min())
    print('xcd', x, 'entry min/max', 0, int(tx[:, :, 0].max()) - t0, 'loop0', int(tx[:, :, 1].min()) - t0, int(tx[:, :, 1].max()) - t0,
          'loop1', int(tx[:, :, 2].min()) - t0, int(tx[:, :, 2].max()) - t0, 'end', int(tx[:, :, 3].min()) - t0, int(tx[:, :, 3].max()) - t0)
    print(' block0 waves', (tx[0] - t0).tolist())
    print(' block5 waves', (tx[5] - t0).tolist())
