#!/usr/bin/env python3
"""In-kernel s_memtime accounting of the UE8M0 (hardware-scaled) kernel on C2, like tools/cycles.py."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg
from deepgemm_amd._lib import lib
from deepgemm_amd.utils.math import pack_ue8m0_to_int, per_block_cast_to_fp8, per_token_cast_to_fp8
m, n, k = 4096, 4096, 7168
sets = []
for i in range(4):
    torch.manual_seed(i)
    a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16); b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
    a_q, sfa = per_token_cast_to_fp8(a, use_ue8m0=True); b_q, sfb = per_block_cast_to_fp8(b, use_ue8m0=True)
    sfb_rows = sfb.repeat_interleave(128, dim=0)[:n].contiguous()
    pa = dg.get_mn_major_tma_aligned_tensor(pack_ue8m0_to_int(sfa).view(torch.float)).view(torch.int)
    pb = dg.get_mn_major_tma_aligned_tensor(pack_ue8m0_to_int(sfb_rows).view(torch.float)).view(torch.int)
    sets.append((a_q, pa, b_q, pb, torch.empty((m, n), device='cuda', dtype=torch.bfloat16)))
dbg = torch.zeros(4096 * 8 * 4, dtype=torch.int64, device='cuda')
lib.dg_set_debug_buffer(dbg.data_ptr())
if len(sys.argv) > 1:
    dg.set_forced_config(sys.argv[1])
def call(s): dg.fp8_gemm_nt((s[0], s[1]), (s[2], s[3]), s[4])
for i in range(5): call(sets[i % 4])
start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); start.record()
for i in range(20): call(sets[i % 4])
end.record(); torch.cuda.synchronize()
lib.dg_set_debug_buffer(None)
t = dbg[:2048 * 4].view(2048, 4).cpu().double()
loop = t[:, 2] - t[:, 1]
print(json.dumps({'kernel': dg.last_config(), 'wall_us': round(start.elapsed_time(end) / 20 * 1e3, 2), 'loop_ticks_mean': round(loop.mean().item()),
                  'loop_ticks_max': loop.max().item(), 'ticks_per_kblock': round(loop.mean().item() / 56, 1),
                  'prologue_ticks_mean': round((t[:, 1] - t[:, 0]).mean().item()), 'epilogue_ticks_mean': round((t[:, 3] - t[:, 2]).mean().item())}))
