#!/usr/bin/env python3
"""Phases of one tile of the K-grouped hardware-scaled kernel from the debug stamps (per wave: entry, K loop begin, K loop end, after the stores;
shader-clock ticks of the wave's own CU -- durations only): 8 groups of 4096 x 7168 x k.   python tools/probes/kgrouped_phase_stamps.py [k]"""
import sys, ctypes
sys.path.insert(0, '.')
import os
import torch, deepgemm_amd as dg
LAYOUT = 2 if os.environ.get('KG_MN') else 1          # KG_MN=1: the operands as [sum_k, mn] (read in place)
from deepgemm_amd._lib import lib, check, current_stream_ptr

k = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g, m, n = 8, 4096, 7168
ks = [k] * g
sum_k = sum(ks)
a = torch.randn((sum_k, m) if LAYOUT == 2 else (m, sum_k), device='cuda').to(torch.float8_e4m3fn); b = torch.randn((sum_k, n) if LAYOUT == 2 else (n, sum_k), device='cuda').to(torch.float8_e4m3fn)
sfa = torch.full((sum_k // 512, m), 0x7f7f7f7f, dtype=torch.int32, device='cuda'); sfb = torch.full((sum_k // 512, n), 0x7f7f7f7f, dtype=torch.int32, device='cuda')
d = torch.zeros((g, m, n), device='cuda')
ks_arr = (ctypes.c_int32 * g)(*ks)
def call():
    check(lib.dg_k_grouped_fp8_gemm_ue8m0(a.data_ptr(), sfa.data_ptr(), b.data_ptr(), sfb.data_ptr(), d.data_ptr(), m, n, sum_k,
                                          ctypes.cast(ks_arr, ctypes.c_void_p), None, g, 128, 128, LAYOUT, a.stride(0), b.stride(0), sfa.stride(0), sfb.stride(0),
                                          current_stream_ptr()))
for _ in range(3): call()
torch.cuda.synchronize()
blocks = g * 16 * 28
dbg = torch.zeros(blocks * 4 * 4, dtype=torch.int64, device='cuda')
lib.dg_set_debug_buffer(dbg.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); call(); e1.record(); torch.cuda.synchronize()
lib.dg_set_debug_buffer(None)
t = dbg.view(blocks * 4, 4).cpu().double()
t = t[t[:, 0] > 0]
total_us = e0.elapsed_time(e1) * 1e3
pro, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
tot = (t[:, 3] - t[:, 0])
# ticks -> us: the sum over a CU's tiles of (end - entry) fills the launch: 14 tiles per CU
scale = total_us / (tot.mean().item() * blocks / 256)
print(f'k {k}: launch {total_us:.0f} us, waves stamped {t.shape[0]}; ticks per us (assuming back-to-back tiles on every CU) {1 / scale:.0f}')
for name, v in (('prologue', pro), ('K loop', loop), ('epilogue', epi), ('tile', tot)):
    q = torch.quantile(v, torch.tensor([0.05, 0.5, 0.95], dtype=torch.double))
    print(f'  {name:9s} median {q[1].item() * scale:7.2f} us   p5 {q[0].item() * scale:7.2f}   p95 {q[2].item() * scale:7.2f}   (ticks median {q[1].item():.0f})')
