#!/usr/bin/env python3
"""Packed-UE8M0 (hardware-scaled) kernels on one dense shape: bit-comparison against the first configuration, sustained time per
launch (bursts, round-robin over the configurations) and in-kernel s_memtime accounting (prologue / K loop / epilogue ticks).
    python tools/e8_sweep.py cfg_a,cfg_b,... [MxNxK] [launches per burst] [rounds]
Configurations: e8_quad_256x256, e8_duo_256x256, e8_ring_256x256 (and e8_quad_v1, v2, v3, v5 in DG_EXPERIMENTS builds:
timing ablations whose results are garbage)."""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                                                     # noqa: E402
from deepgemm_amd._lib import lib                                                             # noqa: E402
from deepgemm_amd.utils.math import per_block_cast_to_fp8, per_token_cast_to_fp8              # noqa: E402

cfgs = sys.argv[1].split(',')
m, n, k = (int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else '4096x4096x7168').split('x'))
burst = int(sys.argv[3]) if len(sys.argv) > 3 else 200
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
GARBAGE = {'e8_quad_v1', 'e8_quad_v2', 'e8_quad_v3', 'e8_quad_v5'}

sets = []
for i in range(4):
    torch.manual_seed(i)
    a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16)
    b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
    a_q, sfa = per_token_cast_to_fp8(a, use_ue8m0=True)
    b_q, sfb = per_block_cast_to_fp8(b, use_ue8m0=True)
    pa = dg.get_mn_major_tma_aligned_packed_ue8m0_tensor(sfa)
    pb = dg.get_mn_major_tma_aligned_packed_ue8m0_tensor(sfb.repeat_interleave(128, dim=0)[:n].contiguous())
    sets.append((a_q, pa, b_q, pb, torch.empty((m, n), device='cuda', dtype=torch.bfloat16)))
    del a, b


def call(s, d=None):
    dg.fp8_gemm_nt((s[0], s[1]), (s[2], s[3]), s[4] if d is None else d)


# bit comparison against the first configuration (accumulation inside the matrix core in the same K order everywhere)
want = None
for cfg in cfgs:
    dg.set_forced_config(cfg)
    d = torch.full((m, n), float('nan'), device='cuda', dtype=torch.bfloat16)
    call(sets[0], d)
    torch.cuda.synchronize()
    if cfg in GARBAGE:
        continue
    if want is None:
        want = d
    else:
        same = torch.equal(d, want)
        print(json.dumps({'config': cfg, 'bit_equal_to': cfgs[0], 'equal': same,
                          'max_abs_diff': (d.float() - want.float()).abs().max().item()}), flush=True)

t_end = time.time() + 1.5
dg.set_forced_config(cfgs[0])
while time.time() < t_end:
    for s in sets:
        call(s)
    torch.cuda.synchronize()

times = {cfg: [] for cfg in cfgs}
for r in range(rounds):
    for cfg in cfgs:
        dg.set_forced_config(cfg)
        for it in range(20):
            call(sets[it % 4])
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for it in range(burst):
            call(sets[it % 4])
        end.record()
        torch.cuda.synchronize()
        times[cfg].append(start.elapsed_time(end) / burst * 1e3)

tiles = -(-m // 256) * -(-n // 256)
blocks = min(tiles, 256)
for cfg in cfgs:
    dg.set_forced_config(cfg)
    waves = 4 if 'quad' in cfg else 8
    dbg = torch.zeros(blocks * waves * 4 + 64, dtype=torch.int64, device='cuda')
    lib.dg_set_debug_buffer(dbg.data_ptr())
    for it in range(6):
        call(sets[it % 4])
    torch.cuda.synchronize()
    lib.dg_set_debug_buffer(None)
    t = dbg[:blocks * waves * 4].view(blocks * waves, 4).cpu().double()
    loop = t[:, 2] - t[:, 1]
    us = statistics.median(times[cfg])
    print(json.dumps({'config': cfg, 'shape': f'{m}x{n}x{k}', 'us_per_launch': [round(x, 1) for x in times[cfg]], 'us_median': round(us, 1),
                      'tflops': round(2.0 * m * n * k / us / 1e6, 1), 'frac_of_5pf': round(2.0 * m * n * k / us / 1e6 / 5000, 3),
                      'ticks_per_kblock': round(loop.mean().item() / (k // 128), 1), 'loop_ticks_max': loop.max().item(),
                      'prologue_ticks': round((t[:, 1] - t[:, 0]).mean().item()),
                      'epilogue_ticks': round((t[:, 3] - t[:, 2]).mean().item())}), flush=True)
dg.set_forced_config('auto')
