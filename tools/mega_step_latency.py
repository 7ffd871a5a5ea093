#!/usr/bin/env python3
"""One decode step of `fp8_mega_moe` on ONE rank (one GPU): the in-kernel dispatch / combine over the (self-)mapped symmetric region -- five launches --
against the scatter / gather path of world size 1, eager wall time per step and device time.  What it shows: the launch count / host cost of the two step
forms; what it cannot show: xGMI (no node).    python tools/mega_step_latency.py [tokens ...]"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg
from deepgemm_amd import mega
from deepgemm_amd.utils.math import per_block_cast_to_fp8, per_token_cast_to_fp8

E, topk, hidden, inter = 8, 4, 7168, 2048
g = torch.Generator(device='cuda').manual_seed(1)
w1 = torch.randn((E, 2 * inter, hidden), dtype=torch.bfloat16, device='cuda', generator=g) / hidden ** 0.5
w2 = torch.randn((E, hidden, inter), dtype=torch.bfloat16, device='cuda', generator=g) / inter ** 0.5
q1 = [per_block_cast_to_fp8(w1[e], use_ue8m0=False) for e in range(E)]
q2 = [per_block_cast_to_fp8(w2[e], use_ue8m0=False) for e in range(E)]
l1 = (torch.stack([q[0] for q in q1]), torch.stack([q[1] for q in q1]))
l2 = (torch.stack([q[0] for q in q2]), torch.stack([q[1] for q in q2]))
del w1, w2
l1, l2 = dg.transform_weights_for_mega_moe(l1, l2)
for tokens in [int(t) for t in sys.argv[1:]] or [16, 64]:
    max_tokens = 64
    x = per_token_cast_to_fp8(torch.randn((tokens, hidden), dtype=torch.bfloat16, device='cuda', generator=g), use_ue8m0=False)
    w, idx = torch.topk(torch.rand((tokens, E), device='cuda', generator=g), topk, dim=1)
    row = {'tokens': tokens, 'experts': E, 'topk': topk, 'hidden': hidden, 'intermediate': inter}
    outs = {}
    for form in ('scatter_gather', 'p2p'):
        buf = mega.SymmBuffer(None, E, max_tokens, topk, hidden, inter, p2p=(form == 'p2p'))
        buf.x[:tokens].copy_(x[0]); buf.x_sf[:tokens].copy_(x[1]); buf.topk_idx[:tokens].copy_(idx); buf.topk_weights[:tokens].copy_(w.float())
        y = torch.empty((tokens, hidden), dtype=torch.bfloat16, device='cuda')
        for _ in range(10):
            dg.fp8_mega_moe(y, l1, l2, buf)
        torch.cuda.synchronize()
        reps = 100
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        s.record()
        for _ in range(reps):
            dg.fp8_mega_moe(y, l1, l2, buf)
        e.record()
        host_us = (time.perf_counter() - t0) / reps * 1e6            # (time to ENQUEUE a step)
        torch.cuda.synchronize()
        row[form] = {'device_us_per_step': round(s.elapsed_time(e) / reps * 1e3, 1), 'host_enqueue_us_per_step': round(host_us, 1)}
        outs[form] = y.clone()
        if form == 'p2p':
            # phase stamps of one step (100 MHz wall clock; dispatch: entry, claims back, stores issued, acknowledged, last workgroup, peers arrived, end;
            # combine: entry, stores issued, acknowledged, last workgroup, end; reduce: entry, flags seen, end) -- relative to the dispatch entry, us
            from deepgemm_amd._lib import lib
            dbg = torch.zeros(65536 + 64, dtype=torch.int64, device='cuda')
            lib.dg_set_debug_buffer(dbg.data_ptr())
            dg.fp8_mega_moe(y, l1, l2, buf)
            torch.cuda.synchronize()
            lib.dg_set_debug_buffer(None)
            st = dbg[65536:65536 + 24].cpu().view(3, 8)
            t0 = int(st[0, 0])
            row['p2p_phase_us'] = {k: [round((int(v) - t0) / 100.0, 2) for v in st[i] if int(v) != 0] for i, k in enumerate(('dispatch', 'combine', 'reduce'))}
        buf.destroy()
    row['same_bits'] = bool(torch.equal(outs['p2p'].view(torch.int16), outs['scatter_gather'].view(torch.int16)))
    print(json.dumps(row), flush=True)
