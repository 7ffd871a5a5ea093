"""Decode-size expert MLP: fused (GEMM1 + SwiGLU + per-token FP8 re-quantisation in one launch, then GEMM2) against the unfused pipeline.
Prints the bench.py secondary record of 'expert_mlp' plus the per-launch split (GEMM1 fused / GEMM2 / unfused GEMM1)."""
import importlib.util
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location('bench_module', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
import deepgemm_amd as dg          # noqa: E402

bench.SECONDARY = ['expert_mlp']
print(json.dumps(bench.run_secondary(2), indent=1))

# the split: each launch alone as a graph replay, rotating over two weight sets
groups, m_max, hidden, inter = 8, 64, 7168, 2048
calls = bench.make_workload('expert_mlp', 2)[0]
sets = [c.__defaults__ for c in calls]                         # (x, y, masked, mid, w1_t, w2_t)
plain = [c.__defaults__ for c in bench.make_workload('expert_mlp_unfused', 2)[0]]     # (x, y, masked, h, w1, w2)
ws = dg.mega.swiglu_workspace(groups, m_max, 2 * inter, 'cuda')        # caller-owned: the library never allocates one under graph capture
for label, fns in (('gemm1_fused', [lambda s=s: dg.m_grouped_fp8_gemm_nt_masked_swiglu(s[0], s[4], s[3], s[2], 48, workspace=ws) for s in sets]),
                   ('gemm1_plain', [lambda s=s: dg.m_grouped_fp8_gemm_nt_masked(s[0], s[4], s[3], s[2], 48) for s in plain]),
                   ('gemm2', [lambda s=s: dg.m_grouped_fp8_gemm_nt_masked(s[3], s[5], s[1], s[2], 48) for s in sets])):
    fns[0]()
    cfg = dg.last_config()
    us = bench.graph_replay_seconds(fns, 20) * 1e6
    nbytes = groups * (2 * inter * hidden if label != 'gemm2' else inter * hidden)
    print(f'{label:12s} {us:7.2f} us  {cfg:24s} weight stream {nbytes / us / 1e3:7.1f} GB/s')
