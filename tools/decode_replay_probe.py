#!/usr/bin/env python3
"""Why does a hipGraph replay of the m = 1 decode call run slower than the eager call it captures?
Eager and replayed calls of `decode_m1_long` (1 x 7168 x 16384) timed over the same NUMBER of back-to-back kernels."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import deepgemm_amd as dg

name = sys.argv[1] if len(sys.argv) > 1 else 'decode_m1_long'
calls, flops, nbytes, desc, check, bound = bench.make_workload(name, 2)
calls[0](); torch.cuda.synchronize()
def eager(n):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): calls[i % len(calls)]()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
def graphed(per_graph, replays):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(2): calls[i % len(calls)]()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(per_graph): calls[i % len(calls)]()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(replays): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (replays * per_graph) * 1e3
out = {'workload': name, 'kernel': None}
for rep in range(2):
    for n in (40, 200, 1000):
        out[f'eager_{n}_us_r{rep}'] = round(eager(n), 2)
    for pg, rp in ((20, 10), (20, 50), (100, 10)):
        out[f'graph_{pg}x{rp}_us_r{rep}'] = round(graphed(pg, rp), 2)
out['kernel'] = dg.last_config()
print(json.dumps(out))
