import json, os, sys, time, torch
sys.path.insert(0, '/root/repo')
import bench, deepgemm_amd as dg
calls, flops, nbytes, desc, check, bound = bench.make_workload('decode_m1_long', 2)
calls[0](); torch.cuda.synchronize()
t_end = time.perf_counter() + 0.25
i = 0
while time.perf_counter() < t_end:
    for _ in range(4):
        calls[i % 2](); i += 1
    torch.cuda.synchronize()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for i in range(2): calls[i % 2]()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for i in range(20): calls[i % 2]()
per = []
for r in range(60):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    per.append(round(s.elapsed_time(e) / 20 * 1e3, 2))
print('per-replay us/call (sync after each):', per)
per2 = []
for r in range(6):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): g.replay()
    e.record(); torch.cuda.synchronize()
    per2.append(round(s.elapsed_time(e) / 200 * 1e3, 2))
print('10 replays back to back:', per2)
