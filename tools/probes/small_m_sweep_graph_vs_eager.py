#!/usr/bin/env python3
"""The reference's forward sweep at m = 1 and m = 128 (tests/generators.py:119-121: seven (n, k) pairs): kernel time as a hipGraph replay over cold
operand sets beside the eager per-call time (host-bound at these sizes), FP32 scales and packed UE8M0 words; optional config names to force.
python tools/probes/small_m_sweep_graph_vs_eager.py [cfg ...]"""
import sys
sys.path.insert(0, '.')
import torch, deepgemm_amd as dg
from deepgemm_amd.testing import generators as gen

NK = [(2112, 7168), (576, 7168), (24576, 1536), (32768, 512), (7168, 16384), (4096, 7168), (7168, 2048)]
forced = sys.argv[1:]


def time_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for m in (1, 128):
    for n, k in NK:
        sets = max(2, min(24, int(320e6 // (n * k)) + 1))
        for packed in (False, True):
            ops = []
            for i in range(sets):
                gen.reset_seed(i)
                c = gen.generate_normal(m, n, k, use_ue8m0=packed)
                c.a_bf16 = c.b_bf16 = None
                if packed:
                    ops.append((gen.packed_ue8m0_operand(*c.a), gen.packed_ue8m0_operand(*c.b, mn_rows=n), c.d))
                else:
                    ops.append(((c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1])), c.b, c.d))
            res = []
            for cfg in ['auto'] + [f for f in forced if f.startswith('e8_') == packed]:
                try:
                    dg.set_forced_config(cfg)
                    dg.fp8_gemm_nt(*ops[0])
                    name = dg.last_config()
                    it = [0]
                    def eager():
                        o = ops[it[0] % sets]; it[0] += 1
                        dg.fp8_gemm_nt(*o)
                    t_eager = time_us(eager, n=40)
                    side = torch.cuda.Stream(); graph = torch.cuda.CUDAGraph()
                    torch.cuda.synchronize()
                    with torch.cuda.graph(graph, stream=side):
                        for o in ops:
                            dg.fp8_gemm_nt(*o)
                    t = time_us(graph.replay, n=10) / sets
                    nbytes = m * k + n * k + 2 * m * n
                    res.append(f'{name} {t:.1f} us graph ({nbytes / t / 1e3:.0f} GB/s) / {t_eager:.1f} eager')
                except RuntimeError as e:
                    res.append(f'{cfg}: {str(e)[:40]}')
                finally:
                    dg.set_forced_config('auto')
            print(f'm={m} n={n} k={k} {"packed" if packed else "fp32  "}: ' + ' | '.join(res), flush=True)
            del ops
