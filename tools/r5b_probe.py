#!/usr/bin/env python3
"""Round 5, second half: same-box A/B of this session's candidates, ONE process, every arm a warmed hipGraph of `calls` operator calls over
rotating input sets, the arms replayed alternately (figure of merit: us per call, median over the replays).
    python tools/r5b_probe.py [pc192] [skinny] [grouped_nn]"""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
from deepgemm_amd._lib import lib                                       # noqa: E402
from deepgemm_amd.testing import generators as gen                      # noqa: E402


STREAM = None


def graph_of(calls):
    global STREAM
    if STREAM is None:
        STREAM = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(STREAM):       # eager once ON the capture stream: caches, the per-stream K-split workspace
        for c in calls:
            c()
    STREAM.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=STREAM):
        for c in calls:
            c()
    return g


def ab(title, arms, replays=40):
    """arms: [(label, setup, calls)]; setup() runs before the capture of that arm (forced config / env knobs)."""
    graphs = []
    for label, setup, calls in arms:
        setup()
        graphs.append((label, graph_of(calls), len(calls)))
        dg.set_forced_config('auto')
    t_end = time.time() + 1.0
    while time.time() < t_end:
        for _, g, _ in graphs:
            g.replay()
        torch.cuda.synchronize()
    times = {label: [] for label, _, _ in graphs}
    for it in range(replays):
        order = graphs if it % 2 == 0 else graphs[::-1]
        for label, g, ncalls in order:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            g.replay()
            e.record()
            e.synchronize()
            times[label].append(s.elapsed_time(e) * 1e3 / ncalls)
    out = {'experiment': title}
    for label, _, _ in graphs:
        us = sorted(times[label])
        out[label] = {'us_median': round(statistics.median(us), 2), 'us_p10': round(us[len(us) // 10], 2), 'us_p90': round(us[len(us) * 9 // 10], 2)}
    print(json.dumps(out), flush=True)


def env_setup(**kv):
    def f():
        for k, v in kv.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        lib.dg_reload_env()
    return f


def forced(name):
    return lambda: dg.set_forced_config(name)


def pc192():
    for m, n, k in ((576, 4096, 7168), (2112, 4096, 7168), (640, 4096, 2048), (1152, 7168, 4096)):
        calls = []
        for i in range(4):
            gen.reset_seed(i)
            case = gen.generate_normal(m, n, k, accumulate=True, out_dtype=torch.float, per_token_b=True)
            a = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1]))
            b = (case.b[0], dg.get_mn_major_tma_aligned_tensor(case.b[1]))
            calls.append(lambda a=a, b=b, c=case: dg.fp8_gemm_nt(a, b, c.d, c=c.d, recipe=(1, 1, 128)))
        names = {}

        def note(tag):
            def f():
                calls[0]()
                names[tag] = dg.last_config()
            return f
        arms = []
        for bm in (256, 192):
            setup = env_setup(DG_PC_BM=bm)
            arms.append((f'bm{bm}', (lambda s=setup, t=f'bm{bm}': (s(), note(t)())), calls))
        ab(f'recipe (1,1,128) wgrad {m}x{n}x{k}', arms)
        env_setup(DG_PC_BM=None)()
        print(json.dumps({'kernels': names}), flush=True)


def skinny():
    for (m, n, k), pairs in (((1, 4096, 7168), [('skinny_16', 'skinny_16c')]),
                             ((1, 7168, 16384), [('skinny_16w', 'skinny_16wc'), ('skinny_16', 'skinny_16c')]),
                             ((16, 4096, 7168), [('skinny_16', 'skinny_16c')]),
                             ((1, 2112, 7168), [('skinny_16', 'skinny_16c')]),
                             ((32, 4096, 7168), [('skinny_32', 'skinny_32c')])):
        calls = []
        for i in range(4):
            gen.reset_seed(i)
            case = gen.generate_normal(m, n, k)
            a = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1]))
            calls.append(lambda a=a, c=case: dg.fp8_gemm_nt(a, c.b, c.d))
        calls = calls * 5
        for plain, coal in pairs:
            ab(f'decode {m}x{n}x{k}', [(plain, forced(plain), calls), (coal, forced(coal), calls)])


def grouped_nn():
    for groups, m_per, n, k in ((8, 512, 4096, 7168), (4, 1024, 7168, 2048)):
        calls_nn, calls_nt = [], []
        for i in range(2):
            gen.reset_seed(i)
            case = gen.generate_m_grouped_contiguous(groups, m_per, n, k, True, False, use_ue8m0=True)
            a = gen.packed_ue8m0_operand(*case.a)
            b = gen.packed_ue8m0_operand(*case.b, mn_rows=n)
            b_nn = b[0].mT.contiguous()
            calls_nn.append(lambda a=a, b_nn=b_nn, sf=b[1], c=case: dg.m_grouped_fp8_gemm_nn_contiguous(a, (b_nn, sf.mT), c.d, c.grouped_layout))
            calls_nt.append(lambda a=a, b=b, c=case: dg.m_grouped_fp8_gemm_nt_contiguous(a, b, c.d, c.grouped_layout))
        names = {}

        def note(tag, call):
            def f():
                call()
                names[tag] = dg.last_config()
            return f
        ab(f'grouped contiguous, packed scales, {groups} x ~{m_per} rows, n={n} k={k}',
           [('nn_in_place(auto)', note('nn_auto', calls_nn[0]), calls_nn),
            ('nn_remajored(e8_quad_128x256)', (lambda: (forced('e8_quad_128x256')(), note('nn_forced', calls_nn[0])())), calls_nn),
            ('nt_k_major(auto)', note('nt_auto', calls_nt[0]), calls_nt)], replays=20)
        print(json.dumps({'kernels': names}), flush=True)


def packed_c4():
    """Packed scales on the contiguous layout: the group-relative tiling (auto) against 128-row tiles on the fixed grid (forced)."""
    for groups, m_per, n, k in ((8, 512, 4096, 7168), (4, 1024, 7168, 2048), (8, 512, 7168, 2048)):
        calls = []
        for i in range(2):
            gen.reset_seed(i)
            case = gen.generate_m_grouped_contiguous(groups, m_per, n, k, True, False, use_ue8m0=True)
            a = gen.packed_ue8m0_operand(*case.a)
            b = gen.packed_ue8m0_operand(*case.b, mn_rows=n)
            calls.append(lambda a=a, b=b, c=case: dg.m_grouped_fp8_gemm_nt_contiguous(a, b, c.d, c.grouped_layout))
        names = {}

        def note(tag, setup):
            def f():
                setup()
                calls[0]()
                names[tag] = dg.last_config()
            return f
        ab(f'grouped contiguous nt, packed scales, {groups} x ~{m_per} rows, n={n} k={k}',
           [('auto', note('auto', lambda: None), calls), ('e8_quad_128x256', note('forced', forced('e8_quad_128x256')), calls)], replays=20)
        print(json.dumps({'kernels': names}), flush=True)


def skinny_w():
    """One N-subtile per workgroup (1.x rounds) against two (one round) with the coalesced loads, where n / 16 lies between one and two rounds."""
    for m, n, k in ((1, 7168, 4096), (16, 8192, 2048), (1, 6144, 7168), (4, 7168, 16384)):
        calls = []
        for i in range(4):
            gen.reset_seed(i)
            case = gen.generate_normal(m, n, k)
            a = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1]))
            calls.append(lambda a=a, c=case: dg.fp8_gemm_nt(a, c.b, c.d))
        calls = calls * 5
        ab(f'decode {m}x{n}x{k}', [('skinny_16c', forced('skinny_16c'), calls), ('skinny_16wc', forced('skinny_16wc'), calls),
                                   ('skinny_16ca', forced('skinny_16ca'), calls)])


def skinny_a():
    """Coalesced activation loads on top of the coalesced weight loads."""
    for (m, n, k), pair in (((1, 4096, 7168), ('skinny_16c', 'skinny_16ca')), ((4, 4096, 7168), ('skinny_16c', 'skinny_16ca')),
                            ((8, 4096, 7168), ('skinny_16c', 'skinny_16ca')), ((16, 4096, 7168), ('skinny_16c', 'skinny_16ca')),
                            ((16, 2112, 7168), ('skinny_16c', 'skinny_16ca')), ((32, 4096, 7168), ('skinny_32c', 'skinny_32ca')),
                            ((24, 4096, 4096), ('skinny_32c', 'skinny_32ca'))):
        calls = []
        for i in range(4):
            gen.reset_seed(i)
            case = gen.generate_normal(m, n, k)
            a = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1]))
            calls.append(lambda a=a, c=case: dg.fp8_gemm_nt(a, c.b, c.d))
        calls = calls * 5
        ab(f'decode {m}x{n}x{k}', [(pair[0], forced(pair[0]), calls), (pair[1], forced(pair[1]), calls)])


def skinny_32():
    for m, n, k in ((32, 4096, 7168), (24, 4096, 4096), (17, 2112, 7168), (32, 4608, 8192)):
        calls = []
        for i in range(4):
            gen.reset_seed(i)
            case = gen.generate_normal(m, n, k)
            a = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1]))
            calls.append(lambda a=a, c=case: dg.fp8_gemm_nt(a, c.b, c.d))
        calls = calls * 5
        ab(f'decode {m}x{n}x{k}', [('skinny_32c', forced('skinny_32c'), calls), ('skinny_32ca', forced('skinny_32ca'), calls)])


if __name__ == '__main__':
    which = sys.argv[1:] or ['pc192', 'skinny', 'grouped_nn']
    for w in which:
        {'pc192': pc192, 'skinny': skinny, 'grouped_nn': grouped_nn, 'packed_c4': packed_c4, 'skinny_w': skinny_w, 'skinny_a': skinny_a, 'skinny_32': skinny_32}[w]()
