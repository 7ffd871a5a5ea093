#!/usr/bin/env python3
"""Headline benchmark: achieved FP8 TFLOPS of ``fp8_gemm_nt`` at M=4096 N=4096 K=7168 (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload dense|contiguous|masked]

A "step" is one pass of the hot path (one GEMM launch) over one batch of synthetic, already HBM-resident input
(``torch.manual_seed(0)`` BF16 randn, quantised with the reference's per-token / per-block casts).  Input sets are
rotated so that consecutive steps do not hit the 256 MiB Infinity Cache with the same operands.  The dense path does not
shard ("replicas only"): with N > 1 every rank runs an independent replica and ``value`` is the whole-job
aggregate (scaling: weak).  Rank 0 prints ONE JSON line carrying ``roofline`` (dominant kernel vs the dense FP8 MFMA
peak, timed with HIP events on the launch stream) and, at N = 1, ``cpu_baseline`` (the reference's CPU-runnable test
expression timed on the host cores of this box).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import deepgemm_amd as dg                                            # noqa: E402
from deepgemm_amd.testing import calc_diff, count_bytes, generators as gen   # noqa: E402

PEAK_FP8_TFLOPS = 5000.0      # dense FP8 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0


def measured_traffic(kernel: str):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/<round>/traffic.json,
    FETCH_SIZE / WRITE_SIZE collected in separate passes and corrected as MI355X_MICROARCH.md prescribes); None if the
    newest committed profile is of a different kernel."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*', 'traffic.json')), reverse=True):
        with open(path) as f:
            rec = json.load(f)
        if rec.get('kernel') == kernel:
            return rec['traffic_bytes']
    return None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default='dense', choices=['dense', 'contiguous', 'masked'])
    ap.add_argument('--config', default='auto', help='force a kernel configuration (tuning)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--clock-warmup-s', type=float, default=1.0, help='untimed load before the warm-up steps (seconds)')
    ap.add_argument('--sets', type=int, default=4, help='rotating input sets (defeats Infinity-Cache residency)')
    return ap.parse_args()


def make_ep_workload(sets: int, world: int, rank: int):
    """BASELINE configs[4] across ranks: 8 experts per rank (8 * world in total, weights resident on their owner), 48
    token rows per rank routed to top-8 experts => about 48 rows per expert; one step = all-to-all dispatch + local masked
    grouped GEMM + all-to-all combine (deepgemm_amd/ep.py)."""
    from deepgemm_amd import ep
    from deepgemm_amd.utils.math import per_block_cast_to_fp8, per_token_cast_to_fp8
    per_rank, max_m, tokens, top_k, n, k = 8, 128, 48, 8, 4096, 7168
    num_experts = per_rank * world
    torch.manual_seed(1000 + rank)
    w = torch.randn((per_rank, n, k), dtype=torch.bfloat16, device='cuda')
    bq = [per_block_cast_to_fp8(w[g], use_ue8m0=False) for g in range(per_rank)]
    b_local = (torch.stack([q[0] for q in bq]), torch.stack([q[1] for q in bq]))
    del w, bq
    calls = []
    for i in range(sets):
        x = per_token_cast_to_fp8(torch.randn((tokens, k), dtype=torch.bfloat16, device='cuda'), use_ue8m0=False)
        ids = torch.stack([torch.randperm(num_experts, device='cuda')[:top_k] for _ in range(tokens)])
        calls.append(lambda x=x, ids=ids: ep.ep_m_grouped_fp8_gemm_nt_masked(x, ids, b_local, num_experts, max_m, expected_m=48))
    flops = 2.0 * tokens * top_k * n * k
    nbytes = float(b_local[0].numel() + 4 * b_local[1].numel() + tokens * top_k * (k + 4 * k // 128 + 2 * n))
    desc = {'workload': f'EP m_grouped_fp8_gemm_nt_masked: {num_experts} experts / {world} GPUs ({per_rank} per rank), about 48 rows per expert, '
                        f'N={n} K={k}; step = RCCL all-to-all dispatch + local masked GEMM + all-to-all combine (BASELINE configs[4])',
            'n': n, 'k': k, 'experts': num_experts}
    return calls, flops, nbytes, desc, lambda: float('nan')


def make_workload(name: str, sets: int, world: int = 1, rank: int = 0):
    """Returns (list of zero-arg callables, flops per step, algorithmic bytes per step, description, checker)."""
    calls, cases = [], []
    if name == 'masked' and world > 1:
        return make_ep_workload(sets, world, rank)
    if name == 'dense':
        m, n, k = 4096, 4096, 7168
        for i in range(sets):
            gen.reset_seed(i)
            case = gen.generate_normal(m, n, k)
            # Producers hand SFA over in the kernel's MN-major layout (zero-copy branch of the reference's layout step,
            # smxx_layout.hpp:124-125), so a step is exactly one GEMM launch.
            case.a = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1]))
            cases.append(case)
            calls.append(lambda c=case: dg.fp8_gemm_nt(c.a, c.b, c.d))
        flops = 2.0 * m * n * k
        nbytes = m * k + n * k + 4 * m * (k // 128) + 4 * (n // 128) * (k // 128) + 2 * m * n
        desc = {'workload': f'fp8_gemm_nt M={m} N={n} K={k} (DeepSeek-V3 dense shape, BASELINE configs[1])', 'm': m, 'n': n, 'k': k}
        check = lambda: calc_diff(cases[0].d, cases[0].ref_d)    # noqa: E731
    elif name == 'contiguous':
        groups, expected, n, k = 8, 512, 4096, 7168
        for i in range(sets):
            gen.reset_seed(i)
            case = gen.generate_m_grouped_contiguous(groups, expected, n, k)
            case.a = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1]))
            cases.append(case)
            calls.append(lambda c=case: dg.m_grouped_fp8_gemm_nt_contiguous(c.a, c.b, c.d, c.grouped_layout))
        m = cases[0].m
        flops = 2.0 * m * n * k
        nbytes = count_bytes(cases[0].a, cases[0].b, cases[0].d)
        desc = {'workload': f'm_grouped_fp8_gemm_nt_contiguous G={groups} M_total={m} N={n} K={k} (BASELINE configs[3])',
                'm': m, 'n': n, 'k': k, 'groups': groups}
        check = lambda: calc_diff(cases[0].d, cases[0].ref_d)    # noqa: E731
    else:
        groups, max_m, expected, n, k = 8, 64, 48, 4096, 7168
        for i in range(sets):
            gen.reset_seed(i)
            case = gen.generate_m_grouped_masked(groups, max_m, expected, n, k, masked_ms=None)
            case.a = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1]))
            cases.append(case)
            calls.append(lambda c=case: dg.m_grouped_fp8_gemm_nt_masked(c.a, c.b, c.d, c.masked_m, expected))
        valid = int(cases[0].masked_m.sum().item())
        flops = 2.0 * valid * n * k
        nbytes = count_bytes(cases[0].a, cases[0].d) * valid / (max_m * groups) + count_bytes(cases[0].b)
        desc = {'workload': f'm_grouped_fp8_gemm_nt_masked G={groups} (64 experts / 8 GPUs) M<=64 N={n} K={k} (BASELINE configs[4], one rank)',
                'valid_m': valid, 'n': n, 'k': k, 'groups': groups}

        def check():
            c = cases[0]
            return max(calc_diff(c.d[g, :int(r)], c.ref_d[g, :int(r)]) for g, r in enumerate(c.masked_m.tolist()) if r)
    return calls, flops, float(nbytes), desc, check


def cpu_baseline(workload: str):
    """The reference's CPU-runnable path for this workload: its test oracle expression (tests/generators.py:312)
    ``(a.float() @ b.float().t()).to(bfloat16)`` on the same synthetic inputs, timed on this box's host cores."""
    m, n, k = 4096, 4096, 7168
    torch.manual_seed(0)
    a = torch.randn((m, k), dtype=torch.bfloat16)
    b = torch.randn((n, k), dtype=torch.bfloat16)
    best = float('inf')
    for i in range(3):                      # 1 warm-up + 2 timed runs (about 7 s each on 8 cores)
        t0 = time.perf_counter()
        (a.float() @ b.float().t()).to(torch.bfloat16)
        dt = time.perf_counter() - t0
        if i:
            best = min(best, dt)
    return {'value': 2.0 * m * n * k / best / 1e12, 'unit': 'TFLOPS', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'full M={m} N={n} K={k} problem, reference test expression (a.float() @ b.float().t()).to(bf16), '
                      f'min of 2 timed runs after 1 warm-up, {best:.2f} s per run, os.cpu_count()={os.cpu_count()}'}


def main():
    args = parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world and world > 1:
        raise SystemExit(f'--gpus {args.gpus} does not match WORLD_SIZE {world}')
    torch.cuda.set_device(local_rank)
    distributed = world > 1
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    dg.set_forced_config(args.config)
    calls, flops, nbytes, desc, check = make_workload(args.workload, args.sets, world, rank)

    calls[0]()
    torch.cuda.synchronize()
    diff = check()                          # parity vs the reference test expression, before anything is timed
    # Untimed clock warm-up, then the W untimed warm-up steps, then -- without an idle gap in between -- the K timed
    # steps: the chip needs a few hundred milliseconds of load to reach its sustained clock / power state, and a result
    # check between warm-up and timing would let it fall back to idle.
    t_warm = time.perf_counter() + args.clock_warmup_s
    i = 0
    while time.perf_counter() < t_warm:
        for _ in range(16):
            calls[i % len(calls)]()
            i += 1
        torch.cuda.synchronize()
    for i in range(args.warmup):
        calls[i % len(calls)]()
    torch.cuda.synchronize()

    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    start.record()                          # HIP events on the stream the kernels are launched on (torch's current stream)
    for i in range(args.steps):
        calls[i % len(calls)]()
    end.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        dist.barrier()
        t = torch.tensor([elapsed], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_s = start.elapsed_time(end) / 1e3 / args.steps       # average launch duration of the dominant kernel

    if rank == 0:
        total_flops = flops * args.steps * world
        value = total_flops / elapsed / 1e12
        achieved = flops / kernel_s / 1e12
        mfma_bound = args.workload != 'masked'
        traffic = measured_traffic(dg.last_config()) if args.workload == 'dense' else None
        roofline = ({'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_FP8_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': achieved / PEAK_FP8_TFLOPS, 'traffic': traffic} if mfma_bound else
                    {'bound': 'hbm', 'achieved': nbytes / kernel_s / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                     'frac': nbytes / kernel_s / 1e9 / PEAK_HBM_GBS, 'traffic': traffic})
        roofline.update({'kernel': dg.last_config(), 'kernel_us': kernel_s * 1e6, 'algorithmic_flops': flops,
                         'algorithmic_bytes': nbytes, 'tflops': achieved, 'gbs': nbytes / kernel_s / 1e9})
        line = {
            'metric': 'achieved FP8 TFLOPS (and % of MFMA roofline) for fp8_gemm_nt M=4096 N=4096 K=7168'
                      if args.workload == 'dense' else f'achieved FP8 TFLOPS for {args.workload} grouped FP8 GEMM',
            'value': value, 'unit': 'TFLOPS', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'fp8_e4m3 (fp32 accumulate, bf16 out)', 'data': 'synthetic',
            'config': dict(desc, parallelism=('single GPU' if world == 1 else f'ep{world}' if args.workload == 'masked' else 'replicas only'), kernel=dg.last_config(),
                           input_sets=len(calls)),
            'pct_of_mfma_peak': 100.0 * value / world / PEAK_FP8_TFLOPS,
            'calc_diff_vs_reference_expr': diff,
            'roofline': roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args.workload)
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
