#!/usr/bin/env python3
"""Headline benchmark: achieved FP8 TFLOPS of ``fp8_gemm_nt`` at M=4096 N=4096 K=7168 (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload dense|contiguous|masked|c3_{nt,nn,tn,tt}|wgrad|wgrad_ksplit|kgrouped|dense_ue8m0|dense_sm100|masked_ue8m0|dgrad_ktail|dgrad_ksplit|decode_m1|decode_m1_long|expert_mlp|expert_mlp_unfused|dense_sfa_rowmajor|dgrad_ktail_ue8m0|dense_m128|c3_nn_ue8m0|contiguous_ue8m0|dense_ue8m0_g32|wgrad_ue8m0|kgrouped_ue8m0|kgrouped_ue8m0_g32|decode_m1_ue8m0]

A "step" is one pass of the hot path (one operator call) over one batch of synthetic, already HBM-resident input
(``torch.manual_seed`` BF16 randn, quantised with the reference's per-token / per-block casts).  Input sets are
rotated so that consecutive steps do not hit the 256 MiB Infinity Cache with the same operands.  The dense path does not
shard ("replicas only"): with N > 1 every rank runs an independent replica and ``value`` is the whole-job
aggregate (scaling: weak); the masked MoE workload shards its experts over the ranks (RCCL all-to-all dispatch / combine,
deepgemm_amd/ep.py) and reports the dispatch / GEMM / combine time split.  ``--gpus N`` without a launcher
(``WORLD_SIZE`` unset) spawns the N ranks itself.  Rank 0 prints ONE JSON line carrying ``roofline`` (dominant kernel vs the
dense FP8 MFMA peak or the HBM peak, timed with HIP events on the launch stream), at N = 1 ``cpu_baseline`` (the reference's
CPU-runnable test expression timed on the host cores of this box) and, on the default run, ``secondary``: the same
measurement, shortened, for the other BASELINE configurations (C3 per layout, C4, C5), the wgrad recipe, the K-grouped
GEMM, the packed-UE8M0 forms of C2 and C5 and two dgrad entries of the reference sweep (K tail, K split).
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
# Recipe (1, 1, 128) -- a distinct scale per accumulator ELEMENT per K block -- costs two VALU operations per element: 4 v_mul + 4 v_fmac per
# 16x16x128 MFMA, measured ~62 matrix-pipe cycles per step in either schedule (profiles/r04_probe/wgrad_duo_pc_ab.log) against the MFMA's
# own 32: the arithmetic roof of that recipe on this part is 32 / 62 of the matrix rate.  Reported beside the MFMA fraction.
RECIPE_1_1_128_ROOF = 32.0 / 62.0
WORKLOADS = ['dense', 'contiguous', 'masked', 'c3_nt', 'c3_nn', 'c3_tn', 'c3_tt', 'wgrad', 'kgrouped', 'dense_ue8m0', 'dgrad_ktail', 'dgrad_ksplit', 'masked_ue8m0', 'wgrad_ksplit',
             'dense_sm100', 'decode_m1', 'decode_m1_long', 'expert_mlp', 'expert_mlp_unfused', 'dense_sfa_rowmajor', 'dgrad_ktail_ue8m0', 'dense_m128', 'c3_nn_ue8m0',
             'contiguous_ue8m0', 'dense_ue8m0_g32', 'wgrad_ue8m0', 'kgrouped_ue8m0', 'kgrouped_ue8m0_g32', 'decode_m1_ue8m0']
SECONDARY = ['c3_nt', 'c3_nn', 'c3_tn', 'c3_tt', 'contiguous', 'masked', 'wgrad', 'kgrouped', 'dense_ue8m0', 'dense_sm100', 'dgrad_ktail', 'dgrad_ksplit',
             'masked_ue8m0', 'wgrad_ksplit', 'decode_m1', 'decode_m1_long', 'expert_mlp', 'dense_sfa_rowmajor', 'dgrad_ktail_ue8m0', 'dense_m128', 'c3_nn_ue8m0',
             'contiguous_ue8m0', 'dense_ue8m0_g32', 'wgrad_ue8m0', 'kgrouped_ue8m0', 'kgrouped_ue8m0_g32', 'decode_m1_ue8m0']
# HBM-bound workloads whose per-call weight stream is smaller than the 256 MiB Infinity Cache (MALL): the rotation must cover more than the
# cache, or the "fraction of 8 TB/s" is a cache-read rate (the reference flushes 8 GB between timed iterations: deep_gemm/testing/bench.py:93,108).
# sets x (bytes not re-used across calls) >= COLD_ROTATION_BYTES; the other HBM-bound lines stream >= 235 MB of weights per call x >= 2 sets.
COLD_ROTATION_BYTES = 320e6


def cold_sets(weight_bytes: float, at_least: int) -> int:
    return max(at_least, int(-(-COLD_ROTATION_BYTES // weight_bytes)))


GRAPHED = {'decode_m1', 'decode_m1_long', 'decode_m1_ue8m0', 'expert_mlp', 'expert_mlp_unfused'}      # launch-bound decode calls: timed as a hipGraph replay (the eager call time is reported beside it)


def measured_counters(kernel: str, workload: str) -> dict:
    """Counter-derived figures of `kernel` ON `workload` from the newest committed rocprofv3 PMC passes (profiles/<round>/traffic_<workload>_<kernel>.json,
    written by tools/make_traffic_json.py from tools/gpu_prof_r06.sh's passes): `traffic` = HBM-side bytes per launch (FETCH_SIZE / WRITE_SIZE,
    separate passes, corrected as MI355X_MICROARCH.md prescribes), `mfma_busy` = SQ_VALU_MFMA_BUSY_CYCLES / (shader cycles x 1024 SIMDs),
    `clock_ghz` = shader cycles of a launch / its kernel-trace duration, `counters_git` = the commit the passes were measured on.  They are
    properties of the kernel on that workload, measured once per round on its own profiling runs -- not of this run; empty if no committed
    profile matches (files of earlier rounds carry no workload: they describe the headline)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*', 'traffic*.json')), reverse=True):
        with open(path) as f:
            rec = json.load(f)
        if rec.get('kernel') == kernel and rec.get('workload', 'dense') == workload:
            return {'traffic': rec['traffic_bytes'], 'mfma_busy': rec.get('mfma_busy'), 'clock_ghz': rec.get('clock_ghz'),
                    'counters_git': rec.get('git'), 'counters_from': os.path.relpath(path, ROOT)}
    return {}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default='dense', choices=WORKLOADS)
    ap.add_argument('--config', default='auto', help='force a kernel configuration (tuning)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the shortened runs of the other configurations')
    ap.add_argument('--clock-warmup-s', type=float, default=1.0, help='untimed load before the warm-up steps (seconds)')
    ap.add_argument('--sets', type=int, default=4, help='rotating input sets (defeats Infinity-Cache residency)')
    ap.add_argument('--ep-capacity', type=int, default=0,
                    help='masked workload with --gpus > 1: rows per (rank, expert) block of the fixed-shape exchange (0 = exact split sizes)')
    return ap.parse_args()


def make_ep_workload(sets: int, world: int, rank: int, phase_events: list, capacity: int = 0):
    """BASELINE configs[4] across ranks: 8 experts per rank (8 * world in total, weights resident on their owner), 48
    token rows per rank routed to top-8 experts => about 48 rows per expert; one step = all-to-all dispatch + local masked
    grouped GEMM + all-to-all combine + top-k weighted reduce on the token's owner (deepgemm_amd/ep.py)."""
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
        weights = torch.softmax(torch.randn((tokens, top_k), device='cuda'), dim=-1)
        calls.append(lambda x=x, ids=ids, weights=weights: ep.ep_m_grouped_fp8_gemm_nt_masked(
            x, ids, b_local, num_experts, max_m, expected_m=48, topk_weights=weights, phase_events=phase_events,
            capacity=capacity if capacity > 0 else None))
    flops = 2.0 * tokens * top_k * n * k
    nbytes = float(b_local[0].numel() + 4 * b_local[1].numel() + tokens * top_k * (k + 4 * k // 128 + 2 * n))
    desc = {'workload': f'EP m_grouped_fp8_gemm_nt_masked: {num_experts} experts / {world} GPUs ({per_rank} per rank), about 48 rows per expert, '
                        f'N={n} K={k}; step = RCCL all-to-all dispatch + local masked GEMM + all-to-all combine + top-{top_k} weighted reduce '
                        '(BASELINE configs[4])',
            'n': n, 'k': k, 'experts': num_experts, 'top_k': top_k}
    return calls, flops, nbytes, desc, lambda: float('nan'), 'hbm'


def make_workload(name: str, sets: int, world: int = 1, rank: int = 0, phase_events=None, ep_capacity: int = 0):
    """Returns (zero-arg callables, flops per step, algorithmic bytes per step, description, checker, roofline bound)."""
    calls, cases = [], []
    bound = 'mfma'
    if name == 'masked' and world > 1:
        return make_ep_workload(sets, world, rank, phase_events, ep_capacity)
    if name == 'dense_sm100':
        # C2 as an SM100-style caller sends it: FP32 power-of-two scales (per_token / per_block casts with use_ue8m0=True, row-major
        # SFA as the cast returns it), scaling-factor mode 'sm100' -- the WHOLE call: the cast branch's pack launches (SFA; SFB with the
        # per-128-row broadcast fused in) + the hardware-scaled GEMM (reference default on SM100, csrc/apis/layout.hpp:48-54)
        m, n, k = 4096, 4096, 7168
        for i in range(sets):
            gen.reset_seed(i)
            case = gen.generate_normal(m, n, k, use_ue8m0=True)
            cases.append(case)

            def call(c=case):
                dg.set_sf_cast_mode('sm100')
                try:
                    dg.fp8_gemm_nt(c.a, c.b, c.d)
                finally:
                    dg.set_sf_cast_mode('sm90')
            calls.append(call)
        flops = 2.0 * m * n * k
        nbytes = m * k + n * k + 4 * m * (k // 128) + 4 * (n // 128) * (k // 128) + 2 * m * n
        desc = {'workload': f'fp8_gemm_nt M={m} N={n} K={k}, FP32 power-of-two scales, sf cast mode sm100: whole call (cast branch + GEMM)',
                'm': m, 'n': n, 'k': k, 'sfa_layout': 'FP32 row-major (as per_token_cast_to_fp8 returns it); cast to packed UE8M0 inside the call'}
        check = lambda: calc_diff(cases[0].d, cases[0].ref_d)    # noqa: E731
    elif name in ('dense', 'dense_ue8m0', 'dense_sfa_rowmajor', 'dense_ue8m0_g32'):
        m, n, k = 4096, 4096, 7168
        packed = name == 'dense_ue8m0'
        for i in range(sets):
            gen.reset_seed(i)
            case = gen.generate_normal(m, n, k, use_ue8m0=packed)
            if name == 'dense_ue8m0_g32':
                # C2 with scales of granularity 32 along K (round 6: the reference's SM100 MX recipe (1, 1, 32), csrc/apis/gemm.hpp:311-312): the
                # reference's quantiser at gran_k = 32, packed words of one 128-K block each, the scaled MFMA's native block size
                from deepgemm_amd.utils.math import pack_ue8m0_to_int, per_token_cast_to_fp8
                qa, qb = per_token_cast_to_fp8(case.a_bf16, True, 32), per_token_cast_to_fp8(case.b_bf16, True, 32)
                a = (qa[0], dg.transform_sf_into_required_layout(pack_ue8m0_to_int(qa[1]), m, k, (1, 32)))
                b = (qb[0], dg.transform_sf_into_required_layout(pack_ue8m0_to_int(qb[1]), n, k, (1, 32)))
                case.a_bf16 = case.b_bf16 = None
                cases.append(case)
                calls.append(lambda a=a, b=b, c=case: dg.fp8_gemm_nt(a, b, c.d, recipe=(1, 1, 32)))
                continue
            if packed:
                # power-of-two scales handed over as packed UE8M0 words (the reference's SM100 input format): hardware-scaled MFMA
                a, b = gen.packed_ue8m0_operand(*case.a), gen.packed_ue8m0_operand(*case.b, mn_rows=n)
            elif name == 'dense_sfa_rowmajor':
                # SFA row-major as per_token_cast_to_fp8 returns it (what the reference's own test passes): every call pays the layout
                # step (csrc/apis/layout.hpp:14-46 -> transpose_fp32): one small transpose launch + the GEMM
                a, b = case.a, case.b
            else:
                # Producers hand SFA over in the kernel's MN-major layout (zero-copy branch of the reference's layout step,
                # smxx_layout.hpp:124-125), so a step is exactly one GEMM launch.
                a, b = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1])), case.b
            cases.append(case)
            calls.append(lambda a=a, b=b, c=case: dg.fp8_gemm_nt(a, b, c.d))
        flops = 2.0 * m * n * k
        nbytes = m * k + n * k + 4 * m * (k // 128) + 4 * (n // 128) * (k // 128) + 2 * m * n
        if name == 'dense_ue8m0_g32':
            nbytes = m * k + n * k + 4 * (m + n) * (k // 128) + 2 * m * n           # one packed word per row and 128-K block, both operands
        desc = {'workload': f'fp8_gemm_nt M={m} N={n} K={k} (DeepSeek-V3 dense shape, BASELINE configs[1])' +
                            (', packed UE8M0 scales (power-of-two scales, recipe (1, 1, 128))' if packed else '') +
                            (', packed UE8M0 scales of granularity 32 along K (recipe (1, 1, 32))' if name == 'dense_ue8m0_g32' else '') +
                            (', SFA row-major as the cast returns it: whole call = layout step (transpose launch) + GEMM' if name == 'dense_sfa_rowmajor' else ''),
                'm': m, 'n': n, 'k': k,
                'sfa_layout': 'packed UE8M0 words, MN-major' if packed or name == 'dense_ue8m0_g32' else 'FP32 row-major [M, K/128]: transposed inside the call' if name == 'dense_sfa_rowmajor' else 'FP32 MN-major (zero-copy branch)'}
        check = lambda: calc_diff(cases[0].d, cases[0].ref_d)    # noqa: E731
    elif name.startswith('c3_'):
        layout, packed = name[3:5], name.endswith('_ue8m0')
        m, n, k = 2048, 7168, 2048
        for i in range(sets):
            gen.reset_seed(i)
            case = gen.generate_normal(m, n, k, layout[0] == 'n', layout[1] == 't', use_ue8m0=packed)
            if packed:          # packed UE8M0 words with the operands' majorness as it is (nn: B [K][N] read in place by e8_duo_bmn_256x256)
                a, b = gen.packed_ue8m0_operand(*case.a), gen.packed_ue8m0_operand(*case.b, mn_rows=n)
            else:
                a, b = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1])), case.b
            cases.append(case)
            calls.append(lambda a=a, b=b, c=case: dg.fp8_gemm_nt(a, b, c.d))
        flops = 2.0 * m * n * k
        nbytes = m * k + n * k + 4 * m * (k // 128) + 4 * (n // 128) * (k // 128) + 2 * m * n
        desc = {'workload': f'fp8_gemm_{layout} M={m} N={n} K={k} (BASELINE configs[2]; whole operator call, majorness from strides)' +
                            (', packed UE8M0 scales' if packed else ''),
                'm': m, 'n': n, 'k': k}
        check = lambda: calc_diff(cases[0].d, cases[0].ref_d)    # noqa: E731
    elif name in ('decode_m1', 'decode_m1_long', 'decode_m1_ue8m0'):
        # batch-1 decode entries of the reference's dense sweep (tests/generators.py:119-121, m = 1): a weight stream, HBM-bound
        # ('_ue8m0', round 6: packed UE8M0 scale words -- the skinny weight-stream kernel with the scaled MFMA)
        bound = 'hbm'
        packed = name == 'decode_m1_ue8m0'
        m, n, k = (1, 7168, 16384) if name == 'decode_m1_long' else (1, 4096, 7168)
        sets = cold_sets(n * k, sets)           # 29 MB of weights a call: 11 sets; 117 MB: 3 (cold: see COLD_ROTATION_BYTES)
        for i in range(sets):
            gen.reset_seed(i)
            case = gen.generate_normal(m, n, k, use_ue8m0=packed)
            if i:
                case.a_bf16 = case.b_bf16 = None        # (only case 0 is checked against the reference expression)
            cases.append(case)
            if packed:
                a, b = gen.packed_ue8m0_operand(*case.a), gen.packed_ue8m0_operand(*case.b, mn_rows=n)
                calls.append(lambda a=a, b=b, c=case: dg.fp8_gemm_nt(a, b, c.d))
            else:
                a = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1]))
                calls.append(lambda a=a, c=case: dg.fp8_gemm_nt(a, c.b, c.d))
        flops = 2.0 * m * n * k
        nbytes = m * k + n * k + (4 * (m + n) * (-(-k // 512)) if packed else 4 * m * (k // 128) + 4 * (n // 128) * (k // 128)) + 2 * m * n
        desc = {'workload': f'fp8_gemm_nt M={m} N={n} K={k} (decode entry of the reference sweep; timed as a hipGraph replay of 20 calls, '
                            'the eager per-call time -- host-bound -- beside it)' + (', packed UE8M0 scales' if packed else ''), 'm': m, 'n': n, 'k': k}
        check = lambda: calc_diff(cases[0].d.float(), cases[0].ref_d.float()) if m * n >= 4096 else float('nan')    # noqa: E731
    elif name in ('expert_mlp', 'expert_mlp_unfused'):
        # decode-size expert MLP of one EP rank (the single-GPU half of the reference's Mega-MoE, deep_gemm/mega/__init__.py:155): 8 local
        # experts x <= 64 tokens, hidden 7168 -> 2 x 2048 (SwiGLU) -> 7168.  'expert_mlp' = GEMM1 with SwiGLU + per-token FP8 re-quantisation
        # fused into its epilogue + masked GEMM2 (2 launches); '_unfused' = masked GEMM1 -> BF16 -> torch SwiGLU -> the reference's
        # per_token_cast_to_fp8 -> masked GEMM2 (bit-identical result, tests/test_mega_gpu.py).  Weight streams: HBM-bound.
        from deepgemm_amd.utils.math import per_block_cast_to_fp8, per_token_cast_to_fp8
        bound = 'hbm'
        groups, m_max, hidden, inter, expected = 8, 64, 7168, 2048, 48
        cast = lambda w: tuple(torch.stack(t) for t in zip(*[per_block_cast_to_fp8(w[g], use_ue8m0=False) for g in range(groups)]))   # noqa: E731
        rows_total = 0
        for i in range(sets):               # every set has its own weights (352 MB a set: two sets do not fit the 256 MB last-level cache)
            gen.reset_seed(i)
            w1 = cast(torch.randn((groups, 2 * inter, hidden), device='cuda', dtype=torch.bfloat16) / hidden ** 0.5)
            w2 = cast(torch.randn((groups, hidden, inter), device='cuda', dtype=torch.bfloat16) / inter ** 0.5)
            x_bf16 = torch.randn((groups, m_max, hidden), device='cuda', dtype=torch.bfloat16)
            xq = [per_token_cast_to_fp8(x_bf16[g], use_ue8m0=False) for g in range(groups)]
            x = (torch.stack([q[0] for q in xq]), dg.get_mn_major_tma_aligned_tensor(torch.stack([q[1] for q in xq])))
            masked = torch.randint(expected - 16, m_max + 1, (groups,), device='cuda', dtype=torch.int)
            rows_total = int(masked.sum().item()) if i == 0 else rows_total
            y = torch.zeros((groups, m_max, hidden), device='cuda', dtype=torch.bfloat16)
            cases.append((x, y, masked, w1, w2))
            if name == 'expert_mlp':
                w1_t, w2_t = dg.transform_weights_for_mega_moe(w1, w2)
                mid = dg.empty_intermediate(groups, m_max, inter, 'cuda')
                # caller-owned exchange workspace: the calls are captured into a hipGraph on torch's capture stream, where the library's
                # per-stream default must not be allocated (deepgemm_amd/mega.py: _exchange_workspace)
                ws = dg.mega.swiglu_workspace(groups, m_max, 2 * inter, 'cuda')
                calls.append(lambda x=x, y=y, masked=masked, mid=mid, w1_t=w1_t, w2_t=w2_t, ws=ws:
                             dg.fp8_mega_moe_local(x, w1_t, w2_t, y, masked, expected, intermediate=mid, workspace=ws))
            else:
                h = torch.zeros((groups, m_max, 2 * inter), device='cuda', dtype=torch.bfloat16)

                def call(x=x, y=y, masked=masked, h=h, w1=w1, w2=w2):
                    dg.m_grouped_fp8_gemm_nt_masked(x, w1, h, masked, expected)
                    act = (torch.nn.functional.silu(h[..., :inter].float()) * h[..., inter:].float()).to(torch.bfloat16)
                    q, sf = per_token_cast_to_fp8(act.view(groups * m_max, inter), use_ue8m0=False)
                    dg.m_grouped_fp8_gemm_nt_masked((q.view(groups, m_max, inter), sf.view(groups, m_max, inter // 128)), w2, y, masked, expected)
                calls.append(call)
        flops = 2.0 * rows_total * hidden * inter * 3
        nbytes = groups * 3 * inter * hidden + rows_total * (hidden + 2 * inter + 2 * hidden)       # both weight streams + x, the FP8 intermediate (written, read), y
        desc = {'workload': f'expert MLP, {groups} local experts x <= {m_max} tokens ({rows_total} valid), {hidden} -> 2 x {inter} -> {hidden}: '
                            + ('GEMM1 with fused SwiGLU + per-token FP8 re-quantisation, then masked GEMM2' if name == 'expert_mlp' else
                               'unfused: masked GEMM1 -> torch SwiGLU -> per_token_cast_to_fp8 -> masked GEMM2')
                            + ' (hipGraph replay of 20 steps; eager beside it)', 'experts': groups, 'hidden': hidden, 'intermediate': inter}

        def check():
            x_, y, masked, w1, w2 = cases[0]
            want = torch.zeros_like(y)
            h = torch.zeros((groups, m_max, 2 * inter), device='cuda', dtype=torch.bfloat16)
            dg.m_grouped_fp8_gemm_nt_masked(x_, w1, h, masked, expected)
            act = (torch.nn.functional.silu(h[..., :inter].float()) * h[..., inter:].float()).to(torch.bfloat16)
            q, sf = per_token_cast_to_fp8(act.view(groups * m_max, inter), use_ue8m0=False)
            dg.m_grouped_fp8_gemm_nt_masked((q.view(groups, m_max, inter), sf.view(groups, m_max, inter // 128)), w2, want, masked, expected)
            return calc_diff(y.float(), want.float())
    elif name == 'dense_m128':
        # a mid-M entry of the reference's forward sweep (tests/generators.py:119-121: m = 128, (n, k) = (4096, 7168)): one weight stream
        # with too few 64-row tiles for 256 CUs -- the 64 x 32 stream tile with loader waves (round 4)
        bound = 'hbm'
        m, n, k = 128, 4096, 7168
        sets = cold_sets(n * k, sets)           # 29 MB of weights a call: 11 sets (cold: see COLD_ROTATION_BYTES)
        for i in range(sets):
            gen.reset_seed(i)
            case = gen.generate_normal(m, n, k)
            if i:
                case.a_bf16 = case.b_bf16 = None
            a = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1]))
            cases.append(case)
            calls.append(lambda a=a, c=case: dg.fp8_gemm_nt(a, c.b, c.d))
        flops = 2.0 * m * n * k
        nbytes = m * k + n * k + 4 * m * (k // 128) + 4 * (n // 128) * (k // 128) + 2 * m * n
        desc = {'workload': f'fp8_gemm_nt M={m} N={n} K={k} (mid-M entry of the reference forward sweep, tests/generators.py:119-121)', 'm': m, 'n': n, 'k': k}
        check = lambda: calc_diff(cases[0].d, cases[0].ref_d)    # noqa: E731
    elif name in ('dgrad_ktail', 'dgrad_ksplit', 'dgrad_ktail_ue8m0'):
        # two dgrad entries of the reference's dense sweep (tests/generators.py:139-145: fp8_gemm_nn with m = 4096 and the (n, k) pairs
        # swapped): K = 2112 is not a multiple of 128 (K-tail stage), n = 512 with K = 32768 fills a quarter of the chip (K split);
        # '_ue8m0': the K-tail entry with packed UE8M0 scale words (the reference's SM100 input format; MN-major B is re-majored inside the call)
        m, n, k = (4096, 512, 32768) if name == 'dgrad_ksplit' else (4096, 7168, 2112)
        packed = name == 'dgrad_ktail_ue8m0'
        for i in range(sets):
            gen.reset_seed(i)
            case = gen.generate_normal(m, n, k, True, False, use_ue8m0=packed)
            if packed:
                a, b = gen.packed_ue8m0_operand(*case.a), gen.packed_ue8m0_operand(*case.b, mn_rows=n)
                cases.append(case)
                calls.append(lambda a=a, b=b, c=case: dg.fp8_gemm_nt(a, b, c.d))
                continue
            a = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1]))
            cases.append(case)
            calls.append(lambda a=a, c=case: dg.fp8_gemm_nt(a, c.b, c.d))
        kb = -(-k // 128)
        flops = 2.0 * m * n * k
        nbytes = m * k + n * k + 4 * m * kb + 4 * (-(-n // 128)) * kb + 2 * m * n
        desc = {'workload': f'fp8_gemm_nn M={m} N={n} K={k} (dgrad entry of the reference sweep, tests/generators.py:139-145; '
                            + ('under-filled launch: K split' if name == 'dgrad_ksplit' else 'K tail on the fast path' + (', packed UE8M0 scales' if packed else '')) + ')',
                'm': m, 'n': n, 'k': k}
        check = lambda: calc_diff(cases[0].d, cases[0].ref_d)    # noqa: E731
    elif name in ('wgrad', 'wgrad_ksplit', 'wgrad_ue8m0'):
        # (wgrad_ksplit: the sweep's wgrad entry of a 576-wide layer -- 48 tiles for 256 CUs: the K pieces run as groups of one launch;
        #  wgrad_ue8m0, round 6: the same GEMM with the reference's SM100 scale format -- packed UE8M0 words per row of both operands -- on the
        #  hardware-scaled kernel: no FP32 promotion, so the (1, 1, 128) VALU roof does not apply)
        m, n, k = (576, 4096, 7168) if name == 'wgrad_ksplit' else (4096, 4096, 7168)
        packed = name == 'wgrad_ue8m0'
        for i in range(sets):
            gen.reset_seed(i)
            case = gen.generate_normal(m, n, k, accumulate=True, out_dtype=torch.float, per_token_b=True, use_ue8m0=packed)
            if packed:
                a, b = gen.packed_ue8m0_operand(*case.a), gen.packed_ue8m0_operand(*case.b)
            else:
                a = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1]))
                b = (case.b[0], dg.get_mn_major_tma_aligned_tensor(case.b[1]))
            case.a_bf16 = case.b_bf16 = None
            cases.append(case)
            calls.append(lambda a=a, b=b, c=case: dg.fp8_gemm_nt(a, b, c.d, c=c.d, recipe=(1, 1, 128)))
        flops = 2.0 * m * n * k
        nbytes = m * k + n * k + 4 * (m + n) * (k // 128) + 8 * m * n           # FP32 D read and written
        desc = {'workload': f'fp8_gemm_nt recipe (1, 1, 128), FP32 accumulate into D, M={m} N={n} K={k} (wgrad form, tests/generators.py:146-153)' +
                            (', packed UE8M0 scales (hardware-scaled MFMA)' if packed else ''),
                'm': m, 'n': n, 'k': k}
        if not packed:
            desc['recipe_roof'] = RECIPE_1_1_128_ROOF
        check = lambda: float('nan')                               # noqa: E731  (D keeps accumulating: parity is the tests' job)
    elif name in ('kgrouped_ue8m0', 'kgrouped_ue8m0_g32'):
        # the reference's SM100 form of the K-grouped GEMM (round 6): MN-major operands, UE8M0 scales of granularity 128 / 32 handed over as packed
        # words; one launch of the hardware-scaled K-grouped kernel, which reads the MN-major operands in place
        import random
        g, m, n, ek = 8, 4096, 7168, 4096
        gran_k = 32 if name.endswith('_g32') else 128
        random.seed(0)
        ks = [max(128, int(ek * random.uniform(0.7, 1.3)) // 128 * 128) for _ in range(g)]
        for i in range(min(sets, 2)):
            gen.reset_seed(i)
            case = gen.generate_k_grouped_contiguous_ue8m0(g, m, n, ks, gran_k)
            a = (case.a[0], gen.pack_k_grouped_ue8m0(case.a[1], ks, gran_k))
            b = (case.b[0], gen.pack_k_grouped_ue8m0(case.b[1], ks, gran_k))
            case.a_groups = case.b_groups = None
            cases.append(case)
            calls.append(lambda a=a, b=b, c=case: dg.k_grouped_fp8_gemm_tn_contiguous(a, b, c.d, c.ks, c.grouped_layout, c=c.d, recipe=(1, 1, gran_k)))
        flops = 2.0 * m * n * sum(ks)
        nbytes = (m + n) * sum(ks) * (1 + 1 / gran_k) + 8.0 * g * m * n
        desc = {'workload': f'k_grouped_fp8_gemm_tn_contiguous G={g} M={m} N={n} sum_k={sum(ks)}, packed UE8M0 scales of granularity {gran_k} '
                            '(the reference\'s SM100 form, tests/generators.py:190-213; MN-major operands in place)',
                'm': m, 'n': n, 'sum_k': sum(ks), 'groups': g}
        check = lambda: float('nan')                               # noqa: E731
    elif name == 'kgrouped':
        import random
        g, m, n, ek = 8, 4096, 7168, 4096
        random.seed(0)
        ks = [max(128, int(ek * random.uniform(0.7, 1.3)) // 128 * 128) for _ in range(g)]
        for i in range(min(sets, 2)):
            gen.reset_seed(i)
            case = gen.generate_k_grouped_contiguous(g, m, n, ks, True)
            cases.append(case)
            calls.append(lambda c=case: dg.k_grouped_fp8_gemm_nt_contiguous(c.a, c.b, c.d, c.ks, c.grouped_layout, c=c.d))
        flops = 2.0 * m * n * sum(ks)
        nbytes = (m + n) * sum(ks) * (1 + 4 / 128) + 8.0 * g * m * n
        desc = {'workload': f'k_grouped_fp8_gemm_nt_contiguous G={g} M={m} N={n} sum_k={sum(ks)} (reference sweep entry, tests/generators.py:200-202)',
                'm': m, 'n': n, 'sum_k': sum(ks), 'groups': g, 'recipe_roof': RECIPE_1_1_128_ROOF}
        check = lambda: float('nan')                               # noqa: E731
    elif name in ('contiguous', 'contiguous_ue8m0'):
        # (contiguous_ue8m0, round 5: C4 with packed UE8M0 scales -- the group-relative tiling of the hardware-scaled kernels)
        groups, expected, n, k = 8, 512, 4096, 7168
        packed = name == 'contiguous_ue8m0'
        for i in range(sets):
            gen.reset_seed(i)
            case = gen.generate_m_grouped_contiguous(groups, expected, n, k, use_ue8m0=packed)
            cases.append(case)
            if packed:
                a, b = gen.packed_ue8m0_operand(*case.a), gen.packed_ue8m0_operand(*case.b, mn_rows=n)
                calls.append(lambda a=a, b=b, c=case: dg.m_grouped_fp8_gemm_nt_contiguous(a, b, c.d, c.grouped_layout))
            else:
                case.a = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1]))
                calls.append(lambda c=case: dg.m_grouped_fp8_gemm_nt_contiguous(c.a, c.b, c.d, c.grouped_layout))
        m = cases[0].m
        flops = 2.0 * m * n * k                                     # the reference's own count: padded M (tests/test_fp8_fp4.py:124)
        nbytes = count_bytes(cases[0].a, cases[0].b, cases[0].d)
        valid_rows = int((cases[0].grouped_layout >= 0).sum().item())      # rows that belong to a group (the rest is alignment padding)
        desc = {'workload': f'm_grouped_fp8_gemm_nt_contiguous G={groups} M_total={m} N={n} K={k} (BASELINE configs[3])' + (', packed UE8M0 scales' if packed else ''),
                'm': m, 'n': n, 'k': k, 'groups': groups, 'valid_rows': valid_rows, 'useful_flops': 2.0 * valid_rows * n * k}
        check = lambda: calc_diff(cases[0].d, cases[0].ref_d)    # noqa: E731
    else:
        bound = 'hbm'
        packed = name == 'masked_ue8m0'
        groups, max_m, expected, n, k = 8, 64, 48, 4096, 7168
        for i in range(sets):
            gen.reset_seed(i)
            case = gen.generate_m_grouped_masked(groups, max_m, expected, n, k, masked_ms=None, use_ue8m0=packed)
            if packed:      # power-of-two scales as packed UE8M0 words (the reference's SM100 input format): hardware-scaled MFMA
                a, b = gen.packed_ue8m0_operand(*case.a), gen.packed_ue8m0_operand(*case.b, mn_rows=n)
            else:
                a, b = (case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1])), case.b
            cases.append(case)
            calls.append(lambda a=a, b=b, c=case: dg.m_grouped_fp8_gemm_nt_masked(a, b, c.d, c.masked_m, expected))
        valid = int(cases[0].masked_m.sum().item())
        flops = 2.0 * valid * n * k
        nbytes = count_bytes(cases[0].a, cases[0].d) * valid / (max_m * groups) + count_bytes(cases[0].b)
        desc = {'workload': f'm_grouped_fp8_gemm_nt_masked G={groups} (64 experts / 8 GPUs) M<=64 N={n} K={k} (BASELINE configs[4], one rank)' +
                            (', packed UE8M0 scales' if packed else ''),
                'valid_m': valid, 'n': n, 'k': k, 'groups': groups}

        def check():
            c = cases[0]
            return max(calc_diff(c.d[g, :int(r)], c.ref_d[g, :int(r)]) for g, r in enumerate(c.masked_m.tolist()) if r)
    return calls, flops, float(nbytes), desc, check, bound


def roofline_record(flops: float, nbytes: float, kernel_s: float, bound: str, kernel: str, counters=None, useful_flops=None, recipe_roof=None):
    tflops, gbs = flops / kernel_s / 1e12, nbytes / kernel_s / 1e9
    rec = ({'bound': 'mfma', 'achieved': tflops, 'peak': PEAK_FP8_TFLOPS, 'unit': 'TFLOP/s', 'frac': tflops / PEAK_FP8_TFLOPS} if bound == 'mfma' else
           {'bound': 'hbm', 'achieved': gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': gbs / PEAK_HBM_GBS})
    counters = counters or {}
    rec.update({'traffic': counters.get('traffic'), 'mfma_busy': counters.get('mfma_busy') and round(counters['mfma_busy'], 4),
                'clock_ghz': counters.get('clock_ghz') and round(counters['clock_ghz'], 3),
                'counters_git': counters.get('counters_git'), 'kernel': kernel, 'kernel_us': kernel_s * 1e6, 'algorithmic_flops': flops,
                'algorithmic_bytes': nbytes, 'tflops': tflops, 'gbs': gbs})
    if useful_flops is not None:        # layouts with padding rows: the fraction on the rows that carry data, beside the reference-style count
        rec.update({'useful_flops': useful_flops, 'frac_useful': useful_flops / kernel_s / 1e12 / PEAK_FP8_TFLOPS})
    if recipe_roof is not None:         # recipe (1, 1, 128): the fraction of that recipe's own VALU-issue roof
        rec.update({'recipe_roof_frac_of_peak': recipe_roof, 'frac_of_recipe_roof': tflops / (PEAK_FP8_TFLOPS * recipe_roof)})
    return rec


def _sig(x: float, digits: int = 4) -> float:
    """x rounded to `digits` significant digits (compact records)."""
    return float(f'{x:.{digits}g}')


def graph_replay_seconds(calls, per_graph: int) -> float:
    """Seconds per call of `per_graph` calls captured into one hipGraph and replayed (every operator call is stream-ordered and
    capturable): what a decode loop that replays its step as a graph pays -- kernel plus kernel boundary, no host path."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(2):
            calls[i % len(calls)]()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(per_graph):
            calls[i % len(calls)]()
    # (capturing leaves the GPU idle for a few milliseconds and its clocks fall back: the first ~11 replays -- ~7 ms of load -- of a fresh graph
    #  run 30.4 us per call, every later one 27.4, the back-to-back eager rate; tools/decode_replay_probe2.py,
    #  profiles/r05_probe/decode_graph_replay_vs_eager.log.  Warm up by TIME, as the main loop does, then measure)
    t_warm = time.perf_counter() + 0.05
    while time.perf_counter() < t_warm:
        for _ in range(4):
            graph.replay()
        torch.cuda.synchronize()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(20):
        graph.replay()
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) / 1e3 / (20 * per_graph)


def run_secondary(sets: int):
    """The other configurations, shortened (about 0.3 s of launches each after a short warm-up): HIP-event time per operator
    call and its roofline fraction (MN-major operands are read as they are: one kernel per call)."""
    out = []
    for name in SECONDARY:
        try:
            calls, flops, nbytes, desc, check, bound = make_workload(name, 2 if not name.startswith('masked') else sets)
            calls[0]()
            torch.cuda.synchronize()
            diff = check()
            t_end = time.perf_counter() + 0.25
            i = 0
            while time.perf_counter() < t_end:
                for _ in range(4):
                    calls[i % len(calls)]()
                    i += 1
                torch.cuda.synchronize()
            steps = 40 if flops < 4e11 else 10
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
            for i in range(steps):
                calls[i % len(calls)]()
            end.record()
            torch.cuda.synchronize()
            call_s = start.elapsed_time(end) / 1e3 / steps
            extra = {}
            if name in GRAPHED:
                extra['eager_call_us'] = call_s * 1e6
                call_s = graph_replay_seconds(calls, 20)
            if name == 'expert_mlp':
                other = make_workload('expert_mlp_unfused', 2)[0]
                other[0]()
                extra['unfused_us'] = graph_replay_seconds(other, 20) * 1e6
                extra['fused_us'] = call_s * 1e6
                other = None
            rec = {'workload': desc['workload'], 'steps': steps, 'calc_diff_vs_reference_expr': diff,
                   'input_sets': len(calls),
                   'roofline': roofline_record(flops, nbytes, call_s, bound, dg.last_config(), measured_counters(dg.last_config(), name),
                                               useful_flops=desc.get('useful_flops'), recipe_roof=desc.get('recipe_roof')), **extra}
            out.append(rec)
        except Exception as e:                                       # noqa: BLE001  (a secondary line must not take the headline down)
            out.append({'workload': name, 'error': f'{type(e).__name__}: {e}'[:200]})
        finally:
            calls = None
            torch.cuda.empty_cache()
    return out


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
            'sample': f'full {m}x{n}x{k}, reference test expr (generators.py:312), min of 2 runs, {best:.2f} s each'}


def run(rank: int, world: int, local_rank: int, args):
    torch.cuda.set_device(local_rank)
    distributed = world > 1
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if not dist.is_initialized():
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))

    dg.set_forced_config(args.config)
    phase_events = []
    calls, flops, nbytes, desc, check, bound = make_workload(args.workload, args.sets, world, rank, phase_events, args.ep_capacity)

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
    phase_events.clear()

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
    split = None
    if phase_events:                        # EP step: HIP-event time of dispatch / local GEMM / combine (+ reduce), mean over the steps
        sums = [0.0, 0.0, 0.0]
        for e0, e1, e2, e3 in phase_events:
            sums[0] += e0.elapsed_time(e1)
            sums[1] += e1.elapsed_time(e2)
            sums[2] += e2.elapsed_time(e3)
        split = [s / len(phase_events) * 1e3 for s in sums]
    if distributed:
        dist.barrier()
        t = torch.tensor([elapsed] + (split or [0.0, 0.0, 0.0]), device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
        if split is not None:
            split = [float(x) for x in t[1:].tolist()]
    kernel_s = start.elapsed_time(end) / 1e3 / args.steps       # average launch duration of the dominant kernel (EP: of the whole step)

    if rank == 0:
        total_flops = flops * args.steps * world
        value = total_flops / elapsed / 1e12
        roofline = roofline_record(flops, nbytes, kernel_s, bound, dg.last_config(), measured_counters(dg.last_config(), args.workload), useful_flops=desc.get('useful_flops'),
                                   recipe_roof=desc.get('recipe_roof'))
        if split is not None:
            # the roofline of the EP step is that of its local GEMM (HBM-bound on the expert weights)
            roofline = roofline_record(flops, nbytes, split[1] / 1e6, bound, dg.last_config(), None)
        headline = args.workload == 'dense'
        line = {
            'metric': 'achieved FP8 TFLOPS (and % of MFMA roofline) for fp8_gemm_nt M=4096 N=4096 K=7168'
                      if headline else f'achieved FP8 TFLOPS for the {args.workload} workload',
            'value': value, 'unit': 'TFLOPS', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'fp8_e4m3 (fp32 accumulate, bf16 out)', 'data': 'synthetic',
            'config': dict(desc, parallelism=('single GPU' if world == 1 else f'ep{world}' if args.workload == 'masked' else 'replicas only'),
                           kernel=dg.last_config(), input_sets=len(calls)),
            'pct_of_mfma_peak': 100.0 * value / world / PEAK_FP8_TFLOPS,
            'calc_diff_vs_reference_expr': diff,
            'roofline': roofline,
        }
        if split is not None:
            line['ep_time_split_us'] = {'dispatch_all_to_all': split[0], 'local_masked_gemm': split[1], 'combine_all_to_all_and_topk_reduce': split[2],
                                        'note': 'HIP events on the launch stream, mean per step, max over ranks'}
        if world == 1 and headline and not args.no_secondary:
            detail = run_secondary(args.sets)
            # The driver keeps the last 2000 characters of stdout: the full records go out FIRST, on a line of their own that is not
            # a JSON line (prefix), the headline line carries {workload: [roofline.frac, us per call(, 'h' = HBM-bound)(, second frac)]} and comes LAST.
            print('secondary_detail: ' + json.dumps(detail), flush=True)
            line['secondary'] = {name: ([_sig(rec['roofline']['frac'], 3), _sig(rec['roofline']['kernel_us'])] +
                                        (['h'] if rec['roofline']['bound'] == 'hbm' else []) +
                                        ([_sig(rec['roofline']['frac_useful'], 3)] if 'frac_useful' in rec['roofline'] else []) +
                                        ([_sig(rec['roofline']['frac_of_recipe_roof'], 3)] if 'frac_of_recipe_roof' in rec['roofline'] else [])
                                        if 'roofline' in rec else rec.get('error', '?')[:40])
                                 for name, rec in zip(SECONDARY, detail)}
            line['secondary_key'] = "[frac of 5 PF ('h': of 8 TB/s), us/call(, frac on data rows | of (1,1,128) roof)]"
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args.workload)
        if 'secondary' in line:         # keep the whole headline line inside the driver's 2000-character tail
            line['roofline'] = {k: (_sig(v, 6) if isinstance(v, float) else v) for k, v in roofline.items()
                                if k not in ('tflops', 'gbs', 'algorithmic_flops', 'algorithmic_bytes')}
            line['pct_of_mfma_peak'] = _sig(line['pct_of_mfma_peak'], 5)
            line['value'], line['ms_per_step'] = _sig(line['value'], 7), _sig(line['ms_per_step'], 6)
            if 'cpu_baseline' in line:
                line['cpu_baseline']['value'] = _sig(line['cpu_baseline']['value'], 4)
            line['calc_diff_vs_reference_expr'] = _sig(diff, 3)
        text = json.dumps(line, separators=(',', ':'))
        if len(text) > 1980 and 'secondary_key' in line:       # (the key is documented in DESIGN.md section 6 as well)
            del line['secondary_key']
            text = json.dumps(line, separators=(',', ':'))
        print(text, flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def _spawned(local_rank: int, world: int, port: int, args):
    os.environ.update({'RANK': str(local_rank), 'LOCAL_RANK': str(local_rank), 'WORLD_SIZE': str(world),
                       'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port)})
    run(local_rank, world, local_rank, args)


def main():
    args = parse_args()
    if 'WORLD_SIZE' in os.environ:          # launched by torch.distributed.run: one process per GPU already exists
        world = int(os.environ['WORLD_SIZE'])
        if args.gpus != world:
            raise SystemExit(f'--gpus {args.gpus} does not match WORLD_SIZE {world}')
        run(int(os.environ.get('RANK', '0')), world, int(os.environ.get('LOCAL_RANK', '0')), args)
        return
    if args.gpus <= 1:
        run(0, 1, 0, args)
        return
    # plain `python bench.py --gpus N`: spawn the N ranks here (as the reference's multi-GPU test does,
    # tests/test_mega_moe.py:312 + deep_gemm/utils/dist.py:10-35) -- and refuse loudly if the node has fewer GPUs
    have = torch.cuda.device_count()
    if have < args.gpus:
        raise SystemExit(f'bench.py --gpus {args.gpus}: only {have} GPU(s) visible on this node')
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_spawned, args=(args.gpus, port, args), nprocs=args.gpus, join=True)


if __name__ == '__main__':
    main()
