#!/usr/bin/env python3
"""Same-box comparator for the C2 headline (fp8_gemm_nt 4096 x 4096 x 7168): this library's kernels next to a third-party FP8 GEMM
(hipBLASLt through ``torch._scaled_mm``) on the SAME reference-quantised bytes, same burst method (back-to-back launches, HIP
events, rotating input sets).  Test-side evidence only -- the product never calls hipBLASLt.

    python tools/ceiling.py [--burst 200] [--rounds 3] [--dump-operands FILE]

Arms (each reported as us per launch, TFLOPS and fraction of the 5 PF dense FP8 peak; the BF16 arm against 2.5 PF):
  dg_fp32_scales     this library, FP32 1x128 / 128x128 scales (the BASELINE headline, duo_p_256x256)
  dg_ue8m0           this library, power-of-two scales as packed UE8M0 words (hardware-scaled MFMA, e8_quad_256x256)
  hipblaslt_tensor   torch._scaled_mm, one FP32 scale per tensor      (NOT the same arithmetic: no per-block scales)
  hipblaslt_rowwise  torch._scaled_mm, one FP32 scale per row of A / B (NOT the same arithmetic)
  hipblaslt_block    torch._scaled_mm with 1x128 / 128x128 FP32 block scales, if this build exposes it
  hipblaslt_mx       torch._scaled_mm with E8M0 1x32 block scales (MX-FP8), if this build exposes it
  hipblaslt_bf16     torch.mm on the BF16 originals
  *_zeros            the same third-party arm on zero-filled operands (DVFS: how much the clock gives back on trivial data)
``--dump-operands FILE`` writes 256 KiB of the quantised operand bytes (A rows, then B rows) for tools/ubench/mfma_rate.
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgemm_amd as dg                                              # noqa: E402
from deepgemm_amd.testing import generators as gen                      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--burst', type=int, default=200)
ap.add_argument('--rounds', type=int, default=3)
ap.add_argument('--sets', type=int, default=4)
ap.add_argument('--dump-operands', default='')
ap.add_argument('--only', default='', help='comma-separated arm names (counter passes: fewer dispatches to sort through)')
args = ap.parse_args()

M, N, K = 4096, 4096, 7168
FLOPS = 2.0 * M * N * K
print(json.dumps({'torch': torch.__version__, 'hip': torch.version.hip, 'device': torch.cuda.get_device_name(0),
                  'cus': torch.cuda.get_device_properties(0).multi_processor_count}), flush=True)

cases, cases_e8 = [], []
for i in range(args.sets):
    gen.reset_seed(i)
    c = gen.generate_normal(M, N, K)
    c.a = (c.a[0], dg.get_mn_major_tma_aligned_tensor(c.a[1]))
    cases.append(c)
    gen.reset_seed(i)
    e = gen.generate_normal(M, N, K, use_ue8m0=True)
    cases_e8.append((gen.packed_ue8m0_operand(*e.a), gen.packed_ue8m0_operand(*e.b, mn_rows=N), e.d))
    del e

if args.dump_operands:
    a_bytes = cases[0].a[0].view(torch.uint8)[:32].reshape(-1)[:128 * 1024]
    b_bytes = cases[0].b[0].view(torch.uint8)[:32].reshape(-1)[:128 * 1024]
    raw = torch.cat([a_bytes, b_bytes]).cpu().numpy().tobytes()
    with open(args.dump_operands, 'wb') as f:
        f.write(raw)
    hist = torch.bincount(cases[0].a[0].view(torch.uint8).reshape(-1)[:1 << 24].int() & 0x7f, minlength=128).float()
    print(json.dumps({'dumped': args.dump_operands, 'bytes': len(raw),
                      'a_exponent_field_histogram_pct': [round(100 * float(hist[e * 8:(e + 1) * 8].sum() / hist.sum()), 2) for e in range(16)]}), flush=True)

arms, kernels = {}, {}


def add(name, fn, peak=5000.0, note=''):
    if args.only and name not in args.only.split(','):
        return
    try:
        fn(0)
        torch.cuda.synchronize()
        arms[name] = (fn, peak, note)
        kernels[name] = dg.last_config() if name.startswith('dg_') else 'hipBLASLt (torch._scaled_mm / torch.mm)'
    except Exception as exc:                                            # noqa: BLE001
        print(json.dumps({'arm': name, 'unavailable': f'{type(exc).__name__}: {exc}'[:300]}), flush=True)


add('dg_fp32_scales', lambda i: dg.fp8_gemm_nt(cases[i % len(cases)].a, cases[i % len(cases)].b, cases[i % len(cases)].d))
add('dg_ue8m0', lambda i: dg.fp8_gemm_nt(cases_e8[i % len(cases_e8)][0], cases_e8[i % len(cases_e8)][1], cases_e8[i % len(cases_e8)][2]))

one = torch.ones((), dtype=torch.float, device='cuda')
out = torch.empty((M, N), dtype=torch.bfloat16, device='cuda')
aq = [c.a[0] for c in cases]
bq_t = [c.b[0].t() for c in cases]                                     # [K, N] column-major view: what _scaled_mm wants for mat2
row_a = [torch.rand((M, 1), device='cuda') + 0.5 for _ in cases]
row_b = [torch.rand((1, N), device='cuda') + 0.5 for _ in cases]
add('hipblaslt_tensor', lambda i: torch._scaled_mm(aq[i % len(aq)], bq_t[i % len(aq)], scale_a=one, scale_b=one, out_dtype=torch.bfloat16, out=out),
    note='one scale per tensor')
add('hipblaslt_rowwise', lambda i: torch._scaled_mm(aq[i % len(aq)], bq_t[i % len(aq)], scale_a=row_a[i % len(aq)], scale_b=row_b[i % len(aq)],
                                                     out_dtype=torch.bfloat16, out=out), note='one scale per row of A / B')
sfa_rm = [c.a[1].contiguous() if c.a[1].stride(-1) != 1 else c.a[1] for c in cases]           # [M, K/128]
sfa_rm = [torch.empty((M, K // 128), device='cuda').copy_(s) for s in sfa_rm]
sfb_t = [c.b[1].t() for c in cases]                                                          # [K/128, N/128]
add('hipblaslt_block', lambda i: torch._scaled_mm(aq[i % len(aq)], bq_t[i % len(aq)], scale_a=sfa_rm[i % len(aq)], scale_b=sfb_t[i % len(aq)],
                                                   out_dtype=torch.bfloat16, out=out), note='1x128 / 128x128 FP32 block scales (DeepSeek recipe)')
if hasattr(torch, 'float8_e8m0fnu'):
    mxa = [torch.full((M, K // 32), 127, dtype=torch.uint8, device='cuda').view(torch.float8_e8m0fnu) for _ in cases]
    mxb = [torch.full((N, K // 32), 127, dtype=torch.uint8, device='cuda').view(torch.float8_e8m0fnu) for _ in cases]
    add('hipblaslt_mx', lambda i: torch._scaled_mm(aq[i % len(aq)], bq_t[i % len(aq)], scale_a=mxa[i % len(aq)], scale_b=mxb[i % len(aq)],
                                                    out_dtype=torch.bfloat16, out=out), note='MX-FP8: E8M0 scale per 32 K elements')
a16 = [c.a_bf16 for c in cases]
b16_t = [c.b_bf16.t() for c in cases]
add('hipblaslt_bf16', lambda i: torch.mm(a16[i % len(a16)], b16_t[i % len(a16)], out=out), peak=2500.0, note='BF16 originals, vs 2.5 PF')
za, zb_t = torch.zeros((M, K), device='cuda').to(torch.float8_e4m3fn), torch.zeros((N, K), device='cuda').to(torch.float8_e4m3fn).t()
add('hipblaslt_tensor_zeros', lambda i: torch._scaled_mm(za, zb_t, scale_a=one, scale_b=one, out_dtype=torch.bfloat16, out=out),
    note='zero-filled operands (DVFS give-back)')
z16 = torch.zeros((M, K), device='cuda', dtype=torch.bfloat16)
add('hipblaslt_bf16_zeros', lambda i: torch.mm(z16, z16[:N].t(), out=out), peak=2500.0, note='zero-filled operands, vs 2.5 PF')

# sanity of the comparator arms' arithmetic where it is comparable (tensor-wise with unit scales = plain FP8 product)
if 'hipblaslt_tensor' in arms:
    arms['hipblaslt_tensor'][0](0)
    want = (cases[0].a[0][:64].float() @ cases[0].b[0].float().t()).to(torch.bfloat16)
    print(json.dumps({'check': 'hipblaslt_tensor rows 0..63 vs torch fp32 product', 'max_abs_diff': float((out[:64].float() - want.float()).abs().max()),
                      'max_abs': float(want.float().abs().max())}), flush=True)

t_end = time.time() + (2.0 if not args.only else 0.3)
first = next(iter(arms.values()))[0]
while time.time() < t_end:
    for i in range(8):
        first(i)
    torch.cuda.synchronize()

times = {name: [] for name in arms}
for r in range(args.rounds):
    for name, (fn, peak, note) in arms.items():
        for it in range(20):
            fn(it)
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for it in range(args.burst):
            fn(it)
        end.record()
        torch.cuda.synchronize()
        times[name].append(start.elapsed_time(end) / args.burst * 1e3)
for name, (fn, peak, note) in arms.items():
    us = statistics.median(times[name])
    print(json.dumps({'arm': name, 'us_per_launch': [round(t, 1) for t in times[name]], 'us_median': round(us, 1),
                      'tflops': round(FLOPS / us / 1e6, 1), 'peak_tflops': peak, 'frac_of_peak': round(FLOPS / us / 1e6 / peak, 3),
                      'kernel': kernels[name], 'note': note}), flush=True)
