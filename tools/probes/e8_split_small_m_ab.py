#!/usr/bin/env python3
"""Packed-scale dense problems of 129 .. 256 rows outside the in-kernel K split's rule: the two-launch split (e8_quad_ks_*) priced against the 128-row
kernel only (DG_E8_SPLIT_QUAD_MODEL_ONLY=1, the rule before the end of round 6) against the rule that prices the stream tiles -- eager, cold sets.
python tools/probes/e8_split_small_m_ab.py"""
import os, sys
sys.path.insert(0, '.')
import torch, deepgemm_amd as dg
from deepgemm_amd._lib import lib
from deepgemm_amd import gemm as gemm_mod
from deepgemm_amd.testing import calc_diff
from deepgemm_amd.utils.math import pack_ue8m0_to_int, per_token_cast_to_fp8


def time_us(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def packed(x, mn, k, gran=128):
    q = per_token_cast_to_fp8(x, True, gran)
    return q[0], dg.transform_sf_into_required_layout(pack_ue8m0_to_int(q[1]), mn, k, (1, gran))


for m, n, k in ((192, 2112, 7168), (256, 2112, 8192), (192, 1536, 7168), (256, 3072, 7168), (160, 7168, 8192), (256, 7168, 16384), (192, 4608, 12288)):
    sets = max(4, min(32, int(320e6 // (n * k)) + 1))
    ops = []
    for i in range(sets):
        torch.manual_seed(i)
        a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16); b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
        ops.append((packed(a, m, k), packed(b, n, k), torch.empty((m, n), device='cuda', dtype=torch.bfloat16)))
    out, ref = [], None
    for old in (True, False, True, False):
        if old: os.environ['DG_E8_SPLIT_QUAD_MODEL_ONLY'] = '1'
        else: os.environ.pop('DG_E8_SPLIT_QUAD_MODEL_ONLY', None)
        lib.dg_reload_env(); gemm_mod._VALIDATED_PACKED.clear()
        dg.fp8_gemm_nt(*ops[0])
        name = dg.last_config()
        res = ops[0][2].float().clone()
        if ref is None: ref = res
        it = [0]
        def call():
            o = ops[it[0] % sets]; it[0] += 1
            dg.fp8_gemm_nt(*o)
        out.append(f'{"before" if old else "now"}={name} {time_us(call):.1f} ({calc_diff(res, ref):.1e})')
    print(f'{m} x {n} x {k} ({sets} sets): ' + ' | '.join(out), flush=True)
    del ops
os.environ.pop('DG_E8_SPLIT_QUAD_MODEL_ONLY', None); lib.dg_reload_env()
