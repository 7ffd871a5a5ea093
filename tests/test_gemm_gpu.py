"""GPU parity tests of the FP8 GEMM path: the HIP kernels (through the public operators / C ABI) against the CPU
oracle on the same seeded inputs, the committed golden fixtures, the reference's own gate, and size-independent
properties at BASELINE.json's full sizes.  Modeled on the reference's tests/test_fp8_fp4.py."""
import os
import random

import pytest
import torch

import deepgemm_amd as dg
import oracle
from deepgemm_amd._lib import lib as dg_lib
from deepgemm_amd.testing import calc_diff, generators as gen
from gpu_helpers import assert_close_fp32, assert_close_to_oracle, cpu_pair, oracle_dense

pytestmark = pytest.mark.gpu
FAST = ['stream_64x128', 'stream_nt_64x128', 'stream2_64x128', 'stream_nt2_64x128', 'stream_64x32', 'stream_l8_64x32', 'duo_256x256', 'duo_p_256x256', 'duo_128x256', 'pipe_256x256', 'pipe_128x256', 'pipe_128x128', 'pipe_64x256', 'pipe_32x256', 'pipe_16x256']
# (superseded forms and ablation variants -- ring, naive, pipe_s*, dabl* ... -- exist only in DG_EXPERIMENTS builds of the library)


@pytest.fixture(autouse=True)
def _auto_config():
    dg.set_forced_config('auto')
    dg.set_mk_alignment_for_contiguous_layout(128)
    yield
    dg.set_forced_config('auto')


def test_c1_unit_scale_exact(golden_gemm):
    """BASELINE.json config 1 on the GPU: integer operands, unit scales -> bit-exact BF16 torch.matmul, every config."""
    a, b = golden_gemm.fp8('c1_a_q').cuda(), golden_gemm.fp8('c1_b_q').cuda()
    sfa, sfb = torch.ones(128, 4, device='cuda'), torch.ones(1, 4, device='cuda')
    want = golden_gemm.bf16('c1_ref_d')
    for cfg in FAST + ['generic_128x128', 'auto']:
        dg.set_forced_config(cfg)
        d = torch.full((128, 128), float('nan'), device='cuda', dtype=torch.bfloat16)
        dg.fp8_gemm_nt((a, sfa), (b, sfb), d)
        assert torch.equal(d.cpu(), want), cfg


def test_golden_fixtures(golden_gemm):
    g = golden_gemm
    for name in ['g64x192x384', 'g33x200x256', 'g128x128x1024', 'g96x136x640_f32']:
        fp32 = name.endswith('_f32')
        a, sfa, b, sfb = g.fp8(f'{name}_a_q'), g.raw(f'{name}_sfa'), g.fp8(f'{name}_b_q'), g.raw(f'{name}_sfb')
        stored = g.raw(f'{name}_oracle_d') if fp32 else g.bf16(f'{name}_oracle_d')
        ref_d = g.raw(f'{name}_ref_d') if fp32 else g.bf16(f'{name}_ref_d')
        for cfg in ('auto', 'generic_128x128'):
            dg.set_forced_config(cfg)
            d = torch.empty(stored.shape, device='cuda', dtype=stored.dtype)
            dg.fp8_gemm_nt((a.cuda(), sfa.cuda()), (b.cuda(), sfb.cuda()), d)
            (assert_close_fp32 if fp32 else assert_close_to_oracle)(d, stored, f'{name}/{cfg}')
            assert calc_diff(d.cpu(), ref_d) < gen.FP8_MAX_DIFF


@pytest.mark.parametrize('m,n,k', [(1, 4096, 7168), (7, 576, 2048), (16, 2112, 7168), (17, 4096, 4096), (32, 2048, 8192), (1, 16, 128), (3, 64, 640)])
def test_skinny_decode_kernel(m, n, k):
    """Decode batches (M <= 32) on the skinny weight-stream kernel (16 columns per workgroup, the 8 waves split K, the partial tiles are
    summed in wave order): oracle parity for BF16, FP32 and accumulating outputs, bit-repeatable, wider D rows, row-major SFA."""
    gen.reset_seed(m + n)
    cfg = 'skinny_16' if m <= 16 else 'skinny_32'
    case = gen.generate_normal(m, n, k)
    want = oracle_dense(case)
    dg.fp8_gemm_nt(case.a, case.b, case.d)                        # automatic pick first (short K loops stay on the stream tiles)
    assert_close_to_oracle(case.d, want, f'auto: {dg.last_config()}')
    if k >= 2048 and (m <= 16 or (4096 <= k <= 8192 and n <= 4608)):
        assert dg.last_config() in (cfg, cfg + 'c', cfg + 'ca'), dg.last_config()          # ('c' / 'ca': the coalesced-load forms, round 5)
    dg.set_forced_config(cfg)
    wide = torch.full((m, n + 24), float('nan'), device='cuda', dtype=torch.bfloat16)
    d = wide[:, :n]
    dg.fp8_gemm_nt(case.a, case.b, d)
    assert dg.last_config() == cfg
    assert_close_to_oracle(d, want, cfg)
    assert bool(torch.isnan(wide[:, n:]).all())
    assert calc_diff(d, case.ref_d) < gen.FP8_MAX_DIFF or m * n < 4096     # (tiny outputs: the reference gate is noise-limited)
    again = torch.empty_like(case.d)
    dg.fp8_gemm_nt((case.a[0], dg.get_mn_major_tma_aligned_tensor(case.a[1])), case.b, again)
    assert torch.equal(again, d.contiguous())                     # MN-major SFA, repeat: same bits
    acc_case = gen.generate_normal(m, n, k, accumulate=True, out_dtype=torch.float)
    c_cpu = acc_case.c.cpu().clone()
    dg.fp8_gemm_nt(acc_case.a, acc_case.b, acc_case.d, c=acc_case.c)
    assert_close_fp32(acc_case.d, oracle_dense(acc_case, c_cpu=c_cpu), f'{cfg} fp32 accumulate')
    with pytest.raises(RuntimeError, match='m <= its row count'):
        big = gen.generate_normal(40, n, k)
        dg.fp8_gemm_nt(big.a, big.b, big.d)


@pytest.mark.parametrize('m,n,k', [(1, 4096, 7168), (1, 7168, 16384), (9, 4112, 2048), (16, 2112, 7168), (24, 1024, 8192), (3, 64, 640), (5, 4100, 1024)])
def test_skinny_coalesced_weight_loads_same_bits(m, n, k):
    """Round 5: the skinny kernels with weight loads in a coalesced lane order (8 rows x 128 contiguous bytes per instruction) that reach the
    MFMA operand layout through wave-private LDS (`skinny_16c / _32c / _16wc`): the same operands in the same order, so the same bits as the
    register-direct forms -- BF16 and accumulating FP32 outputs, ragged N (rows clamped at n - 1), against the oracle."""
    gen.reset_seed(m + n + k)
    case = gen.generate_normal(m, n, k)
    want = oracle_dense(case)
    pairs = [('skinny_16', 'skinny_16c'), ('skinny_16', 'skinny_16ca')] if m <= 16 else []
    pairs += [('skinny_32', 'skinny_32c'), ('skinny_32', 'skinny_32ca')]          # ('ca': coalesced activation loads as well)
    pairs += [('skinny_16w', 'skinny_16wc')] if m <= 16 and n % 4 == 0 else []
    for plain, coal in pairs:
        outs = []
        for cfg in (plain, coal):
            dg.set_forced_config(cfg)
            wide = torch.full((m, n + 24), float('nan'), device='cuda', dtype=torch.bfloat16)
            dg.fp8_gemm_nt(case.a, case.b, wide[:, :n])
            assert dg.last_config() == cfg
            assert bool(torch.isnan(wide[:, n:]).all())
            outs.append(wide[:, :n].contiguous())
        dg.set_forced_config('auto')
        assert torch.equal(outs[0], outs[1]), f'{coal} differs from {plain}'
        assert_close_to_oracle(outs[1], want, coal)
    if m <= 16:
        acc_case = gen.generate_normal(m, n, k, accumulate=True, out_dtype=torch.float)
        c0 = acc_case.c.clone()
        outs = []
        for cfg in ('skinny_16', 'skinny_16c'):
            dg.set_forced_config(cfg)
            d = c0.clone()
            dg.fp8_gemm_nt(acc_case.a, acc_case.b, d, c=d)
            outs.append(d)
        dg.set_forced_config('auto')
        assert torch.equal(outs[0], outs[1])


DENSE_SHAPES = [(1, 128, 128), (7, 136, 256), (128, 2112, 512), (129, 576, 384), (256, 256, 1024), (300, 520, 384),
                (16, 4096, 512), (64, 256, 7168), (384, 768, 256)]


@pytest.mark.parametrize('m,n,k', DENSE_SHAPES)
def test_dense_nt_vs_oracle(m, n, k):
    gen.reset_seed(m * 7 + n)
    case = gen.generate_normal(m, n, k)
    want = oracle_dense(case)
    configs = ['auto', 'generic_128x128'] + FAST
    for cfg in configs:
        dg.set_forced_config(cfg)
        case.d.fill_(float('nan'))
        dg.fp8_gemm_nt(case.a, case.b, case.d)
        assert_close_to_oracle(case.d, want, f'{(m, n, k)}/{cfg}')
        assert calc_diff(case.d, case.ref_d) < gen.FP8_MAX_DIFF, cfg


@pytest.mark.parametrize('layout', ['nt', 'nn', 'tn', 'tt'])
@pytest.mark.parametrize('m,n,k', [(256, 384, 512), (130, 200, 384), (2048, 1024, 256)])
def test_dense_layouts_vs_oracle(layout, m, n, k):
    """nn / tn / tt are transposed views (csrc/apis/gemm.hpp:126-164); both the strided-view and the alias entry."""
    gen.reset_seed(1)
    a_k_major, b_k_major = layout[0] == 'n', layout[1] == 't'
    case = gen.generate_normal(m, n, k, a_k_major, b_k_major)
    want = oracle_dense(case)
    dg.fp8_gemm_nt(case.a, case.b, case.d)                                   # MN-major operands as strided views
    assert_close_to_oracle(case.d, want, f'{layout} view')
    if m * n * k >= dg.gemm.REMAJOR_MIN_MACS:                                # large: MN-major operands go to the fast kernels
        native = {'nt': '', 'nn': 'duo_bmn_', 'tt': 'duo_amn_', 'tn': 'duo_abmn_'}[layout] if m > 256 else ''
        assert not dg.last_config().startswith('generic') and dg.last_config().startswith(native), (layout, dg.last_config())
        saved, dg.gemm.REMAJOR_MIN_MACS = dg.gemm.REMAJOR_MIN_MACS, 0       # ... and the in-place layout-agnostic kernel agrees
        dg.set_forced_config('generic_128x128')
        try:
            d_generic = torch.full_like(case.d, float('nan'))
            dg.fp8_gemm_nt(case.a, case.b, d_generic)
            assert dg.last_config() == 'generic_128x128'
            assert_close_to_oracle(d_generic, want, f'{layout} generic')
        finally:
            dg.gemm.REMAJOR_MIN_MACS = saved
            dg.set_forced_config('auto')
    a = case.a if a_k_major else (case.a[0].T, case.a[1].T)
    b = case.b if b_k_major else (case.b[0].T, case.b[1].T)
    assert a[0].is_contiguous() and b[0].is_contiguous()
    case.d.fill_(float('nan'))
    getattr(dg, f'fp8_gemm_{layout}')(a, b, case.d)
    assert_close_to_oracle(case.d, want, f'{layout} alias')
    assert calc_diff(case.d, case.ref_d) < gen.FP8_MAX_DIFF


@pytest.mark.parametrize('out_dtype', [torch.bfloat16, torch.float])
@pytest.mark.parametrize('cfg', ['auto', 'generic_128x128', 'pipe_128x128', 'duo_256x256', 'duo_p_256x256', 'stream_64x128'])
def test_accumulate_and_fp32_out(out_dtype, cfg):
    gen.reset_seed(2)
    dg.set_forced_config(cfg)
    case = gen.generate_normal(200, 384, 512, accumulate=True, out_dtype=out_dtype)
    c_cpu = case.c.cpu().clone()
    want = oracle_dense(case, c_cpu=c_cpu)
    dg.fp8_gemm_nt(case.a, case.b, case.d, c=case.c)                         # c is d: accumulate in place
    if out_dtype == torch.float:
        assert_close_fp32(case.d, want, 'in place')
    else:
        assert_close_to_oracle(case.d, want, 'in place', addend=c_cpu)
    assert calc_diff(case.d, case.ref_d) < gen.FP8_MAX_DIFF
    # c in a different buffer: copied into d first (gemm.hpp:43-44), c itself untouched
    d2 = torch.empty_like(case.d)
    c2 = c_cpu.cuda()
    dg.fp8_gemm_nt(case.a, case.b, d2, c=c2)
    assert torch.equal(d2, case.d) and torch.equal(c2.cpu(), c_cpu)
    # plain FP32 output without accumulation
    if out_dtype == torch.float:
        d3 = torch.empty_like(case.d)
        dg.fp8_gemm_nt(case.a, case.b, d3)
        d3_want = torch.empty(d3.shape, dtype=torch.float)
        oracle.fp8_gemm_nt(*cpu_pair(case.a), *cpu_pair(case.b), d3_want)
        assert_close_fp32(d3, d3_want, 'fp32 out')


def test_wgrad_recipe_per_column_sfb():
    """Recipe (1, 1, 128): per-row SFA x per-column SFB, FP32 accumulate (reference sm90_fp8_gemm_1d1d)."""
    gen.reset_seed(3)
    for a_k, b_k in ((True, True), (False, False)):
        case = gen.generate_normal(192, 264, 640, a_k, b_k, accumulate=True, out_dtype=torch.float, per_token_b=True)
        want = oracle_dense(case, gran_n=1, c_cpu=case.c.cpu().clone())
        dg.fp8_gemm_nt(case.a, case.b, case.d, c=case.c, recipe=(1, 1, 128))
        assert_close_fp32(case.d, want, 'recipe (1,1,128)')
        assert calc_diff(case.d, case.ref_d) < gen.FP8_MAX_DIFF
    case = gen.generate_normal(64, 128, 256, per_token_b=True)
    want = oracle_dense(case, gran_n=1)
    dg.fp8_gemm_nt(case.a, case.b, case.d, recipe_a=(1, 128), recipe_b=(1, 128))
    assert_close_to_oracle(case.d, want, 'recipe_a/recipe_b')


def per_col_tile_name(m, n, split=False):
    """The K-major recipe-(1, 1, 128) kernel takes 192-row tiles where they mean fewer computed rows (K split: in total; otherwise per round
    of resident tiles) -- the rule of dg_api.hip's per_col_bm, 256 CUs."""
    nt = -(-n // 256)
    cost = lambda bm: -(-m // bm) * bm if split else -(-(-(-m // bm) * nt) // 256) * bm
    return ('pipe_pc_ks_' if split else 'pipe_pc_') + ('192x256' if cost(192) < cost(256) else '256x256')


@pytest.mark.parametrize('m,n,k', [(256, 256, 128), (300, 520, 896), (1024, 768, 2048), (65, 4096, 512), (576, 512, 1024), (190, 300, 384)])
@pytest.mark.parametrize('out_dtype,accumulate', [(torch.float, True), (torch.bfloat16, False)])
def test_per_column_sfb_fast_kernel(m, n, k, out_dtype, accumulate):
    """Recipe (1, 1, 128) on the LDS-DMA path: both scale vectors of a K block ride along as 1 KiB pieces, the scale
    product is formed per accumulator element (reference arithmetic: impls/sm90_fp8_gemm_1d1d.cuh:279-311)."""
    gen.reset_seed(m + n + k)
    case = gen.generate_normal(m, n, k, accumulate=accumulate, out_dtype=out_dtype, per_token_b=True)
    c_cpu = case.c.cpu().clone() if accumulate else None
    want = oracle_dense(case, gran_n=1, c_cpu=c_cpu)
    dg.fp8_gemm_nt(case.a, case.b, case.d, c=case.c if accumulate else None, recipe=(1, 1, 128))
    assert dg.last_config() == per_col_tile_name(m, n)
    if out_dtype == torch.float:
        assert_close_fp32(case.d, want, 'per-column SFB')
    else:
        assert_close_to_oracle(case.d, want, 'per-column SFB')
    assert calc_diff(case.d, case.ref_d) < gen.FP8_MAX_DIFF
    # same answer as the layout-agnostic kernel (different summation machinery, same arithmetic order)
    dg.set_forced_config('generic_128x128')
    d2 = case.c.clone() if accumulate else torch.empty_like(case.d)
    if accumulate:
        d2.copy_(c_cpu.cuda())
    dg.fp8_gemm_nt(case.a, case.b, d2, c=d2 if accumulate else None, recipe=(1, 1, 128))
    dg.set_forced_config('auto')
    assert torch.equal(d2, case.d)


@pytest.mark.parametrize('m,n,k', [(256, 256, 256), (304, 528, 896), (1024, 768, 2048)])
def test_per_column_sfb_mn_major_operands(m, n, k):
    """fp8_gemm_tn with recipe (1, 1, 128) (dense wgrad in the TN form): both operands MN-major go into the kernel as they
    are (LDS-DMA of k-rows + hardware transpose reads), same result as the K-major form."""
    gen.reset_seed(m + n + k + 1)
    case = gen.generate_normal(m, n, k, a_k_major=False, b_k_major=False, accumulate=True, out_dtype=torch.float, per_token_b=True)
    c_cpu = case.c.cpu().clone()
    want = oracle_dense(case, gran_n=1, c_cpu=c_cpu)
    dg.fp8_gemm_nt(case.a, case.b, case.d, c=case.c, recipe=(1, 1, 128))
    assert dg.last_config() == 'pipe_pc_mn_256x256'
    assert_close_fp32(case.d, want, 'per-column SFB, MN-major operands')
    # the K-major form of the same problem: same arithmetic order, same bits
    d2 = c_cpu.cuda()
    a_km = (case.a[0].contiguous(), case.a[1])
    b_km = (case.b[0].contiguous(), case.b[1])
    dg.fp8_gemm_nt(a_km, b_km, d2, c=d2, recipe=(1, 1, 128))
    assert dg.last_config() == per_col_tile_name(m, n)
    assert torch.equal(d2, case.d)


@pytest.mark.parametrize('m,n,k,mn_major', [(576, 4096, 7168, False), (576, 4096, 7168, True), (320, 1032, 4096, False),
                                             (320, 1040, 4096, True), (200, 518, 3072, False)])
@pytest.mark.parametrize('out_dtype,accumulate', [(torch.float, True), (torch.bfloat16, False), (torch.bfloat16, True), (torch.float, False)])
def test_per_column_sfb_k_split(m, n, k, mn_major, out_dtype, accumulate):
    """Under-filled recipe (1, 1, 128) launches with long K loops (the wgrad entries of the reference sweep with M = 576,
    tests/generators.py:150-153): the K axis runs as the groups of one K-grouped launch into FP32 partial matrices, a second kernel adds
    them in piece order and performs the output step (FP32 / BF16, plain / reduce-add).  Against the oracle, and against the one-launch
    kernel (same arithmetic, the K blocks summed in a different association: equal up to FP32 rounding)."""
    gen.reset_seed(m + n + k)
    case = gen.generate_normal(m, n, k, not mn_major, not mn_major, accumulate=accumulate, out_dtype=out_dtype, per_token_b=True)
    c_cpu = case.c.cpu().clone() if accumulate else None
    want = oracle_dense(case, gran_n=1, c_cpu=c_cpu)
    dg.fp8_gemm_nt(case.a, case.b, case.d, c=case.c if accumulate else None, recipe=(1, 1, 128))
    assert dg.last_config() == ('pipe_pc_mn_ks_256x256' if mn_major else per_col_tile_name(m, n, split=True)), dg.last_config()
    if out_dtype == torch.float:
        assert_close_fp32(case.d, want, 'per-column SFB, K split')
    else:
        assert_close_to_oracle(case.d, want, 'per-column SFB, K split', addend=c_cpu)
    assert calc_diff(case.d, case.ref_d) < gen.FP8_MAX_DIFF
    first = case.d.clone()
    # bit-repeatable, and next to the one-launch kernel
    d2 = c_cpu.cuda() if accumulate else torch.full_like(case.d, float('nan'))
    dg.fp8_gemm_nt(case.a, case.b, d2, c=d2 if accumulate else None, recipe=(1, 1, 128))
    assert torch.equal(d2, first)
    dg.set_forced_config('pipe_pc_mn_256x256' if mn_major else 'pipe_pc_256x256')
    d3 = c_cpu.cuda() if accumulate else torch.empty_like(case.d)
    dg.fp8_gemm_nt(case.a, case.b, d3, c=d3 if accumulate else None, recipe=(1, 1, 128))
    dg.set_forced_config('auto')
    assert dg.last_config() == ('pipe_pc_mn_256x256' if mn_major else 'pipe_pc_256x256')
    assert calc_diff(d3, first) < (1e-6 if out_dtype == torch.bfloat16 else 1e-9)      # (BF16: a rounding flips now and then)


@pytest.mark.parametrize('m,n,k', [(512, 512, 512), (1040, 784, 896), (4096, 2048, 1024)])
@pytest.mark.parametrize('out_dtype,accumulate', [(torch.bfloat16, False), (torch.float, True)])
def test_mn_major_b_native_path(m, n, k, out_dtype, accumulate):
    """fp8_gemm_nn on large problems: the MN-major B goes into the duo kernels as it is (LDS-DMA of k-rows + hardware
    transpose reads, natural column order in the epilogue); same bits as the K-major form of the same operand."""
    gen.reset_seed(m + n + k)
    case = gen.generate_normal(m, n, k, a_k_major=True, b_k_major=False, accumulate=accumulate, out_dtype=out_dtype)
    c_cpu = case.c.cpu().clone() if accumulate else None
    want = oracle_dense(case, c_cpu=c_cpu)
    dg.fp8_gemm_nt(case.a, case.b, case.d, c=case.c if accumulate else None)
    assert dg.last_config().startswith('duo_bmn_'), dg.last_config()
    if out_dtype == torch.float:
        assert_close_fp32(case.d, want, 'MN-major B')
    else:
        assert_close_to_oracle(case.d, want, 'MN-major B')
    assert calc_diff(case.d, case.ref_d) < gen.FP8_MAX_DIFF
    d2 = c_cpu.cuda() if accumulate else torch.empty_like(case.d)
    dg.fp8_gemm_nt(case.a, (case.b[0].contiguous(), case.b[1]), d2, c=d2 if accumulate else None)
    assert not dg.last_config().startswith('duo_bmn_')
    assert torch.equal(d2, case.d)


@pytest.mark.parametrize('m,n,k', [(512, 512, 512), (1040, 784, 896), (4096, 2048, 1024), (304, 272, 256)])
@pytest.mark.parametrize('b_k_major', [True, False])
@pytest.mark.parametrize('out_dtype,accumulate', [(torch.bfloat16, False), (torch.float, True)])
def test_mn_major_a_native_path(m, n, k, b_k_major, out_dtype, accumulate):
    """fp8_gemm_tt / _tn on large problems: the MN-major A goes into the 256 x 256 duo kernel as it is (LDS-DMA of k-rows +
    hardware transpose reads, natural row order, per-row dword scale loads) -- with a K-major B (tt) or an MN-major B read the
    same way (tn); same bits as the K-major form of the same operands."""
    gen.reset_seed(m + n + k + 7)
    case = gen.generate_normal(m, n, k, a_k_major=False, b_k_major=b_k_major, accumulate=accumulate, out_dtype=out_dtype)
    c_cpu = case.c.cpu().clone() if accumulate else None
    want = oracle_dense(case, c_cpu=c_cpu)
    dg.fp8_gemm_nt(case.a, case.b, case.d, c=case.c if accumulate else None)
    assert dg.last_config() == ('duo_amn_256x256' if b_k_major else 'duo_abmn_256x256'), dg.last_config()
    if out_dtype == torch.float:
        assert_close_fp32(case.d, want, 'MN-major A')
    else:
        assert_close_to_oracle(case.d, want, 'MN-major A')
    assert calc_diff(case.d, case.ref_d) < gen.FP8_MAX_DIFF
    d2 = c_cpu.cuda() if accumulate else torch.empty_like(case.d)
    dg.fp8_gemm_nt((case.a[0].contiguous(), case.a[1]), (case.b[0].contiguous(), case.b[1]), d2, c=d2 if accumulate else None)
    assert not dg.last_config().startswith('duo_a')
    assert torch.equal(d2, case.d)
    # the alias entries hand over the same views
    d3 = c_cpu.cuda() if accumulate else torch.empty_like(case.d)
    a_t = (case.a[0].T, case.a[1].T)
    b_arg = case.b if b_k_major else (case.b[0].T, case.b[1].T)
    (dg.fp8_gemm_tt if b_k_major else dg.fp8_gemm_tn)(a_t, b_arg, d3, c=d3 if accumulate else None)
    assert dg.last_config().startswith('duo_a') and torch.equal(d3, case.d)


@pytest.mark.parametrize('b_k_major', [True, False])
def test_mn_major_a_with_k_tail_is_re_majored_onto_a_tail_kernel(b_k_major):
    """The MN-major-A kernels have no K-tail stage: dg_operand_plan sends A (and only A) through dg_transpose_fp8 so that the call lands
    on duo_kt / duo_bmn_kt instead of the layout-agnostic kernel (what the former Python-side predicate let happen)."""
    m, n, k = 512, 768, 2112
    gen.reset_seed(7)
    case = gen.generate_normal(m, n, k, a_k_major=False, b_k_major=b_k_major)
    assert case.a[0].stride(0) == 1
    want = oracle_dense(case)
    dg.fp8_gemm_nt(case.a, case.b, case.d)
    assert dg.last_config().startswith('duo_kt_' if b_k_major else 'duo_bmn_kt_'), dg.last_config()
    assert_close_to_oracle(case.d, want, 'mn-major a, k tail')
    d2 = torch.empty_like(case.d)
    dg.set_forced_config('generic_128x128')
    dg.fp8_gemm_nt(case.a, case.b, d2)
    dg.set_forced_config('auto')
    assert torch.equal(d2, case.d)


@pytest.mark.parametrize('m,n,k', [(512, 512, 576), (1040, 784, 2112), (4096, 1024, 320), (130, 4096, 1088)])
@pytest.mark.parametrize('b_k_major', [True, False])
def test_k_tail_fast_path(m, n, k, b_k_major):
    """K not a multiple of 128 (the reference's dgrad sweep has K = 2112 and 576): the duo kernels compute the partial last block
    after their K loop -- chunks at and beyond K are masked through the buffer range check (zeros in the LDS), so neither the next
    row's bytes nor NaN patterns in row padding reach the result.  Same bits as the layout-agnostic kernel."""
    gen.reset_seed(m + n + k)
    case = gen.generate_normal(m, n, k, a_k_major=True, b_k_major=b_k_major)
    # operands inside wider buffers whose padding bytes are FP8 NaNs (0x7f): a kernel that read past K would produce NaNs
    a_wide = torch.full((m, k + 64), 0x7f, dtype=torch.uint8, device='cuda').view(torch.float8_e4m3fn)
    a_wide[:, :k].copy_(case.a[0])
    a = (a_wide[:, :k], case.a[1])
    if b_k_major:
        b_wide = torch.full((n, k + 64), 0x7f, dtype=torch.uint8, device='cuda').view(torch.float8_e4m3fn)
        b_wide[:, :k].copy_(case.b[0])
        b = (b_wide[:, :k], case.b[1])
    else:
        b_wide = torch.full((k + 16, n), 0x7f, dtype=torch.uint8, device='cuda').view(torch.float8_e4m3fn)     # [K][N] storage + NaN rows
        b_wide[:k].copy_(case.b[0].T)
        b = (b_wide[:k].T, case.b[1])
    want = oracle_dense(case)
    dg.fp8_gemm_nt(a, b, case.d)
    picked = dg.last_config()
    assert picked.startswith('duo_kt_' if b_k_major else 'duo_bmn_kt_') or (not b_k_major and m <= 256), picked
    assert_close_to_oracle(case.d, want, f'k tail {picked}')
    assert calc_diff(case.d, case.ref_d) < gen.FP8_MAX_DIFF
    dg.set_forced_config('generic_128x128')
    d2 = torch.empty_like(case.d)
    dg.fp8_gemm_nt(a, b, d2)
    dg.set_forced_config('auto')
    assert torch.equal(d2, case.d), picked
    for cfg in (['duo_kt_256x256', 'duo_kt_128x256'] if b_k_major else (['duo_bmn_kt_256x256', 'duo_bmn_kt_128x256'] if m > 256 else [])):
        dg.set_forced_config(cfg)
        d3 = torch.full_like(case.d, float('nan'))
        dg.fp8_gemm_nt(a, b, d3)
        assert dg.last_config() == cfg and torch.equal(d3, case.d), cfg
    # whole K blocks through the tail-capable kernels: the stage is skipped
    dg.set_forced_config('duo_kt_256x256' if b_k_major else 'duo_bmn_kt_256x256')
    if b_k_major or m > 256:
        whole = gen.generate_normal(m, n, 384, a_k_major=True, b_k_major=b_k_major)
        dg.fp8_gemm_nt(whole.a, whole.b, whole.d)
        assert_close_to_oracle(whole.d, oracle_dense(whole), 'whole blocks through the tail kernel')


def test_k_tail_sub_views_and_wide_d():
    gen.reset_seed(4)
    # K not a multiple of 128 (quantisers zero-pad the last block): generic path
    case = gen.generate_normal(70, 136, 200)
    dg.fp8_gemm_nt(case.a, case.b, case.d)
    assert_close_to_oracle(case.d, oracle_dense(case), 'k tail')
    assert dg.last_config() == 'generic_128x128'
    # operands that are column sub-views of wider buffers (leading dimension > k) and D wider than N (A5)
    big = gen.generate_normal(160, 256, 1024)
    a_view, b_view = big.a[0][:, 128:640], big.b[0][:, 128:640]
    sfa_view, sfb_view = big.a[1][:, 1:5].contiguous(), big.b[1][:, 1:5].contiguous()
    d_wide = torch.full((160, 300), -5.0, device='cuda', dtype=torch.bfloat16)
    d = d_wide[:, :256]
    for cfg in ('auto', 'generic_128x128'):
        dg.set_forced_config(cfg)
        dg.fp8_gemm_nt((a_view, sfa_view), (b_view, sfb_view), d)
        want = torch.empty((160, 256), dtype=torch.bfloat16)
        oracle.fp8_gemm_nt(a_view.cpu(), sfa_view.cpu(), b_view.cpu(), sfb_view.cpu(), want)
        assert_close_to_oracle(d, want, f'sub views/{cfg}')
        assert bool((d_wide[:, 256:] == -5.0).all())
    # SFA already MN-major (zero-copy path) and SFB transposed-contiguous
    dg.set_forced_config('auto')
    case = gen.generate_normal(130, 256, 512)
    sfa_t = dg.get_mn_major_tma_aligned_tensor(case.a[1])
    assert sfa_t.stride() == (1, 132)
    sfb_t = case.b[1].t().contiguous().t()
    dg.fp8_gemm_nt((case.a[0], sfa_t), (case.b[0], sfb_t), case.d)
    assert_close_to_oracle(case.d, oracle_dense(case), 'pre-transposed SF')


def test_back_to_back_launches_and_side_stream():
    """Calls are stream-ordered and non-blocking."""
    gen.reset_seed(5)
    case = gen.generate_normal(512, 512, 1024)
    want = oracle_dense(case)
    for _ in range(5):
        dg.fp8_gemm_nt(case.a, case.b, case.d)
    assert_close_to_oracle(case.d, want, 'back to back')
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    d2 = torch.empty_like(case.d)
    with torch.cuda.stream(side):
        dg.fp8_gemm_nt(case.a, case.b, d2)
    side.synchronize()
    assert torch.equal(d2, case.d)


@pytest.mark.parametrize('use_psum', [False, True])
@pytest.mark.parametrize('b_k_major', [True, False])
def test_m_grouped_contiguous_vs_oracle(use_psum, b_k_major):
    gen.reset_seed(6)
    for actual_ms, n, k in (([100, 0, 130, 256], 256, 384), ([300, 77], 520, 256), ([128] * 8, 4096, 512),
                            ([128, 384, 0, 0, 200, 640], 512, 256)):
        case = gen.generate_m_grouped_contiguous(len(actual_ms), 0, n, k, b_k_major, use_psum, actual_ms=actual_ms)
        want = torch.full(case.d.shape, float('nan'), dtype=torch.bfloat16)
        oracle.m_grouped_fp8_gemm_nt_contiguous(*cpu_pair(case.a), *cpu_pair(case.b), want, case.grouped_layout.cpu(), use_psum)
        fast = ['pipe_128x256', 'pipe_128x128', 'pipe_64x256', 'duo_128x256'] + ([] if use_psum else ['duo_256x256', 'duo_p_256x256'])   # duo: two-pass 256-row tiles
        fast_nn = (['duo_bmn_128x256'] + ([] if use_psum else ['duo_bmn_256x256'])) if n % 16 == 0 else []   # MN-major B read natively
        for cfg in (['auto', 'generic_128x128'] + (fast if b_k_major else fast_nn)):
            dg.set_forced_config(cfg)
            case.d.fill_(float('nan'))
            if b_k_major:
                dg.m_grouped_fp8_gemm_nt_contiguous(case.a, case.b, case.d, case.grouped_layout, use_psum_layout=use_psum)
            else:
                b_alias = (case.b[0].mT, case.b[1].mT)
                assert b_alias[0].is_contiguous()
                dg.m_grouped_fp8_gemm_nn_contiguous(case.a, b_alias, case.d, case.grouped_layout, use_psum_layout=use_psum)
            start = 0
            for actual, aligned in zip(case.actual_ms, case.aligned_ms):
                rows = slice(start, start + actual)
                assert_close_to_oracle(case.d[rows], want[rows], f'{cfg} rows {rows}')
                pad = case.d[start + actual:start + aligned]
                assert bool((pad == 0).all()), f'{cfg}: padding rows must be zeros'
                start += aligned
            assert calc_diff(torch.nan_to_num(case.d), torch.nan_to_num(case.ref_d)) < gen.FP8_MAX_DIFF


def test_m_grouped_contiguous_reference_shapes_sampled():
    """One reference-sized case (8 groups x ~512 rows, N=4096, K=7168 = BASELINE config 4): reference gate on all rows,
    oracle on a row sample per group."""
    gen.reset_seed(0)
    case = gen.generate_m_grouped_contiguous(8, 512, 4096, 7168)
    dg.m_grouped_fp8_gemm_nt_contiguous(case.a, case.b, case.d, case.grouped_layout)
    assert calc_diff(case.d, case.ref_d) < gen.FP8_MAX_DIFF
    start = 0
    for g, (actual, aligned) in enumerate(zip(case.actual_ms, case.aligned_ms)):
        rows = torch.tensor(sorted(random.sample(range(start, start + actual), 8)))
        want = oracle.fp8_gemm_nt_blockwise_torch(case.a[0][rows.cuda()].cpu(), case.a[1][rows.cuda()].cpu(),
                                                  case.b[0][g].cpu(), case.b[1][g].cpu())
        assert_close_to_oracle(case.d[rows.cuda()], want, f'group {g}')
        assert bool((case.d[start + actual:start + aligned] == 0).all())
        start += aligned


def _split_k_case(actual_ms, n, k, use_psum, alignment):
    dg.set_mk_alignment_for_contiguous_layout(alignment)
    case = gen.generate_m_grouped_contiguous(len(actual_ms), 0, n, k, True, use_psum, actual_ms=actual_ms)
    tiles = (case.a[0].size(0) // 128) * ((n + 255) // 256)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    return case, tiles, cus


@pytest.mark.parametrize('use_psum', [False, True])
@pytest.mark.parametrize('actual_ms,n,k,alignment', [
    ([520, 500, 640, 400, 512, 700, 384, 512], 2048, 512, 128),      # 280 tiles: 24 tail tiles x 4 K pieces
    ([300, 77, 1000, 128, 129, 1, 2000, 640], 2304, 384, 128),       # 333 tiles: 77 tail tiles x 3 pieces, ragged groups
    ([100, 1300, 30, 900, 1500], 4096, 256, 256),                    # alignment 256: all-padding 128-row tiles among the pieces
    ([512] * 7 + [640], 2048, 896, 128),                             # 264 tiles: 8 tail tiles x 7 pieces (one K block each)
])
def test_m_grouped_contiguous_split_k_tail(use_psum, actual_ms, n, k, alignment):
    """The K-split tail of the persistent 128 x 256 duo kernel (the partial last round cut along K over the idle CUs, FP32
    partials through the caller's workspace, summed in piece order by the reduction kernel behind it): every row against the oracle,
    padding rows zero, nothing outside D, bit-repeatable."""
    from deepgemm_amd import gemm as gemm_mod
    gen.reset_seed(11)
    case, tiles, cus = _split_k_case(actual_ms, n, k, use_psum, alignment)
    assert tiles > cus and 0 < tiles % cus <= cus // 2, 'the case must leave a partial last round'
    want = torch.full(case.d.shape, float('nan'), dtype=torch.bfloat16)
    oracle.m_grouped_fp8_gemm_nt_contiguous(*cpu_pair(case.a), *cpu_pair(case.b), want, case.grouped_layout.cpu(), use_psum)
    m = case.d.size(0)
    dg.m_grouped_fp8_gemm_nt_contiguous(case.a, case.b, case.d, case.grouped_layout, use_psum_layout=use_psum)
    assert dg.last_config() != 'duo_sk_128x256', 'short K loops: the exchange of partial tiles costs more than the split saves'
    for cfg in ('duo_sk_128x256',):
        dg.set_forced_config(cfg)
        guarded = torch.full((m + 256, n), 777.0, device='cuda', dtype=torch.bfloat16)
        outs = []
        for _ in range(4):
            d = guarded[128:128 + m]
            d.fill_(float('nan'))
            dg.m_grouped_fp8_gemm_nt_contiguous(case.a, case.b, d, case.grouped_layout, use_psum_layout=use_psum)
            outs.append(d.clone())
        assert dg.last_config() == 'duo_sk_128x256', (cfg, dg.last_config())
        assert bool((guarded[:128] == 777.0).all()) and bool((guarded[128 + m:] == 777.0).all()), 'wrote outside D'
        assert all(torch.equal(o, outs[0]) for o in outs[1:]), 'the piece-order reduction must be bit-repeatable'
        start = 0
        for actual, aligned in zip(case.actual_ms, case.aligned_ms):
            assert_close_to_oracle(outs[0][start:start + actual], want[start:start + actual], f'{cfg} rows {start}+{actual}')
            assert bool((outs[0][start + actual:start + aligned] == 0).all()), f'{cfg}: padding rows must be zeros'
            start += aligned
    assert gemm_mod._SPLIT_K_WORKSPACES, 'the K split went through the host layer\'s workspace'
    # the plain persistent walk of the same kernel (no workspace: dense entry) and the non-split kernel agree with it to rounding
    dg.set_forced_config('duo_128x256')
    plain = torch.empty_like(case.d)
    dg.m_grouped_fp8_gemm_nt_contiguous(case.a, case.b, plain, case.grouped_layout, use_psum_layout=use_psum)
    assert dg.last_config() == 'duo_128x256'
    assert calc_diff(torch.nan_to_num(plain), torch.nan_to_num(outs[0])) < 1e-6


def test_m_grouped_contiguous_nn_split_k_tail():
    """m_grouped_fp8_gemm_nn_contiguous (MN-major B read natively) with a partial last round and a long K loop: the K-split tail of
    the MN-major-B form (duo_sk_bmn_128x256) -- reference gate on all rows, oracle on a row sample per group, padding rows zero,
    bit-repeatable."""
    gen.reset_seed(21)
    actual_ms = [520, 500, 640, 400, 512, 700, 384, 512]
    dg.set_mk_alignment_for_contiguous_layout(128)
    case = gen.generate_m_grouped_contiguous(len(actual_ms), 0, 2048, 7168, False, False, actual_ms=actual_ms)
    tiles = (case.a[0].size(0) // 128) * (2048 // 256)
    assert tiles % torch.cuda.get_device_properties(0).multi_processor_count != 0
    outs = []
    for _ in range(3):
        d = torch.full_like(case.d, float('nan'))
        dg.m_grouped_fp8_gemm_nt_contiguous(case.a, case.b, d, case.grouped_layout)
        outs.append(d)
    assert dg.last_config() == 'duo_sk_bmn_128x256', dg.last_config()
    assert all(torch.equal(o, outs[0]) for o in outs[1:])
    assert calc_diff(torch.nan_to_num(outs[0]), torch.nan_to_num(case.ref_d)) < gen.FP8_MAX_DIFF
    start = 0
    for g, (actual, aligned) in enumerate(zip(case.actual_ms, case.aligned_ms)):
        rows = torch.tensor(sorted(random.sample(range(start, start + actual), 6)))
        want = oracle.fp8_gemm_nt_blockwise_torch(case.a[0][rows.cuda()].cpu(), case.a[1][rows.cuda()].cpu(),
                                                  case.b[0][g].cpu(), case.b[1][g].cpu())
        assert_close_to_oracle(outs[0][rows.cuda()], want, f'group {g}')
        assert bool((outs[0][start + actual:start + aligned] == 0).all())
        start += aligned


@pytest.mark.parametrize('m,n,k', [(1024, 512, 8192), (4096, 576, 7168), (200, 1024, 16384)])
@pytest.mark.parametrize('b_k_major', [True, False])
@pytest.mark.parametrize('out_dtype,accumulate', [(torch.bfloat16, False), (torch.float, True)])
def test_dense_split_k_under_filled_launch(m, n, k, b_k_major, out_dtype, accumulate):
    """Dense problems whose 128 x 256 tiles fill a fraction of the chip while the K loop is long (the dgrad shape 4096 x 512 x 32768 is
    the reference-sweep example): every tile is cut along K over the idle CUs -- the K-split of the contiguous tail with all tiles
    in the "last round" -- for K-major and MN-major B; oracle, repeatability, reduce-add applied once, same result as the unsplit
    kernel to rounding."""
    from deepgemm_amd import gemm as gemm_mod
    gen.reset_seed(m + n + k + 3)
    case = gen.generate_normal(m, n, k, a_k_major=True, b_k_major=b_k_major, accumulate=accumulate, out_dtype=out_dtype)
    c_cpu = case.c.cpu().clone() if accumulate else None
    want = oracle_dense(case, c_cpu=c_cpu)
    outs = []
    for _ in range(4):
        d = c_cpu.cuda() if accumulate else torch.full_like(case.d, float('nan'))
        dg.fp8_gemm_nt(case.a, case.b, d, c=d if accumulate else None)
        outs.append(d)
    # (an MN-major B of a problem with m <= 256 is re-majored first: the K-major kernels -- since the end of round 6 the 64 x 32 stream tile cut
    #  along K inside the kernel where its tiles fill at most half the chip: 200 x 1024 x 16384 27.5 -> 20.7 us, profiles/r06_probe/ks_vs_duo_sk_ab.log)
    assert dg.last_config() == ('stream_ks_64x32' if m <= 256 else 'duo_sk_128x256' if b_k_major else 'duo_sk_bmn_128x256'), dg.last_config()
    assert all(torch.equal(o, outs[0]) for o in outs[1:]), 'the piece-order reduction must be bit-repeatable'
    if out_dtype == torch.float:
        assert_close_fp32(outs[0], want, 'dense split K')
    else:
        assert_close_to_oracle(outs[0], want, 'dense split K')
    assert gemm_mod._SPLIT_K_WORKSPACES, 'the K split went through the host layer\'s workspace'
    dg.set_forced_config('duo_128x256' if b_k_major or m <= 256 else 'duo_bmn_128x256')
    plain = c_cpu.cuda() if accumulate else torch.empty_like(case.d)
    dg.fp8_gemm_nt(case.a, case.b, plain, c=plain if accumulate else None)
    assert calc_diff(plain, outs[0]) < 1e-6


def test_split_k_tail_full_size_and_small_workspace(monkeypatch):
    """BASELINE config 4 (8 groups x ~512 rows, N = 4096, K = 7168) with a 2.25-round tile count: the default path takes the
    K-split tail; with a workspace too small for the partials the same kernel walks all tiles unsplit.  Dense calls (no
    workspace) forced onto the kernel take the unsplit walk too."""
    from deepgemm_amd import gemm as gemm_mod
    gen.reset_seed(12)
    case, tiles, cus = _split_k_case([520, 500, 640, 400, 512, 700, 384, 512], 4096, 7168, False, 128)
    assert tiles % cus != 0
    # (round 3: the automatic selection takes the group-relative tile table for this shape -- test_m_grouped_contiguous_group_relative_tiles;
    # the K-split tail of the 128-row walk is what is under test here, by name)
    dg.m_grouped_fp8_gemm_nt_contiguous(case.a, case.b, case.d, case.grouped_layout)
    assert dg.last_config() == 'duo_tab_256x256' and calc_diff(case.d, case.ref_d) < gen.FP8_MAX_DIFF
    dg.set_forced_config('duo_sk_128x256')
    case.d.fill_(float('nan'))
    dg.m_grouped_fp8_gemm_nt_contiguous(case.a, case.b, case.d, case.grouped_layout)
    assert dg.last_config() == 'duo_sk_128x256'
    assert calc_diff(case.d, case.ref_d) < gen.FP8_MAX_DIFF
    split = case.d.clone()
    again = torch.empty_like(case.d)
    for _ in range(5):
        dg.m_grouped_fp8_gemm_nt_contiguous(case.a, case.b, again, case.grouped_layout)
        assert torch.equal(again, split)
    start = 0
    for g, (actual, aligned) in enumerate(zip(case.actual_ms, case.aligned_ms)):
        rows = torch.tensor(sorted(random.sample(range(start, start + actual), 6)))
        want = oracle.fp8_gemm_nt_blockwise_torch(case.a[0][rows.cuda()].cpu(), case.a[1][rows.cuda()].cpu(),
                                                  case.b[0][g].cpu(), case.b[1][g].cpu())
        assert_close_to_oracle(split[rows.cuda()], want, f'group {g}')
        assert bool((split[start + actual:start + aligned] == 0).all())
        start += aligned
    # other data through the same workspace: a reducer must never see a stale cached partial of the launch before
    gen.reset_seed(13)
    other, _, _ = _split_k_case([520, 500, 640, 400, 512, 700, 384, 512], 4096, 7168, False, 128)
    dg.m_grouped_fp8_gemm_nt_contiguous(other.a, other.b, other.d, other.grouped_layout)
    assert calc_diff(other.d, other.ref_d) < gen.FP8_MAX_DIFF
    other_first = other.d.clone()
    for _ in range(4):
        dg.m_grouped_fp8_gemm_nt_contiguous(case.a, case.b, again, case.grouped_layout)
        dg.m_grouped_fp8_gemm_nt_contiguous(other.a, other.b, other.d, other.grouped_layout)
        assert torch.equal(again, split) and torch.equal(other.d, other_first)
    tiny = torch.zeros(8192, dtype=torch.uint8, device='cuda')
    monkeypatch.setattr(gemm_mod, '_split_k_workspace', lambda device, stream: tiny)
    dg.set_forced_config('duo_sk_128x256')
    unsplit = torch.empty_like(case.d)
    dg.m_grouped_fp8_gemm_nt_contiguous(case.a, case.b, unsplit, case.grouped_layout)
    assert calc_diff(unsplit, case.ref_d) < gen.FP8_MAX_DIFF and calc_diff(unsplit, split) < 1e-6
    dg.set_forced_config('duo_128x256')
    plain = torch.empty_like(case.d)
    dg.m_grouped_fp8_gemm_nt_contiguous(case.a, case.b, plain, case.grouped_layout)
    assert torch.equal(plain, unsplit), 'the unsplit walk accumulates in the same order as the one-tile-per-block kernel'
    dense = gen.generate_normal(4480, 4096, 1024)
    dg.set_forced_config('duo_sk_128x256')
    dg.fp8_gemm_nt(dense.a, dense.b, dense.d)
    assert dg.last_config() == 'duo_sk_128x256' and calc_diff(dense.d, dense.ref_d) < gen.FP8_MAX_DIFF


@pytest.mark.parametrize('masked_ms,max_m,n,k', [([5, 0, 64, 33], 64, 256, 384), ([200, 1, 129], 256, 520, 256),
                                                  ([20] * 6 + [0, 64], 64, 4096, 512)])
def test_m_grouped_masked_vs_oracle(masked_ms, max_m, n, k):
    gen.reset_seed(7)
    case = gen.generate_m_grouped_masked(len(masked_ms), max_m, 0, n, k, masked_ms=masked_ms)
    want = torch.full(case.d.shape, float('nan'), dtype=torch.bfloat16)
    oracle.m_grouped_fp8_gemm_nt_masked(*cpu_pair(case.a), *cpu_pair(case.b), want, case.masked_m.cpu())
    expected_m = max(1, int(sum(masked_ms) / len(masked_ms)))
    for cfg in ['auto', 'generic_128x128'] + FAST:
        dg.set_forced_config(cfg)
        case.d.fill_(float('nan'))
        dg.m_grouped_fp8_gemm_nt_masked(case.a, case.b, case.d, case.masked_m, expected_m)
        for g, rows in enumerate(masked_ms):
            if rows:
                assert_close_to_oracle(case.d[g, :rows], want[g, :rows], f'{cfg} group {g}')
                assert calc_diff(case.d[g, :rows], case.ref_d[g, :rows]) < gen.FP8_MAX_DIFF
            assert bool(torch.isnan(case.d[g, rows:]).all()), f'{cfg}: rows >= masked_m must not be written'


@pytest.mark.parametrize('groups,expected_m,max_m,n,k', [(6, 1024, 4096, 4096, 4096), (32, 20, 4096, 4096, 2048), (8, 48, 64, 4096, 7168)])
def test_m_grouped_masked_reference_shapes(groups, expected_m, max_m, n, k):
    """The reference's masked sweep (tests/generators.py:174-187: max_m 4096, masked_m = int(expected_m * U(0.7, 1.3))) and
    BASELINE config 5's per-rank shape (8 local experts, M <= 64): reference gate on the valid rows, oracle on a row
    sample, rows >= masked_m untouched (NaN poison)."""
    gen.reset_seed(groups)
    case = gen.generate_m_grouped_masked(groups, max_m, expected_m, n, k)
    case.d.fill_(float('nan'))
    dg.m_grouped_fp8_gemm_nt_masked(case.a, case.b, case.d, case.masked_m, expected_m)
    for g, rows in enumerate(case.masked_m.tolist()):
        rows = int(rows)
        if rows:
            assert calc_diff(case.d[g, :rows], case.ref_d[g, :rows]) < gen.FP8_MAX_DIFF, (g, rows, dg.last_config())
            pick = torch.tensor(sorted(random.sample(range(rows), min(rows, 4))))
            want = oracle.fp8_gemm_nt_blockwise_torch(case.a[0][g][pick.cuda()].cpu(), case.a[1][g][pick.cuda()].cpu(),
                                                      case.b[0][g].cpu(), case.b[1][g].cpu())
            assert_close_to_oracle(case.d[g][pick.cuda()], want, f'group {g}')
        assert bool(torch.isnan(case.d[g, rows:]).all()), f'group {g}: rows >= masked_m must not be written'


@pytest.mark.parametrize('num_sms', [40, 200])
def test_set_num_sms_limits_the_persistent_kernels(num_sms):
    """set_num_sms (csrc/apis/runtime.hpp:12-41): the persistent launches, the masked walk and the K split size their grids and their
    piece counts from it; results do not depend on it (K-split sums: to rounding)."""
    gen.reset_seed(31)
    dense = gen.generate_normal(1024, 1024, 1024)
    skinny = gen.generate_normal(1024, 512, 8192)
    masked = gen.generate_m_grouped_masked(6, 256, 0, 512, 512, masked_ms=[200, 3, 0, 256, 77, 130])
    contig = gen.generate_m_grouped_contiguous(4, 0, 1024, 1024, actual_ms=[300, 77, 512, 129])

    def run_all():
        outs = []
        for case, call in ((dense, lambda c, d: dg.fp8_gemm_nt(c.a, c.b, d)), (skinny, lambda c, d: dg.fp8_gemm_nt(c.a, c.b, d)),
                           (masked, lambda c, d: dg.m_grouped_fp8_gemm_nt_masked(c.a, c.b, d, c.masked_m, 128)),
                           (contig, lambda c, d: dg.m_grouped_fp8_gemm_nt_contiguous(c.a, c.b, d, c.grouped_layout))):
            d = torch.zeros_like(case.d)
            call(case, d)
            outs.append((d, dg.last_config()))
        return outs
    want = run_all()
    dg.set_num_sms(num_sms)
    try:
        assert dg.get_num_sms() == num_sms
        got = run_all()
    finally:
        dg.set_num_sms(0)
    for (d0, cfg0), (d1, cfg1) in zip(want, got):
        if 'sk' in cfg0 or 'sk' in cfg1 or cfg0 != cfg1:
            assert calc_diff(d1, d0) < 1e-6, (cfg0, cfg1)
        else:
            assert torch.equal(d1, d0), (cfg0, cfg1)
    assert calc_diff(got[0][0], dense.ref_d) < gen.FP8_MAX_DIFF and calc_diff(got[1][0], skinny.ref_d) < gen.FP8_MAX_DIFF


@pytest.mark.parametrize('seed', [101, 202, 303])
def test_dense_random_shapes_and_layouts_vs_oracle(seed):
    """Randomised dense problems through the automatic selection -- every majorness, K tails (multiples of 16), odd M / N, BF16 and
    FP32 outputs with and without accumulation, shapes that reach the stream, duo, K-tail, K-split and layout-agnostic kernels -- each against the
    oracle.  (Fixed seeds: a failure reproduces.)"""
    rng = random.Random(seed)
    picked = set()
    for _ in range(14):
        m = rng.choice([1, 7, 64, 100, 129, 256, 300, 520, 777, 1024, 1536])
        n = rng.choice([16, 136, 256, 384, 520, 1024, 2048])
        k = rng.choice([128, 144, 256, 320, 512, 1088, 2112, 4096])
        a_k, b_k = rng.random() < 0.6, rng.random() < 0.6
        accumulate = rng.random() < 0.35
        out_dtype = torch.float if rng.random() < (0.5 if accumulate else 0.2) else torch.bfloat16      # (BF16 accumulation: the reduce-add in D's dtype)
        if not a_k and m % 16:
            m = (m + 15) // 16 * 16          # an MN-major view needs a 16-byte row pitch to be a legal operand of the fast kernels
        gen.reset_seed(seed + m + n + k)
        case = gen.generate_normal(m, n, k, a_k, b_k, accumulate=accumulate, out_dtype=out_dtype)
        c_cpu = case.c.cpu().clone() if accumulate else None
        want = oracle_dense(case, c_cpu=c_cpu)
        dg.fp8_gemm_nt(case.a, case.b, case.d, c=case.c if accumulate else None)
        label = f'm={m} n={n} k={k} a_k={a_k} b_k={b_k} acc={accumulate} {out_dtype} [{dg.last_config()}]'
        picked.add(dg.last_config())
        if out_dtype == torch.float:
            assert_close_fp32(case.d, want, label)
        else:
            assert_close_to_oracle(case.d, want, label, addend=c_cpu)
        if m * n >= 16384:              # (the reference's gate is statistical: a handful of outputs is FP8 quantisation noise)
            assert calc_diff(case.d, case.ref_d) < gen.FP8_MAX_DIFF, label
    assert len(picked) >= 3, picked


def test_import_is_fork_safe_and_bench_runs():
    """(i) importing the package must not initialise the GPU runtime: a forked child can still pick its device (reference
    tests/test_lazy_init.py:7-20); (ii) bench.py prints one well-formed JSON line carrying roofline and cpu_baseline."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = (
        "import os, sys\n"
        f"sys.path.insert(0, {root!r})\n"
        "import torch, deepgemm_amd\n"
        "assert not torch.cuda.is_initialized()\n"
        "pids = []\n"
        "for i in range(4):\n"
        "    pid = os.fork()\n"
        "    if pid == 0:\n"
        "        torch.cuda.set_device(0); torch.zeros(1, device='cuda'); os._exit(0)\n"
        "    pids.append(pid)\n"
        "assert all(os.waitpid(p, 0)[1] == 0 for p in pids)\n"
        "print('fork ok')\n")
    out = subprocess.run([sys.executable, '-c', script], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'fork ok' in out.stdout, out.stderr[-2000:]
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '4', '--warmup', '2'],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 1 and line['steps'] == 4 and line['unit'] == 'TFLOPS' and line['value'] > 0
    assert line['roofline']['bound'] == 'mfma' and 0 < line['roofline']['frac'] < 1 and line['cpu_baseline']['value'] > 0
    assert 'zero-copy' in line['config']['sfa_layout']
    # the driver-visible record of the other configurations (C3 per layout, C4, C5, wgrad, K-grouped, packed UE8M0, dgrad entries ...):
    # compact {workload: [frac, us(, 'h')(, frac on data rows)]} on the headline line, which must fit the driver's 2000-character tail;
    # the full records on a prefixed (non-JSON) line before it
    assert len(out.stdout.splitlines()[-1]) < 2000, len(out.stdout.splitlines()[-1])
    secondary = line['secondary']
    assert len(secondary) == 27 and not [v for v in secondary.values() if isinstance(v, str)], secondary
    for name, rec in secondary.items():             # [frac, us(, 'h' = HBM-bound)(, frac on data rows | of the recipe's roof)]
        assert 0 < rec[0] < 1 and rec[1] > 0 and all(v == 'h' or 0 < v < 1 for v in rec[2:]), (name, rec)
    assert secondary['masked'][2] == 'h' and len(secondary['c3_nt']) == 2 and len(secondary['contiguous']) == 3 and len(secondary['kgrouped_ue8m0']) == 2
    detail = [ln for ln in out.stdout.splitlines() if ln.startswith('secondary_detail: ')]
    assert len(detail) == 1
    detail = json.loads(detail[0][len('secondary_detail: '):])
    assert len(detail) == 27 and all(0 < rec['roofline']['frac'] < 1 and rec['roofline']['kernel_us'] > 0 for rec in detail)
    # --gpus N without a launcher must not silently run one rank
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '64', '--steps', '2', '--warmup', '1'],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and 'GPU(s) visible' in out.stderr


def test_full_size_c2_properties():
    """BASELINE.json config 2 (4096 x 4096 x 7168): reference gate on the whole output, oracle on sampled rows, and
    size-independent properties that must hold bit-exactly: power-of-two scaling of SFA/SFB and row permutation of A."""
    gen.reset_seed(0)
    case = gen.generate_normal(4096, 4096, 7168)
    dg.fp8_gemm_nt(case.a, case.b, case.d)
    assert dg.last_config().split('_')[0] in ('duo', 'ring', 'pipe')
    assert calc_diff(case.d, case.ref_d) < gen.FP8_MAX_DIFF
    rows = torch.tensor(sorted(random.sample(range(4096), 48)), device='cuda')
    want = oracle.fp8_gemm_nt_blockwise_torch(case.a[0][rows].cpu(), case.a[1][rows].cpu(), case.b[0].cpu(), case.b[1].cpu())
    assert_close_to_oracle(case.d[rows], want, 'sampled rows')
    # scaling SFA by 2 and SFB by 1/4 scales every product exactly by 1/2
    d2 = torch.empty_like(case.d)
    dg.fp8_gemm_nt((case.a[0], case.a[1] * 2), (case.b[0], case.b[1] * 0.25), d2)
    assert torch.equal(d2.float() * 2, case.d.float())
    # permuting the rows of A (and SFA) permutes the rows of D, whatever tile / wave / lane each row lands in
    perm = torch.randperm(4096, device='cuda')
    d3 = torch.empty_like(case.d)
    dg.fp8_gemm_nt((case.a[0][perm].contiguous(), case.a[1][perm].contiguous()), case.b, d3)
    assert torch.equal(d3, case.d[perm])
    # every dense configuration agrees bit-for-bit on the same problem (same per-element arithmetic order)
    for cfg in ('duo_256x256', 'duo_p_256x256', 'pipe_256x256', 'pipe_128x256', 'pipe_128x128', 'pipe_64x256'):
        dg.set_forced_config(cfg)
        d4 = torch.empty_like(case.d)
        dg.fp8_gemm_nt(case.a, case.b, d4)
        assert torch.equal(d4, case.d), cfg


def test_repeatability_full_size():
    """Same inputs, same configuration, many launches: every output must be bit-identical.  Guards the hand-managed
    asynchrony (counted vmcnt, raw barriers, asm loads whose destination registers hipcc may copy before they land):
    a race shows up as a few wrong tiles in a few launches, far below what a single calc_diff check notices."""
    gen.reset_seed(3)
    case = gen.generate_normal(4096, 4096, 7168)
    for cfg in ('duo_256x256', 'duo_p_256x256', 'pipe_256x256', 'stream_64x128'):
        dg.set_forced_config(cfg)
        first = torch.empty_like(case.d)
        dg.fp8_gemm_nt(case.a, case.b, first)
        for _ in range(6 if cfg == 'stream_64x128' else 12):
            again = torch.empty_like(case.d)
            dg.fp8_gemm_nt(case.a, case.b, again)
            assert torch.equal(again, first), cfg
    # a multi-tile-per-CU problem through the persistent launch
    case2 = gen.generate_normal(4096, 16384, 1024)
    want = None
    for cfg in ('duo_256x256', 'duo_p_256x256', 'duo_p_256x256', 'duo_p_256x256'):
        dg.set_forced_config(cfg)
        d = torch.empty_like(case2.d)
        dg.fp8_gemm_nt(case2.a, case2.b, d)
        want = d if want is None else want
        assert torch.equal(d, want), cfg
    assert calc_diff(want, case2.ref_d) < gen.FP8_MAX_DIFF


def test_repeatability_other_kernels():
    """The same bit-repeatability check for the kernels with hand-placed waits that the dense default does not reach: the
    128-row duo tile, the per-column-SFB kernel in its K-major and transpose-read forms (FP32 accumulate), the K-grouped
    single launch and the hardware-scaled UE8M0 kernel."""
    from deepgemm_amd.utils import per_block_cast_to_fp8, per_token_cast_to_fp8
    from deepgemm_amd.utils.math import pack_ue8m0_to_int
    gen.reset_seed(5)
    case = gen.generate_normal(2048, 4096, 3584)
    dg.set_forced_config('duo_128x256')
    first = torch.empty_like(case.d)
    dg.fp8_gemm_nt(case.a, case.b, first)
    for _ in range(8):
        again = torch.empty_like(case.d)
        dg.fp8_gemm_nt(case.a, case.b, again)
        assert torch.equal(again, first), 'duo_128x256'
    dg.set_forced_config('auto')
    for k_major in (True, False):
        pc = gen.generate_normal(2048, 2304, 3584, a_k_major=k_major, b_k_major=k_major, accumulate=True, out_dtype=torch.float,
                                 per_token_b=True)
        c0 = pc.c.clone()
        outs = []
        for _ in range(6):
            d = c0.clone()
            dg.fp8_gemm_nt(pc.a, pc.b, d, c=d, recipe=(1, 1, 128))
            outs.append(d)
        assert dg.last_config() == (per_col_tile_name(2048, 2304) if k_major else 'pipe_pc_mn_256x256')
        assert all(torch.equal(o, outs[0]) for o in outs[1:]), dg.last_config()
    for k_major in (True, False):
        kg = gen.generate_k_grouped_contiguous(3, 1024, 1280, [1024, 512, 1536], k_major)
        fn = dg.k_grouped_fp8_gemm_nt_contiguous if k_major else dg.k_grouped_fp8_gemm_tn_contiguous
        outs = []
        for _ in range(5):
            d = kg.c.clone()
            fn(kg.a, kg.b, d, kg.ks, kg.grouped_layout, c=d)
            outs.append(d)
        assert all(torch.equal(o, outs[0]) for o in outs[1:]), ('k-grouped', k_major)
    torch.manual_seed(6)
    m, n, k = 2048, 2048, 3584
    a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16)
    b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
    a_q, sfa = per_token_cast_to_fp8(a, use_ue8m0=True)
    b_q, sfb = per_block_cast_to_fp8(b, use_ue8m0=True)
    pa = dg.get_mn_major_tma_aligned_packed_ue8m0_tensor(sfa)
    pb = dg.get_mn_major_tma_aligned_packed_ue8m0_tensor(sfb.repeat_interleave(128, dim=0)[:n].contiguous())
    outs = []
    for _ in range(8):
        d = torch.empty((m, n), device='cuda', dtype=torch.bfloat16)
        dg.fp8_gemm_nt((a_q, pa), (b_q, pb), d)
        outs.append(d)
    assert dg.last_config().startswith('e8_')
    assert all(torch.equal(o, outs[0]) for o in outs[1:]), 'e8'


def test_reference_sweep_subset_gate():
    """A slice of the reference's dense sweep (tests/generators.py:119-121) at the reference's own gate."""
    gen.reset_seed(0)
    for m in (1, 128, 4096):
        for n, k in ((2112, 7168), (576, 7168), (7168, 2048), (24576, 1536)):
            case = gen.generate_normal(m, n, k)
            dg.fp8_gemm_nt(case.a, case.b, case.d)
            diff = calc_diff(case.d, case.ref_d)
            assert diff < gen.FP8_MAX_DIFF, (m, n, k, diff, dg.last_config())


@pytest.mark.parametrize('m,n,k', [(256, 512, 1024), (300, 520, 1536), (64, 136, 512), (1024, 2048, 7168)])
def test_packed_ue8m0_scales_hw_path(m, n, k):
    """Power-of-two scales in the reference's packed UE8M0 format (SM100 input format, recipe (1, 1, 128);
    deep_gemm/utils/math.py:13-23): hardware-scaled MFMA path against the oracle fed with the same scales as FP32."""
    from deepgemm_amd.utils.math import pack_ue8m0_to_int, per_block_cast_to_fp8, per_token_cast_to_fp8
    gen.reset_seed(m + n)
    a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16)
    b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
    ref = (a.float() @ b.float().t()).to(torch.bfloat16)
    a_q, sfa = per_token_cast_to_fp8(a, use_ue8m0=True)
    b_q, sfb_blocks = per_block_cast_to_fp8(b, use_ue8m0=True)
    sfb_rows = sfb_blocks.repeat_interleave(128, dim=0)[:n].contiguous()             # per-row scales of B
    d = torch.full((m, n), float('nan'), device='cuda', dtype=torch.bfloat16)
    dg.fp8_gemm_nt((a_q, pack_ue8m0_to_int(sfa)), (b_q, pack_ue8m0_to_int(sfb_rows)), d)
    assert dg.last_config().startswith('e8_')
    want = torch.empty((m, n), dtype=torch.bfloat16)
    oracle.fp8_gemm_nt(a_q.cpu(), sfa.cpu(), b_q.cpu(), sfb_rows.cpu(), want, gran_n=1)
    assert_close_to_oracle(d, want, 'packed ue8m0')
    assert calc_diff(d, ref) < gen.FP8_MAX_DIFF
    # the same scales as FP32 tensors through the FP32-scale path give the same result up to summation order
    d2 = torch.empty_like(d)
    dg.fp8_gemm_nt((a_q, sfa), (b_q, sfb_blocks), d2)
    assert calc_diff(d, d2) < 2e-6
    # FP32 output and accumulation
    c32 = torch.randn((m, n), device='cuda', dtype=torch.float)
    d32 = c32.clone()
    dg.fp8_gemm_nt((a_q, pack_ue8m0_to_int(sfa)), (b_q, pack_ue8m0_to_int(sfb_rows)), d32, c=d32)
    want32 = torch.empty((m, n), dtype=torch.float)
    oracle.fp8_gemm_nt(a_q.cpu(), sfa.cpu(), b_q.cpu(), sfb_rows.cpu(), want32, c=c32.cpu(), gran_n=1)
    assert_close_fp32(d32, want32, 'packed ue8m0 fp32 accumulate')


E8_QUAD_256 = ['e8_quad_256x256', 'e8_quad_h_256x256', 'e8_quad_h2_256x256']     # whole K quads only; _h*: the register-resident schedule (round 5)
E8_DENSE_CONFIGS = ['auto', *E8_QUAD_256, 'e8_quad_128x256', 'e8_duo_256x256', 'e8_stream_64x128', 'e8_stream_nt_64x128', 'e8_stream2_64x128', 'e8_stream_nt2_64x128', 'e8_stream_64x32']


@pytest.mark.parametrize('m,n,k', [(512, 768, 1024), (300, 520, 896), (4096, 4096, 1536), (129, 4096, 384)])
def test_packed_ue8m0_every_kernel_and_per_row_sfb(m, n, k):
    """Packed UE8M0 scales with one scale per ROW of B (recipe (1, 1, 128), the general SM100 form) through every
    hardware-scaled kernel: oracle with the same scales as FP32 (gran_n = 1), reference gate, and bit-equality between the
    kernels -- the matrix core accumulates the K blocks in place, in the same order, in all of them."""
    gen.reset_seed(m + k)
    case = gen.generate_normal(m, n, k, per_token_b=True, use_ue8m0=True)
    a, b = gen.packed_ue8m0_operand(*case.a), gen.packed_ue8m0_operand(*case.b)
    want = torch.empty((m, n), dtype=torch.bfloat16)
    oracle.fp8_gemm_nt(*cpu_pair(case.a), *cpu_pair(case.b), want, gran_n=1)
    first = None
    for cfg in E8_DENSE_CONFIGS:
        if cfg in E8_QUAD_256 and k % 512 != 0:
            continue                                    # whole packed words only (the other kernels take the K tail)
        dg.set_forced_config(cfg)
        d = torch.full((m, n), float('nan'), device='cuda', dtype=torch.bfloat16)
        dg.fp8_gemm_nt(a, b, d)
        assert dg.last_config().startswith('e8_'), dg.last_config()
        assert_close_to_oracle(d, want, cfg)
        assert calc_diff(d, case.ref_d) < gen.FP8_MAX_DIFF
        first = d if first is None else first
        assert torch.equal(d, first), f'{cfg} differs from {E8_DENSE_CONFIGS[0]}'
    # round 4: the nn layout -- B MN-major ([K][N]) with packed scales -- read in place by the 8-wave hardware-scaled kernel (transpose
    # reads, natural column order, one scale word per weight row): the same bits as every K-major kernel above; the automatic choice
    # between that and a re-majoring pass in front of the quad kernel gives those bits either way
    b_kn = b[0].t().contiguous()                                            # [K, N] storage
    sfb_kn = b[1].t().contiguous()                                          # the scale words travel transposed with it
    dg.set_forced_config('e8_duo_bmn_256x256')
    d = torch.full((m, n), float('nan'), device='cuda', dtype=torch.bfloat16)
    if n % 16 == 0:
        dg.fp8_gemm_nn(a, (b_kn, sfb_kn), d)
        assert dg.last_config() == 'e8_duo_bmn_256x256', dg.last_config()
        assert torch.equal(d, first), 'MN-major B read in place differs from the K-major kernels'
    else:                               # k-rows of B not 16-byte aligned: the host layer re-majors, and the MN-major kernel refuses a K-major B
        with pytest.raises(RuntimeError, match='MN-major'):
            dg.fp8_gemm_nn(a, (b_kn, sfb_kn), d)
    dg.set_forced_config('auto')
    d.fill_(float('nan'))
    dg.fp8_gemm_nn(a, (b_kn, sfb_kn), d)
    assert dg.last_config().startswith('e8_') and torch.equal(d, first)
    if (m, n, k) == (4096, 4096, 1536):                                      # (the model's verdict at this size: in place)
        assert dg.last_config() == 'e8_duo_bmn_256x256', dg.last_config()
    dg.set_forced_config('e8_quad_128x256')                                  # a K-major kernel forced onto the MN-major operand: re-majored first
    d.fill_(float('nan'))
    dg.fp8_gemm_nn(a, (b_kn, sfb_kn), d)
    assert dg.last_config() == 'e8_quad_128x256' and torch.equal(d, first)
    # ... and the tt / tn layouts: A MN-major ([K][M]; its scale words in natural row order), alone and together with an MN-major B
    a_km, sfa_km = a[0].t().contiguous(), a[1].t().contiguous()
    for op, name, b_arg, ok in ((dg.fp8_gemm_tt, 'e8_duo_amn_256x256', b, m % 16 == 0),
                                (dg.fp8_gemm_tn, 'e8_duo_abmn_256x256', (b_kn, sfb_kn), m % 16 == 0 and n % 16 == 0)):
        dg.set_forced_config(name)
        d.fill_(float('nan'))
        if ok:
            op((a_km, sfa_km), b_arg, d)
            assert dg.last_config() == name, dg.last_config()
            assert torch.equal(d, first), f'{name}: MN-major operands read in place differ from the K-major kernels'
        else:
            with pytest.raises(RuntimeError, match='MN-major'):
                op((a_km, sfa_km), b_arg, d)
        dg.set_forced_config('auto')
        d.fill_(float('nan'))
        op((a_km, sfa_km), b_arg, d)
        assert dg.last_config().startswith('e8_') and torch.equal(d, first), dg.last_config()
    dg.set_forced_config('e8_quad_256x256')
    if k % 512 != 0:
        with pytest.raises(RuntimeError, match='k % 512'):
            dg.fp8_gemm_nt(a, b, torch.empty_like(first))


@pytest.mark.parametrize('use_psum', [False, True])
def test_packed_ue8m0_m_grouped_contiguous(use_psum):
    """m_grouped_fp8_gemm_nt_contiguous / _nn_contiguous with int scale tensors (csrc/apis/gemm.hpp:217-231): the tests of the
    FP32-scale form mirrored -- oracle per group, zero padding rows, both tile forms (two-pass 256-row tiles where legal)."""
    gen.reset_seed(16)
    for actual_ms, n, k in (([100, 0, 130, 256], 256, 384), ([300, 77], 520, 512), ([128] * 8, 4096, 512),
                            ([128, 384, 0, 0, 200, 640], 512, 1024)):
        case = gen.generate_m_grouped_contiguous(len(actual_ms), 0, n, k, True, use_psum, actual_ms=actual_ms, use_ue8m0=True)
        want = torch.full(case.d.shape, float('nan'), dtype=torch.bfloat16)
        oracle.m_grouped_fp8_gemm_nt_contiguous(*cpu_pair(case.a), *cpu_pair(case.b), want, case.grouped_layout.cpu(), use_psum)
        a = gen.packed_ue8m0_operand(*case.a)
        b = gen.packed_ue8m0_operand(*case.b, mn_rows=n)
        cfgs = ['auto', 'e8_quad_128x256'] + (E8_QUAD_256 if not use_psum and k % 512 == 0 else [])
        # round 5: the nn form's MN-major weights read in place by the 8-wave kernel (two-pass 256-row tiles; contiguous layout without psum)
        cfgs += ['e8_duo_bmn_256x256'] if not use_psum and n % 16 == 0 else []
        for cfg in cfgs:
            for nn in ((True,) if cfg == 'e8_duo_bmn_256x256' else (False, True)):
                dg.set_forced_config(cfg)
                case.d.fill_(float('nan'))
                if nn:
                    b_nn = (b[0].mT.contiguous(), b[1])                # [G, K, N] storage; scales stay per row of B
                    dg.m_grouped_fp8_gemm_nn_contiguous(a, (b_nn[0], b_nn[1].mT), case.d, case.grouped_layout, use_psum_layout=use_psum)
                else:
                    dg.m_grouped_fp8_gemm_nt_contiguous(a, b, case.d, case.grouped_layout, use_psum_layout=use_psum)
                assert dg.last_config() == cfg if cfg == 'e8_duo_bmn_256x256' else dg.last_config().startswith('e8_quad'), dg.last_config()
                start = 0
                for actual, aligned in zip(case.actual_ms, case.aligned_ms):
                    rows = slice(start, start + actual)
                    assert_close_to_oracle(case.d[rows], want[rows], f'{cfg} nn={nn} rows {rows}')
                    assert bool((case.d[start + actual:start + aligned] == 0).all()), f'{cfg}: padding rows must be zeros'
                    start += aligned
                assert calc_diff(torch.nan_to_num(case.d), torch.nan_to_num(case.ref_d)) < gen.FP8_MAX_DIFF


def test_packed_ue8m0_m_grouped_nn_weights_in_place():
    """m_grouped_fp8_gemm_nn_contiguous with packed scales at a size where a pass over every group's weights costs more than the 8-wave
    kernel's slower K loop (dg_ue8m0_grouped_operand_plan answers 0): the automatic selection reads B [G, K, N] in place
    (csrc/apis/gemm.hpp:234-248: the SM100 path takes either majorness through its descriptors) -- same bits as the K-major call,
    padding rows zero, rows of every group against the oracle."""
    gen.reset_seed(23)
    actual_ms, n, k = [500, 384, 130, 640, 0, 256], 4096, 1024
    case = gen.generate_m_grouped_contiguous(len(actual_ms), 0, n, k, True, False, actual_ms=actual_ms, use_ue8m0=True)
    a = gen.packed_ue8m0_operand(*case.a)
    b = gen.packed_ue8m0_operand(*case.b, mn_rows=n)
    dg.m_grouped_fp8_gemm_nt_contiguous(a, b, case.d, case.grouped_layout)
    assert dg.last_config().startswith('e8_quad'), dg.last_config()
    k_major = case.d.clone()
    b_nn = b[0].mT.contiguous()                                    # [G, K, N] storage
    d = torch.full_like(case.d, float('nan'))
    dg.m_grouped_fp8_gemm_nn_contiguous(a, (b_nn, b[1].mT), d, case.grouped_layout)
    assert dg.last_config() == 'e8_duo_bmn_256x256', dg.last_config()
    want = torch.full(case.d.shape, float('nan'), dtype=torch.bfloat16)
    oracle.m_grouped_fp8_gemm_nt_contiguous(*cpu_pair(case.a), *cpu_pair(case.b), want, case.grouped_layout.cpu(), False)
    start = 0
    for actual, aligned in zip(case.actual_ms, case.aligned_ms):
        rows = slice(start, start + actual)
        assert torch.equal(d[rows], k_major[rows]), f'rows {rows}: in place vs K-major'
        if actual:
            assert_close_to_oracle(d[rows], want[rows], f'nn in place, rows {rows}')
        assert bool((d[start + actual:start + aligned] == 0).all()), 'padding rows must be zeros'
        start += aligned


@pytest.mark.parametrize('m,n,k', [(256, 384, 576), (130, 264, 2112), (1040, 784, 2112), (512, 1024, 144), (300, 520, 656)])
def test_packed_ue8m0_k_tail(m, n, k):
    """Packed UE8M0 scales with K not a multiple of 128 (the reference's SM100 kernels take any K through TMA zero-fill,
    csrc/jit_kernels/impls/sm100_fp8_fp4_gemm_1d1d.hpp:93): K-major operands with whole 16-byte chunks stay on the hardware-scaled path
    (round 4: e8_quad_kt_128x256, the partial last block zero-filled by the buffer range check) -- oracle parity, accumulation included,
    row padding full of FP8 NaNs never reaches the matrix core; operands that path does not take (MN-major: re-majored first; rows off
    16 bytes: exponents expanded to exact FP32 scales, recipe (1, 1, 128)) give the same answer."""
    gen.reset_seed(m + k)
    for accumulate in (False, True):
        case = gen.generate_normal(m, n, k, use_ue8m0=True, accumulate=accumulate, out_dtype=torch.float if accumulate else torch.bfloat16)
        c_cpu = case.c.cpu().clone() if accumulate else None
        want = oracle_dense(case, c_cpu=c_cpu)
        a, b = gen.packed_ue8m0_operand(*case.a), gen.packed_ue8m0_operand(*case.b, mn_rows=n)
        assert a[1].dtype == torch.int and a[1].shape == (m, -(-k // 512))
        # operands inside wider buffers whose padding holds FP8 NaN bytes (0x7f): rows 16-byte aligned, K-major
        a_wide = torch.full((m, k + 48), 0x7f, dtype=torch.uint8, device='cuda')
        b_wide = torch.full((n, k + 80), 0x7f, dtype=torch.uint8, device='cuda')
        a_wide[:, :k] = a[0].view(torch.uint8)
        b_wide[:, :k] = b[0].view(torch.uint8)
        a_pad, b_pad = (a_wide[:, :k].view(torch.float8_e4m3fn), a[1]), (b_wide[:, :k].view(torch.float8_e4m3fn), b[1])
        dg.fp8_gemm_nt(a_pad, b_pad, case.d, c=case.c if accumulate else None)
        assert dg.last_config() == 'e8_quad_kt_128x256', dg.last_config()
        if accumulate:
            assert_close_fp32(case.d, want, 'packed scales, K tail, accumulate')
        else:
            assert_close_to_oracle(case.d, want, 'packed scales, K tail')
        assert calc_diff(case.d, case.ref_d) < gen.FP8_MAX_DIFF
        first = case.d.clone()
        if not accumulate:
            # bit-repeatable; contiguous operands: same bits; MN-major B (fp8_gemm_nn: re-majored, then the same kernel): same bits
            d2 = torch.full_like(first, float('nan'))
            dg.fp8_gemm_nt(a, b, d2)
            assert dg.last_config() == 'e8_quad_kt_128x256' and torch.equal(d2, first)
            # round 5: MN-major B ([K][N]: fp8_gemm_nn, the packed-scale dgrad layout) read IN PLACE by the 8-wave hardware-scaled kernel with
            # the same zero-filled partial block: the same bits (the matrix core accumulates the K blocks in place, in the same order)
            if n % 16 == 0:
                b_kn, sfb_kn = b[0].t().contiguous(), b[1].t().contiguous()         # [K, N] storage; the scale words travel transposed with it
                dg.set_forced_config('e8_duo_bmn_kt_256x256')
                d3 = torch.full_like(first, float('nan'))
                dg.fp8_gemm_nn(a, (b_kn, sfb_kn), d3)
                assert dg.last_config() == 'e8_duo_bmn_kt_256x256', dg.last_config()
                assert torch.equal(d3, first), 'MN-major B in place differs from the K-major tail kernel'
                dg.set_forced_config('auto')
                d3.fill_(float('nan'))
                dg.fp8_gemm_nn(a, (b_kn, sfb_kn), d3)                                # automatic: in place where that pays, re-majored otherwise
                assert dg.last_config() in ('e8_duo_bmn_kt_256x256', 'e8_quad_kt_128x256') and torch.equal(d3, first)
            # rows off 16 bytes: the expanded-scale fallback (FP32 promotion instead of in-core accumulation: equal up to FP32 rounding)
            a_off = torch.empty((a[0].numel() + 1,), dtype=torch.uint8, device='cuda')[1:].view(torch.float8_e4m3fn).view(a[0].shape)
            a_off.copy_(a[0])
            d4 = torch.full_like(first, float('nan'))
            dg.fp8_gemm_nt((a_off, a[1]), b, d4)
            assert not dg.last_config().startswith('e8_'), dg.last_config()
            assert_close_to_oracle(d4, want, 'packed scales, K tail, unaligned rows')
    dg.set_forced_config('e8_quad_128x256')
    with pytest.raises(RuntimeError, match='k % 128'):
        dg.fp8_gemm_nt(a, b, torch.empty((m, n), device='cuda', dtype=torch.bfloat16))
    dg.set_forced_config('auto')


@pytest.mark.parametrize('masked_ms,max_m,n,k', [([5, 0, 64, 33], 64, 256, 384), ([200, 1, 129], 256, 520, 512),
                                                  ([20] * 6 + [0, 64], 64, 4096, 512), ([700, 130], 1024, 768, 1024)])
def test_packed_ue8m0_m_grouped_masked(masked_ms, max_m, n, k):
    """m_grouped_fp8_gemm_nt_masked with int scale tensors (csrc/apis/gemm.hpp:280-296): oracle on the valid rows, NaN poison
    on the rows >= masked_m."""
    gen.reset_seed(17)
    case = gen.generate_m_grouped_masked(len(masked_ms), max_m, 0, n, k, masked_ms=masked_ms, use_ue8m0=True)
    want = torch.full(case.d.shape, float('nan'), dtype=torch.bfloat16)
    oracle.m_grouped_fp8_gemm_nt_masked(*cpu_pair(case.a), *cpu_pair(case.b), want, case.masked_m.cpu())
    a, b = gen.packed_ue8m0_operand(*case.a), gen.packed_ue8m0_operand(*case.b, mn_rows=n)
    expected_m = max(1, int(sum(masked_ms) / len(masked_ms)))
    for cfg in ['auto', 'e8_quad_128x256', 'e8_stream_64x128', 'e8_stream_nt_64x128', 'e8_stream2_64x128', 'e8_stream_nt2_64x128', 'e8_stream_64x32'] + (E8_QUAD_256 if k % 512 == 0 else []):
        dg.set_forced_config(cfg)
        case.d.fill_(float('nan'))
        dg.m_grouped_fp8_gemm_nt_masked(a, b, case.d, case.masked_m, expected_m)
        assert dg.last_config().startswith('e8_'), dg.last_config()
        for g, rows in enumerate(masked_ms):
            if rows:
                assert_close_to_oracle(case.d[g, :rows], want[g, :rows], f'{cfg} group {g}')
                assert calc_diff(case.d[g, :rows], case.ref_d[g, :rows]) < gen.FP8_MAX_DIFF
            assert bool(torch.isnan(case.d[g, rows:]).all()), f'{cfg}: rows >= masked_m must not be written'


@pytest.mark.parametrize('k_major', [True, False])
@pytest.mark.parametrize('num_groups,m,n,ks', [(3, 256, 384, [256, 0, 512]), (2, 200, 264, [128, 384]), (2, 304, 272, [256, 384]),
                                               (4, 512, 1024, [1024, 896, 1152, 768])])
def test_k_grouped_contiguous(k_major, num_groups, m, n, ks):
    """k_grouped_fp8_gemm_{nt,tn}_contiguous (csrc/apis/gemm.hpp:299-400; reference test: tests/test_fp8_fp4.py:193-215):
    FP32 ``d[g] = c[g] + A_g @ B_g^T`` with per-channel scales; an empty group leaves ``d[g] = c[g]``."""
    gen.reset_seed(sum(ks) + m)
    case = gen.generate_k_grouped_contiguous(num_groups, m, n, ks, k_major)
    fn = dg.k_grouped_fp8_gemm_nt_contiguous if k_major else dg.k_grouped_fp8_gemm_tn_contiguous
    c_before = case.c.clone()
    fn(case.a, case.b, case.d, ks, case.grouped_layout, c=case.c)
    assert torch.equal(case.c, c_before)                       # c is a different buffer here: read, never written
    if m > 64 and m % 16 == 0 and n % 16 == 0:
        # one launch over all groups; MN-major operands go in as they are (hardware transpose reads), no re-majoring pass
        assert dg.last_config() == ('pipe_pc_256x256' if k_major else 'pipe_pc_mn_256x256')
    for g, k in enumerate(ks):
        if k == 0:
            assert torch.equal(case.d[g], case.c[g])
            continue
        (a_g, sfa_g), (b_g, sfb_g) = case.a_groups[g], case.b_groups[g]
        want = torch.empty((m, n), dtype=torch.float)
        oracle.fp8_gemm_nt(a_g.cpu(), sfa_g.cpu(), b_g.cpu(), sfb_g.cpu(), want, c=case.c[g].cpu(), gran_n=1)
        assert_close_fp32(case.d[g], want, f'k-grouped group {g}')
    assert calc_diff(case.d, case.ref_d) < gen.FP8_MAX_DIFF
    # in place (c is d), the reference test's calling convention
    d2 = case.c.clone()
    fn(case.a, case.b, d2, ks, case.grouped_layout, c=d2)
    assert torch.equal(d2, case.d)


@pytest.mark.parametrize('num_groups,m,n,real_ks', [(4, 256, 384, [300, 0, 129, 512]), (2, 304, 272, [1000, 77]), (3, 512, 1024, [128, 1, 640]),
                                                    (2, 64, 128, [200, 256]),
                                                    (72, 128, 256, [(37 * i) % 300 for i in range(72)])])      # more groups than fit in kernel arguments
def test_k_grouped_tn_psum_layout(num_groups, m, n, real_ks):
    """k_grouped_fp8_gemm_tn_contiguous(use_psum_layout=True) (csrc/apis/gemm.hpp:299-346, scheduler/gemm.cuh:74-85; reference test:
    tests/test_fp8_fp4.py:198-225): K ranges read from the device tensor of group ENDS -- with ``ks_cpu`` given, missing or empty
    ("unsynced" calls) -- groups of any real K inside zero-padded 128-row blocks, empty groups, in-place accumulation."""
    gen.reset_seed(sum(real_ks) + m)
    case = gen.generate_k_grouped_contiguous_psum(num_groups, m, n, real_ks)
    results = []
    for ks_cpu in (case.ks, None, []):
        if m <= 64 and not ks_cpu:
            with pytest.raises(RuntimeError, match='ks_cpu.has_value'):       # no single-launch kernel at this size: the host must know the extents
                dg.k_grouped_fp8_gemm_tn_contiguous(case.a, case.b, case.c.clone(), ks_cpu, case.grouped_layout, c=case.c, use_psum_layout=True)
            continue
        d = case.c.clone()
        dg.k_grouped_fp8_gemm_tn_contiguous(case.a, case.b, d, ks_cpu, case.grouped_layout, c=d, use_psum_layout=True)
        if m > 64 and m % 16 == 0 and n % 16 == 0:
            assert dg.last_config() == 'pipe_pc_mn_256x256', dg.last_config()     # operands in place, ranges from the device
        results.append(d)
    for other in results[1:]:
        assert torch.equal(other, results[0])
    d = results[0]
    for g, k in enumerate(real_ks):
        if k == 0:
            assert torch.equal(d[g], case.c[g])
            continue
        (a_g, sfa_g), (b_g, sfb_g) = case.a_groups[g], case.b_groups[g]
        want = torch.empty((m, n), dtype=torch.float)
        oracle.fp8_gemm_nt(a_g.cpu(), sfa_g.cpu(), b_g.cpu(), sfb_g.cpu(), want, c=case.c[g].cpu(), gran_n=1)
        assert_close_fp32(d[g], want, f'k-grouped psum group {g}')
    assert calc_diff(d, case.ref_d) < gen.FP8_MAX_DIFF
    # the same problem through the non-psum operator (host extents = the aligned sizes): same bits
    d2 = case.c.clone()
    dg.k_grouped_fp8_gemm_tn_contiguous(case.a, case.b, d2, case.ks, torch.tensor(case.ks, device='cuda', dtype=torch.int32), c=d2)
    assert torch.equal(d2, d)
    # operands the in-place form does not take (rows off 16 bytes): re-majored, still device-side ranges
    if m > 64:
        a_off = torch.empty((case.a[0].numel() + 1,), dtype=torch.uint8, device='cuda')[1:].view(torch.float8_e4m3fn).view(case.a[0].shape)
        a_off.copy_(case.a[0])
        d3 = case.c.clone()
        dg.k_grouped_fp8_gemm_tn_contiguous((a_off, case.a[1]), case.b, d3, None, case.grouped_layout, c=d3, use_psum_layout=True)
        assert dg.last_config() in ('pipe_pc_256x256', per_col_tile_name(m, n)) and torch.equal(d3, d)


@pytest.mark.parametrize('k_alignment', [32, 160, 192, 224])
@pytest.mark.parametrize('num_groups,m,n,real_ks', [(4, 256, 384, [300, 0, 129, 512]), (3, 512, 1024, [128, 1, 640]), (2, 304, 272, [1000, 77]),
                                                    (70, 128, 256, [(37 * i) % 300 for i in range(70)])])
def test_k_grouped_tn_psum_layout_at_other_k_alignments(num_groups, m, n, real_ks, k_alignment):
    """The psum form with a K alignment that is not the scale granularity (round 6; the reference's SM100 sweep, tests/generators.py:192-194:
    gran_k 128 with alignments 160 / 224, here also 32 / 192; scheduler/gemm.cuh:74-85, 238-261): groups start at multiples of the alignment,
    scale rows are compact and count from each group's start, the last 128-block of a group is partial -- the rows behind a group's end belong
    to the NEXT group (alignment 32) and must not contribute.  Every group against the oracle on its own operands."""
    gen.reset_seed(sum(real_ks) + m + k_alignment)
    dg.set_mk_alignment_for_contiguous_layout(k_alignment)
    try:
        case = gen.generate_k_grouped_contiguous_psum(num_groups, m, n, real_ks, k_alignment)
        ends = case.grouped_layout.tolist()
        a_q, b_q = case.a[0].clone(), case.b[0].clone()
        results = []
        for ks_cpu in (case.ks, None):
            d = case.c.clone()
            dg.k_grouped_fp8_gemm_tn_contiguous((a_q, case.a[1]), (b_q, case.b[1]), d, ks_cpu, case.grouped_layout, c=d, use_psum_layout=True)
            assert dg.last_config() == 'pipe_pc_mn_256x256', dg.last_config()
            results.append(d)
        assert torch.equal(results[0], results[1])
        d = results[0]
        for g, k in enumerate(real_ks):
            if k == 0:
                assert torch.equal(d[g], case.c[g])
                continue
            (a_g, sfa_g), (b_g, sfb_g) = case.a_groups[g], case.b_groups[g]
            want = torch.empty((m, n), dtype=torch.float)
            oracle.fp8_gemm_nt(a_g.cpu(), sfa_g.cpu(), b_g.cpu(), sfb_g.cpu(), want, c=case.c[g].cpu(), gran_n=1)
            assert_close_fp32(d[g], want, f'k-grouped psum, K alignment {k_alignment}, group {g}')
        assert calc_diff(d, case.ref_d) < gen.FP8_MAX_DIFF
        # garbage in the rows between a group's end and the next group's start must not reach any result (the reference zero-fills them through
        # TMA; here the operand descriptors end at the group's end)
        prev = 0
        for e in ends:
            start = -(-prev // k_alignment) * k_alignment
            a_q[prev:start] = torch.full((1,), 448.0, device='cuda').to(torch.float8_e4m3fn)
            b_q[prev:start] = torch.full((1,), -448.0, device='cuda').to(torch.float8_e4m3fn)
            prev = e
        d2 = case.c.clone()
        dg.k_grouped_fp8_gemm_tn_contiguous((a_q, case.a[1]), (b_q, case.b[1]), d2, None, case.grouped_layout, c=d2, use_psum_layout=True)
        assert torch.equal(d2, d)
    finally:
        dg.set_mk_alignment_for_contiguous_layout(128)


@pytest.mark.parametrize('m,n,k', [(256, 224, 256), (512, 448, 512), (300, 672, 384), (1024, 1568, 1024), (256, 7168, 512), (520, 500, 640)])
def test_duo_256x224_tiles(m, n, k):
    """The 256 x 224 tile family (round 6: wave tile 64 x 112, two SFB values per wave tile -- the reference's straddle logic,
    sm90_fp8_gemm_1d2d.cuh:232-237, 290-291, 342-346 -- and a seventh N-subtile in natural column order): every wave-tile / SFB-boundary
    alignment (n0 mod 128 takes 0, 96, 64, 32 over four consecutive tiles), ragged M, N that is not a multiple of 224, BF16 and FP32 outputs
    with accumulation -- against the oracle, and bit-identical to the 256 x 256 kernel (same K-block order per accumulator)."""
    if 'duo_p_256x224' not in dg.list_configs():
        pytest.skip('a measured negative (profiles/r06_probe/n224_tile_family_negative.log): DG_EXPERIMENTS builds only')
    gen.reset_seed(m + n + k)
    case = gen.generate_normal(m, n, k)
    want = oracle_dense(case)
    dg.set_forced_config('duo_p_256x224')
    try:
        d = torch.full((m, n), float('nan'), device='cuda', dtype=torch.bfloat16)
        dg.fp8_gemm_nt(case.a, case.b, d)
        assert dg.last_config() == 'duo_p_256x224'
        c32 = torch.randn((m, n), device='cuda', dtype=torch.float)
        d32 = c32.clone()
        dg.fp8_gemm_nt(case.a, case.b, d32, c=d32)
        cb = torch.randn((m, n), device='cuda', dtype=torch.bfloat16)
        db = cb.clone()
        dg.fp8_gemm_nt(case.a, case.b, db, c=db)
    finally:
        dg.set_forced_config('auto')
    assert_close_to_oracle(d, want, '256 x 224 tiles')
    assert calc_diff(d, case.ref_d) < gen.FP8_MAX_DIFF
    dg.set_forced_config('duo_p_256x256')
    try:
        d256 = torch.empty_like(d)
        dg.fp8_gemm_nt(case.a, case.b, d256)
        e32 = c32.clone()
        dg.fp8_gemm_nt(case.a, case.b, e32, c=e32)
        eb = cb.clone()
        dg.fp8_gemm_nt(case.a, case.b, eb, c=eb)
    finally:
        dg.set_forced_config('auto')
    assert torch.equal(d.view(torch.int16), d256.view(torch.int16))
    assert torch.equal(d32, e32) and torch.equal(db.view(torch.int16), eb.view(torch.int16))


@pytest.mark.parametrize('m,n,k', [(128, 4096, 7168), (65, 512, 1024), (200, 1000, 2048), (256, 2112, 7168), (128, 7168, 2048), (96, 128, 512)])
def test_stream_tile_with_in_kernel_k_split(m, n, k):
    """`stream_ks_64x128` (round 6): the 64 x 128 stream tile with every tile cut along K into pieces that exchange their FP32 partials
    inside the kernel (the last piece of a tile sums them in piece order): against the oracle, bit-repeatable, BF16 / FP32 outputs with
    accumulation, ragged M and N, K ranges that do not divide evenly -- and without a workspace (whole tiles) the bits of the plain tile."""
    gen.reset_seed(m + n + k)
    case = gen.generate_normal(m, n, k)
    want = oracle_dense(case)
    dg.set_forced_config('stream_ks_64x128')
    try:
        d = torch.full((m, n), float('nan'), device='cuda', dtype=torch.bfloat16)
        dg.fp8_gemm_nt(case.a, case.b, d)
        assert dg.last_config() == 'stream_ks_64x128'
        again = torch.full_like(d, float('nan'))
        dg.fp8_gemm_nt(case.a, case.b, again)
        c32 = torch.randn((m, n), device='cuda', dtype=torch.float)
        d32 = c32.clone()
        dg.fp8_gemm_nt(case.a, case.b, d32, c=d32)
    finally:
        dg.set_forced_config('auto')
    assert_close_to_oracle(d, want, 'stream tile, in-kernel K split')
    assert torch.equal(d.view(torch.int16), again.view(torch.int16)), 'piece order is fixed: bit-repeatable'
    assert calc_diff(d, case.ref_d) < gen.FP8_MAX_DIFF
    want32 = torch.empty((m, n), dtype=torch.float)
    oracle.fp8_gemm_nt(*cpu_pair(case.a), *cpu_pair(case.b), want32, c=c32.cpu())
    assert_close_fp32(d32, want32, 'stream tile, in-kernel K split, fp32 accumulate')


@pytest.mark.parametrize('m,n,k', [(192, 4096, 7168), (256, 2112, 7168), (129, 4096, 4096), (192, 1536, 10240), (128, 6144, 7168), (100, 4096, 10240), (65, 7168, 8192)])
def test_mid_m_dense_calls_take_the_k_split_stream_tile(m, n, k):
    """The automatic selection (round 6): 129 .. 256 rows whose 64 x 128 tiles fill at most half the chip and K >= 4096 run `stream_ks_64x128`
    through the plain entry (the host layer lends the workspace: dg_dense_wants_workspace), with the oracle's result; repeated calls (the
    exchange epoch advances, the workspace is reused dirty) give the same bits.  Last session: 33 .. 63 tiles from K = 10240, and 65 .. 128 rows on
    wide layers (from 96 tiles with K = 7168 .. 10240, 64 .. 95 tiles from K = 10240)."""
    gen.reset_seed(3 * m + n + k)
    case = gen.generate_normal(m, n, k)
    want = oracle_dense(case)
    outs = []
    for _ in range(4):
        d = torch.full((m, n), float('nan'), device='cuda', dtype=torch.bfloat16)
        dg.fp8_gemm_nt(case.a, case.b, d)
        assert dg.last_config() == 'stream_ks_64x128'
        outs.append(d)
    assert_close_to_oracle(outs[0], want, 'mid-M dense, automatic K-split stream tile')
    assert all(torch.equal(outs[0].view(torch.int16), o.view(torch.int16)) for o in outs[1:])


def test_mid_m_k_split_in_a_hip_graph():
    """A captured `stream_ks_64x128` launch carries ONE exchange epoch: every replay must wait for ITS partials (the last piece takes the flags
    back) -- replays over changing inputs against eager calls, bit for bit."""
    m, n, k = 192, 4096, 4096
    cases = []
    for i in range(3):
        gen.reset_seed(40 + i)
        cases.append(gen.generate_normal(m, n, k))
    eager = []
    for c in cases:
        d = torch.empty((m, n), device='cuda', dtype=torch.bfloat16)
        dg.fp8_gemm_nt(c.a, c.b, d)
        assert dg.last_config() == 'stream_ks_64x128'
        eager.append(d)
    a = (cases[0].a[0].clone(), dg.get_mn_major_tma_aligned_tensor(cases[0].a[1]).clone())
    b = (cases[0].b[0].clone(), cases[0].b[1].clone())
    d = torch.empty((m, n), device='cuda', dtype=torch.bfloat16)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        dg.fp8_gemm_nt(a, b, d)                                     # (warm: plan caches, the stream's workspace)
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        dg.fp8_gemm_nt(a, b, d)
    for which in (1, 2, 0, 2, 1):
        c = cases[which]
        a[0].copy_(c.a[0]); a[1].copy_(dg.get_mn_major_tma_aligned_tensor(c.a[1])); b[0].copy_(c.b[0]); b[1].copy_(c.b[1])
        d.fill_(float('nan'))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(d.view(torch.int16), eager[which].view(torch.int16)), which


@pytest.mark.parametrize('seed', [11, 12, 13])
def test_mid_m_k_split_random_shapes(seed):
    """Random problems inside the `stream_ks_64x128` rule (129 .. 256 rows, ragged N, K = 4096 .. 8192 in whole blocks), BF16 / FP32 outputs with
    and without accumulation, each against the oracle."""
    rng = random.Random(seed)
    taken = 0
    for _ in range(6):
        m = rng.randrange(129, 257)
        n = rng.choice([3000, 3584, 4096])          # 72 .. 128 tiles of 64 x 128: inside the rule
        k = 128 * rng.randrange(32, 65)
        accumulate = rng.random() < 0.4
        out_dtype = torch.float if rng.random() < 0.4 else torch.bfloat16
        gen.reset_seed(seed + m + n + k)
        case = gen.generate_normal(m, n, k, accumulate=accumulate, out_dtype=out_dtype)
        c_cpu = case.c.cpu().clone() if accumulate else None
        want = oracle_dense(case, c_cpu=c_cpu)
        dg.fp8_gemm_nt(case.a, case.b, case.d, c=case.c if accumulate else None)
        label = f'm={m} n={n} k={k} acc={accumulate} {out_dtype} [{dg.last_config()}]'
        taken += dg.last_config() == 'stream_ks_64x128'
        if out_dtype == torch.float:
            assert_close_fp32(case.d, want, label)
        else:
            assert_close_to_oracle(case.d, want, label, addend=c_cpu)
    assert taken == 6, taken


@pytest.mark.parametrize('m,n,k', [(128, 576, 7168), (33, 4096, 7168), (65, 520, 4096), (1, 40, 4608), (200, 96, 5120), (256, 576, 16384), (24, 576, 7168),
                                   (17, 1536, 8192)])
def test_narrow_stream_tile_with_in_kernel_k_split(m, n, k):
    """`stream_ks_64x32` (end of round 6): the 64 x 32 stream tile (four K blocks per stage) cut along K inside the kernel -- narrow layers at small
    M, e.g. the MLA down-projection n = 576 of the reference's sweep: forced by name against the oracle, bit-repeatable, FP32 accumulation, ragged M
    and N, piece boundaries inside a stage; through the plain entry the automatic selection takes it (the host layer lends the workspace) and
    repeated calls -- the epoch advances, the workspace is reused dirty -- give the same bits."""
    gen.reset_seed(m + n + k)
    case = gen.generate_normal(m, n, k)
    want = oracle_dense(case)
    dg.set_forced_config('stream_ks_64x32')
    try:
        d = torch.full((m, n), float('nan'), device='cuda', dtype=torch.bfloat16)
        dg.fp8_gemm_nt(case.a, case.b, d)
        assert dg.last_config() == 'stream_ks_64x32'
        again = torch.full_like(d, float('nan'))
        dg.fp8_gemm_nt(case.a, case.b, again)
        c32 = torch.randn((m, n), device='cuda', dtype=torch.float)
        d32 = c32.clone()
        dg.fp8_gemm_nt(case.a, case.b, d32, c=d32)
    finally:
        dg.set_forced_config('auto')
    assert_close_to_oracle(d, want, 'narrow stream tile, in-kernel K split')
    assert torch.equal(d.view(torch.int16), again.view(torch.int16)), 'piece order is fixed: bit-repeatable'
    assert calc_diff(d, case.ref_d) < gen.FP8_MAX_DIFF
    want32 = torch.empty((m, n), dtype=torch.float)
    oracle.fp8_gemm_nt(*cpu_pair(case.a), *cpu_pair(case.b), want32, c=c32.cpu())
    assert_close_fp32(d32, want32, 'narrow stream tile, in-kernel K split, fp32 accumulate')
    # (17 .. 32 rows: narrow layers -- at most 48 tiles -- with K >= 7168 leave the skinny kernel for this tile)
    takes_64x64 = 32 < m <= 128 and k >= 7168 and 48 <= -(-m // 64) * -(-n // 64) <= 85       # (three or more pieces of a 64 x 64 tile: the test below)
    if (m > 32 or (m > 16 and k >= 7168 and -(-n // 32) <= 48)) and -(-m // 64) * -(-n // 32) * 2 <= 256 and not takes_64x64:
        outs = []
        for _ in range(3):
            o = torch.full((m, n), float('nan'), device='cuda', dtype=torch.bfloat16)
            dg.fp8_gemm_nt(case.a, case.b, o)
            assert dg.last_config() == 'stream_ks_64x32', dg.last_config()
            outs.append(o)
        assert all(torch.equal(d.view(torch.int16), o.view(torch.int16)) for o in outs)


@pytest.mark.parametrize('m,n,k', [(128, 2112, 7168), (33, 4096, 7168), (100, 1544, 8192), (64, 5000, 7168), (1, 72, 4608), (200, 200, 5120)])
def test_64x64_stream_tile_with_in_kernel_k_split(m, n, k):
    """`stream_ks_64x64` (last session of round 6): a 64 x 64 stream tile (two K blocks per stage, four stages) cut along K inside the kernel, for
    33 .. 128 rows where its tiles get three or more pieces (48 .. CUs / 3 tiles, K >= 7168) -- forced by name against the oracle (ragged M and N,
    odd K block counts, FP32 accumulation), bit-repeatable; inside the rule the plain entry takes it and repeated calls on the dirty workspace
    give the same bits."""
    gen.reset_seed(m + n + k + 64)
    case = gen.generate_normal(m, n, k)
    want = oracle_dense(case)
    dg.set_forced_config('stream_ks_64x64')
    try:
        d = torch.full((m, n), float('nan'), device='cuda', dtype=torch.bfloat16)
        dg.fp8_gemm_nt(case.a, case.b, d)
        assert dg.last_config() == 'stream_ks_64x64'
        again = torch.full_like(d, float('nan'))
        dg.fp8_gemm_nt(case.a, case.b, again)
        c32 = torch.randn((m, n), device='cuda', dtype=torch.float)
        d32 = c32.clone()
        dg.fp8_gemm_nt(case.a, case.b, d32, c=d32)
    finally:
        dg.set_forced_config('auto')
    assert_close_to_oracle(d, want, '64 x 64 stream tile, in-kernel K split')
    assert torch.equal(d.view(torch.int16), again.view(torch.int16)), 'piece order is fixed: bit-repeatable'
    assert calc_diff(d, case.ref_d) < gen.FP8_MAX_DIFF
    want32 = torch.empty((m, n), dtype=torch.float)
    oracle.fp8_gemm_nt(*cpu_pair(case.a), *cpu_pair(case.b), want32, c=c32.cpu())
    assert_close_fp32(d32, want32, '64 x 64 stream tile, in-kernel K split, fp32 accumulate')
    if 32 < m <= 128 and k >= 7168 and 48 <= -(-m // 64) * -(-n // 64) <= 85:
        outs = []
        for _ in range(3):
            o = torch.full((m, n), float('nan'), device='cuda', dtype=torch.bfloat16)
            dg.fp8_gemm_nt(case.a, case.b, o)
            assert dg.last_config() == 'stream_ks_64x64', dg.last_config()
            outs.append(o)
        assert all(torch.equal(d.view(torch.int16), o.view(torch.int16)) for o in outs)


def test_narrow_k_split_in_a_hip_graph():
    """A captured `stream_ks_64x32` launch (one exchange epoch per capture): replays over changing inputs against eager calls, bit for bit."""
    m, n, k = 128, 576, 7168
    cases = []
    for i in range(3):
        gen.reset_seed(70 + i)
        cases.append(gen.generate_normal(m, n, k))
    eager = []
    for c in cases:
        d = torch.empty((m, n), device='cuda', dtype=torch.bfloat16)
        dg.fp8_gemm_nt(c.a, c.b, d)
        assert dg.last_config() == 'stream_ks_64x32'
        eager.append(d)
    a = (cases[0].a[0].clone(), dg.get_mn_major_tma_aligned_tensor(cases[0].a[1]).clone())
    b = (cases[0].b[0].clone(), cases[0].b[1].clone())
    d = torch.empty((m, n), device='cuda', dtype=torch.bfloat16)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        dg.fp8_gemm_nt(a, b, d)                                     # (warm: plan caches, the stream's workspace)
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        dg.fp8_gemm_nt(a, b, d)
    for which in (1, 2, 0, 2, 1):
        c = cases[which]
        a[0].copy_(c.a[0]); a[1].copy_(dg.get_mn_major_tma_aligned_tensor(c.a[1])); b[0].copy_(c.b[0]); b[1].copy_(c.b[1])
        d.fill_(float('nan'))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(d.view(torch.int16), eager[which].view(torch.int16)), which


def test_k_grouped_argument_checks():
    gen.reset_seed(1)
    case = gen.generate_k_grouped_contiguous(2, 128, 128, [128, 256], True)
    with pytest.raises(RuntimeError, match='c.has_value'):
        dg.k_grouped_fp8_gemm_nt_contiguous(case.a, case.b, case.d, case.ks, case.grouped_layout)
    with pytest.raises(RuntimeError, match='k % k_alignment == 0'):
        dg.k_grouped_fp8_gemm_nt_contiguous(case.a, case.b, case.d, [100, 284], case.grouped_layout, c=case.c)
    with pytest.raises(RuntimeError, match='ks_cpu'):
        dg.k_grouped_fp8_gemm_nt_contiguous(case.a, case.b, case.d, None, case.grouped_layout, c=case.c)
    with pytest.raises(RuntimeError, match='recipe'):
        dg.k_grouped_fp8_gemm_nt_contiguous(case.a, case.b, case.d, case.ks, case.grouped_layout, c=case.c, recipe=(1, 128, 128))


def _apply_skip_head_mid(d: torch.Tensor, head_splits, fill: float):
    """tests/test_attention.py:19-31 with a recognisable filler in the middle columns."""
    left, mid, right = head_splits
    m, n = d.shape
    heads = n // (left + right)
    d = d.view(m, heads, -1)
    pad = torch.full((m, heads, mid), fill, dtype=d.dtype, device=d.device)
    return torch.cat([d[:, :, :left], pad, d[:, :, left:]], dim=2).reshape(m, -1)


@pytest.mark.parametrize('head_splits', [(128, 64, 128), (64, 8, 192), (24, 4, 40)])
@pytest.mark.parametrize('m,heads,k', [(128, 32, 512), (300, 5, 384), (4096, 8, 512)])
@pytest.mark.parametrize('out_dtype', [torch.bfloat16, torch.float])
def test_gemm_skip_head_mid(head_splits, m, heads, k, out_dtype):
    """fp8_gemm_nt_skip_head_mid (reference test: tests/test_attention.py:34-53): same values as fp8_gemm_nt, scattered
    around the reserved middle columns, which must stay untouched."""
    left, mid, right = head_splits
    n = heads * (left + right)
    gen.reset_seed(m + n)
    case = gen.generate_normal(m, n, k, out_dtype=out_dtype)
    want = oracle_dense(case)
    plain = torch.empty_like(case.d)
    dg.fp8_gemm_nt(case.a, case.b, plain)
    d = _apply_skip_head_mid(torch.zeros_like(case.d), head_splits, fill=-7.0).contiguous()
    dg.fp8_gemm_nt_skip_head_mid(case.a, case.b, d, head_splits)
    assert torch.equal(d, _apply_skip_head_mid(plain, head_splits, fill=-7.0))      # same kernel, same bits, gaps kept
    got_cols = d.view(m, heads, left + mid + right)
    compact = torch.cat([got_cols[:, :, :left], got_cols[:, :, left + mid:]], dim=2).reshape(m, n)
    if out_dtype == torch.float:
        assert_close_fp32(compact, want, 'skip_head_mid')
    else:
        assert_close_to_oracle(compact, want, 'skip_head_mid')
    assert calc_diff(compact, case.ref_d) < gen.FP8_MAX_DIFF
    with pytest.raises(RuntimeError, match='left \\+ right'):
        dg.fp8_gemm_nt_skip_head_mid(case.a, case.b, d[:, :-8], head_splits)


def test_hip_graph_capture_and_replay():
    """The operators never allocate or synchronise once their SF operands are in the kernel's layout, so a sequence of them
    can be captured into a hipGraph and replayed (reference: CUDA-graph capturable after the first call,
    jit/kernel_runtime.hpp:139-162)."""
    gen.reset_seed(11)
    dense = gen.generate_normal(512, 768, 1024)
    dense.a = (dense.a[0], dg.get_mn_major_tma_aligned_tensor(dense.a[1]))
    masked = gen.generate_m_grouped_masked(4, 128, 48, 512, 512)
    masked_a = (masked.a[0], dg.get_mn_major_tma_aligned_tensor(masked.a[1]))
    skinny = gen.generate_normal(1024, 512, 8192)          # K split: two kernels per call and the per-stream workspace
    skinny.a = (skinny.a[0], dg.get_mn_major_tma_aligned_tensor(skinny.a[1]))
    # eager results (also warms every lazily initialised piece of host state, the K-split workspace of this stream included)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        dg.fp8_gemm_nt(skinny.a, skinny.b, skinny.d)
        assert dg.last_config() == 'duo_sk_128x256'
    side.synchronize()
    dg.fp8_gemm_nt(dense.a, dense.b, dense.d)
    dg.m_grouped_fp8_gemm_nt_masked(masked_a, masked.b, masked.d, masked.masked_m, 48)
    torch.cuda.synchronize()
    want_dense, want_masked, want_skinny = dense.d.clone(), masked.d.clone(), skinny.d.clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        dg.fp8_gemm_nt(dense.a, dense.b, dense.d)
        dg.m_grouped_fp8_gemm_nt_masked(masked_a, masked.b, masked.d, masked.masked_m, 48)
        dg.fp8_gemm_nt(skinny.a, skinny.b, skinny.d)
    for _ in range(3):
        dense.d.zero_(), masked.d.zero_(), skinny.d.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(dense.d, want_dense) and torch.equal(skinny.d, want_skinny)
        for g in range(4):
            rows = int(masked.masked_m[g])
            assert torch.equal(masked.d[g, :rows], want_masked[g, :rows])
    # the K-grouped GEMM with device-side K ranges: captured once, replayed with different group ends in the device tensor
    kg = gen.generate_k_grouped_contiguous_psum(3, 256, 384, [300, 129, 512])
    layouts = [kg.grouped_layout.clone(), kg.grouped_layout.clone()]
    layouts[1][2] -= 256                                        # the last group ends two scale blocks earlier
    wants = []
    for layout in layouts:
        d = kg.c.clone()
        dg.k_grouped_fp8_gemm_tn_contiguous(kg.a, kg.b, d, None, layout, c=d, use_psum_layout=True)
        wants.append(d)
    assert not torch.equal(wants[0][2], wants[1][2]) and torch.equal(wants[0][:2], wants[1][:2])
    live_layout, live_d = layouts[0].clone(), kg.c.clone()
    graph2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph2, stream=side):
        dg.k_grouped_fp8_gemm_tn_contiguous(kg.a, kg.b, live_d, [], live_layout, c=live_d, use_psum_layout=True)
    for which in (0, 1, 0):
        live_layout.copy_(layouts[which])
        live_d.copy_(kg.c)
        graph2.replay()
        torch.cuda.synchronize()
        assert torch.equal(live_d, wants[which])


@pytest.mark.parametrize('actual_ms,n,k', [
    ([617, 591, 487, 437, 515, 482, 599, 451], 4096, 2048),      # C4's layout: 16 tile pairs + 4 remainders
    ([128, 1, 0, 256, 129, 1000, 384, 3000, 77, 640], 2304, 1024),  # odd / even runs, empty group, one-block groups
    ([4000, 4200], 2048, 1024),                                  # two long groups: everything pairs up but one block
])
def test_m_grouped_contiguous_group_relative_tiles(actual_ms, n, k):
    """The tabled path of the contiguous layout (launch_contiguous_tabled: device-built tile tables, 256-row tiles that never straddle
    two groups + K-split 128-row remainders): every row against the oracle's device restatement, padding rows zero, nothing outside D,
    bit-repeatable, and the same bits as the 128-row kernel on the fixed grid."""
    gen.reset_seed(23)
    case = gen.generate_m_grouped_contiguous(len(actual_ms), 0, n, k, actual_ms=actual_ms)
    m = case.d.size(0)
    guarded = torch.full((m + 256, n), 777.0, device='cuda', dtype=torch.bfloat16)
    outs = []
    for _ in range(3):
        d = guarded[128:128 + m]
        d.fill_(float('nan'))
        dg.m_grouped_fp8_gemm_nt_contiguous(case.a, case.b, d, case.grouped_layout)
        outs.append(d.clone())
    assert dg.last_config() == 'duo_tab_256x256', dg.last_config()
    assert bool((guarded[:128] == 777.0).all()) and bool((guarded[128 + m:] == 777.0).all()), 'wrote outside D'
    assert all(torch.equal(o, outs[0]) for o in outs[1:])
    start = 0
    for g, (actual, aligned) in enumerate(zip(case.actual_ms, case.aligned_ms)):
        if actual:
            rows = slice(start, start + actual)
            want = oracle.fp8_gemm_nt_blockwise_torch(case.a[0][rows], case.a[1][rows], case.b[0][g], case.b[1][g])
            assert_close_to_oracle(outs[0][rows], want, f'group {g}')
        assert bool((outs[0][start + actual:start + aligned] == 0).all()), f'group {g}: padding rows must be zeros'
        start += aligned
    # round 4: the 256-row walk and the remainder walk share ONE launch; the two-launch form of round 3 must give the same bits
    os.environ['DG_TAB_UNFUSED'] = '1'
    dg_lib.dg_reload_env()                          # (the library reads its tuning variables once)
    try:
        two = guarded[128:128 + m]
        two.fill_(float('nan'))
        dg.m_grouped_fp8_gemm_nt_contiguous(case.a, case.b, two, case.grouped_layout)
        assert dg.last_config() == 'duo_tab_256x256'
        assert torch.equal(two, outs[0]), 'one launch and two launches differ'
    finally:
        del os.environ['DG_TAB_UNFUSED']
        dg_lib.dg_reload_env()
    dg.set_forced_config('duo_128x256')
    fixed = torch.empty_like(case.d)
    dg.m_grouped_fp8_gemm_nt_contiguous(case.a, case.b, fixed, case.grouped_layout)
    assert calc_diff(outs[0], fixed) < 2e-6           # (K pieces are summed in piece order: not the same bits as an unsplit K loop)


@pytest.mark.parametrize('actual_ms,n,k', [
    ([617, 591, 487, 437, 515, 482, 599, 451], 4096, 4096),      # C4's layout
    ([128, 1, 0, 256, 129, 1000, 384, 3000, 77, 640], 2304, 4096),  # odd / even runs, empty group, one-block groups
    ([4000, 4050], 2048, 4608),                                  # two long groups (64 blocks: the largest in-kernel tile list)
])
def test_packed_ue8m0_m_grouped_contiguous_group_relative_tiles(actual_ms, n, k):
    """Round 5: the group-relative tiling for packed UE8M0 scales (launch_e8_contiguous_tabled: e8_quad_256x256 over the 256-row tiles of the
    in-kernel tile list, e8_quad_128x256 over the remainders and padding blocks, the remainders cut along K into whole-quad pieces whose FP32
    partial tiles a second kernel sums in piece order): every group against the oracle, padding rows zero, nothing outside D, bit-repeatable;
    with whole remainder tiles (DG_E8_TAB_UNSPLIT) the same bits as the 128-row quad kernel on the fixed grid (the scaled MFMA accumulates in
    K-block order whatever the tile), with K pieces equal up to the FP32 association of the piece sums."""
    gen.reset_seed(29)
    case = gen.generate_m_grouped_contiguous(len(actual_ms), 0, n, k, True, False, actual_ms=actual_ms, use_ue8m0=True)
    a = gen.packed_ue8m0_operand(*case.a)
    b = gen.packed_ue8m0_operand(*case.b, mn_rows=n)
    m = case.d.size(0)
    guarded = torch.full((m + 256, n), 777.0, device='cuda', dtype=torch.bfloat16)
    d = guarded[128:128 + m]
    d.fill_(float('nan'))
    dg.m_grouped_fp8_gemm_nt_contiguous(a, b, d, case.grouped_layout)
    assert dg.last_config() == 'e8_quad_tab_256x256', dg.last_config()
    assert bool((guarded[:128] == 777.0).all()) and bool((guarded[128 + m:] == 777.0).all()), 'wrote outside D'
    first = d.clone()
    d.fill_(float('nan'))
    dg.m_grouped_fp8_gemm_nt_contiguous(a, b, d, case.grouped_layout)
    assert torch.equal(torch.nan_to_num(d), torch.nan_to_num(first)), 'not bit-repeatable'
    dg.set_forced_config('e8_quad_128x256')
    fixed = torch.full_like(case.d, float('nan'))
    dg.m_grouped_fp8_gemm_nt_contiguous(a, b, fixed, case.grouped_layout)
    dg.set_forced_config('auto')
    assert dg.last_config() == 'e8_quad_128x256'
    os.environ['DG_E8_TAB_UNSPLIT'] = '1'
    dg_lib.dg_reload_env()
    try:
        whole = torch.full_like(case.d, float('nan'))
        dg.m_grouped_fp8_gemm_nt_contiguous(a, b, whole, case.grouped_layout)
        assert dg.last_config() == 'e8_quad_tab_256x256'
    finally:
        del os.environ['DG_E8_TAB_UNSPLIT']
        dg_lib.dg_reload_env()
    start = 0
    for g, (actual, aligned) in enumerate(zip(case.actual_ms, case.aligned_ms)):
        rows = slice(start, start + actual)
        assert torch.equal(whole[rows], fixed[rows]), f'group {g}: whole group-relative tiles vs fixed grid'
        if actual:
            assert calc_diff(d[rows], fixed[rows]) < 2e-6, f'group {g}: K pieces vs fixed grid'
        if actual:
            want = oracle.fp8_gemm_nt_blockwise_torch(case.a[0][rows], case.a[1][rows], case.b[0][g], case.b[1][g])
            assert_close_to_oracle(d[rows], want, f'group {g}')
        assert bool((d[start + actual:start + aligned] == 0).all()), f'group {g}: padding rows must be zeros'
        start += aligned


def test_packed_ue8m0_m_grouped_nn_in_place_alignment_256():
    """The in-place grouped nn form at an M alignment of 256 (every 256-row tile belongs to ONE group: a single pass per tile) -- same bits as
    the K-major call, padding rows zero."""
    gen.reset_seed(31)
    dg.set_mk_alignment_for_contiguous_layout(256)
    try:
        actual_ms, n, k = [300, 256, 0, 513, 100, 700], 2048, 1024
        case = gen.generate_m_grouped_contiguous(len(actual_ms), 0, n, k, True, False, actual_ms=actual_ms, use_ue8m0=True)
        a = gen.packed_ue8m0_operand(*case.a)
        b = gen.packed_ue8m0_operand(*case.b, mn_rows=n)
        ref = torch.full_like(case.d, float('nan'))
        dg.m_grouped_fp8_gemm_nt_contiguous(a, b, ref, case.grouped_layout)
        dg.set_forced_config('e8_duo_bmn_256x256')
        d = torch.full_like(case.d, float('nan'))
        dg.m_grouped_fp8_gemm_nn_contiguous(a, (b[0].mT.contiguous(), b[1].mT), d, case.grouped_layout)
        dg.set_forced_config('auto')
        assert dg.last_config() == 'e8_duo_bmn_256x256', dg.last_config()
        start = 0
        for g, (actual, aligned) in enumerate(zip(case.actual_ms, case.aligned_ms)):
            assert torch.equal(d[start:start + actual], ref[start:start + actual]), f'group {g}'
            if actual:
                want = oracle.fp8_gemm_nt_blockwise_torch(case.a[0][start:start + actual], case.a[1][start:start + actual], case.b[0][g], case.b[1][g])
                assert_close_to_oracle(d[start:start + actual], want, f'group {g}')
            assert bool((d[start + actual:start + aligned] == 0).all()), f'group {g}: padding rows must be zeros'
            start += aligned
    finally:
        dg.set_forced_config('auto')
        dg.set_mk_alignment_for_contiguous_layout(128)


def test_packed_ue8m0_m_grouped_contiguous_in_a_hip_graph():
    """The group-relative tiling with packed scales inside a hipGraph: captured on a stream that already owns its K-split workspace it keeps
    the K pieces (three launches per call), captured on a fresh stream it falls back to whole remainder tiles (no allocation during capture);
    both replay to the eager result's bits / tolerance."""
    gen.reset_seed(37)
    actual_ms, n, k = [617, 591, 487, 437, 515, 482, 599, 451], 4096, 4096
    case = gen.generate_m_grouped_contiguous(len(actual_ms), 0, n, k, True, False, actual_ms=actual_ms, use_ue8m0=True)
    a = gen.packed_ue8m0_operand(*case.a)
    b = gen.packed_ue8m0_operand(*case.b, mn_rows=n)
    torch.cuda.synchronize()                    # (the operands were produced on the default stream; the side stream does not wait for it)
    warm = torch.cuda.Stream()
    with torch.cuda.stream(warm):
        dg.m_grouped_fp8_gemm_nt_contiguous(a, b, case.d, case.grouped_layout)
        assert dg.last_config() == 'e8_quad_tab_256x256'
    warm.synchronize()
    want = case.d.clone()
    assert calc_diff(torch.nan_to_num(want), torch.nan_to_num(case.ref_d)) < gen.FP8_MAX_DIFF
    for stream, exact in ((warm, True), (torch.cuda.Stream(), False)):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            dg.m_grouped_fp8_gemm_nt_contiguous(a, b, case.d, case.grouped_layout)
        for _ in range(2):
            case.d.fill_(float('nan'))
            graph.replay()
            torch.cuda.synchronize()
            start = 0
            for actual, aligned in zip(case.actual_ms, case.aligned_ms):
                rows = slice(start, start + actual)
                if exact:
                    assert torch.equal(case.d[rows], want[rows])
                else:
                    assert calc_diff(case.d[rows], want[rows]) < 2e-6
                assert bool((case.d[start + actual:start + aligned] == 0).all())
                start += aligned


def test_stream_kernels_need_aligned_scale_rows():
    """The stream kernels fetch the row scales of four K blocks with 16-byte requests (round 3): an MN-major SFA whose K-block rows do
    not start on 16-byte boundaries (a direct C-ABI caller: the host layer always produces the padded layout) takes another kernel --
    same result, bit for bit -- and forcing a stream kernel onto it is refused."""
    from deepgemm_amd._lib import lib, current_stream_ptr
    gen.reset_seed(5)
    m, n, k = 50, 4096, 1024
    case = gen.generate_normal(m, n, k)
    a, sfa = case.a
    b, sfb = case.b
    aligned = dg.get_mn_major_tma_aligned_tensor(sfa)                       # strides (1, 52)
    dg.fp8_gemm_nt((a, aligned), case.b, case.d)
    assert dg.last_config().startswith('stream_')
    want = case.d.clone()
    packed = torch.empty((k // 128) * m + 3, dtype=torch.float, device='cuda')[1:1 + (k // 128) * m]      # base 4 bytes off, K-block rows 200 bytes apart
    odd = torch.as_strided(packed, (m, k // 128), (1, m))
    odd.copy_(sfa)

    def raw(d):
        return lib.dg_fp8_gemm_nt(a.data_ptr(), odd.data_ptr(), b.data_ptr(), sfb.data_ptr(), d.data_ptr(), m, n, k, a.stride(0), a.stride(1),
                                  b.stride(0), b.stride(1), odd.stride(0), odd.stride(1), sfb.stride(0), sfb.stride(1), 128, d.stride(0), 0, 0,
                                  current_stream_ptr())
    got = torch.full_like(want, float('nan'))
    assert raw(got) == 0
    assert not dg.last_config().startswith('stream_')
    assert torch.equal(got, want)
    dg.set_forced_config('stream_64x128')
    assert raw(got) != 0
    dg.set_forced_config('auto')


@pytest.mark.parametrize('m,n,k', [(1, 7168, 4096), (9, 4112, 2048), (16, 8192, 2048), (2, 4992, 2560)])
def test_skinny_two_subtile_form(m, n, k):
    """n / 16 column tiles between one and two rounds of the chip: workgroups of 17 .. 32 columns as two N-subtiles (`skinny_16w`); the
    overlap of neighbouring workgroups writes identical bits; accumulating calls stay on the 16-column form; same bits as `skinny_16`."""
    gen.reset_seed(m + n)
    case = gen.generate_normal(m, n, k)
    want = oracle_dense(case)
    wide = torch.full((m, n + 40), float('nan'), device='cuda', dtype=torch.bfloat16)
    d = wide[:, :n]
    dg.fp8_gemm_nt(case.a, case.b, d)
    assert dg.last_config() in ('skinny_16w', 'skinny_16wc', 'skinny_16ca'), dg.last_config()      # (round 5: two subtiles only with short K loops)
    assert_close_to_oracle(d, want, 'skinny_16w')
    assert bool(torch.isnan(wide[:, n:]).all())
    dg.set_forced_config('skinny_16')
    dg.fp8_gemm_nt(case.a, case.b, case.d)
    assert torch.equal(case.d, d.contiguous())
    dg.set_forced_config('auto')
    for _ in range(3):
        again = torch.empty_like(case.d)
        dg.fp8_gemm_nt(case.a, case.b, again)
        assert torch.equal(again, case.d)
    acc_case = gen.generate_normal(m, n, k, accumulate=True, out_dtype=torch.float)
    dg.fp8_gemm_nt(acc_case.a, acc_case.b, acc_case.d, c=acc_case.c)
    assert dg.last_config() in ('skinny_16', 'skinny_16c', 'skinny_16ca')
    dg.set_forced_config('skinny_16w')
    with pytest.raises(RuntimeError):
        dg.fp8_gemm_nt(acc_case.a, acc_case.b, acc_case.d, c=acc_case.c)
    dg.set_forced_config('auto')


def test_single_ulp_flip_on_a_tiny_output_motivates_the_frobenius_gate():
    """tests/gpu_helpers.py holds outputs below 256 elements to rel-Frobenius 4e-3 instead of 1e-3.  Why, demonstrated: the matrix core
    sums the 128 products of a K block in its own order, the oracle rounds the exact block sum; when the two FP32 results straddle a BF16
    rounding boundary ONE output element lands on the neighbouring BF16 value (2^-8 relative).  On a 16-element output that single,
    legitimate flip is 2^-8 |w_i| / (4 rms) of the norm -- above 1e-3 whenever the element is larger than the rms -- while the
    element-wise bound (one BF16 ulp + noise) holds.  The test finds such a case among 150 seeds and checks exactly that."""
    flips_seen, worst = 0, None
    for seed in range(150):
        gen.reset_seed(1000 + seed)
        case = gen.generate_normal(1, 16, 2048)
        dg.fp8_gemm_nt(case.a, case.b, case.d)
        want = oracle_dense(case).float()
        got = case.d.float().cpu()
        flips = int((got != want).sum())
        if flips == 0:
            continue
        flips_seen += flips
        assert bool(((got - want).abs() <= want.abs() * 2.0 ** -7 + 1e-30).all()), f'seed {seed}: a flip is at most one BF16 ulp'
        assert_close_to_oracle(case.d, want.to(torch.bfloat16), f'seed {seed}: {flips} flip(s)')      # the 4e-3 gate + element-wise bound hold
        rel = float((got - want).norm() / want.norm())
        if worst is None or rel > worst[1]:
            worst = (seed, rel, flips)
        if flips == 1 and rel > 1e-3:
            break
    assert flips_seen > 0, 'no BF16 flip in 150 seeds of a 1 x 16 output'
    assert worst[1] > 1e-3, f'flips found ({flips_seen}) but none beyond the old 1e-3 gate: worst {worst}'
    print(f'seed {1000 + worst[0]}: {worst[2]} single-ulp flip(s) on 16 elements -> rel-Frobenius {worst[1]:.2e} (> 1e-3, < 4e-3)')
