"""The reference's own test sweeps at their stated sizes (tests/test_fp8_fp4.py:32-189 over tests/generators.py:115-187) and the
BASELINE.json configurations at their exact sizes, through the public operators.

Every case is checked (a) at the reference's gate -- calc_diff < 1e-3 against the FP32 matmul of the unquantised inputs,
tests/generators.py:65-70 -- on the whole output and (b) against the oracle's arithmetic (torch restatement of the C oracle,
float64 block products) on a sample of rows, with the tolerances of tests/gpu_helpers.py.  Zero-padding and untouched-row
guarantees are checked with NaN poison, as in the small-shape tests."""
import random

import pytest
import torch

import deepgemm_amd as dg
from deepgemm_amd.testing import calc_diff, generators as gen

import oracle
from gpu_helpers import assert_close_fp32, assert_close_to_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _reset_knobs():
    dg.set_forced_config('auto')
    dg.set_mk_alignment_for_contiguous_layout(128)
    yield
    dg.set_forced_config('auto')
    dg.set_mk_alignment_for_contiguous_layout(128)


def _oracle_rows(a_pair, b_pair, rows, gran_n, out_dtype):
    """Oracle result for the given rows of A (operands in any majorness; logical indexing)."""
    a, sfa = a_pair[0][rows].cpu(), a_pair[1][rows].cpu()
    return oracle.fp8_gemm_nt_blockwise_torch(a, sfa, b_pair[0].cpu(), b_pair[1].cpu(), gran_n=gran_n, out_dtype=out_dtype)


def _check_sampled(d, case, gran_n, label, addend=None, rows_n=4):
    m = d.shape[0]
    rows = torch.tensor(sorted(random.sample(range(m), min(m, rows_n))), device='cuda')
    want = _oracle_rows(case.a, case.b, rows, gran_n, torch.float if d.dtype == torch.float else torch.bfloat16)
    if addend is not None:
        add = addend[rows].cpu()
        if d.dtype == torch.float:
            want = want + add
        else:                                   # reduce-add in BF16: the GEMM result is rounded before the addition
            want = (want.float() + add.float()).to(torch.bfloat16)
    if d.dtype == torch.float:
        assert_close_fp32(d[rows], want, label)
    else:
        assert_close_to_oracle(d[rows], want, label, addend=None if addend is None else addend[rows].cpu())


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE config 3: fp8_gemm_{nt,nn,tn,tt} at M=2048, N=7168, K=2048 (reference sweep entry: tests/test_fp8_fp4.py:32-55)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('layout', ['nt', 'nn', 'tn', 'tt'])
def test_c3_layouts_at_stated_size(layout):
    m, n, k = 2048, 7168, 2048
    gen.reset_seed(3)
    a_k_major, b_k_major = layout[0] == 'n', layout[1] == 't'
    case = gen.generate_normal(m, n, k, a_k_major, b_k_major)
    # strided-view entry: majorness inferred from the strides (csrc/utils/layout.hpp:13-24)
    case.d.fill_(float('nan'))
    dg.fp8_gemm_nt(case.a, case.b, case.d)
    view_cfg = dg.last_config()
    assert not view_cfg.startswith('generic'), (layout, view_cfg)
    assert calc_diff(case.d, case.ref_d) < gen.FP8_MAX_DIFF, (layout, view_cfg)
    _check_sampled(case.d, case, 128, f'{layout} view', rows_n=8)
    # alias entry on the transposed contiguous tensors (tests/test_fp8_fp4.py:47-52)
    a = case.a if a_k_major else (case.a[0].T, case.a[1].T)
    b = case.b if b_k_major else (case.b[0].T, case.b[1].T)
    assert a[0].is_contiguous() and b[0].is_contiguous()
    d_alias = torch.full_like(case.d, float('nan'))
    getattr(dg, f'fp8_gemm_{layout}')(a, b, d_alias)
    assert torch.equal(d_alias, case.d), f'{layout}: alias entry differs from the strided-view entry'
    # the same operands made K-major by hand go through the plain NT kernels: identical promotion order => identical bits
    a_km = (case.a[0].contiguous(), case.a[1])
    b_km = (case.b[0].contiguous(), case.b[1])
    d_km = torch.full_like(case.d, float('nan'))
    dg.fp8_gemm_nt(a_km, b_km, d_km)
    assert torch.equal(d_km, case.d), f'{layout}: {view_cfg} differs from {dg.last_config()} on K-major copies'


# ---------------------------------------------------------------------------------------------------------------------
# The reference's dense sweep: forward (with and without accumulation), dgrad, wgrad (tests/generators.py:115-154)
# ---------------------------------------------------------------------------------------------------------------------
def test_reference_dense_sweep_full():
    gen.reset_seed(0)
    count = 0
    for m, n, k, a_k_major, b_k_major, accumulate, out_dtype, per_token_b in gen.enumerate_normal():
        case = gen.generate_normal(m, n, k, a_k_major, b_k_major, accumulate, out_dtype, per_token_b)
        recipe = (1, 1, 128) if per_token_b else None
        c0 = case.c.clone() if accumulate else None
        label = f'm={m} n={n} k={k} a_k={a_k_major} b_k={b_k_major} acc={accumulate} {out_dtype} per_token_b={per_token_b}'
        dg.fp8_gemm_nt(case.a, case.b, case.d, c=case.c, recipe=recipe)
        label += f' [{dg.last_config()}]'
        assert calc_diff(case.d, case.ref_d) < gen.FP8_MAX_DIFF, label
        _check_sampled(case.d, case, 1 if per_token_b else 128, label, addend=c0)
        count += 1
        del case
    assert count == 2 * len(gen.DENSE_M_FWD) * len(gen.DENSE_NK) + 3 * len(gen.DENSE_NK)


# ---------------------------------------------------------------------------------------------------------------------
# The reference's M-grouped contiguous sweep (tests/generators.py:157-171): (4, 8192) and (8, 4096) x 4 (n, k) x B major x psum
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('use_psum', [False, True])
@pytest.mark.parametrize('groups,expected_m', gen.CONTIGUOUS_GROUPS)
def test_reference_contiguous_sweep_full(groups, expected_m, use_psum):
    gen.reset_seed(groups)
    for n, k in gen.GROUPED_NK:
        for b_k_major in (True, False):
            case = gen.generate_m_grouped_contiguous(groups, expected_m, n, k, b_k_major, use_psum)
            case.d.fill_(float('nan'))
            if b_k_major:
                dg.m_grouped_fp8_gemm_nt_contiguous(case.a, case.b, case.d, case.grouped_layout, use_psum_layout=use_psum)
            else:
                b_alias = (case.b[0].mT, case.b[1].mT)
                assert b_alias[0].is_contiguous()
                dg.m_grouped_fp8_gemm_nn_contiguous(case.a, b_alias, case.d, case.grouped_layout, use_psum_layout=use_psum)
            label = f'g={groups} m~{expected_m} n={n} k={k} b_k_major={b_k_major} psum={use_psum} [{dg.last_config()}]'
            start = 0
            for g, (actual, aligned) in enumerate(zip(case.actual_ms, case.aligned_ms)):
                assert calc_diff(case.d[start:start + actual], case.ref_d[start:start + actual]) < gen.FP8_MAX_DIFF, (label, g)
                rows = torch.tensor(sorted(random.sample(range(start, start + actual), 2)), device='cuda')
                want = oracle.fp8_gemm_nt_blockwise_torch(case.a[0][rows].cpu(), case.a[1][rows].cpu(),
                                                          case.b[0][g].cpu(), case.b[1][g].cpu())
                assert_close_to_oracle(case.d[rows], want, f'{label} group {g}')
                assert bool((case.d[start + actual:start + aligned] == 0).all()), f'{label}: padding rows of group {g} must be zeros'
                start += aligned
            del case


# ---------------------------------------------------------------------------------------------------------------------
# The reference's masked sweep (tests/generators.py:174-187): 4 (groups, expected_m) x 4 (n, k), max_m = 4096
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('groups,expected_m', gen.MASKED_GROUPS)
def test_reference_masked_sweep_full(groups, expected_m):
    gen.reset_seed(groups + expected_m)
    for n, k in gen.GROUPED_NK:
        case = gen.generate_m_grouped_masked(groups, gen.MASKED_MAX_M, expected_m, n, k)
        case.d.fill_(float('nan'))
        dg.m_grouped_fp8_gemm_nt_masked(case.a, case.b, case.d, case.masked_m, expected_m)
        label = f'g={groups} m~{expected_m} n={n} k={k} [{dg.last_config()}]'
        for g, rows in enumerate(case.masked_m.tolist()):
            rows = int(rows)
            if rows:
                assert calc_diff(case.d[g, :rows], case.ref_d[g, :rows]) < gen.FP8_MAX_DIFF, (label, g)
            assert bool(torch.isnan(case.d[g, rows:]).all()), f'{label}: rows >= masked_m of group {g} must not be written'
        for g in random.sample(range(groups), 3):
            rows = int(case.masked_m[g])
            if rows:
                pick = torch.tensor(sorted(random.sample(range(rows), min(rows, 2))), device='cuda')
                want = oracle.fp8_gemm_nt_blockwise_torch(case.a[0][g][pick].cpu(), case.a[1][g][pick].cpu(),
                                                          case.b[0][g].cpu(), case.b[1][g].cpu())
                assert_close_to_oracle(case.d[g][pick], want, f'{label} group {g}')
        del case


# ---------------------------------------------------------------------------------------------------------------------
# set_mk_alignment_for_contiguous_layout != 128 reaching a kernel (the reference's SM100 tests run alignments down to 32)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('use_psum', [False, True])
def test_contiguous_layout_with_alignment_64(use_psum):
    dg.set_mk_alignment_for_contiguous_layout(64)
    assert dg.get_mk_alignment_for_contiguous_layout() == 64
    gen.reset_seed(64)
    for actual_ms, n, k in (([60, 0, 70, 200], 512, 384), ([64, 1, 129], 4096, 1024), ([33] * 8, 520, 256)):
        case = gen.generate_m_grouped_contiguous(len(actual_ms), 0, n, k, True, use_psum, actual_ms=actual_ms)
        assert case.m == sum(-(-x // 64) * 64 for x in actual_ms)
        want = torch.full(case.d.shape, float('nan'), dtype=torch.bfloat16)
        oracle.m_grouped_fp8_gemm_nt_contiguous(case.a[0].cpu(), case.a[1].cpu(), case.b[0].cpu(), case.b[1].cpu(), want,
                                                case.grouped_layout.cpu(), use_psum, m_alignment=64)
        # canary rows behind D: a tile taller than the alignment must not write past m (m is not a multiple of 128 here)
        storage = torch.full((case.m + 256, n), float('nan'), device='cuda', dtype=torch.bfloat16)
        d = storage[:case.m]
        dg.m_grouped_fp8_gemm_nt_contiguous(case.a, case.b, d, case.grouped_layout, use_psum_layout=use_psum)
        cfg = dg.last_config()
        assert bool(torch.isnan(storage[case.m:]).all()), f'{cfg}: wrote past the end of D'
        start = 0
        for actual, aligned in zip(case.actual_ms, case.aligned_ms):
            assert_close_to_oracle(d[start:start + actual], want[start:start + actual], f'{cfg} rows {start}+{actual}')
            assert bool((d[start + actual:start + aligned] == 0).all()), f'{cfg}: padding rows must be zeros'
            start += aligned


def test_trailing_padding_does_not_write_past_d():
    """a contiguous layout whose row count is not a multiple of the tile height and whose tail is padding (-1 rows /
    the psum gap): zero rows stop at m.  The tile walk is forced to 128- and 256-row tiles on a 128-aligned layout cut short."""
    gen.reset_seed(5)
    n, k = 512, 256
    case = gen.generate_m_grouped_contiguous(3, 0, n, k, actual_ms=[100, 128, 30])        # aligned: 128 + 128 + 128 = 384 rows
    m_cut = 300                                 # rows 286 .. 299 of the last group's padding stay, its tile reaches row 383
    a = (case.a[0][:m_cut], case.a[1][:m_cut].contiguous())
    layout = case.grouped_layout[:m_cut].contiguous()
    for cfg in ('auto', 'duo_128x256', 'duo_256x256', 'pipe_128x128', 'generic_128x128'):
        dg.set_forced_config(cfg)
        storage = torch.full((m_cut + 300, n), float('nan'), device='cuda', dtype=torch.bfloat16)
        dg.m_grouped_fp8_gemm_nt_contiguous(a, case.b, storage[:m_cut], layout)
        assert bool(torch.isnan(storage[m_cut:]).all()), f'{cfg}: wrote past the end of D'
        assert bool((storage[286:m_cut] == 0).all()) and bool((storage[100:128] == 0).all()), cfg
        for lo, hi in ((0, 100), (128, 256), (256, 286)):
            assert calc_diff(storage[lo:hi], case.ref_d[lo:hi]) < gen.FP8_MAX_DIFF, (cfg, lo)


def test_packed_ue8m0_at_c2_size():
    """The hardware-scaled kernels at BASELINE config 2's size: reference gate, oracle row sample, quad == duo bit for bit."""
    m, n, k = 4096, 4096, 7168
    gen.reset_seed(2)
    case = gen.generate_normal(m, n, k, use_ue8m0=True)
    a, b = gen.packed_ue8m0_operand(*case.a), gen.packed_ue8m0_operand(*case.b, mn_rows=n)
    outs = {}
    for cfg in ('auto', 'e8_quad_256x256', 'e8_quad_h_256x256', 'e8_quad_h2_256x256', 'e8_quad_128x256', 'e8_duo_256x256'):
        dg.set_forced_config(cfg)
        d = torch.full((m, n), float('nan'), device='cuda', dtype=torch.bfloat16)
        dg.fp8_gemm_nt(a, b, d)
        outs[cfg] = d
        if cfg == 'auto':
            assert dg.last_config() in ('e8_quad_256x256', 'e8_quad_h_256x256', 'e8_quad_h2_256x256')
    assert calc_diff(outs['auto'], case.ref_d) < gen.FP8_MAX_DIFF
    _check_sampled(outs['auto'], case, 128, 'e8 c2', rows_n=16)
    for cfg, d in outs.items():
        assert torch.equal(d, outs['auto']), cfg


@pytest.mark.parametrize('use_psum', [False, True])
@pytest.mark.parametrize('gran_k,k_alignment', [(32, 32), (32, 128), (32, 160), (32, 224), (128, 128), (128, 160), (128, 224)])
def test_reference_k_grouped_sm100_sweep_full(gran_k, k_alignment, use_psum):
    """The reference's SM100 K-grouped sweep at its stated sizes (tests/generators.py:190-213 x tests/test_fp8_fp4.py:193-225): every
    (gran_k, K alignment) pair, with and without the psum layout, the three test variants (as generated / an empty group / a shortened first
    group in the psum form), ``ks_cpu`` given, missing and empty in the psum form -- UE8M0 scales, MN-major operands read in place by the
    hardware-scaled K-grouped kernels; the reference's own gate (calc_diff < 1e-3).  Two of the six (groups, m, n, k) entries per pair, to keep
    the run at seconds: the largest K per group and the largest group count."""
    import random
    dg.set_mk_alignment_for_contiguous_layout(k_alignment)
    try:
        for num_groups, m, n, expected_k in ((4, 4096, 7168, 8192), (16, 7168, 2048, 2048)):
            random.seed(num_groups + k_alignment + gran_k)
            real_ks = [max(1, int(expected_k * random.uniform(0.7, 1.3))) for _ in range(num_groups)]
            for variant in range(2):
                ks = list(real_ks)
                if variant == 1:
                    ks[random.randint(0, num_groups - 1)] = 0                      # an empty group
                elif use_psum:
                    ks[0] -= random.randint(1, min(k_alignment - 1, ks[0] - 1))    # a group that ends off the alignment
                if not use_psum:
                    ks = [gen.align(k, k_alignment) for k in ks]
                gen.reset_seed(sum(ks))
                case = gen.generate_k_grouped_contiguous_ue8m0(num_groups, m, n, ks, gran_k, k_alignment, use_psum_layout=use_psum)
                case.a_groups = case.b_groups = None
                for ks_cpu in ((case.ks, None, []) if use_psum else (case.ks,)):
                    d = case.c.clone()
                    dg.k_grouped_fp8_gemm_tn_contiguous(case.a, case.b, d, ks_cpu, case.grouped_layout, c=d, recipe=(1, 1, gran_k), use_psum_layout=use_psum) \
                        if gran_k == 32 else _sm100_mode(lambda: dg.k_grouped_fp8_gemm_tn_contiguous(case.a, case.b, d, ks_cpu, case.grouped_layout, c=d,
                                                                                                      recipe=(1, 1, gran_k), use_psum_layout=use_psum))
                    assert dg.last_config().startswith('e8_quad_kg_mn_'), dg.last_config()
                    diff = calc_diff(d, case.ref_d)
                    assert diff < 1e-3, (num_groups, m, n, ks, gran_k, k_alignment, use_psum, ks_cpu is None, diff)
                del case
                torch.cuda.empty_cache()
    finally:
        dg.set_mk_alignment_for_contiguous_layout(128)


def _sm100_mode(fn):
    mode = dg.get_sf_cast_mode()
    dg.set_sf_cast_mode('sm100')
    try:
        return fn()
    finally:
        dg.set_sf_cast_mode(mode)
