"""The FP32 -> UE8M0 cast branch of ``transform_sf_into_required_layout`` (reference csrc/apis/layout.hpp:48-54, its SM100 default):
``set_sf_cast_mode('sm100')`` casts FP32 scales to packed UE8M0 words (row broadcast + pack in one HIP kernel) and routes the GEMM
to the hardware-scaled MFMA kernels; ``disable_ue8m0_cast=True`` and the default ``'sm90'`` mode keep FP32 scales FP32."""
import pytest
import torch

import deepgemm_amd as dg
import oracle
from deepgemm_amd.layout import get_mn_major_tma_aligned_packed_ue8m0_tensor, transform_sf_into_required_layout
from deepgemm_amd.testing import calc_diff, generators as gen
from gpu_helpers import assert_close_fp32, assert_close_to_oracle, cpu_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _modes():
    dg.set_forced_config('auto')
    dg.set_sf_cast_mode('sm90')
    dg.set_mk_alignment_for_contiguous_layout(128)
    yield
    dg.set_forced_config('auto')
    dg.set_sf_cast_mode('sm90')


def _reference_cast(sf: torch.Tensor, mn: int, gran_mn: int) -> torch.Tensor:
    """The reference's statement of the branch, in torch: index_select broadcast (layout.hpp:52-53), then the torch twin of the
    pack (smxx_layout.hpp:156-179): exponent byte = bits >> 23 (mantissa dropped), four to a little-endian word, zero padding."""
    rows = torch.arange(mn, device=sf.device) // gran_mn
    b = sf.index_select(-2, rows)
    exps = (b.view(torch.int) >> 23).to(torch.uint8)
    pad = (-exps.size(-1)) % 4
    exps = torch.nn.functional.pad(exps, (0, pad))
    return exps.contiguous().view(torch.int)                    # [..., mn, ceil(sf_k / 4)]


@pytest.mark.parametrize('mn,k,gran_mn,groups', [(300, 896, 1, None), (520, 1024, 128, None), (4096, 7168, 128, None), (4096, 7168, 1, None),
                                                  (129, 512, 128, 3), (64, 384, 1, 5), (7, 128, 128, None)])
def test_cast_branch_bit_exact_vs_reference_statement(mn, k, gran_mn, groups):
    """transform_sf_into_required_layout in 'sm100' mode == index_select + pack, bit for bit, for power-of-two AND arbitrary positive
    scales (truncation of the mantissa, `>> 23`), in the MN-major aligned layout."""
    torch.manual_seed(mn + k)
    shape = ((groups,) if groups else ()) + (-(-mn // gran_mn), -(-k // 128))
    for pow2 in (True, False):
        sf = torch.rand(shape, device='cuda') * 3 + 1e-3
        if pow2:
            sf = dg.utils.math.ceil_to_ue8m0(sf)
        dg.set_sf_cast_mode('sm100')
        got = transform_sf_into_required_layout(sf, mn, k, (gran_mn, 128), groups)
        want = _reference_cast(sf, mn, gran_mn)
        assert got.dtype == torch.int and tuple(got.shape) == tuple(want.shape)
        aligned = (mn + 3) // 4 * 4
        assert got.stride(-2) == 1 and got.stride(-1) == aligned
        assert torch.equal(got, want), (pow2, mn, k, gran_mn)
        # 3-tuple recipe + is_sfa selects the same granularity
        got3 = transform_sf_into_required_layout(sf, mn, k, (gran_mn, 77, 128), groups, True)
        assert torch.equal(got3, want)
        # disable_ue8m0_cast (and the default mode) keep the FP32 path
        kept = transform_sf_into_required_layout(sf, mn, k, (gran_mn, 128), groups, None, True)
        assert kept.dtype == torch.float
        dg.set_sf_cast_mode('sm90')
        assert transform_sf_into_required_layout(sf, mn, k, (gran_mn, 128), groups).dtype == torch.float


def test_pack_with_psum_layout_zeroes_gap_rows():
    """get_mn_major_tma_aligned_packed_ue8m0_tensor(sf, psum_layout) (smxx_layout.hpp:181-246, kernel smxx_layout.cuh:76-94): rows
    outside every group's [align(end[g-1]), end[g]) -- uninitialised in the psum layout, here NaN-poisoned -- come out as zero words."""
    torch.manual_seed(3)
    for alignment, ends in ((128, [100, 128 + 77, 256 + 128, 384 + 128]), (64, [10, 64 + 64, 128 + 1])):
        dg.set_mk_alignment_for_contiguous_layout(alignment)
        mn = (ends[-1] + alignment - 1) // alignment * alignment + alignment          # one whole trailing gap block as well
        sf = dg.utils.math.ceil_to_ue8m0(torch.rand((mn, 7), device='cuda') + 0.01)
        valid = torch.zeros(mn, dtype=torch.bool, device='cuda')
        start = 0
        for e in ends:
            valid[start:e] = True
            start = (e + alignment - 1) // alignment * alignment
        sf[~valid] = float('nan')
        layout = torch.tensor(ends, dtype=torch.int, device='cuda')
        got = get_mn_major_tma_aligned_packed_ue8m0_tensor(sf, layout)
        want = _reference_cast(torch.where(valid[:, None], sf, torch.zeros_like(sf)), mn, 1)
        assert torch.equal(got, want)
        with pytest.raises(RuntimeError, match='num_sf_batches == 1'):
            get_mn_major_tma_aligned_packed_ue8m0_tensor(sf.unsqueeze(0).repeat(2, 1, 1), layout)


@pytest.mark.parametrize('m,n,k', [(256, 512, 1024), (300, 520, 1536), (64, 136, 512), (2048, 2048, 2048)])
def test_dense_sm100_mode_is_the_packed_int_call(m, n, k):
    """Power-of-two FP32 scales in 'sm100' mode == the same call with the scales handed over as packed ints, bit for bit (dense, BF16
    and FP32 accumulate), on the hardware-scaled kernels; `disable_ue8m0_cast` restores the FP32-scale kernels."""
    gen.reset_seed(m + n)
    case = gen.generate_normal(m, n, k, use_ue8m0=True)
    a_int, b_int = gen.packed_ue8m0_operand(*case.a), gen.packed_ue8m0_operand(*case.b, mn_rows=n)
    d_int = torch.empty_like(case.d)
    dg.fp8_gemm_nt(a_int, b_int, d_int)
    cfg_int = dg.last_config()
    dg.set_sf_cast_mode('sm100')
    d = torch.full_like(case.d, float('nan'))
    dg.fp8_gemm_nt(case.a, case.b, d)
    assert dg.last_config() == cfg_int and cfg_int.startswith('e8_')
    assert torch.equal(d, d_int)
    assert calc_diff(d, case.ref_d) < gen.FP8_MAX_DIFF
    # accumulate, FP32 output
    c32 = torch.randn((m, n), device='cuda')
    d32, d32_int = c32.clone(), c32.clone()
    dg.fp8_gemm_nt(case.a, case.b, d32, c=d32)
    dg.fp8_gemm_nt(a_int, b_int, d32_int, c=d32_int)
    assert torch.equal(d32, d32_int)
    # the keyword keeps the FP32-scale kernels in this mode
    d_fp32 = torch.empty_like(case.d)
    dg.fp8_gemm_nt(case.a, case.b, d_fp32, disable_ue8m0_cast=True)
    assert not dg.last_config().startswith('e8_')
    assert calc_diff(d_fp32, d) < 2e-6
    # nn / tn / tt reach the same kernels (operands re-majored by the host layer) with the same bits
    a_mn = (case.a[0].mT.contiguous().mT, case.a[1])
    b_mn = (case.b[0].mT.contiguous().mT, case.b[1])
    for aa, bb in ((case.a, b_mn), (a_mn, b_mn), (a_mn, case.b)):
        d2 = torch.empty_like(case.d)
        dg.fp8_gemm_nt(aa, bb, d2)
        assert torch.equal(d2, d)


def test_dense_sm100_mode_truncates_non_power_of_two_scales():
    """Arbitrary positive FP32 scales in 'sm100' mode: the cast keeps the exponent byte only (`>> 23`) -- the result is the oracle's
    with the scales truncated to 2^(e - 127), NOT the FP32-scale result."""
    m, n, k = 256, 384, 1024
    gen.reset_seed(5)
    case = gen.generate_normal(m, n, k)                                   # reference SM90 casts: scales are not powers of two
    dg.set_sf_cast_mode('sm100')
    d = torch.empty_like(case.d)
    dg.fp8_gemm_nt(case.a, case.b, d)
    assert dg.last_config().startswith('e8_')
    trunc = lambda s: (s.view(torch.int) & 0x7f800000).view(torch.float)   # noqa: E731
    want = torch.empty((m, n), dtype=torch.bfloat16)
    oracle.fp8_gemm_nt(case.a[0].cpu(), trunc(case.a[1]).cpu(), case.b[0].cpu(), trunc(case.b[1]).cpu(), want)
    assert_close_to_oracle(d, want, 'sm100 mode, truncated scales')
    # K tail in this mode: the same truncated scales through the FP32-scale path
    kt = 576
    case = gen.generate_normal(130, 264, kt)
    d = torch.empty_like(case.d)
    dg.fp8_gemm_nt(case.a, case.b, d)
    want = torch.empty(case.d.shape, dtype=torch.bfloat16)
    oracle.fp8_gemm_nt(case.a[0].cpu(), trunc(case.a[1]).cpu(), case.b[0].cpu(), trunc(case.b[1]).cpu(), want)
    assert_close_to_oracle(d, want, 'sm100 mode, truncated scales, K tail')


def test_sm100_mode_is_one_arithmetic_at_every_entry():
    """'sm100' mode with scales that are not powers of two, at the entries that cannot take the hardware-scaled kernels (a K tail on the
    grouped entries, the head-split epilogue): the FP32-scale kernels run on the truncated scales -- bit-identical to the same call in
    'sm90' mode with the truncation done by the caller, and different from the untruncated result (round 3 let these fall through)."""
    trunc = lambda s: (s.view(torch.int) & 0x7f800000).view(torch.float)   # noqa: E731
    n, k = 512, 576                                                        # 4.5 K blocks
    gen.reset_seed(11)
    cont = gen.generate_m_grouped_contiguous(3, 0, n, k, actual_ms=[130, 256, 90])
    gen.reset_seed(12)
    masked = gen.generate_m_grouped_masked(3, 64, 40, n, k, masked_ms=[33, 64, 0])
    gen.reset_seed(13)
    head = gen.generate_normal(96, 4 * (128 + 64), 1024)
    splits = (128, 64, 64)

    def run_all(ta, tb):
        outs = []
        d = torch.zeros_like(cont.d)
        dg.m_grouped_fp8_gemm_nt_contiguous((cont.a[0], ta(cont.a[1])), (cont.b[0], tb(cont.b[1])), d, cont.grouped_layout)
        outs.append(d)
        d = torch.zeros_like(masked.d)
        dg.m_grouped_fp8_gemm_nt_masked((masked.a[0], ta(masked.a[1])), (masked.b[0], tb(masked.b[1])), d, masked.masked_m, 40)
        outs.append(d)
        d = torch.zeros((96, 4 * (128 + 64 + 64)), device='cuda', dtype=torch.bfloat16)
        dg.fp8_gemm_nt_skip_head_mid((head.a[0], ta(head.a[1])), (head.b[0], tb(head.b[1])), d, splits)
        outs.append(d)
        return outs

    ident = lambda s: s                                                     # noqa: E731
    plain = run_all(ident, ident)                                           # 'sm90': FP32 scales as they are
    by_hand = run_all(trunc, trunc)                                         # 'sm90' on truncated scales
    dg.set_sf_cast_mode('sm100')
    mode = run_all(ident, ident)
    for name, got, want, other in zip(('contiguous', 'masked', 'skip_head_mid'), mode, by_hand, plain):
        assert torch.equal(got, want), f'{name}: sm100 mode != truncated scales'
        assert not torch.equal(got, other), f'{name}: truncation had no effect (scales already powers of two?)'
    # and the keyword of the reference switches it off per call
    d = torch.zeros_like(cont.d)
    dg.m_grouped_fp8_gemm_nt_contiguous(cont.a, cont.b, d, cont.grouped_layout, disable_ue8m0_cast=True)
    assert torch.equal(d, plain[0])


@pytest.mark.parametrize('use_psum', [False, True])
def test_contiguous_sm100_mode_is_the_packed_int_call(use_psum):
    gen.reset_seed(16)
    for actual_ms, n, k in (([100, 0, 130, 256], 256, 384), ([128, 384, 0, 0, 200, 640], 512, 1024)):
        case = gen.generate_m_grouped_contiguous(len(actual_ms), 0, n, k, True, use_psum, actual_ms=actual_ms, use_ue8m0=True)
        a_int, b_int = gen.packed_ue8m0_operand(*case.a), gen.packed_ue8m0_operand(*case.b, mn_rows=n)
        d_int = torch.full_like(case.d, float('nan'))
        dg.m_grouped_fp8_gemm_nt_contiguous(a_int, b_int, d_int, case.grouped_layout, use_psum_layout=use_psum)
        dg.set_sf_cast_mode('sm100')
        sfa = case.a[1].clone()
        if use_psum:                    # the psum layout's gap rows are uninitialised memory: poison their scales
            start = 0
            for actual, aligned in zip(case.actual_ms, case.aligned_ms):
                sfa[start + actual:start + aligned] = float('nan')
                start += aligned
        for nn in (False, True):
            d = torch.full_like(case.d, float('nan'))
            if nn:
                dg.m_grouped_fp8_gemm_nn_contiguous((case.a[0], sfa), (case.b[0].mT.contiguous(), case.b[1].mT), d, case.grouped_layout,
                                                    use_psum_layout=use_psum)
            else:
                dg.m_grouped_fp8_gemm_nt_contiguous((case.a[0], sfa), case.b, d, case.grouped_layout, use_psum_layout=use_psum)
            assert dg.last_config().startswith('e8_quad'), dg.last_config()
            start = 0
            for actual, aligned in zip(case.actual_ms, case.aligned_ms):
                assert torch.equal(d[start:start + actual], d_int[start:start + actual])
                assert bool((d[start + actual:start + aligned] == 0).all())
                start += aligned
        dg.set_sf_cast_mode('sm90')


def test_masked_sm100_mode_is_the_packed_int_call():
    gen.reset_seed(17)
    for masked_ms, max_m, n, k in (([5, 0, 64, 33], 64, 256, 384), ([200, 1, 129], 256, 520, 512), ([20] * 6 + [0, 64], 64, 4096, 512)):
        case = gen.generate_m_grouped_masked(len(masked_ms), max_m, 0, n, k, masked_ms=masked_ms, use_ue8m0=True)
        a_int, b_int = gen.packed_ue8m0_operand(*case.a), gen.packed_ue8m0_operand(*case.b, mn_rows=n)
        expected_m = max(1, int(sum(masked_ms) / len(masked_ms)))
        d_int = torch.full_like(case.d, float('nan'))
        dg.m_grouped_fp8_gemm_nt_masked(a_int, b_int, d_int, case.masked_m, expected_m)
        cfg = dg.last_config()
        dg.set_sf_cast_mode('sm100')
        d = torch.full_like(case.d, float('nan'))
        dg.m_grouped_fp8_gemm_nt_masked(case.a, case.b, d, case.masked_m, expected_m)
        assert dg.last_config() == cfg and cfg.startswith('e8_')
        for g, rows in enumerate(masked_ms):
            assert torch.equal(d[g, :rows], d_int[g, :rows])
            assert bool(torch.isnan(d[g, rows:]).all())
        dg.set_sf_cast_mode('sm90')
        d90 = torch.full_like(case.d, float('nan'))
        dg.m_grouped_fp8_gemm_nt_masked(case.a, case.b, d90, case.masked_m, expected_m)
        assert not dg.last_config().startswith('e8_')


def test_c2_sm100_mode_whole_call_matches_oracle_rows():
    """BASELINE configs[1] through the mode: power-of-two FP32 scales, whole call (two pack launches + the GEMM)."""
    m, n, k = 4096, 4096, 7168
    gen.reset_seed(0)
    case = gen.generate_normal(m, n, k, use_ue8m0=True)
    dg.set_sf_cast_mode('sm100')
    dg.fp8_gemm_nt(case.a, case.b, case.d)
    assert dg.last_config() == 'e8_quad_256x256'
    assert calc_diff(case.d, case.ref_d) < gen.FP8_MAX_DIFF
    rows = torch.tensor(sorted({0, 1, 127, 128, 255, 256, 2047, 2048, 4095} | set(range(1000, 1016))))
    want = torch.empty((len(rows), n), dtype=torch.bfloat16)
    oracle.fp8_gemm_nt(case.a[0][rows.cuda()].cpu(), case.a[1][rows.cuda()].cpu(), case.b[0].cpu(), case.b[1].cpu(), want)
    assert_close_to_oracle(case.d[rows.cuda()], want, 'C2 sm100 mode')
