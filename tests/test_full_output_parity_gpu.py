"""EVERY element of EVERY BASELINE.json configuration against the oracle's arithmetic.

The C oracle takes seconds per row block at these sizes, so the other GPU tests compare a few dozen rows with it and gate the whole
output only through the reference's ``calc_diff < 1e-3`` -- which does not notice a few wrong tiles.  Here the oracle's torch
restatement (``oracle.fp8_gemm_nt_blockwise_torch``: FP64 block products, exact for FP8 operands, FP32 promotion in K-block order)
runs ON THE DEVICE over the whole problem and every output element is held to the tolerance of ``assert_close_to_oracle`` (one BF16
ulp + 2e-4 rms).  Anchor: the same restatement, on the device, is first pinned bit-for-bit to the C oracle on sampled rows of the same
inputs -- so "device restatement" cannot drift from ``oracle/fp8_gemm_oracle.c`` unnoticed."""
import pytest
import torch

import deepgemm_amd as dg
import oracle
from deepgemm_amd.testing import calc_diff, generators as gen
from gpu_helpers import assert_close_fp32, assert_close_to_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _auto_config():
    dg.set_forced_config('auto')
    dg.set_sf_cast_mode('sm90')
    dg.set_mk_alignment_for_contiguous_layout(128)
    yield
    dg.set_forced_config('auto')


def device_oracle(a, sfa, b, sfb, gran_n=128, out_dtype=torch.bfloat16, c=None, chunk=1024):
    """The oracle's arithmetic for all rows, in row chunks (bounds the FP64 temporaries)."""
    out = torch.empty((a.size(0), b.size(0)), dtype=out_dtype, device=a.device)
    for r0 in range(0, a.size(0), chunk):
        rows = slice(r0, min(a.size(0), r0 + chunk))
        out[rows] = oracle.fp8_gemm_nt_blockwise_torch(a[rows], sfa[rows], b, sfb, gran_n=gran_n, out_dtype=out_dtype,
                                                       c=None if c is None else c[rows])
    return out


def pin_to_c_oracle(a, sfa, b, sfb, want_dev, rows, gran_n=128, c=None):
    """The device restatement == the C oracle on `rows`: BF16 outputs bit for bit except where the two FP32 sums straddle a rounding
    boundary (the C oracle promotes with fmaf, torch with multiply-then-add: one FP32 ulp apart at most), FP32 outputs to 1e-6."""
    rows_dev = torch.as_tensor(rows, device=a.device)
    want_c = torch.empty((len(rows), b.size(0)), dtype=want_dev.dtype)
    oracle.fp8_gemm_nt(a[rows_dev].cpu(), sfa[rows_dev].cpu(), b.cpu(), sfb.cpu(), want_c, c=None if c is None else c[rows_dev].cpu(), gran_n=gran_n)
    got = want_dev[rows_dev].cpu()
    if want_dev.dtype == torch.bfloat16:
        differ = (got != want_c)
        assert differ.float().mean().item() < 2e-3, f'device restatement differs from the C oracle in {differ.float().mean().item():.2e} of the sampled elements'
        assert ((got.float() - want_c.float()).abs() <= want_c.float().abs() * 2.0 ** -7 + 1e-30).all()       # and then by one BF16 ulp only
    else:
        assert torch.allclose(got, want_c, rtol=1e-6, atol=1e-6 * want_c.abs().max().item())


def test_c2_every_element():
    """BASELINE configs[1]: fp8_gemm_nt 4096 x 4096 x 7168, all 16.8 M outputs, FP32-scale kernel and (power-of-two scales) the
    hardware-scaled kernel."""
    m, n, k = 4096, 4096, 7168
    for use_ue8m0 in (False, True):
        gen.reset_seed(0)
        case = gen.generate_normal(m, n, k, use_ue8m0=use_ue8m0)
        want = device_oracle(case.a[0], case.a[1], case.b[0], case.b[1])
        pin_to_c_oracle(case.a[0], case.a[1], case.b[0], case.b[1], want, [0, 1, 255, 256, 2047, 4095] + list(range(1000, 1010)))
        case.d.fill_(float('nan'))
        if use_ue8m0:
            dg.fp8_gemm_nt(gen.packed_ue8m0_operand(*case.a), gen.packed_ue8m0_operand(*case.b, mn_rows=n), case.d)
            assert dg.last_config() == 'e8_quad_256x256'
        else:
            dg.fp8_gemm_nt(case.a, case.b, case.d)
            assert dg.last_config() == 'duo_p_256x256'
        assert_close_to_oracle(case.d, want, f'C2 every element (ue8m0={use_ue8m0})')
        assert calc_diff(case.d, case.ref_d) < gen.FP8_MAX_DIFF
        del case, want
        torch.cuda.empty_cache()


@pytest.mark.parametrize('layout', ['nt', 'nn', 'tn', 'tt'])
def test_c3_every_element(layout):
    """BASELINE configs[2]: 2048 x 7168 x 2048 in all four layouts (operands materialised MN-major where the layout says so)."""
    m, n, k = 2048, 7168, 2048
    gen.reset_seed(1)
    case = gen.generate_normal(m, n, k, layout[0] == 'n', layout[1] == 't')
    want = device_oracle(case.a[0], case.a[1], case.b[0], case.b[1])
    pin_to_c_oracle(case.a[0], case.a[1], case.b[0], case.b[1], want, [0, 255, 256, 1023, 2047])
    case.d.fill_(float('nan'))
    dg.fp8_gemm_nt(case.a, case.b, case.d)
    assert_close_to_oracle(case.d, want, f'C3 {layout} every element ({dg.last_config()})')
    assert calc_diff(case.d, case.ref_d) < gen.FP8_MAX_DIFF


def test_c3_fp32_accumulate_every_element():
    """The accumulating FP32-output form at the C3 size (reference sweep: `accumulate` cases of tests/generators.py:126-131)."""
    m, n, k = 2048, 7168, 2048
    gen.reset_seed(2)
    case = gen.generate_normal(m, n, k, accumulate=True, out_dtype=torch.float)
    c0 = case.c.clone()
    want = device_oracle(case.a[0], case.a[1], case.b[0], case.b[1], out_dtype=torch.float, c=c0)
    dg.fp8_gemm_nt(case.a, case.b, case.d, c=case.c)
    assert_close_fp32(case.d, want, 'C3 FP32 accumulate every element')
    # element-wise: matrix-core accumulation noise is absolute (about 3e-5 rms of the product at K = 7168), not relative to a sum that
    # may cancel against C -- the bound of assert_close_to_oracle without its BF16 ulp
    rms = want.pow(2).mean().sqrt().item()
    worst = ((case.d - want).abs() - 2.0 ** -20 * want.abs()).max().item()
    assert worst <= 2e-4 * rms, (worst, rms)


def test_c4_every_element():
    """BASELINE configs[3]: m_grouped_fp8_gemm_nt_contiguous, 8 groups x ~512 rows, N 4096, K 7168: every valid row against the oracle,
    every padding row exactly zero."""
    gen.reset_seed(0)
    case = gen.generate_m_grouped_contiguous(8, 512, 4096, 7168)
    case.d.fill_(float('nan'))
    dg.m_grouped_fp8_gemm_nt_contiguous(case.a, case.b, case.d, case.grouped_layout)
    cfg = dg.last_config()
    start = 0
    for g, (actual, aligned) in enumerate(zip(case.actual_ms, case.aligned_ms)):
        rows = slice(start, start + actual)
        want = device_oracle(case.a[0][rows], case.a[1][rows], case.b[0][g], case.b[1][g])
        if g in (0, 5):
            pin_to_c_oracle(case.a[0][rows], case.a[1][rows], case.b[0][g], case.b[1][g], want, [0, 1, actual // 2, actual - 1])
        assert_close_to_oracle(case.d[rows], want, f'C4 group {g} every element ({cfg})')
        assert bool((case.d[start + actual:start + aligned] == 0).all()), f'C4 group {g}: padding rows must be zeros'
        start += aligned
    assert calc_diff(torch.nan_to_num(case.d), torch.nan_to_num(case.ref_d)) < gen.FP8_MAX_DIFF


@pytest.mark.parametrize('packed', [False, True])
def test_c5_every_element(packed):
    """BASELINE configs[4], one rank: masked grouped GEMM, 8 local experts, M <= 64 valid rows each, N 4096, K 7168: every valid element
    against the oracle, rows >= masked_m untouched (NaN poison)."""
    gen.reset_seed(0)
    case = gen.generate_m_grouped_masked(8, 64, 48, 4096, 7168, use_ue8m0=packed)
    case.d.fill_(float('nan'))
    a, b = (gen.packed_ue8m0_operand(*case.a), gen.packed_ue8m0_operand(*case.b, mn_rows=4096)) if packed else (case.a, case.b)
    dg.m_grouped_fp8_gemm_nt_masked(a, b, case.d, case.masked_m, 48)
    cfg = dg.last_config()
    for g, rows in enumerate(case.masked_m.tolist()):
        if rows:
            want = device_oracle(case.a[0][g, :rows], case.a[1][g, :rows], case.b[0][g], case.b[1][g])
            if g == 0:
                pin_to_c_oracle(case.a[0][g, :rows], case.a[1][g, :rows], case.b[0][g], case.b[1][g], want, [0, rows - 1])
            assert_close_to_oracle(case.d[g, :rows], want, f'C5 group {g} every element ({cfg})')
        assert bool(torch.isnan(case.d[g, rows:]).all()), f'C5 group {g}: rows >= masked_m must not be written'


def test_c1_every_element_is_exact():
    """BASELINE configs[0] at its own size is bit-exact (tests/test_gemm_gpu.py::test_c1_unit_scale_exact); here the device restatement
    itself is checked on it: unit scales, integer operands -> exactly the FP32 matmul."""
    torch.manual_seed(0)
    a = torch.randint(-8, 9, (128, 512), device='cuda').float().to(torch.float8_e4m3fn)
    b = torch.randint(-8, 9, (128, 512), device='cuda').float().to(torch.float8_e4m3fn)
    sfa, sfb = torch.ones((128, 4), device='cuda'), torch.ones((1, 4), device='cuda')
    want = device_oracle(a, sfa, b, sfb)
    assert torch.equal(want, (a.float() @ b.float().t()).to(torch.bfloat16))
    d = torch.empty((128, 128), device='cuda', dtype=torch.bfloat16)
    dg.fp8_gemm_nt((a, sfa), (b, sfb), d)
    assert torch.equal(d, want)
