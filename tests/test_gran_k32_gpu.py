"""Scale granularity 32 along K -- the reference's SM100 MX recipe for FP8 x FP8 operands (csrc/apis/gemm.hpp:311-312 ``gran_k == 32 or
gran_k == 128``; csrc/apis/layout.hpp:48-58; ``per_token_cast_to_fp8(..., gran_k=32, use_ue8m0=True, use_packed_ue8m0=True)``,
deep_gemm/utils/math.py:26-38; sweep tests/generators.py:192-194,230): packed UE8M0 words hold the four exponents of ONE 128-K block, the scaled
MFMA takes one byte per lane group natively (csrc/fp8_gemm_quad.hpp, G32).  Every result is checked against the C oracle evaluated with 32-K blocks
on the same bytes and the same scales as FP32 (tolerance of the packed gran-128 tests: the only freedom is the matrix core's summation order;
products with power-of-two scales are exact) and against the reference's own gate vs the unquantised matmul."""
import pytest
import torch

import deepgemm_amd as dg
import oracle
from deepgemm_amd.testing import calc_diff, generators as gen
from deepgemm_amd.utils.math import pack_ue8m0_to_int, per_token_cast_to_fp8
from gpu_helpers import assert_close_to_oracle, assert_close_fp32

pytestmark = pytest.mark.gpu


def _cast32(x: torch.Tensor):
    """(fp8, FP32 power-of-two scales [rows, k / 32], packed words [rows, k / 128]) with the reference's quantiser at gran_k = 32."""
    q, sf = per_token_cast_to_fp8(x, use_ue8m0=True, gran_k=32)
    return q, sf, pack_ue8m0_to_int(sf)


def _device_oracle32(a, sfa, b, sfb, out_dtype=torch.bfloat16, chunk=512):
    """The oracle's statement on the device for full-size outputs: per 32-K block the exactly scaled partial product (power-of-two scales: exact in
    FP32), summed in FP64 and rounded once -- within the tolerance below of any FP32 summation order."""
    m, k = a.shape
    out = torch.empty((m, b.size(0)), dtype=out_dtype, device=a.device)
    bd = (b.float().view(b.size(0), k // 32, 32) * sfb.unsqueeze(-1)).view(b.size(0), k).double()
    for r0 in range(0, m, chunk):
        ad = (a[r0:r0 + chunk].float().view(-1, k // 32, 32) * sfa[r0:r0 + chunk].unsqueeze(-1)).view(-1, k).double()
        out[r0:r0 + chunk] = (ad @ bd.t()).to(out_dtype)
    return out


@pytest.mark.parametrize('m,n,k', [(256, 512, 1024), (300, 520, 1536), (64, 136, 512), (129, 4096, 384), (1024, 2048, 7168)])
def test_dense_packed_gran_k_32(m, n, k):
    gen.reset_seed(m + n + k)
    a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16)
    b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
    ref = (a.float() @ b.float().t()).to(torch.bfloat16)
    (a_q, sfa, pa), (b_q, sfb, pb) = _cast32(a), _cast32(b)
    assert pa.shape == (m, k // 128) and pb.shape == (n, k // 128)
    d = torch.full((m, n), float('nan'), device='cuda', dtype=torch.bfloat16)
    dg.fp8_gemm_nt((a_q, pa), (b_q, pb), d, recipe=(1, 1, 32))
    # (decode-sized M: the granularity-32 stream tiles, e8_stream*_g32_*; else the four-wave forms)
    assert dg.last_config().startswith('e8_') and '_g32_' in dg.last_config() and (m > 256 or 'stream' in dg.last_config()), dg.last_config()
    want = torch.empty((m, n), dtype=torch.bfloat16)
    oracle.fp8_gemm_nt(a_q.cpu(), sfa.cpu(), b_q.cpu(), sfb.cpu(), want, gran_n=1, gran_k=32)
    assert_close_to_oracle(d, want, 'packed ue8m0, gran_k 32')
    assert calc_diff(d, ref) < gen.FP8_MAX_DIFF
    # recipe_a / recipe_b spelling of the same call; both tile forms agree bit for bit (in-place accumulation in K order in both)
    d2 = torch.empty_like(d)
    dg.fp8_gemm_nt((a_q, pa), (b_q, pb), d2, recipe_a=(1, 32), recipe_b=(1, 32))
    assert torch.equal(d.view(torch.int16), d2.view(torch.int16))
    for name in ('e8_quad_g32_256x256', 'e8_quad_g32_128x256'):
        dg.set_forced_config(name)
        try:
            d3 = torch.full_like(d, float('nan'))
            dg.fp8_gemm_nt((a_q, pa), (b_q, pb), d3, recipe=(1, 1, 32))
            assert dg.last_config() == name
        finally:
            dg.set_forced_config('auto')
        assert torch.equal(d.view(torch.int16), d3.view(torch.int16)), name
    # FP32 output with accumulation
    c32 = torch.randn((m, n), device='cuda', dtype=torch.float)
    d32 = c32.clone()
    dg.fp8_gemm_nt((a_q, pa), (b_q, pb), d32, c=d32, recipe=(1, 1, 32))
    want32 = torch.empty((m, n), dtype=torch.float)
    oracle.fp8_gemm_nt(a_q.cpu(), sfa.cpu(), b_q.cpu(), sfb.cpu(), want32, c=c32.cpu(), gran_n=1, gran_k=32)
    assert_close_fp32(d32, want32, 'packed ue8m0 gran_k 32, fp32 accumulate')


def test_gran_k_32_differs_from_gran_k_128_scales():
    """Guards against a kernel that ignores three of the four bytes: scales that differ between the 32-K groups of a block."""
    gen.reset_seed(7)
    m, n, k = 128, 256, 512
    a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16)
    a[:, 32:64] *= 64.0                     # the second 32-K group of every row gets a much larger scale
    a[:, 448:480] *= 1.0 / 64.0
    b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
    b[:, 96:128] *= 32.0
    (a_q, sfa, pa), (b_q, sfb, pb) = _cast32(a), _cast32(b)
    assert int((sfa[:, 0] != sfa[:, 1]).sum()) > m // 2
    d = torch.empty((m, n), device='cuda', dtype=torch.bfloat16)
    dg.fp8_gemm_nt((a_q, pa), (b_q, pb), d, recipe=(1, 1, 32))
    want = torch.empty((m, n), dtype=torch.bfloat16)
    oracle.fp8_gemm_nt(a_q.cpu(), sfa.cpu(), b_q.cpu(), sfb.cpu(), want, gran_n=1, gran_k=32)
    assert_close_to_oracle(d, want, 'gran_k 32, uneven groups')


@pytest.mark.parametrize('layout', ['nn', 'tn', 'tt'])
def test_gran_k_32_mn_major_operands(layout):
    """MN-major operands are re-majored by the host layer in front of the G32 kernels (csrc/apis/gemm.hpp:126-164 semantics)."""
    gen.reset_seed(11)
    m, n, k = 384, 512, 1024
    a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16)
    b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
    (a_q, sfa, pa), (b_q, sfb, pb) = _cast32(a), _cast32(b)
    a_in = a_q if layout[0] == 'n' else a_q.t().contiguous().t()
    b_in = b_q if layout[1] == 't' else b_q.t().contiguous().t()
    d = torch.empty((m, n), device='cuda', dtype=torch.bfloat16)
    dg.fp8_gemm_nt((a_in, pa), (b_in, pb), d, recipe=(1, 1, 32))
    want = torch.empty((m, n), dtype=torch.bfloat16)
    oracle.fp8_gemm_nt(a_q.cpu(), sfa.cpu(), b_q.cpu(), sfb.cpu(), want, gran_n=1, gran_k=32)
    assert_close_to_oracle(d, want, f'gran_k 32 {layout}')


def test_gran_k_32_fp32_scales_in_sm100_mode():
    """FP32 power-of-two scales with recipe (1, 32, 32) in the 'sm100' cast mode: the layout step packs them (csrc/apis/layout.hpp:48-54, the
    row broadcast of the 32 x 32 weight blocks fused into the pack kernel) -- same bits as handing over the packed words."""
    from deepgemm_amd.utils.math import per_block_cast_to_fp8
    gen.reset_seed(3)
    m, n, k = 256, 384, 1024
    a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16)
    b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
    a_q, sfa, pa = _cast32(a)
    b_q, sfb_blocks = per_block_cast_to_fp8(b, use_ue8m0=True, gran_k=32)                # [n / 32, k / 32]
    sfb_rows = sfb_blocks.repeat_interleave(32, dim=0)[:n].contiguous()
    d_packed = torch.empty((m, n), device='cuda', dtype=torch.bfloat16)
    dg.fp8_gemm_nt((a_q, pa), (b_q, pack_ue8m0_to_int(sfb_rows)), d_packed, recipe=(1, 1, 32))
    dg.set_sf_cast_mode('sm100')
    try:
        d = torch.empty_like(d_packed)
        dg.fp8_gemm_nt((a_q, sfa), (b_q, sfb_blocks), d, recipe=(1, 32, 32))
        assert dg.last_config().startswith('e8_') and '_g32_' in dg.last_config()
    finally:
        dg.set_sf_cast_mode('sm90')
    assert torch.equal(d.view(torch.int16), d_packed.view(torch.int16))
    with pytest.raises(RuntimeError):                   # FP32 scales of granularity 32 consumed as FP32: no such arithmetic (reference: SM100 only)
        dg.fp8_gemm_nt((a_q, sfa), (b_q, sfb_blocks), d, recipe=(1, 32, 32))


@pytest.mark.parametrize('use_psum', [False, True])
def test_m_grouped_contiguous_gran_k_32(use_psum):
    gen.reset_seed(5)
    n, k = 512, 1024
    actual_ms = [100, 256, 0, 130]
    case = gen.generate_m_grouped_contiguous(len(actual_ms), 0, n, k, True, use_psum, actual_ms=actual_ms)
    m = case.m
    # the generator's layout with fresh BF16 operands quantised at granularity 32 (its own casts are gran-128); padding rows of A are zero as
    # the reference's generator leaves them (tests/generators.py:343-355)
    a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16)
    if not use_psum:
        a[case.grouped_layout < 0] = 0
    b = torch.randn((len(actual_ms), n, k), device='cuda', dtype=torch.bfloat16)
    a_q, sfa, pa = _cast32(a)
    bq = [_cast32(b[g]) for g in range(len(actual_ms))]
    b_q, sfb, pb = torch.stack([x[0] for x in bq]), torch.stack([x[1] for x in bq]), torch.stack([x[2] for x in bq])
    d = torch.full((m, n), float('nan'), device='cuda', dtype=torch.bfloat16)
    dg.m_grouped_fp8_gemm_nt_contiguous((a_q, pa), (b_q, pb), d, case.grouped_layout, recipe=(1, 1, 32), use_psum_layout=use_psum)
    assert 'g32' in dg.last_config(), dg.last_config()
    want = torch.full((m, n), float('nan'), dtype=torch.bfloat16)
    oracle.m_grouped_fp8_gemm_nt_contiguous(a_q.cpu(), sfa.cpu(), b_q.cpu(), sfb.cpu(), want, case.grouped_layout.cpu(), use_psum,
                                            gran_k=32, gran_n=1)
    written = ~torch.isnan(want.float())
    assert bool(written.any())
    assert_close_to_oracle(torch.where(written.cuda(), d, torch.zeros_like(d)), torch.where(written, want, torch.zeros_like(want)), 'contiguous gran_k 32')
    if not use_psum:        # padding rows (-1) are exact zeros (tests/test_fp8_fp4.py:22-29)
        pad = (case.grouped_layout < 0)
        assert bool((d[pad] == 0).all())


@pytest.mark.parametrize('masked_ms,max_m', [([5, 0, 64, 33], 64), ([200, 17, 256], 256)])
def test_m_grouped_masked_gran_k_32(masked_ms, max_m):
    gen.reset_seed(9)
    n, k = 512, 768
    g = len(masked_ms)
    a = torch.randn((g, max_m, k), device='cuda', dtype=torch.bfloat16)
    b = torch.randn((g, n, k), device='cuda', dtype=torch.bfloat16)
    aq, bq = [_cast32(a[i]) for i in range(g)], [_cast32(b[i]) for i in range(g)]
    a_q, sfa, pa = (torch.stack([x[j] for x in aq]) for j in range(3))
    b_q, sfb, pb = (torch.stack([x[j] for x in bq]) for j in range(3))
    masked = torch.tensor(masked_ms, dtype=torch.int, device='cuda')
    d = torch.full((g, max_m, n), float('nan'), device='cuda', dtype=torch.bfloat16)
    dg.m_grouped_fp8_gemm_nt_masked((a_q, pa), (b_q, pb), d, masked, max(1, sum(masked_ms) // g), recipe=(1, 1, 32))
    assert 'g32' in dg.last_config(), dg.last_config()
    want = torch.full((g, max_m, n), float('nan'), dtype=torch.bfloat16)
    oracle.m_grouped_fp8_gemm_nt_masked(a_q.cpu(), sfa.cpu(), b_q.cpu(), sfb.cpu(), want, masked.cpu(), gran_k=32, gran_n=1)
    for i, rows in enumerate(masked_ms):
        if rows:
            assert_close_to_oracle(d[i, :rows], want[i, :rows], f'masked gran_k 32, group {i}')
        assert bool(torch.isnan(d[i, rows:]).all()), 'rows >= masked_m stay untouched'


def test_full_size_c2_c4_c5_gran_k_32():
    """BASELINE configs[1], [3], [4] (one rank) at size with granularity-32 scales, every element against the device statement of the oracle
    (sampled rows of which are pinned to the C oracle)."""
    # C2
    gen.reset_seed(0)
    m, n, k = 4096, 4096, 7168
    a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16)
    b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
    (a_q, sfa, pa), (b_q, sfb, pb) = _cast32(a), _cast32(b)
    d = torch.empty((m, n), device='cuda', dtype=torch.bfloat16)
    dg.fp8_gemm_nt((a_q, pa), (b_q, pb), d, recipe=(1, 1, 32))
    assert dg.last_config() == 'e8_quad_g32_256x256'
    want = _device_oracle32(a_q, sfa, b_q, sfb)
    assert calc_diff(d, want) < 2e-6
    rows = [0, 1, 255, 256, 2047, 4095]
    pin = torch.empty((len(rows), n), dtype=torch.bfloat16)
    oracle.fp8_gemm_nt(a_q[rows].cpu(), sfa[rows].cpu(), b_q.cpu(), sfb.cpu(), pin, gran_n=1, gran_k=32)
    assert calc_diff(want[rows].cpu(), pin) < 2e-6 and calc_diff(d[rows].cpu(), pin) < 2e-6
    del a, b, want
    # C4: 8 groups x ~512 rows (the group-relative tiling of the contiguous layout)
    import random
    random.seed(0)
    actual_ms = [int(512 * random.uniform(0.7, 1.3)) for _ in range(8)]
    case = gen.generate_m_grouped_contiguous(8, 512, 4096, 7168, actual_ms=actual_ms)
    mm = case.m
    a = torch.randn((mm, k), device='cuda', dtype=torch.bfloat16)
    a[case.grouped_layout < 0] = 0
    a_q, sfa, pa = _cast32(a)
    wq = [_cast32(torch.randn((n, k), device='cuda', dtype=torch.bfloat16)) for _ in range(8)]
    w_q, sfw, pw = (torch.stack([x[j] for x in wq]) for j in range(3))
    d = torch.full((mm, n), float('nan'), device='cuda', dtype=torch.bfloat16)
    dg.m_grouped_fp8_gemm_nt_contiguous((a_q, pa), (w_q, pw), d, case.grouped_layout, recipe=(1, 1, 32))
    assert dg.last_config() == 'e8_quad_g32_tab_256x256', dg.last_config()
    start = 0
    for g, (actual, aligned) in enumerate(zip(case.actual_ms, case.aligned_ms)):
        want = _device_oracle32(a_q[start:start + actual], sfa[start:start + actual], w_q[g], sfw[g])
        assert calc_diff(d[start:start + actual], want) < 2e-6, g
        assert bool((d[start + actual:start + aligned] == 0).all()), 'padding rows are exact zeros'
        start += aligned
    del a, wq, case
    # C5: 8 local experts, M <= 64
    masked_ms = [int(48 * random.uniform(0.7, 1.3)) for _ in range(8)]
    aq = [_cast32(torch.randn((64, k), device='cuda', dtype=torch.bfloat16)) for _ in range(8)]
    a_q, sfa, pa = (torch.stack([x[j] for x in aq]) for j in range(3))
    masked = torch.tensor(masked_ms, dtype=torch.int, device='cuda')
    d = torch.full((8, 64, n), float('nan'), device='cuda', dtype=torch.bfloat16)
    dg.m_grouped_fp8_gemm_nt_masked((a_q, pa), (w_q, pw), d, masked, 48, recipe=(1, 1, 32))
    for g, rows in enumerate(masked_ms):
        want = _device_oracle32(a_q[g, :rows], sfa[g, :rows], w_q[g], sfw[g])
        assert calc_diff(d[g, :rows], want) < 2e-6, g
        assert bool(torch.isnan(d[g, rows:]).all())


@pytest.mark.parametrize('m,n,k', [(40, 4096, 7168), (17, 520, 1408), (64, 4096, 2048), (200, 264, 384), (96, 7168, 1536)])
def test_decode_sized_gran_k_32_runs_on_stream_tiles(m, n, k):
    """Decode-sized M at granularity 32 (round 6): the stream tiles with every stage carrying its K block's words (e8_stream*_g32_*) instead of the
    128-row four-wave tile (68 -> 21 us at 4096 x 7168) -- against the oracle, and bit for bit against the four-wave form forced by name (both
    accumulate in the matrix core in K order)."""
    gen.reset_seed(m + n + k)
    a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16)
    b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
    (a_q, sfa, pa), (b_q, sfb, pb) = _cast32(a), _cast32(b)
    d = torch.full((m, n), float('nan'), device='cuda', dtype=torch.bfloat16)
    dg.fp8_gemm_nt((a_q, pa), (b_q, pb), d, recipe=(1, 1, 32))
    picked = dg.last_config()
    assert picked.startswith('e8_stream') and '_g32_' in picked, picked
    if m * n * k <= 64 * 4096 * 2048:
        want = torch.empty((m, n), dtype=torch.bfloat16)
        oracle.fp8_gemm_nt(a_q.cpu(), sfa.cpu(), b_q.cpu(), sfb.cpu(), want, gran_n=1, gran_k=32)
        assert_close_to_oracle(d, want, 'decode-sized gran_k 32')
    assert calc_diff(d, (a.float() @ b.float().t()).to(torch.bfloat16)) < gen.FP8_MAX_DIFF
    dg.set_forced_config('e8_quad_g32_128x256')
    try:
        d2 = torch.full_like(d, float('nan'))
        dg.fp8_gemm_nt((a_q, pa), (b_q, pb), d2, recipe=(1, 1, 32))
        assert dg.last_config() == 'e8_quad_g32_128x256'
    finally:
        dg.set_forced_config('auto')
    if '_ks_' in picked:        # (cut along K inside the kernel: the same products in another summation split)
        assert calc_diff(d.float(), d2.float()) < 2e-6
    else:
        assert torch.equal(d.view(torch.int16), d2.view(torch.int16))


@pytest.mark.parametrize('gran_k', [128, 32])
@pytest.mark.parametrize('m,n,k,accumulate,out_dtype', [(1, 4096, 7168, False, torch.bfloat16), (16, 528, 2048, False, torch.bfloat16),
                                                       (7, 272, 2560, True, torch.float), (32, 2048, 7168, False, torch.bfloat16),
                                                       (20, 4608, 6144, True, torch.bfloat16)])
def test_batch_decode_with_packed_scales_runs_the_skinny_kernel(gran_k, m, n, k, accumulate, out_dtype):
    """Batch-1 .. 32 decode with packed UE8M0 scale words (round 6, both granularities): the skinny weight-stream kernel with the scaled MFMA
    (e8_skinny_*: a wave accumulates its eighth of K in the matrix core, the eight partial tiles are summed in wave order) -- against the oracle
    and against the stream tile forced by name (same products, another summation split: tolerance of the K-split paths)."""
    gen.reset_seed(m + n + k + gran_k)
    a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16)
    b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
    qa, qb = per_token_cast_to_fp8(a, use_ue8m0=True, gran_k=gran_k), per_token_cast_to_fp8(b, use_ue8m0=True, gran_k=gran_k)
    pa = dg.transform_sf_into_required_layout(pack_ue8m0_to_int(qa[1]), m, k, (1, gran_k))
    pb = dg.transform_sf_into_required_layout(pack_ue8m0_to_int(qb[1]), n, k, (1, gran_k))
    c0 = (torch.randn((m, n), device='cuda', dtype=out_dtype) * 8) if accumulate else None
    d = c0.clone() if accumulate else torch.full((m, n), float('nan'), device='cuda', dtype=out_dtype)
    dg.fp8_gemm_nt((qa[0], pa), (qb[0], pb), d, c=d if accumulate else None, recipe=(1, 1, gran_k))
    want_name = ('e8_skinny_g32_' if gran_k == 32 else 'e8_skinny_') + ('16' if m <= 16 else '32')
    assert dg.last_config() == want_name, dg.last_config()
    want = torch.empty((m, n), dtype=out_dtype)
    oracle.fp8_gemm_nt(qa[0].cpu(), qa[1].cpu(), qb[0].cpu(), qb[1].cpu(), want, c=c0.cpu() if accumulate else None, gran_n=1, gran_k=gran_k)
    if out_dtype == torch.float:
        assert_close_fp32(d, want, f'packed skinny gran {gran_k}')
    else:
        assert_close_to_oracle(d, want, f'packed skinny gran {gran_k}', addend=c0)
    dg.set_forced_config('e8_stream_g32_64x32' if gran_k == 32 else 'e8_stream_64x32')
    try:
        d2 = c0.clone() if accumulate else torch.full_like(d, float('nan'))
        dg.fp8_gemm_nt((qa[0], pa), (qb[0], pb), d2, c=d2 if accumulate else None, recipe=(1, 1, gran_k))
        assert 'stream' in dg.last_config()
    finally:
        dg.set_forced_config('auto')
    assert calc_diff(d.float(), d2.float()) < 2e-6
    # repeatable to the bit (the eight partial tiles are summed in wave order)
    d3 = c0.clone() if accumulate else torch.full_like(d, float('nan'))
    dg.fp8_gemm_nt((qa[0], pa), (qb[0], pb), d3, c=d3 if accumulate else None, recipe=(1, 1, gran_k))
    assert torch.equal(d3, d)


def test_randomized_decode_sized_packed_shapes_both_granularities():
    """Randomised decode-sized problems with packed scale words (end of round 6: the skinny kernels with the scaled MFMA, the stream tiles with a K
    quad's words in the group ring, the 64 x 32 tile with loader waves): every case against the device statement of the oracle (exactly scaled
    32- / 128-K partial products summed in FP64, rounded once) and -- where the automatic pick is a stream tile -- bit for bit against the four-wave
    128-row tile forced by name (both accumulate in the matrix core in K order)."""
    import random
    rng = random.Random(2026)
    seen = set()
    for case in range(28):
        gran_k = rng.choice([128, 32])
        m = rng.choice([1, 2, 7, 16, 17, 31, 32, 33, 48, 64, 65, 100, 128, 129, 200, 256])
        n = rng.choice([16, 48, 272, 528, 1040, 2112, 4096, 7168]) + rng.choice([0, 0, 16, 8])
        k = rng.choice([512, 1024, 1536, 2048, 2560, 4096, 7168])
        accumulate = rng.random() < 0.3
        out_dtype = torch.float if accumulate and rng.random() < 0.5 else torch.bfloat16
        gen.reset_seed(case)
        a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16)
        b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
        qa, qb = per_token_cast_to_fp8(a, use_ue8m0=True, gran_k=gran_k), per_token_cast_to_fp8(b, use_ue8m0=True, gran_k=gran_k)
        pa = dg.transform_sf_into_required_layout(pack_ue8m0_to_int(qa[1]), m, k, (1, gran_k))
        pb = dg.transform_sf_into_required_layout(pack_ue8m0_to_int(qb[1]), n, k, (1, gran_k))
        c0 = (torch.randn((m, n), device='cuda', dtype=out_dtype) * 4) if accumulate else None
        d = c0.clone() if accumulate else torch.full((m, n), float('nan'), device='cuda', dtype=out_dtype)
        dg.fp8_gemm_nt((qa[0], pa), (qb[0], pb), d, c=d if accumulate else None, recipe=(1, 1, gran_k))
        picked = dg.last_config()
        seen.add(picked)
        ad = (qa[0].float().view(m, k // gran_k, gran_k) * qa[1].unsqueeze(-1)).view(m, k).double()
        bd = (qb[0].float().view(n, k // gran_k, gran_k) * qb[1].unsqueeze(-1)).view(n, k).double()
        exact = ad @ bd.t()
        label = f'case {case}: {m} x {n} x {k} gran {gran_k} {picked} acc={accumulate} {out_dtype}'
        assert not bool(torch.isnan(d).any()), label
        if out_dtype == torch.float:
            want = exact + (c0.double() if accumulate else 0)
            assert float((d.double() - want).norm() / want.norm()) < 5e-5, label      # (gpu_helpers: the matrix core does not round a block sum correctly)
        else:           # (BF16 reduce-add: the GEMM result is rounded to BF16, then added -- gpu_helpers.assert_close_to_oracle)
            want = (exact.to(torch.bfloat16).float() + c0.float()).to(torch.bfloat16) if accumulate else exact.to(torch.bfloat16)
            assert_close_to_oracle(d, want, label, addend=c0)
        if 'stream' in picked and '_ks_' not in picked:         # (the K-split forms sum the pieces' partials: test_packed_stream_tiles_cut_along_k)
            dg.set_forced_config('e8_quad_g32_128x256' if gran_k == 32 else 'e8_quad_128x256')
            try:
                d2 = c0.clone() if accumulate else torch.full_like(d, float('nan'))
                dg.fp8_gemm_nt((qa[0], pa), (qb[0], pb), d2, c=d2 if accumulate else None, recipe=(1, 1, gran_k))
            finally:
                dg.set_forced_config('auto')
            assert torch.equal(d.view(torch.int32 if out_dtype == torch.float else torch.int16), d2.view(torch.int32 if out_dtype == torch.float else torch.int16)), label
    assert any(s.startswith('e8_skinny') for s in seen) and any('stream' in s for s in seen), seen


@pytest.mark.parametrize('gran_k', [128, 32])
@pytest.mark.parametrize('m,n,k', [(128, 576, 7168), (33, 4096, 7168), (65, 520, 4096), (200, 96, 5120), (256, 576, 16384), (192, 4096, 7168), (256, 2112, 4608),
                                   (24, 1536, 7168), (40, 4096, 4096), (192, 1536, 10240), (128, 6144, 7168), (65, 4096, 12288)])
def test_packed_stream_tiles_cut_along_k(gran_k, m, n, k):
    """`e8_stream_ks_64x32` / `_64x128` and their granularity-32 forms (end of round 6): the packed-scale stream tiles cut along K inside the kernel
    in pieces of whole K quads (the FP32-scale rules of `stream_ks_*`) -- the automatic pick through the plain entry (the host layer lends the
    stream's workspace) against the oracle and the FP64 statement, against the unsplit tile forced by name (same products, another summation
    split), repeated calls on the dirty workspace bit for bit, FP32 accumulation, and each form forced by name on a shape outside its rule."""
    gen.reset_seed(m + n + k + gran_k)
    a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16)
    b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
    qa, qb = per_token_cast_to_fp8(a, use_ue8m0=True, gran_k=gran_k), per_token_cast_to_fp8(b, use_ue8m0=True, gran_k=gran_k)
    pa = dg.transform_sf_into_required_layout(pack_ue8m0_to_int(qa[1]), m, k, (1, gran_k))
    pb = dg.transform_sf_into_required_layout(pack_ue8m0_to_int(qb[1]), n, k, (1, gran_k))
    g = '_g32' if gran_k == 32 else ''
    tiles128 = -(-m // 64) * -(-n // 128)
    wide = (m > 128 and (tiles128 >= 64 or (tiles128 > 32 and k >= 10240))) or (64 < m <= 128 and ((tiles128 >= 96 and k >= 7168) or (64 <= tiles128 < 96 and k >= 12288)))       # (17 .. 32 rows: where the skinny kernel does not apply, or narrow layers with K >= 7168)
    want_name = f'e8_stream_ks{g}_64x128' if wide else f'e8_stream_ks{g}_64x32'
    outs = []
    for _ in range(3):
        d = torch.full((m, n), float('nan'), device='cuda', dtype=torch.bfloat16)
        dg.fp8_gemm_nt((qa[0], pa), (qb[0], pb), d, recipe=(1, 1, gran_k))
        assert dg.last_config() == want_name, dg.last_config()
        outs.append(d)
    d = outs[0]
    assert all(torch.equal(d.view(torch.int16), o.view(torch.int16)) for o in outs[1:]), 'piece order is fixed: bit-repeatable'
    ad = (qa[0].float().view(m, k // gran_k, gran_k) * qa[1].unsqueeze(-1)).view(m, k).double()
    bd = (qb[0].float().view(n, k // gran_k, gran_k) * qb[1].unsqueeze(-1)).view(n, k).double()
    exact = ad @ bd.t()
    assert_close_to_oracle(d, exact.to(torch.bfloat16), f'{want_name} {m} x {n} x {k}')
    if m * n * k <= 128 * 576 * 7168:
        want = torch.empty((m, n), dtype=torch.bfloat16)
        oracle.fp8_gemm_nt(qa[0].cpu(), qa[1].cpu(), qb[0].cpu(), qb[1].cpu(), want, gran_n=1, gran_k=gran_k)
        assert_close_to_oracle(d, want, f'{want_name} against the C oracle')
    assert calc_diff(d, (a.float() @ b.float().t()).to(torch.bfloat16)) < gen.FP8_MAX_DIFF
    dg.set_forced_config(f'e8_stream{g}_64x32')
    try:
        d2 = torch.full_like(d, float('nan'))
        dg.fp8_gemm_nt((qa[0], pa), (qb[0], pb), d2, recipe=(1, 1, gran_k))
        assert dg.last_config() == f'e8_stream{g}_64x32'
    finally:
        dg.set_forced_config('auto')
    assert calc_diff(d.float(), d2.float()) < 2e-6
    # FP32 accumulation into the output (the last piece reads the addend)
    c32 = torch.randn((m, n), device='cuda', dtype=torch.float) * 4
    d32 = c32.clone()
    dg.fp8_gemm_nt((qa[0], pa), (qb[0], pb), d32, c=d32, recipe=(1, 1, gran_k))
    assert dg.last_config() == want_name, dg.last_config()
    want32 = exact + c32.double()
    assert float((d32.double() - want32).norm() / want32.norm()) < 5e-5
    # the other form forced by name (outside its rule: any tile count up to 1024)
    other = f'e8_stream_ks{g}_64x32' if wide else f'e8_stream_ks{g}_64x128'
    dg.set_forced_config(other)
    try:
        d3 = torch.full_like(d, float('nan'))
        dg.fp8_gemm_nt((qa[0], pa), (qb[0], pb), d3, recipe=(1, 1, gran_k))
        assert dg.last_config() == other
    finally:
        dg.set_forced_config('auto')
    assert calc_diff(d.float(), d3.float()) < 2e-6


@pytest.mark.parametrize('gran_k', [128, 32])
def test_packed_k_split_in_a_hip_graph(gran_k):
    """A captured `e8_stream_ks_*` launch carries one exchange epoch: replays over changing inputs against eager calls, bit for bit."""
    m, n, k = 128, 576, 7168
    g = '_g32' if gran_k == 32 else ''

    def operands(seed):
        torch.manual_seed(seed)
        a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16)
        b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
        qa, qb = per_token_cast_to_fp8(a, use_ue8m0=True, gran_k=gran_k), per_token_cast_to_fp8(b, use_ue8m0=True, gran_k=gran_k)
        return (qa[0], dg.transform_sf_into_required_layout(pack_ue8m0_to_int(qa[1]), m, k, (1, gran_k)),
                qb[0], dg.transform_sf_into_required_layout(pack_ue8m0_to_int(qb[1]), n, k, (1, gran_k)))
    cases = [operands(80 + i) for i in range(3)]
    eager = []
    for c in cases:
        d = torch.empty((m, n), device='cuda', dtype=torch.bfloat16)
        dg.fp8_gemm_nt((c[0], c[1]), (c[2], c[3]), d, recipe=(1, 1, gran_k))
        assert dg.last_config() == f'e8_stream_ks{g}_64x32', dg.last_config()
        eager.append(d)
    held = [t.clone() for t in cases[0]]
    d = torch.empty((m, n), device='cuda', dtype=torch.bfloat16)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        dg.fp8_gemm_nt((held[0], held[1]), (held[2], held[3]), d, recipe=(1, 1, gran_k))      # (warm: plan caches, the stream's workspace)
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        dg.fp8_gemm_nt((held[0], held[1]), (held[2], held[3]), d, recipe=(1, 1, gran_k))
    for which in (1, 2, 0, 2, 1):
        for dst, src in zip(held, cases[which]):
            dst.copy_(src)
        d.fill_(float('nan'))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(d.view(torch.int16), eager[which].view(torch.int16)), which
