"""K-grouped GEMM with UE8M0 scales on the hardware-scaled kernels (round 6): the reference's SM100 form of
``k_grouped_fp8_gemm_tn_contiguous`` (csrc/apis/gemm.hpp:299-346; reference test tests/test_fp8_fp4.py:193-231 over
tests/generators.py:190-213: gran_k 32 / 128, K alignments 32 / 128 / 160 / 224, with and without the psum layout, empty groups, K tails).

Every group is compared with the oracle run on that group's own K-major operands and scales (recipe (1, 1, gran_k), FP32 accumulate into C) at
the FP32-output tolerance of tests/gpu_helpers.py, the whole output with the reference's own test expression (calc_diff < 1e-3), and the
layout step (FP32 -> packed words) bit for bit with its host-side statement."""
import pytest
import torch

import oracle
import deepgemm_amd as dg
from deepgemm_amd.testing import calc_diff, generators as gen
from gpu_helpers import assert_close_fp32

pytestmark = pytest.mark.gpu


def _with_fp32_scales(gran_k, fn):
    """FP32 power-of-two scale tensors: granularity 32 always takes the UE8M0 cast of the layout step; granularity 128 in the 'sm100'
    scaling-factor mode (the default mode keeps FP32 scales on the FP32-promotion kernels)."""
    if gran_k == 32:
        return fn()
    mode = dg.get_sf_cast_mode()
    dg.set_sf_cast_mode('sm100')
    try:
        return fn()
    finally:
        dg.set_sf_cast_mode(mode)


def _check_groups(d, case, real_ks, gran_k, label):
    for g, k in enumerate(real_ks):
        if k == 0:
            assert torch.equal(d[g], case.c[g]), f'{label}: empty group {g} must leave d[g] = c[g]'
            continue
        (a_g, sfa_g), (b_g, sfb_g) = case.a_groups[g], case.b_groups[g]
        want = torch.empty(d.shape[1:], dtype=torch.float)
        oracle.fp8_gemm_nt(a_g.cpu(), sfa_g.cpu(), b_g.cpu(), sfb_g.cpu(), want, c=case.c[g].cpu(), gran_n=1, gran_k=gran_k)
        assert_close_fp32(d[g], want, f'{label} group {g}')
    assert calc_diff(d, case.ref_d) < gen.FP8_MAX_DIFF


@pytest.mark.parametrize('gran_k', [128, 32])
@pytest.mark.parametrize('num_groups,m,n,ks', [(3, 256, 384, [256, 0, 512]), (2, 200, 264, [128, 384]), (4, 512, 1024, [1024, 896, 1152, 768]),
                                               (2, 128, 256, [640, 128]), (3, 64, 128, [384, 128, 256]), (5, 304, 272, [128, 256, 384, 512, 640])])
def test_k_grouped_ue8m0_host_extents(gran_k, num_groups, m, n, ks):
    """Extents on the host (``ks_cpu``), K alignment 128: packed int words and FP32 power-of-two scales (cast by the layout step) give the same
    bits; any number of K blocks per group (not only whole quads of four); 128- and 256-row tiles; in-place accumulation."""
    gen.reset_seed(sum(ks) + m + gran_k)
    case = gen.generate_k_grouped_contiguous_ue8m0(num_groups, m, n, ks, gran_k)
    recipe = (1, 1, gran_k)
    packed_a = (case.a[0], gen.pack_k_grouped_ue8m0(case.a[1], ks, gran_k))
    packed_b = (case.b[0], gen.pack_k_grouped_ue8m0(case.b[1], ks, gran_k))
    d = case.c.clone()
    dg.k_grouped_fp8_gemm_tn_contiguous(packed_a, packed_b, d, ks, case.grouped_layout, c=d, recipe=recipe)
    # MN-major operands in place (transposing fragment reads) where the library takes them: m > 128 and 16-byte aligned k-rows; else re-majored
    in_place = m > 128 and m % 16 == 0 and n % 16 == 0
    assert dg.last_config() == ('e8_quad_kg_' + ('mn_' if in_place else '') + ('g32_' if gran_k == 32 else '') + ('256x256' if m > 128 else '128x256')), dg.last_config()
    _check_groups(d, case, ks, gran_k, f'k-grouped ue8m0 gran {gran_k}')
    # c in another buffer: read, never written
    d2, c_before = torch.empty_like(case.c), case.c.clone()
    dg.k_grouped_fp8_gemm_tn_contiguous(packed_a, packed_b, d2, ks, case.grouped_layout, c=case.c, recipe=recipe)
    assert torch.equal(case.c, c_before) and torch.equal(d2, d)
    # FP32 scale tensors: granularity 32 always takes the cast; granularity 128 in the 'sm100' scaling-factor mode
    d3 = case.c.clone()
    _with_fp32_scales(gran_k, lambda: dg.k_grouped_fp8_gemm_tn_contiguous(case.a, case.b, d3, ks, case.grouped_layout, c=d3, recipe=recipe))
    assert dg.last_config().startswith('e8_quad_kg_') and torch.equal(d3, d)


@pytest.mark.parametrize('gran_k', [128, 32])
@pytest.mark.parametrize('use_psum,k_alignment', [(False, 128), (True, 128), (True, 160), (True, 32), (False, 96)])
def test_k_grouped_ue8m0_in_place_equals_re_majored(gran_k, use_psum, k_alignment):
    """The MN-major tensors read in place (e8_quad_kg_mn_*: LDS-DMA of [k][m] rows, ds_read_b64_tr_b8 fragments, natural row order) against the
    K-major kernel on re-majored copies: the same K-block order per accumulator -- the same bits; both against the oracle.  A row pitch that is
    not a multiple of 16 bytes (a view with an odd offset) must fall back to the re-majored path."""
    real_ks = [300, 0, 129, 512, 96, 1000] if use_psum else [k_alignment * q for q in (3, 0, 1, 5, 2, 7)]
    m, n = 272, 528
    gen.reset_seed(gran_k + k_alignment + 5)
    dg.set_mk_alignment_for_contiguous_layout(k_alignment)
    try:
        case = gen.generate_k_grouped_contiguous_ue8m0(len(real_ks), m, n, real_ks, gran_k, k_alignment, use_psum_layout=use_psum)
        a = (case.a[0], gen.pack_k_grouped_ue8m0(case.a[1], real_ks, gran_k))
        b = (case.b[0], gen.pack_k_grouped_ue8m0(case.b[1], real_ks, gran_k))
        d = case.c.clone()
        dg.k_grouped_fp8_gemm_tn_contiguous(a, b, d, case.ks, case.grouped_layout, c=d, recipe=(1, 1, gran_k), use_psum_layout=use_psum)
        assert dg.last_config() == ('e8_quad_kg_mn_g32_256x256' if gran_k == 32 else 'e8_quad_kg_mn_256x256'), dg.last_config()
        _check_groups(d, case, real_ks, gran_k, f'in place, gran {gran_k}, alignment {k_alignment}, psum {use_psum}')
        # the same operands behind a base that is off 16 bytes: re-majored, K-major kernel
        a_off = torch.empty((case.a[0].numel() + 1,), dtype=torch.uint8, device='cuda')[1:].view(torch.float8_e4m3fn).view(case.a[0].shape)
        a_off.copy_(case.a[0])
        d2 = case.c.clone()
        dg.k_grouped_fp8_gemm_tn_contiguous((a_off, a[1]), b, d2, case.ks, case.grouped_layout, c=d2, recipe=(1, 1, gran_k), use_psum_layout=use_psum)
        assert dg.last_config() == ('e8_quad_kg_g32_256x256' if gran_k == 32 else 'e8_quad_kg_256x256'), dg.last_config()
        assert torch.equal(d2, d)
    finally:
        dg.set_mk_alignment_for_contiguous_layout(128)


@pytest.mark.parametrize('gran_k,k_alignment', [(32, 32), (32, 128), (32, 160), (32, 224), (128, 128), (128, 160), (128, 224), (128, 32), (128, 192)])
@pytest.mark.parametrize('num_groups,m,n,real_ks', [(4, 256, 384, [300, 0, 129, 512]), (3, 512, 1024, [128, 1, 640]), (2, 304, 272, [1000, 77]),
                                                    (70, 128, 256, [(37 * i) % 300 for i in range(70)])])
def test_k_grouped_ue8m0_psum_layout(gran_k, k_alignment, num_groups, m, n, real_ks):
    """The psum form over the reference's SM100 sweep of (gran_k, K alignment) pairs (tests/generators.py:192-194): ranges read on the device
    with ``ks_cpu`` given or missing, groups of any real K, empty groups, partial last blocks whose neighbours in memory belong to the NEXT group
    (alignment 32): the layout's contract is zeros up to a group's aligned end, beyond it the next group starts -- those bytes must not reach
    this group's sums (every group is checked against the oracle on its own operands)."""
    gen.reset_seed(sum(real_ks) + m + k_alignment + gran_k)
    dg.set_mk_alignment_for_contiguous_layout(k_alignment)
    try:
        case = gen.generate_k_grouped_contiguous_ue8m0(num_groups, m, n, real_ks, gran_k, k_alignment, use_psum_layout=True)
        recipe = (1, 1, gran_k)
        results = []
        for ks_cpu in (case.ks, None, []):
            d = case.c.clone()
            _with_fp32_scales(gran_k, lambda: dg.k_grouped_fp8_gemm_tn_contiguous(case.a, case.b, d, ks_cpu, case.grouped_layout, c=d, recipe=recipe,
                                                                                   use_psum_layout=True))
            assert dg.last_config().startswith('e8_quad_kg_'), dg.last_config()
            results.append(d)
        for other in results[1:]:
            assert torch.equal(other, results[0])
        _check_groups(results[0], case, real_ks, gran_k, f'k-grouped ue8m0 psum gran {gran_k} alignment {k_alignment}')
        # packed words handed over by the caller (group rows from the REAL extents, as the layout step packs them)
        packed_a = (case.a[0], gen.pack_k_grouped_ue8m0(case.a[1], real_ks, gran_k))
        packed_b = (case.b[0], gen.pack_k_grouped_ue8m0(case.b[1], real_ks, gran_k))
        d2 = case.c.clone()
        dg.k_grouped_fp8_gemm_tn_contiguous(packed_a, packed_b, d2, None, case.grouped_layout, c=d2, recipe=recipe, use_psum_layout=True)
        assert torch.equal(d2, results[0])
    finally:
        dg.set_mk_alignment_for_contiguous_layout(128)




@pytest.mark.parametrize('gran_k', [128, 32])
@pytest.mark.parametrize('use_psum,k_alignment', [(False, 128), (True, 128), (True, 160), (False, 32)])
def test_k_grouped_sf_pack_matches_host_statement(gran_k, use_psum, k_alignment):
    """dg_pack_sf_k_grouped_ue8m0 bit for bit against the host-side statement of the reference's pack kernel (impls/smxx_layout.cuh:148-246): a new
    word row per group, zero bytes beyond a group's last block, empty groups own no row."""
    from deepgemm_amd.gemm import _k_grouped_packed_sf
    real_ks = [300, 0, 129, 512, 32, 640] if use_psum else [k_alignment * q for q in (3, 0, 1, 5, 2, 7)]
    m = 264
    gen.reset_seed(gran_k + k_alignment)
    case = gen.generate_k_grouped_contiguous_ue8m0(len(real_ks), m, 128, real_ks, gran_k, k_alignment, use_psum_layout=use_psum)
    want = gen.pack_k_grouped_ue8m0(case.a[1], real_ks, gran_k)
    for ks_cpu in ((case.ks, None) if use_psum else (case.ks,)):
        got = _k_grouped_packed_sf(case.a[1], m, ks_cpu, case.grouped_layout, len(real_ks), gran_k, k_alignment, use_psum)
        assert got.dtype == torch.int and got.size(0) >= want.size(0)
        assert torch.equal(got[:want.size(0)], want)


def test_k_grouped_ue8m0_full_size_group():
    """One group of the reference's sweep at full size (tests/generators.py:200-202: m 4096, n 7168, k ~ 4096 per group) beside a short one: the
    256-row tiles over many rounds; reference test expression only (the oracle takes minutes at this size)."""
    gen.reset_seed(7)
    ks = [4224, 128, 3968]
    for gran_k in (128, 32):
        case = gen.generate_k_grouped_contiguous_ue8m0(3, 4096, 7168, ks, gran_k)
        d = case.c.clone()
        _with_fp32_scales(gran_k, lambda: dg.k_grouped_fp8_gemm_tn_contiguous(case.a, case.b, d, ks, case.grouped_layout, c=d, recipe=(1, 1, gran_k)))
        assert dg.last_config().startswith('e8_quad_kg_')
        assert calc_diff(d, case.ref_d) < gen.FP8_MAX_DIFF
        # sampled rows of the short group against the oracle
        (a_g, sfa_g), (b_g, sfb_g) = case.a_groups[1], case.b_groups[1]
        rows = torch.arange(0, 4096, 509)
        want = torch.empty((rows.numel(), 7168), dtype=torch.float)
        oracle.fp8_gemm_nt(a_g[rows].cpu(), sfa_g[rows].cpu(), b_g.cpu(), sfb_g.cpu(), want, c=case.c[1][rows].cpu(), gran_n=1, gran_k=gran_k)
        assert_close_fp32(d[1][rows], want, f'full-size k-grouped ue8m0 gran {gran_k}, short group')


@pytest.mark.parametrize('gran_k', [128, 32])
@pytest.mark.parametrize('m,n,k,accumulate,out_dtype', [(320, 512, 8192, False, torch.bfloat16), (576, 1024, 14336, True, torch.float),
                                                       (1000, 264, 16384, False, torch.float), (512, 512, 16384, True, torch.bfloat16)])
def test_packed_dense_k_split_runs_as_k_groups(gran_k, m, n, k, accumulate, out_dtype):
    """Under-filled packed-scale dense problems with a long K loop (round 6): the K axis cut into pieces that run as the groups of one launch of the
    K-grouped hardware-scaled kernel (e8_quad_ks_*), FP32 partials summed in piece order by dg_sum_partials_kernel -- against the oracle, and
    against the same call as ONE launch (forced kernel: whole K loops; the split changes only where the FP32 partial sums are cut)."""
    from deepgemm_amd.utils.math import pack_ue8m0_to_int, per_token_cast_to_fp8
    gen.reset_seed(m + k + gran_k)
    case = gen.generate_normal(m, n, k, accumulate=accumulate, out_dtype=out_dtype, per_token_b=True, use_ue8m0=True)
    if gran_k == 32:
        qa, qb = per_token_cast_to_fp8(case.a_bf16, True, 32), per_token_cast_to_fp8(case.b_bf16, True, 32)
        a = (qa[0], dg.transform_sf_into_required_layout(pack_ue8m0_to_int(qa[1]), m, k, (1, 32)))
        b = (qb[0], dg.transform_sf_into_required_layout(pack_ue8m0_to_int(qb[1]), n, k, (1, 32)))
        fp32_a, fp32_b = qa, qb
    else:
        a, b = gen.packed_ue8m0_operand(*case.a), gen.packed_ue8m0_operand(*case.b)
        fp32_a, fp32_b = case.a, case.b
    c0 = case.d.clone() if accumulate else None
    kw = dict(c=case.d if accumulate else None, recipe=(1, 1, gran_k))
    dg.fp8_gemm_nt(a, b, case.d, **kw)
    assert dg.last_config() == ('e8_quad_ks_g32_256x256' if gran_k == 32 else 'e8_quad_ks_256x256'), dg.last_config()
    want = torch.empty((m, n), dtype=out_dtype)
    oracle.fp8_gemm_nt(fp32_a[0].cpu(), fp32_a[1].cpu(), fp32_b[0].cpu(), fp32_b[1].cpu(), want, c=c0.cpu() if accumulate else None, gran_n=1, gran_k=gran_k)
    if out_dtype == torch.float:
        assert_close_fp32(case.d, want, f'packed dense K split {m}x{n}x{k} gran {gran_k}')
    else:
        from gpu_helpers import assert_close_to_oracle
        assert_close_to_oracle(case.d, want, f'packed dense K split {m}x{n}x{k} gran {gran_k}', addend=c0)
    # one launch over the whole K loop (a forced kernel takes no K split)
    dg.set_forced_config('e8_quad_g32_128x256' if gran_k == 32 else 'e8_quad_128x256')
    try:
        d1 = c0.clone() if accumulate else torch.empty_like(case.d)
        dg.fp8_gemm_nt(a, b, d1, c=d1 if accumulate else None, recipe=(1, 1, gran_k))
        assert dg.last_config() in ('e8_quad_g32_128x256', 'e8_quad_128x256')
    finally:
        dg.set_forced_config('auto')
    assert calc_diff(case.d.float(), d1.float()) < 2e-6
    # a second call gives the same bits (piece order is fixed)
    d2 = c0.clone() if accumulate else torch.empty_like(case.d)
    dg.fp8_gemm_nt(a, b, d2, c=d2 if accumulate else None, recipe=(1, 1, gran_k))
    assert torch.equal(d2, case.d)


def test_k_grouped_ue8m0_and_packed_k_split_in_a_hip_graph():
    """Both round-6 paths inside a hipGraph: the K-grouped call with packed words (operands in place: one kernel, nothing allocated) and with FP32
    scales in the psum form (the layout step's pack kernel + the GEMM, ranges read on the device) replay to the eager bits; the packed-scale dense K
    split captured on a stream that owns its workspace keeps the pieces, on a fresh stream it runs as one launch (no allocation during capture)."""
    gen.reset_seed(41)
    real_ks = [384, 0, 512, 200]
    case = gen.generate_k_grouped_contiguous_ue8m0(4, 272, 528, real_ks, 32, 128, use_psum_layout=True)
    packed_a = (case.a[0], gen.pack_k_grouped_ue8m0(case.a[1], real_ks, 32))
    packed_b = (case.b[0], gen.pack_k_grouped_ue8m0(case.b[1], real_ks, 32))
    want = case.c.clone()
    dg.k_grouped_fp8_gemm_tn_contiguous(packed_a, packed_b, want, None, case.grouped_layout, c=want, recipe=(1, 1, 32), use_psum_layout=True)
    torch.cuda.synchronize()
    for a, b in ((packed_a, packed_b), (case.a, case.b)):
        d = case.c.clone()
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            dg.k_grouped_fp8_gemm_tn_contiguous(a, b, d, None, case.grouped_layout, c=d, recipe=(1, 1, 32), use_psum_layout=True)
        for _ in range(2):
            d.copy_(case.c)
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(d, want)
    # dense K split (over 256 rows: up to 256 the stream tile cut along K inside the kernel takes these shapes -- e8_stream_ks_*)
    m, n, k = 320, 512, 8192
    dense = gen.generate_normal(m, n, k, per_token_b=True, use_ue8m0=True)
    a, b = gen.packed_ue8m0_operand(*dense.a), gen.packed_ue8m0_operand(*dense.b)
    torch.cuda.synchronize()
    warm = torch.cuda.Stream()
    with torch.cuda.stream(warm):
        dg.fp8_gemm_nt(a, b, dense.d, recipe=(1, 1, 128))
        assert dg.last_config() == 'e8_quad_ks_256x256'
    warm.synchronize()
    eager = dense.d.clone()
    for stream, name in ((warm, 'e8_quad_ks_256x256'), (torch.cuda.Stream(), None)):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            dg.fp8_gemm_nt(a, b, dense.d, recipe=(1, 1, 128))
        assert name is None or dg.last_config() == name
        assert name is not None or not dg.last_config().startswith('e8_quad_ks_'), dg.last_config()
        dense.d.fill_(float('nan'))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(dense.d, eager) if name else calc_diff(dense.d.float(), eager.float()) < 2e-6
