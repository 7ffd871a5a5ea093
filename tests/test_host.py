"""CPU tests of the host layer: C-ABI surface, argument validation, trivial cases, knobs.  No kernel is launched."""
import ctypes
import os
import re

import pytest
import torch

import deepgemm_amd as dg
from deepgemm_amd import _lib
from deepgemm_amd.testing import generators as gen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, 'include', 'deepgemm_amd.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(dg_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    names = _header_functions()
    assert len(names) >= 10
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in names:
        assert hasattr(raw, name), f'{name} declared in include/deepgemm_amd.h but not exported'
        assert name in _lib.SIGNATURES, f'{name} has no ctypes signature'
    assert sorted(_lib.SIGNATURES) == names
    assert b'gfx950' in _lib.lib.dg_version()


def test_reference_operator_names_exist():
    import deep_gemm
    for name in ['fp8_gemm_nt', 'fp8_gemm_nn', 'fp8_gemm_tn', 'fp8_gemm_tt', 'fp8_fp4_gemm_nt', 'fp8_fp4_gemm_nn',
                 'fp8_fp4_gemm_tn', 'fp8_fp4_gemm_tt', 'm_grouped_fp8_gemm_nt_contiguous', 'm_grouped_fp8_gemm_nn_contiguous',
                 'm_grouped_fp8_gemm_nt_masked', 'm_grouped_fp8_fp4_gemm_nt_contiguous', 'm_grouped_fp8_fp4_gemm_nn_contiguous',
                 'm_grouped_fp8_fp4_gemm_nt_masked', 'fp8_m_grouped_gemm_nt_masked', 'transform_sf_into_required_layout',
                 'get_tma_aligned_size', 'get_mn_major_tma_aligned_tensor', 'set_num_sms', 'get_num_sms', 'set_tc_util',
                 'get_tc_util', 'set_pdl', 'get_pdl', 'set_ignore_compile_dims', 'set_block_size_multiple_of',
                 'set_mk_alignment_for_contiguous_layout', 'get_mk_alignment_for_contiguous_layout',
                 'get_theoretical_mk_alignment_for_contiguous_layout', 'get_m_alignment_for_contiguous_layout',
                 'per_token_cast_to_fp8', 'per_block_cast_to_fp8', 'per_channel_cast_to_fp8', 'ceil_to_ue8m0', 'align', 'ceil_div']:
        assert hasattr(deep_gemm, name), name
    from deep_gemm.testing import bench, bench_kineto, calc_diff, count_bytes, assert_bitwise_equal, get_arch_major  # noqa: F401
    from deep_gemm.utils import per_custom_dims_cast_to_fp8  # noqa: F401


def test_knobs_round_trip():
    assert dg.get_num_sms() == 256
    dg.set_num_sms(128)
    assert dg.get_num_sms() == 128
    dg.set_num_sms(0)
    assert dg.get_num_sms() == 256
    dg.set_tc_util(80), dg.set_pdl(True)
    assert dg.get_tc_util() == 80 and dg.get_pdl() is True
    dg.set_tc_util(100), dg.set_pdl(False)
    assert dg.get_mk_alignment_for_contiguous_layout() == 128 == dg.get_theoretical_mk_alignment_for_contiguous_layout(20)
    dg.set_mk_alignment_for_contiguous_layout(64)
    assert dg.get_m_alignment_for_contiguous_layout() == 64
    dg.set_mk_alignment_for_contiguous_layout(128)
    dg.set_block_size_multiple_of(2), dg.set_ignore_compile_dims(True), dg.set_ignore_compile_dims(False)
    with pytest.raises(RuntimeError, match='unknown kernel configuration'):
        dg.set_forced_config('bogus')
    dg.set_forced_config('generic_128x128'), dg.set_forced_config('auto')
    assert 'pipe_256x256' in dg.list_configs()
    # scaling-factor mode (the role of the reference's arch_major in csrc/apis/layout.hpp:22,40-58): FP32 scales stay FP32 by default
    assert dg.get_sf_cast_mode() == 'sm90'
    dg.set_sf_cast_mode('sm100')
    assert dg.get_sf_cast_mode() == 'sm100'
    with pytest.raises(ValueError, match='sm90'):
        dg.set_sf_cast_mode('sm80')
    assert dg.get_sf_cast_mode() == 'sm100'
    dg.set_sf_cast_mode('sm90')


def test_sf_cast_mode_on_the_host():
    """The cast branch's host logic without a device: which calls take it, what it asserts (csrc/apis/layout.hpp:40-54)."""
    from deepgemm_amd.gemm import _casts_to_ue8m0, _truncate_to_ue8m0
    from deepgemm_amd.layout import get_mn_major_tma_aligned_packed_ue8m0_tensor, transform_sf_into_required_layout
    f, i = torch.ones((4, 2)), torch.ones((4, 1), dtype=torch.int)
    assert not _casts_to_ue8m0(f, f, False)
    dg.set_sf_cast_mode('sm100')
    try:
        assert _casts_to_ue8m0(f, f, False) and not _casts_to_ue8m0(f, f, True) and not _casts_to_ue8m0(i, i, False)
        # the keyword and the default mode keep FP32 scales FP32 (no device needed: zero-copy / check-only branches)
        sfb = torch.ones((1, 2))
        assert transform_sf_into_required_layout(sfb, 128, 256, (128, 128), None, None, True) is sfb
        # the cast itself runs on the device only
        with pytest.raises(RuntimeError, match='must live on the GPU'):
            transform_sf_into_required_layout(sfb, 128, 256, (128, 128))
        with pytest.raises(RuntimeError, match='sf.size\\(-2\\) == ceil_div\\(mn, gran_mn\\)'):
            get_mn_major_tma_aligned_packed_ue8m0_tensor(torch.ones((3, 2)), None, _gran_mn=128, _mn=128)
    finally:
        dg.set_sf_cast_mode('sm90')
    assert transform_sf_into_required_layout(torch.ones((1, 2)), 128, 256, (128, 128)).dtype == torch.float
    x = torch.tensor([1.0, 1.5, 3.999, 0.3, 448.0 / 3])
    assert torch.equal(_truncate_to_ue8m0(x), torch.tensor([1.0, 1.0, 2.0, 0.25, 128.0]))


def test_c_abi_reports_errors_without_launching():
    lib = _lib.lib
    # empty problems return success before any pointer is looked at
    assert lib.dg_fp8_gemm_nt(None, None, None, None, None, 0, 128, 128, 128, 1, 128, 1, 1, 4, 1, 1, 128, 128, 0, 0, None) == 0
    rc = lib.dg_fp8_gemm_nt(1, 1, 1, 1, 1, 16, 16, 128, 128, 1, 128, 1, 1, 16, 1, 1, 7, 16, 0, 0, None)
    assert rc != 0 and b'sfb_gran_n' in lib.dg_last_error()
    rc = lib.dg_fp8_gemm_nt(1, 1, 1, 1, 1, 16, 16, 128, 128, 2, 128, 1, 1, 16, 1, 1, 128, 16, 0, 0, None)
    assert rc != 0 and b'a_stride_m == 1 || a_stride_k == 1' in lib.dg_last_error()
    rc = lib.dg_m_grouped_fp8_gemm_nt_masked(1, 1, 1, 1, 1, 1, 2, 64, 128, 128, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, None)
    assert rc != 0 and b'expected_m > 0' in lib.dg_last_error()
    with pytest.raises(RuntimeError, match='Assertion error'):
        _lib.check(rc)


def _case(m=32, n=64, k=256, **kw):
    torch.manual_seed(0)
    return gen.generate_normal(m, n, k, device='cpu', **kw)


def test_cpu_tensors_fail_loudly():
    c = _case()
    with pytest.raises(RuntimeError, match='no CPU path'):
        dg.fp8_gemm_nt(c.a, c.b, c.d)


def test_validation_messages():
    c = _case()
    with pytest.raises(RuntimeError, match=r'Assertion error \(gemm.py:\d+\): ab.scalar_type'):
        dg.fp8_gemm_nt((c.a[0].view(torch.uint8), c.a[1]), c.b, c.d)
    with pytest.raises(RuntimeError, match='m == m_ and n == n_ and k == k_'):
        dg.fp8_gemm_nt(c.a, c.b, torch.empty(32, 65, dtype=torch.bfloat16))
    with pytest.raises(RuntimeError, match='d.scalar_type'):
        dg.fp8_gemm_nt(c.a, c.b, torch.empty(32, 64, dtype=torch.float16))
    with pytest.raises(RuntimeError, match=r't.stride\(-1\) == 1'):
        dg.fp8_gemm_nt(c.a, c.b, torch.empty(64, 32, dtype=torch.bfloat16).t())
    with pytest.raises(RuntimeError, match='ceil_div'):
        dg.fp8_gemm_nt((c.a[0], c.a[1][:, :1]), c.b, c.d)                       # SFA with too few K blocks
    with pytest.raises(RuntimeError, match='recipe_a.has_value'):
        dg.fp8_gemm_nt(c.a, c.b, c.d, recipe=(1, 128, 128), recipe_a=(1, 128), recipe_b=(128, 128))
    with pytest.raises(RuntimeError, match=r'sfa.scalar_type\(\) == torch::kInt and sfb.scalar_type\(\) == torch::kInt'):
        dg.fp8_gemm_nt((c.a[0], c.a[1].to(torch.int)), c.b, c.d)          # packed UE8M0 scales come in pairs
    with pytest.raises(RuntimeError, match=r'd.scalar_type\(\) == c'):
        dg.fp8_gemm_nt(c.a, c.b, c.d, c=torch.zeros(32, 64))
    # grouped: A must be K-major, D must be BF16, layout length must match
    torch.manual_seed(0)
    g = gen.generate_m_grouped_contiguous(2, 0, 64, 256, device='cpu', actual_ms=[128, 128])
    with pytest.raises(RuntimeError, match='major_a'):
        dg.m_grouped_fp8_gemm_nt_contiguous((g.a[0].t().contiguous().t(), g.a[1]), g.b, g.d, g.grouped_layout)
    with pytest.raises(RuntimeError, match='m == m__'):
        dg.m_grouped_fp8_gemm_nt_contiguous(g.a, g.b, g.d, g.grouped_layout[:-1].contiguous())
    with pytest.raises(RuntimeError, match='kBFloat16'):
        dg.m_grouped_fp8_gemm_nt_contiguous(g.a, g.b, g.d.float(), g.grouped_layout)
    mk = gen.generate_m_grouped_masked(2, 64, 0, 64, 256, device='cpu', masked_ms=[3, 64])
    with pytest.raises(RuntimeError, match='expected_m > 0'):
        dg.m_grouped_fp8_gemm_nt_masked(mk.a, mk.b, mk.d, mk.masked_m, 0)
    with pytest.raises(RuntimeError, match='kInt'):
        dg.m_grouped_fp8_gemm_nt_masked(mk.a, mk.b, mk.d, mk.masked_m.long(), 8)


def test_trivial_cases_need_no_kernel():
    """csrc/apis/gemm.hpp:19-46: m == 0 or n == 0 -> no-op; k == 0 -> D = C or 0."""
    fp8 = torch.float8_e4m3fn
    d = torch.full((0, 64), 7.0, dtype=torch.bfloat16)
    dg.fp8_gemm_nt((torch.empty(0, 256, dtype=fp8), torch.empty(0, 2)), (torch.empty(64, 256, dtype=fp8), torch.empty(1, 2)), d)
    d = torch.full((4, 8), 7.0, dtype=torch.bfloat16)
    a0, b0 = (torch.empty(4, 0, dtype=fp8), torch.empty(4, 0)), (torch.empty(8, 0, dtype=fp8), torch.empty(1, 0))
    dg.fp8_gemm_nt(a0, b0, d)
    assert bool((d == 0).all())
    c = torch.arange(32, dtype=torch.float32).view(4, 8).to(torch.bfloat16)
    dg.fp8_gemm_nt(a0, b0, d, c=c)
    assert torch.equal(d, c)
    g = gen.generate_m_grouped_contiguous(2, 0, 64, 256, device='cpu', actual_ms=[0, 0])
    dg.m_grouped_fp8_gemm_nt_contiguous(g.a, g.b, g.d, g.grouped_layout)       # m == 0: no-op


def test_sf_layout_helpers_on_host():
    assert dg.get_tma_aligned_size(4097, 4) == 4100 and dg.get_tma_aligned_size(5, 2) == 8
    sf = torch.rand(4097, 7)
    out = dg.get_mn_major_tma_aligned_tensor(sf)
    assert torch.equal(out, sf) and out.stride() == (1, 4100)
    assert dg.get_mn_major_tma_aligned_tensor(out).data_ptr() == out.data_ptr()       # already in layout: zero copy
    grouped = torch.rand(3, 130, 5)
    out = dg.get_mn_major_tma_aligned_tensor(grouped)
    assert torch.equal(out, grouped) and out.stride() == (132 * 5, 1, 132)
    t = dg.transform_sf_into_required_layout(sf, 4097, 7 * 128, (1, 128, 128), is_sfa=True)
    assert t.stride() == (1, 4100)
    sfb = torch.rand(33, 7)
    assert dg.transform_sf_into_required_layout(sfb, 33 * 128, 7 * 128, (1, 128, 128), is_sfa=False) is sfb
    with pytest.raises(RuntimeError, match='SFB must be contiguous'):
        dg.transform_sf_into_required_layout(torch.rand(33, 14)[:, ::2], 33 * 128, 7 * 128, (128, 128))
    # packed UE8M0 words (INT, 1, 128): checked and brought to the MN-major layout (csrc/apis/layout.hpp:60-62)
    packed = dg.transform_sf_into_required_layout(torch.zeros(64, 2, dtype=torch.int), 64, 1024, (1, 128))
    assert packed.dtype == torch.int and packed.shape == (64, 2) and packed.stride() == (1, 64)
    with pytest.raises(RuntimeError, match='ceil_div'):
        dg.transform_sf_into_required_layout(torch.zeros(64, 3, dtype=torch.int), 64, 1024, (1, 128))
    # ... and at granularity 32 (round 6: the SM100 MX recipe, csrc/apis/layout.hpp:56-58 with gran_k == 32): a word per 128-K block
    packed32 = dg.transform_sf_into_required_layout(torch.zeros(64, 8, dtype=torch.int), 64, 1024, (1, 32))
    assert packed32.dtype == torch.int and packed32.shape == (64, 8) and packed32.stride() == (1, 64)
    with pytest.raises(RuntimeError, match='ceil_div'):
        dg.transform_sf_into_required_layout(torch.zeros(64, 2, dtype=torch.int), 64, 1024, (1, 32))
    with pytest.raises(RuntimeError, match='Unknown SF transformation'):      # FP32 scales of granularity 32 consumed as FP32: no such arithmetic
        dg.transform_sf_into_required_layout(torch.ones(64, 32), 64, 1024, (1, 32))
    with pytest.raises(RuntimeError, match='Unknown SF transformation'):
        dg.transform_sf_into_required_layout(torch.zeros(64, 4, dtype=torch.int), 64, 1024, (1, 64))


def test_generators_follow_reference_conventions():
    gen.reset_seed(0)
    g = gen.generate_m_grouped_contiguous(4, 200, 64, 256, device='cpu')
    assert g.m == sum(g.aligned_ms) and all(x % 128 == 0 for x in g.aligned_ms)
    start = 0
    for i, (actual, aligned) in enumerate(zip(g.actual_ms, g.aligned_ms)):
        assert bool((g.grouped_layout[start:start + actual] == i).all())
        assert bool((g.grouped_layout[start + actual:start + aligned] == -1).all())
        assert bool((g.a[0][start + actual:start + aligned].view(torch.uint8) == 0).all())
        start += aligned
    shapes = list(gen.enumerate_normal())
    assert (4096, 4096, 7168, True, True, False, torch.bfloat16, False) in shapes and len(shapes) == 3 * 7 * 2 + 7 * 3


def test_newer_operators_exist_and_validate_on_the_host():
    """K-grouped GEMM, skip_head_mid, packed-UE8M0 layout helper, fused casts: names (incl. the deep_gemm alias) and the
    argument checks that run before any device work (reference: csrc/apis/gemm.hpp:48-69,299-400, attention.hpp:19-73)."""
    import deep_gemm
    for name in ['k_grouped_fp8_gemm_nt_contiguous', 'k_grouped_fp8_gemm_tn_contiguous', 'fp8_gemm_nt_skip_head_mid',
                 'get_mn_major_tma_aligned_packed_ue8m0_tensor', 'fused_per_token_cast_to_fp8', 'fused_per_block_cast_to_fp8',
                 'fused_per_channel_cast_to_fp8']:
        assert hasattr(deep_gemm, name) and hasattr(dg, name), name
    kg = gen.generate_k_grouped_contiguous(2, 128, 128, [128, 256], True, device='cpu')
    with pytest.raises(RuntimeError, match='c.has_value'):
        dg.k_grouped_fp8_gemm_nt_contiguous(kg.a, kg.b, kg.d, kg.ks, kg.grouped_layout)
    with pytest.raises(RuntimeError, match='k % k_alignment == 0'):
        dg.k_grouped_fp8_gemm_nt_contiguous(kg.a, kg.b, kg.d, [100, 284], kg.grouped_layout, c=kg.c)
    with pytest.raises(RuntimeError, match='ks_cpu'):
        dg.k_grouped_fp8_gemm_nt_contiguous(kg.a, kg.b, kg.d, None, kg.grouped_layout, c=kg.c)
    with pytest.raises(RuntimeError, match='not use_psum_layout'):
        dg.k_grouped_fp8_gemm_nt_contiguous(kg.a, kg.b, kg.d, kg.ks, kg.grouped_layout, c=kg.c, use_psum_layout=True)
    with pytest.raises(RuntimeError, match='num_groups'):
        dg.k_grouped_fp8_gemm_nt_contiguous(kg.a, kg.b, kg.d, kg.ks, kg.grouped_layout[:1].contiguous(), c=kg.c)
    with pytest.raises(RuntimeError, match='sum_mk'):
        dg.k_grouped_fp8_gemm_nt_contiguous((kg.a[0][:-128], kg.a[1]), kg.b, kg.d, kg.ks, kg.grouped_layout, c=kg.c)
    with pytest.raises(RuntimeError, match='no CPU path'):                  # everything valid: stops at the device check
        dg.k_grouped_fp8_gemm_nt_contiguous(kg.a, kg.b, kg.d, kg.ks, kg.grouped_layout, c=kg.c)
    kt = gen.generate_k_grouped_contiguous(2, 128, 128, [128, 256], False, device='cpu')
    with pytest.raises(RuntimeError, match='gran_k == 32 or gran_k == 128'):
        dg.k_grouped_fp8_gemm_tn_contiguous(kt.a, kt.b, kt.d, kt.ks, kt.grouped_layout, c=kt.c, recipe=(1, 1, 64))
    # granularity 32 is the UE8M0 form (round 6): these scale tensors have one row per 128 K, not per 32 -- the layout step's own check
    with pytest.raises(RuntimeError, match='ref_sf_k == sf_k'):
        dg.k_grouped_fp8_gemm_tn_contiguous(kt.a, kt.b, kt.d, kt.ks, kt.grouped_layout, c=kt.c, recipe=(1, 1, 32))
    with pytest.raises(RuntimeError, match='packed_sf_k >= aligned_packed_sf_k'):       # packed words: too few rows for the groups
        dg.k_grouped_fp8_gemm_tn_contiguous((kt.a[0], torch.zeros((1, 128), dtype=torch.int)), (kt.b[0], torch.zeros((1, 128), dtype=torch.int)),
                                            kt.d, kt.ks, kt.grouped_layout, c=kt.c)
    with pytest.raises(RuntimeError, match='no CPU path'):                  # packed words, everything valid: stops at the device check
        dg.k_grouped_fp8_gemm_tn_contiguous((kt.a[0], torch.zeros((2, 128), dtype=torch.int)), (kt.b[0], torch.zeros((2, 128), dtype=torch.int)),
                                            kt.d, kt.ks, kt.grouped_layout, c=kt.c)
    with pytest.raises(RuntimeError, match='sum_k'):
        dg.k_grouped_fp8_gemm_tn_contiguous((kt.a[0][:256], kt.a[1]), kt.b, kt.d, kt.ks, kt.grouped_layout, c=kt.c)
    # skip_head_mid: the width of d must reserve the middle columns
    c = _case(32, 256, 256)
    with pytest.raises(RuntimeError, match=r'left \+ right'):
        dg.fp8_gemm_nt_skip_head_mid(c.a, c.b, torch.empty(32, 256, dtype=torch.bfloat16), (64, 32, 64))
    with pytest.raises(RuntimeError, match='no CPU path'):
        dg.fp8_gemm_nt_skip_head_mid(c.a, c.b, torch.empty(32, 256 + 2 * 32, dtype=torch.bfloat16), (64, 32, 64))
    # fused casts and the pack helper take BF16 / FP32 device tensors only
    with pytest.raises(RuntimeError, match='kBFloat16'):
        dg.fused_per_token_cast_to_fp8(torch.zeros(4, 128))
    with pytest.raises(RuntimeError, match='no CPU path'):
        dg.fused_per_block_cast_to_fp8(torch.zeros(4, 128, dtype=torch.bfloat16))
    with pytest.raises(RuntimeError, match='gran_k'):
        dg.fused_per_channel_cast_to_fp8(torch.zeros(100, 128, dtype=torch.bfloat16))
    with pytest.raises(RuntimeError, match='kFloat'):
        dg.get_mn_major_tma_aligned_packed_ue8m0_tensor(torch.zeros(4, 4, dtype=torch.int))
    with pytest.raises(RuntimeError, match='no CPU path'):
        dg.get_mn_major_tma_aligned_packed_ue8m0_tensor(torch.ones(4, 4))
    # the C ABI of the K-grouped entry: argument checks without a launch
    lib = _lib.lib
    ks = (ctypes.c_int32 * 2)(128, 100)
    rc = lib.dg_k_grouped_fp8_gemm_nt_contiguous(1, 1, 1, 1, 1, 128, 128, ctypes.cast(ks, ctypes.c_void_p), 2, 0, 0, 0, 1, 128, 1, 128, None)
    assert rc != 0 and b'% 128 == 0' in lib.dg_last_error()
    rc = lib.dg_k_grouped_fp8_gemm_nt_contiguous(1, 1, 1, 1, 1, 128, 128, ctypes.cast(ks, ctypes.c_void_p), 2, 7, 0, 0, 1, 128, 1, 128, None)
    assert rc != 0 and b'ab_layout' in lib.dg_last_error()
    assert lib.dg_k_grouped_fp8_gemm_nt_contiguous(None, None, None, None, None, 0, 128, None, 2, 0, 0, 0, 1, 1, 1, 1, None) == 0


def test_every_reference_keyword_by_name():
    """Drop-in boundary: every operator takes the reference's keyword names (csrc/apis/gemm.hpp:645-717, attention.hpp m.def)
    -- all of them passed BY NAME on host tensors; a call that gets through the argument checks stops at the device check."""
    c = _case(128, 256, 256)
    with pytest.raises(RuntimeError, match='no CPU path'):
        dg.fp8_gemm_nt(a=c.a, b=c.b, d=c.d, c=None, recipe=None, recipe_a=None, recipe_b=None, compiled_dims='nk', disable_ue8m0_cast=False)
    for name, dims in (('fp8_gemm_nn', 'nk'), ('fp8_gemm_tn', 'mn'), ('fp8_gemm_tt', 'mn')):
        a = c.a if name[-2] == 'n' else (c.a[0].T.contiguous(), c.a[1].T.contiguous())
        b = c.b if name[-1] == 't' else (c.b[0].T.contiguous(), c.b[1].T.contiguous())
        with pytest.raises(RuntimeError, match='no CPU path'):
            getattr(dg, name)(a=a, b=b, d=c.d, c=None, recipe=None, recipe_a=None, recipe_b=None, compiled_dims=dims, disable_ue8m0_cast=False)
    g = gen.generate_m_grouped_contiguous(2, 100, 128, 256, device='cpu')
    with pytest.raises(RuntimeError, match='no CPU path'):
        dg.m_grouped_fp8_gemm_nt_contiguous(a=g.a, b=g.b, d=g.d, grouped_layout=g.grouped_layout, recipe=None, recipe_a=None, recipe_b=None,
                                            compiled_dims='nk', disable_ue8m0_cast=False, use_psum_layout=False,
                                            ensure_zero_padding=True, expected_m_for_psum_layout=None)
    with pytest.raises(RuntimeError, match='no CPU path'):
        dg.m_grouped_fp8_gemm_nn_contiguous(a=g.a, b=(g.b[0].mT.contiguous(), g.b[1].mT.contiguous()), d=g.d, grouped_layout=g.grouped_layout,
                                            recipe=None, recipe_a=None, recipe_b=None, compiled_dims='nk', disable_ue8m0_cast=False,
                                            use_psum_layout=False, ensure_zero_padding=True)
    mk = gen.generate_m_grouped_masked(2, 64, 0, 128, 256, device='cpu', masked_ms=[3, 64])
    with pytest.raises(RuntimeError, match='no CPU path'):
        dg.m_grouped_fp8_gemm_nt_masked(a=mk.a, b=mk.b, d=mk.d, masked_m=mk.masked_m, expected_m=32, recipe=None, recipe_a=None,
                                        recipe_b=None, compiled_dims='nk', disable_ue8m0_cast=False)
    kg = gen.generate_k_grouped_contiguous(2, 128, 128, [128, 256], True, device='cpu')
    with pytest.raises(RuntimeError, match='no CPU path'):
        dg.k_grouped_fp8_gemm_nt_contiguous(a=kg.a, b=kg.b, d=kg.d, ks_cpu=kg.ks, grouped_layout=kg.grouped_layout, c=kg.c,
                                            recipe=(1, 1, 128), compiled_dims='mn', use_psum_layout=False)
    kt = gen.generate_k_grouped_contiguous(2, 128, 128, [128, 256], False, device='cpu')
    with pytest.raises(RuntimeError, match='no CPU path'):
        dg.k_grouped_fp8_gemm_tn_contiguous(a=kt.a, b=kt.b, d=kt.d, ks_cpu=kt.ks, grouped_layout=kt.grouped_layout, c=kt.c,
                                            recipe=(1, 1, 128), compiled_dims='mn', use_psum_layout=False)
    # ks_cpu missing: legal only together with the psum layout (csrc/apis/gemm.hpp:66-68) -- the K ranges then come from the device
    # tensor
    for missing in (None, []):
        with pytest.raises(RuntimeError, match=r'\): use_psum_layout'):
            dg.k_grouped_fp8_gemm_tn_contiguous(kt.a, kt.b, kt.d, missing, kt.grouped_layout, c=kt.c)
        with pytest.raises(RuntimeError, match='no CPU path'):
            dg.k_grouped_fp8_gemm_tn_contiguous(kt.a, kt.b, kt.d, missing, kt.grouped_layout, c=kt.c, use_psum_layout=True)
        # (round 6: any K alignment that is a multiple of 32 -- the reference's SM100 sweep 32 / 160 / 192 / 224 -- is taken by the psum form)
        dg.set_mk_alignment_for_contiguous_layout(64)
        try:
            with pytest.raises(RuntimeError, match='no CPU path'):
                dg.k_grouped_fp8_gemm_tn_contiguous(kt.a, kt.b, kt.d, missing, kt.grouped_layout, c=kt.c, use_psum_layout=True)
        finally:
            dg.set_mk_alignment_for_contiguous_layout(128)
    with pytest.raises(RuntimeError, match='sum_k == sum_k_'):     # host extents that disagree with the operands
        dg.k_grouped_fp8_gemm_tn_contiguous(kt.a, kt.b, kt.d, [128, 128], kt.grouped_layout, c=kt.c, use_psum_layout=True)
    with pytest.raises(RuntimeError, match='no CPU path'):
        dg.fp8_gemm_nt_skip_head_mid(a=c.a, b=c.b, d=torch.empty(128, 256 + 2 * 32, dtype=torch.bfloat16), head_splits=(64, 32, 64),
                                     recipe=None, compiled_dims='nk', disable_ue8m0_cast=False)
    # int (packed UE8M0) scale tensors: routed by dtype, default recipe (1, 1, 128) (csrc/utils/layout.hpp:64-77)
    pa, pb = (c.a[0], torch.zeros(128, 1, dtype=torch.int)), (c.b[0], torch.zeros(256, 1, dtype=torch.int))
    with pytest.raises(RuntimeError, match='no CPU path'):
        dg.fp8_gemm_nt(pa, pb, c.d)
    with pytest.raises(RuntimeError, match='recipe == .1, 1, gran_k.'):
        dg.fp8_gemm_nt(pa, pb, c.d, recipe=(1, 128, 128))
    with pytest.raises(RuntimeError, match='no CPU path'):
        dg.m_grouped_fp8_gemm_nt_contiguous((g.a[0], torch.zeros(g.m, 1, dtype=torch.int)), (g.b[0], torch.zeros(2, 128, 1, dtype=torch.int)),
                                            g.d, g.grouped_layout)
    with pytest.raises(RuntimeError, match='no CPU path'):
        dg.m_grouped_fp8_gemm_nt_masked((mk.a[0], torch.zeros(2, 64, 1, dtype=torch.int)), (mk.b[0], torch.zeros(2, 128, 1, dtype=torch.int)),
                                        mk.d, mk.masked_m, 32)
    with pytest.raises(RuntimeError, match='gran_n == 128'):           # per-column FP32 SFB is a dense-only recipe
        dg.m_grouped_fp8_gemm_nt_masked(mk.a, (mk.b[0], torch.ones(2, 128, 2)), mk.d, mk.masked_m, 32, recipe=(1, 1, 128))


def test_operand_majorness_decisions_on_the_host():
    """Which FP8 operands the operators hand over as they are: dg_operand_plan (include/deepgemm_amd.h), the one owner of the alignment
    and tile rules (amn_eligible / bmn_eligible / per_col_mn_eligible in dg_api.hip) -- decided from shapes, strides and pointers
    alone, so CPU tensors suffice; the re-majoring kernel itself is never reached here (REMAJOR_MIN_MACS keeps the operands in place)."""
    from deepgemm_amd import gemm
    a_bit, b_bit = 1, 2

    def fp8(rows, cols):
        return torch.empty((rows, cols), dtype=torch.uint8).view(torch.float8_e4m3fn)
    k_major = fp8(1024, 1024)
    mn_major = fp8(1024, 1024).t()                                                  # [1024, 1024] view, unit stride along m / n
    sfa = torch.empty((8, 1024), dtype=torch.float).t()                           # MN-major SFA

    def plan(a, b, m=1024, n=1024, k=1024, gran_n=128, sf=sfa, gemm_type=0, alignment=128):
        return gemm._operand_plan(gemm_type, a, b, sf, gran_n, m, n, k, alignment)
    assert plan(k_major, k_major) == 0
    assert plan(mn_major, k_major) == 0 and plan(mn_major, mn_major) == 0          # tn / tt: transpose reads of A (duo_amn / duo_abmn)
    assert plan(k_major, mn_major) == 0                                            # nn: duo_bmn
    # M <= 256 is not a 256-row-tile problem: nothing reads MN-major operands in place
    assert plan(mn_major[:256], k_major, m=256) == a_bit and plan(k_major[:256], mn_major, m=256) == b_bit
    # A's pitch not a multiple of 16 bytes: A is re-majored, B still read in place by the B_MN kernels
    odd_pitch = fp8(1024, 1032).t()[:1024]
    assert plan(odd_pitch, k_major) == a_bit and plan(odd_pitch, mn_major) == a_bit
    # N % 16 != 0 with an MN-major B: re-major B, keep the MN-major A
    assert plan(mn_major, mn_major[:1000], n=1000) == b_bit and plan(k_major, mn_major[:1000], n=1000) == b_bit
    # a partial last K block (K = 1040 = 16 x 65): the MN-major-A kernels have no tail stage, the B_MN ones do
    a_t, b_t = fp8(1040, 1024).t(), fp8(1040, 1024).t()
    sf_t = torch.empty((9, 1024)).t()
    assert plan(a_t, b_t, k=1040, sf=sf_t) == a_bit and plan(a_t, fp8(1024, 1040), k=1040, sf=sf_t) == a_bit
    assert plan(fp8(1024, 1040), b_t, k=1040, sf=sf_t) == 0
    assert plan(fp8(1024, 1016), fp8(1016, 1024).t(), k=1016) == b_bit                # K not in whole 16-byte chunks: generic kernel, K-major
    # few 256 x 256 tiles and a long K loop: A is re-majored so that the K split can use the idle CUs (a rule, not alignment)
    for (m, n, k), want in (((576, 4096, 7168), a_bit), ((2048, 4096, 7168), a_bit), ((2048, 7168, 2048), 0), ((576, 4096, 1024), 0)):
        assert plan(fp8(k, m).t(), fp8(n, k), m, n, k, sf=torch.empty((k // 128, m)).t()) == want, (m, n, k)
    # MN-major SFA is what the fast kernels read: a K-major SFA sends everything to the K-major generic path
    assert plan(mn_major, mn_major, sf=torch.empty((1024, 8))) == a_bit | b_bit
    # recipe (1, 1, 128): only BOTH operands MN-major are read in place (pipe_pc_mn)
    assert plan(mn_major, mn_major, gran_n=1) == 0 and plan(mn_major, k_major, gran_n=1) == a_bit and plan(k_major, mn_major, gran_n=1) == b_bit
    # contiguous layout (A K-major by contract): B in place for alignments the 128-row tiles divide
    b3 = torch.empty((4, 1024, 1024), dtype=torch.uint8).view(torch.float8_e4m3fn).transpose(1, 2)       # [4, 1024, 1024], unit stride along n
    a_c = fp8(4096, 1024)
    sf_c = torch.empty((8, 4096)).t()
    assert plan(a_c, b3, m=4096, sf=sf_c, gemm_type=1) == 0 and plan(a_c, b3, m=4096, sf=sf_c, gemm_type=2) == 0
    assert plan(a_c, b3, m=4096, sf=sf_c, gemm_type=1, alignment=64) == b_bit
    saved, gemm.REMAJOR_MIN_MACS = gemm.REMAJOR_MIN_MACS, 1 << 62
    try:
        a, b = gemm._dense_operands(mn_major, k_major, sfa, 128, 1024, 1024, 1024)
        assert a is mn_major and b is k_major
        a, b = gemm._dense_operands(k_major, mn_major, sfa, 128, 1024, 1024, 1024)
        assert a is k_major and b is mn_major
        a, b = gemm._dense_operands(odd_pitch, mn_major, sfa, 128, 1024, 1024, 1024)   # (would be re-majored above the size threshold)
        assert a is odd_pitch and b is mn_major
    finally:
        gemm.REMAJOR_MIN_MACS = saved


def test_split_k_workspace_is_only_requested_when_the_model_says_so():
    """deepgemm_amd/gemm.py:_dense_split_k_workspace asks the library (dg_dense_wants_workspace: split_k_pieces / split_k_pays / the
    stream-vs-split model / per_col_split_pieces in dg_api.hip): ordinary calls must neither create nor pass a buffer (creating one
    needs a device: not reached for these shapes)."""
    from deepgemm_amd import gemm
    from deepgemm_amd._lib import lib
    cpu = torch.device('cpu')
    for m, n, k in ((4096, 4096, 7168), (2048, 7168, 2048), (128, 4096, 7168), (64, 4096, 2048), (4096, 4096, 128), (4096, 7168, 2112)):
        assert gemm._dense_split_k_workspace(m, n, k, 128, cpu) is None, (m, n, k)
    # recipe (1, 1, 128): only under-filled launches with long K loops (wgrad of a narrow layer)
    for m, n, k in ((4096, 4096, 7168), (2112, 4096, 7168), (576, 4096, 1024), (64, 4096, 7168)):
        assert gemm._dense_split_k_workspace(m, n, k, 1, cpu) is None, (m, n, k)
    wants = lambda m, n, k, gran_n=128, a_mn=0, b_mn=0: lib.dg_dense_wants_workspace(m, n, k, a_mn, b_mn, gran_n)      # noqa: E731
    assert wants(4096, 512, 32768) == 1 and wants(4096, 512, 32768, b_mn=1) == 1          # 64 tiles of 128 x 256, 256 K blocks
    assert wants(1024, 1024, 16384) == 1 and wants(512, 4096, 7168) == 1
    # (end of round 6: 64 x 32 stream tiles that fill at most half the chip, K >= 4096, are cut along K inside the kernel: stream_ks_64x32)
    assert wants(128, 576, 7168) == 1 and wants(64, 4096, 7168) == 1 and wants(128, 576, 2048) == 0 and wants(128, 3072, 7168) == 0 and wants(128, 2112, 7168) == 1      # (last session: 128 x 2112 x 7168 runs the K-split 64 x 64 tile)
    assert wants(576, 4096, 7168, 1) == 1 and wants(576, 4096, 7168, 1, 1, 1) == 1        # 48 tiles of 256 x 256, 56 K blocks: K pieces as groups
    assert wants(4096, 512, 32768, 1) == 1
    assert wants(576, 4096, 7168, 1, 1, 0) == 0                                            # mixed majorness: the layout-agnostic kernel, no split


def test_packed_ue8m0_words_expand_to_exact_powers_of_two():
    from deepgemm_amd import gemm
    from deepgemm_amd.utils.math import pack_ue8m0_to_int
    k = 2112                                                                                       # 17 K blocks: a partial last word
    exps = torch.randint(100, 150, (37, 20), dtype=torch.int32)                                    # 5 whole words per row
    sf = (exps << 23).view(torch.float)
    packed = pack_ue8m0_to_int(sf)
    assert packed.dtype == torch.int and packed.shape == (37, 5)
    back = gemm._unpack_ue8m0(packed, k)
    assert back.shape == (37, 17) and torch.equal(back, sf[:, :17])


def test_grouped_operand_plan_of_packed_scales():
    """dg_ue8m0_grouped_operand_plan (round 5): MN-major weights [G][K][N] of a packed-scale contiguous call stay in place where the in-place
    kernel is eligible (no psum, n % 16 == 0, M alignment 128 or a multiple of 256) and a pass over every group's weights costs more than its
    slower K loop; 2 = re-major first.  Answers without a device."""
    from deepgemm_amd._lib import lib
    a, b = 1 << 20, 1 << 21

    def plan(groups, m, n, k, psum=0, alignment=128, b_sn=1, b_sk=None):
        return lib.dg_ue8m0_grouped_operand_plan(a, b, groups, m, n, k, k, n * k, b_sn, n if b_sk is None else b_sk, psum, alignment)
    assert plan(8, 4608, 4096, 7168) == 0                               # C4's shape in the nn form: 235 MB of weights would be re-majored
    assert plan(8, 4608, 4096, 7168, psum=1) == 2                       # the psum walk is written for tiles that divide the alignment
    assert plan(4, 32768, 4096, 7168) == 2                              # long groups: the pass is cheap next to the faster quad kernel
    assert plan(8, 4608, 4104, 7168) == 2 and plan(8, 4608, 4096, 7168, alignment=384) == 2
    assert plan(8, 4608, 4096, 7168, alignment=256) == 0 and plan(2, 256, 4096, 7168) == 2      # (decode-sized M: the stream tiles want K-major weights)
    assert plan(8, 4608, 4096, 7168, b_sn=7168, b_sk=1) == 0            # K-major weights: nothing to decide


def test_automatic_kernel_selection_is_pinned():
    """The tile-selection heuristics (dg_api.hip: select_config / select_e8_config, the analogue of get_best_config,
    csrc/jit_kernels/heuristics/common.hpp:14-52) on BASELINE's configurations and on the entries of the reference's sweeps that each
    rule exists for -- through dg_select_config, which launches nothing.  A changed choice must be a decision, not an accident."""
    from deepgemm_amd._lib import lib

    def pick(gemm_type, m, n, k, groups=1, expected_m=0, a_mn=0, b_mn=0, gran_n=128, alignment=0, workspace=1, packed=0):
        return lib.dg_select_config(gemm_type, m, n, k, groups, expected_m, a_mn, b_mn, gran_n, alignment, workspace, packed).decode()
    dense, contiguous, masked = 0, 1, 3
    # BASELINE configs[1..4]
    assert pick(dense, 4096, 4096, 7168) == 'duo_p_256x256'
    assert [pick(dense, 2048, 7168, 2048, a_mn=a, b_mn=b) for a, b in ((0, 0), (0, 1), (1, 1), (1, 0))] == \
        ['duo_p_256x256', 'duo_bmn_256x256', 'duo_abmn_256x256', 'duo_amn_256x256']
    assert pick(contiguous, 4608, 4096, 7168, groups=8, alignment=128) == 'duo_tab_256x256'             # group-relative 256-row tiles + K-split remainders
    assert pick(contiguous, 4608, 4096, 7168, groups=8, alignment=128, workspace=0) == 'duo_128x256'
    assert pick(masked, 64, 4096, 7168, groups=8, expected_m=48) == 'stream_nt2_64x128'                 # two workgroups per CU; 235 MB of weights: non-temporal stream
    # packed UE8M0 scales
    assert pick(dense, 4096, 4096, 7168, packed=1) == 'e8_quad_256x256'
    assert pick(masked, 64, 4096, 7168, groups=8, expected_m=48, packed=1) == 'e8_stream_nt2_64x128'
    assert pick(dense, 128, 4096, 7168, packed=1) == 'e8_stream_l8_64x32'      # (end of round 6: four loader waves, the FP32-scale rule)
    # round 6: decode batches with packed scales run the skinny weight-stream kernel with the scaled MFMA (the rule of the FP32-scale skinny kernels)
    assert pick(dense, 1, 4096, 7168, packed=1) == 'e8_skinny_16' and pick(dense, 24, 4096, 7168, packed=1) == 'e8_skinny_32' and pick(dense, 33, 4096, 7168, packed=1, workspace=0) == 'e8_stream_l8_64x32'
    # (end of round 6: the stream tiles cut along K inside the kernel with packed words too -- the FP32-scale rules)
    assert pick(dense, 33, 4096, 7168, packed=1) == 'e8_stream_ks_64x32' and pick(dense, 128, 576, 7168, packed=1) == 'e8_stream_ks_64x32' and pick(dense, 192, 4096, 7168, packed=1) == 'e8_stream_ks_64x128'
    # (17 .. 32 rows on narrow layers with K >= 7168 leave the skinny kernel for the K-split 64 x 32 tile; 33 .. 63 tiles of 64 x 128 from K = 10240)
    assert pick(dense, 24, 1536, 7168, packed=1) == 'e8_stream_ks_64x32' and pick(dense, 24, 1536, 7168, packed=1, workspace=0) == 'e8_skinny_32' and pick(dense, 32, 2112, 7168, packed=1) == 'e8_skinny_32'
    assert pick(dense, 16, 576, 7168, packed=1) == 'e8_skinny_16' and pick(dense, 192, 1536, 16384, packed=1) == 'e8_stream_ks_64x128' and pick(dense, 192, 2112, 7168, packed=1) == 'e8_stream_l8_64x32'
    assert pick(dense, 24, 1536, 7168) == 'stream_ks_64x32' and pick(dense, 24, 1536, 7168, workspace=0) == 'skinny_32ca' and pick(dense, 32, 2112, 7168) == 'skinny_32ca' and pick(dense, 24, 576, 4096) == 'skinny_32c'
    assert pick(dense, 192, 1536, 16384) == 'stream_ks_64x128' and pick(dense, 192, 2048, 16384) == 'stream_ks_64x128' and pick(dense, 200, 1024, 16384) == 'stream_ks_64x32'
    # (65 .. 128 rows on wide layers: from 96 tiles of 64 x 128 the K-split 64 x 128 tile -- FP32 scales up to K = 10240, where the 8-wave split takes over)
    assert pick(dense, 128, 6144, 7168) == 'stream_ks_64x128' and pick(dense, 128, 7168, 8192) == 'stream_ks_64x128' and pick(dense, 128, 4096, 10240) == 'stream_ks_64x128'
    assert pick(dense, 128, 7168, 16384) == 'duo_sk_128x256' and pick(dense, 128, 4096, 7168) == 'stream_l8_64x32' and pick(dense, 64, 7168, 16384) == 'stream_l8_64x32'
    assert pick(dense, 128, 7168, 16384, packed=1) == 'e8_stream_ks_64x128' and pick(dense, 96, 6144, 7168, packed=1) == 'e8_stream_ks_64x128' and pick(dense, 128, 4096, 16384, packed=1) == 'e8_stream_ks_64x128'
    assert pick(dense, 128, 4096, 10240, packed=1) == 'e8_stream_l8_64x32' and pick(dense, 64, 7168, 16384, packed=1) == 'e8_stream_l8_64x32'
    # (packed scales at 17 .. 32 rows: the skinny kernel from 32 K blocks, as with FP32 scales)
    assert pick(dense, 24, 4096, 4096, packed=1) == 'e8_skinny_32' and pick(dense, 24, 4096, 3968, packed=1) != 'e8_skinny_32' and pick(dense, 33, 4096, 4096, packed=1) == 'e8_stream_ks_64x32'
    # (33 .. 128 rows, K >= 7168: the K-split 64 x 64 tile where it gets three or more pieces -- 48 .. 85 tiles)
    assert pick(dense, 128, 2112, 7168) == 'stream_ks_64x64' and pick(dense, 128, 1536, 7168) == 'stream_ks_64x64' and pick(dense, 64, 4096, 7168) == 'stream_ks_64x64'
    assert pick(dense, 128, 3072, 7168) == 'stream_l8_64x32' and pick(dense, 128, 2112, 4096) == 'stream_l8_64x32' and pick(dense, 128, 2112, 7168, workspace=0) == 'stream_l8_64x32'
    assert pick(dense, 128, 5120, 7168) == 'stream_ks_64x128' and pick(dense, 128, 5120, 7168, packed=1) == 'e8_stream_ks_64x128'
    from deepgemm_amd._lib import lib as _l
    assert _l.dg_ue8m0_dense_wants_workspace(192, 2112, 7168) == 0 and _l.dg_ue8m0_dense_wants_workspace(320, 512, 8192) == 1      # (the two-launch split prices the stream tiles up to 256 rows)
    assert _l.dg_ue8m0_dense_wants_workspace(128, 576, 7168) == 1 and _l.dg_ue8m0_dense_wants_workspace(192, 4096, 7168) == 1 and _l.dg_ue8m0_dense_wants_workspace(128, 4096, 7168) == 0 and _l.dg_ue8m0_dense_wants_workspace(1, 576, 7168) == 0
    # packed scales with MN-major operands: read in place where that beats a re-majoring pass (e8_mn_pays); a K tail in the nn layout (the
    # packed-scale dgrad shapes) always stays in place (round 5)
    assert pick(dense, 2048, 7168, 2048, b_mn=1, packed=1) == 'e8_duo_bmn_256x256' and pick(dense, 4096, 4096, 7168, b_mn=1, packed=1) == 'e8_quad_256x256'
    assert pick(dense, 4096, 7168, 2112, b_mn=1, packed=1) == 'e8_duo_bmn_kt_256x256' and pick(dense, 4096, 7168, 2112, packed=1) == 'e8_quad_kt_128x256'
    # recipe (1, 1, 128): the per-column kernel; the narrow-layer wgrad entry of the reference sweep runs its K pieces as groups
    assert pick(dense, 4096, 4096, 7168, gran_n=1) == 'pipe_pc_256x256' and pick(dense, 4096, 4096, 7168, a_mn=1, b_mn=1, gran_n=1) == 'pipe_pc_mn_256x256'
    # (round 5: 192-row tiles where they compute fewer rows -- 576 = 3 x 192, 2112 = 11 x 192 -- or fewer rows per round: per_col_bm)
    assert pick(dense, 576, 4096, 7168, gran_n=1) == 'pipe_pc_ks_192x256' and pick(dense, 576, 4096, 7168, a_mn=1, b_mn=1, gran_n=1) == 'pipe_pc_mn_ks_256x256'
    assert pick(dense, 576, 4096, 7168, gran_n=1, workspace=0) == 'pipe_pc_192x256' and pick(dense, 2112, 4096, 7168, gran_n=1) == 'pipe_pc_192x256'
    assert pick(dense, 3264, 4096, 7168, gran_n=1) == 'pipe_pc_256x256' and pick(dense, 640, 4096, 2048, gran_n=1) == 'pipe_pc_192x256'
    assert pick(contiguous, 4608, 4096, 7168, groups=8, alignment=128, packed=1) == 'e8_quad_tab_256x256'       # (round 5: group-relative 256-row tiles + 128-row remainders)
    # recipe (1, 1, 128)
    assert pick(dense, 4096, 4096, 7168, gran_n=1) == 'pipe_pc_256x256'
    assert pick(dense, 4096, 4096, 7168, gran_n=1, a_mn=1, b_mn=1) == 'pipe_pc_mn_256x256'
    # the reference's dense sweep: small M, K tails, few tiles with long K loops, tile-count quantisation
    assert pick(dense, 1, 7168, 16384) == 'skinny_16ca' and pick(dense, 16, 8192, 2048) == 'skinny_16wc' and pick(dense, 1, 4096, 16384) == 'skinny_16ca' and pick(dense, 128, 4096, 7168) == 'stream_l8_64x32'
    # decode batches: the skinny weight-stream kernel for long K loops, the stream tiles for short ones / wide N
    assert pick(dense, 16, 4096, 7168) == 'skinny_16ca' and pick(dense, 17, 4096, 7168) == 'skinny_32ca' and pick(dense, 24, 4096, 4096) == 'skinny_32c' and pick(dense, 33, 4096, 7168) == 'stream_ks_64x64' and pick(dense, 33, 2112, 7168) == 'stream_ks_64x32' and pick(dense, 33, 4096, 7168, workspace=0) == 'stream_l8_64x32'
    # (end of round 6, cold weights: non-temporal weight stream from 16 MB per launch; dense m <= 256 on the stream tile up to one resident round of two per CU)
    assert pick(dense, 1, 24576, 1536) == 'stream_nt2_64x128' and pick(dense, 1, 32768, 512) == 'stream_nt2_64x128'
    assert pick(dense, 128, 32768, 512) == 'stream_nt2_64x128' and pick(dense, 128, 24576, 1536, packed=1) == 'e8_stream_nt2_64x128'
    assert pick(dense, 32, 7168, 16384) == 'stream_l8_64x32' and not pick(dense, 1, 4104, 7168).startswith('skinny_16')
    assert pick(dense, 128, 24576, 1536) == 'stream_nt2_64x128' and pick(dense, 128, 7168, 2048) == 'stream2_64x128' and pick(dense, 256, 32768, 512) == 'duo_128x256'
    assert pick(dense, 128, 7168, 16384) == 'duo_sk_128x256'                                            # K split beats one stream tile per CU
    # round 6: 129 .. 256 rows, 64 .. CUs / 2 tiles of 64 x 128, K >= 4096: the stream tile cut along K inside the kernel (profiles/r06_probe/stream_ks_mid_m_ab.log)
    assert pick(dense, 128, 576, 7168) == 'stream_ks_64x32' and pick(dense, 128, 576, 16384) == 'stream_ks_64x32' and pick(dense, 128, 576, 7168, workspace=0) == 'stream_l8_64x32'
    assert pick(dense, 192, 4096, 7168) == 'stream_ks_64x128' and pick(dense, 256, 4096, 7168) == 'stream_ks_64x128' and pick(dense, 256, 2112, 7168) == 'stream_ks_64x128'
    assert pick(dense, 192, 4096, 7168, workspace=0) == 'stream_l8_64x32' and pick(dense, 192, 2112, 7168) == 'stream_l8_64x32' and pick(dense, 256, 4096, 2048) == 'stream2_64x128'
    assert pick(dense, 192, 7168, 2048) == 'stream2_64x128' and pick(masked, 192, 4096, 7168, groups=1, expected_m=192) != 'stream_ks_64x128'
    assert pick(dense, 4096, 7168, 2112, b_mn=1) == 'duo_bmn_kt_256x256' and pick(dense, 4096, 7168, 2112) == 'duo_kt_256x256'
    assert pick(dense, 4096, 512, 32768, b_mn=1) == 'duo_sk_bmn_128x256' and pick(dense, 4096, 576, 7168) == 'duo_sk_128x256'
    assert pick(dense, 4096, 576, 7168, workspace=0) == 'duo_128x256'
    assert pick(dense, 4096, 2048, 7168, b_mn=1) == 'duo_bmn_128x256'                                   # 128 / 256 tiles: one round either way
    assert pick(dense, 4096, 2112, 7168) == 'duo_p_256x256'                                             # 144 tiles: one round of 256-row tiles
    assert pick(dense, 70, 136, 200) == 'generic_128x128'                                               # K not in whole 16-byte chunks
    # the reference's grouped sweeps
    assert pick(contiguous, 34048, 4096, 2048, groups=8, alignment=128) == 'duo_tab_256x256'            # many rounds: group-relative tiles too
    assert pick(contiguous, 34048, 4096, 2048, groups=8, alignment=128, workspace=0) == 'duo_p_256x256'  # no workspace: persistent two-pass walk
    assert pick(contiguous, 34048, 4096, 2048, groups=8, alignment=128, b_mn=1) == 'duo_bmn_256x256'
    assert pick(masked, 4096, 4096, 4096, groups=32, expected_m=192) == 'duo_p_256x256'
    assert pick(masked, 4096, 6144, 7168, groups=6, expected_m=20) == 'stream_nt2_64x128'               # 288 stream tiles: one resident round at two per CU (round 5)
    assert pick(masked, 64, 7168, 2048, groups=8, expected_m=48) == 'stream_nt2_64x128'                 # the expert MLP's GEMM2: 448 tiles, 117 MB
    assert pick(masked, 4096, 4096, 4096, groups=6, expected_m=20) == 'stream_nt2_64x128' and pick(masked, 4096, 4096, 2048, groups=6, expected_m=20) == 'stream_nt2_64x128'
    assert pick(masked, 4096, 6144, 7168, groups=32, expected_m=20) == 'stream_nt2_64x128'


def test_k_grouped_packed_ue8m0_scale_layout():
    """get_k_grouped_mn_major_tma_aligned_packed_ue8m0_tensor against the reference's own check (tests/test_layout.py:82-98: every group's
    rows equal the reference's torch statement of the packing, :20-42, applied to that group alone), incl. an empty group and both scale
    granularities; argument checks in the reference's order (csrc/jit_kernels/impls/smxx_layout.hpp:261-287)."""
    import random
    from deepgemm_amd.utils.math import align, ceil_div, per_channel_cast_to_fp8

    def torch_statement(x):                       # [mn, k] FP32 -> [mn, ceil(k / 4)] int32, byte j of word q = exponent of column 4 q + j
        mn, k = x.shape
        padded = torch.zeros((mn, align(k, 4)), dtype=torch.uint8)
        padded[:, :k] = (x.view(torch.int) >> 23).to(torch.uint8)
        return padded.view(torch.int)
    random.seed(0)
    for mn in (64, 260):
        for groups, avg_k in ((16, 2048), (3, 384), (128, 256)):
            for gran_k in (32, 128):
                ks = [align(int(random.uniform(0.7, 1.3) * avg_k), gran_k) for _ in range(groups)]
                ks[1] = 0
                x = torch.randn((sum(ks), mn), dtype=torch.bfloat16)
                _, sf = per_channel_cast_to_fp8(x, use_ue8m0=True, gran_k=gran_k)
                layout = torch.tensor(ks, dtype=torch.int)
                packed = dg.get_k_grouped_mn_major_tma_aligned_packed_ue8m0_tensor(sf, layout, ks, gran_k, gran_k)
                sf_ks, packed_ks = [k // gran_k for k in ks], [ceil_div(k, gran_k * 4) for k in ks]
                assert packed.shape == (sum(packed_ks), mn) and packed.dtype == torch.int and packed.is_contiguous()
                for got, part in zip(packed.split(packed_ks), sf.split(sf_ks)):
                    if part.size(0):
                        assert torch.equal(got, torch_statement(part.T.contiguous()).T), (mn, groups, gran_k)
    sf, layout = torch.ones((6, 8)), torch.tensor([256, 512], dtype=torch.int)
    assert dg.utils.layout.get_k_grouped_mn_major_tma_aligned_packed_ue8m0_tensor is dg.get_k_grouped_mn_major_tma_aligned_packed_ue8m0_tensor
    assert dg.get_k_grouped_mn_major_tma_aligned_packed_ue8m0_tensor(sf, layout, [256, 512], 128, 128).shape == (2, 8)
    for args, message in (((sf, layout, [256, 512], 64, 128), 'gran_k == 32 or gran_k == 128'),
                          ((sf, layout, [256, 512], 128, 48), 'k_alignment % 32 == 0'),
                          ((sf[:, :6].contiguous(), layout, [256, 512], 128, 128), 'mn % 4 == 0'),
                          ((sf, layout, [256], 128, 128), r'ks_cpu.value\(\).size\(\)\) == num_groups'),
                          ((sf, layout, None, 128, 128), 'use_psum_layout'),
                          ((sf, layout, [256, 384], 128, 128), 'ref_sf_k == sf_k')):
        with pytest.raises(RuntimeError, match=message):
            dg.get_k_grouped_mn_major_tma_aligned_packed_ue8m0_tensor(*args)
    with pytest.raises(RuntimeError, match='Unsupported architecture'):
        dg.get_k_grouped_mn_major_tma_aligned_packed_ue8m0_tensor(sf, layout, None, 128, 128, use_psum_layout=True)


def test_bench_workload_tables_are_consistent():
    """bench.py: every secondary workload is a selectable workload, the headline is not among them, and the argument parser accepts the
    driver's flags with defaults that finish in minutes (the timed numbers themselves are the GPU tests' and the driver's business)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('bench_module', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert set(bench.SECONDARY) <= set(bench.WORKLOADS) and 'dense' in bench.WORKLOADS and 'dense' not in bench.SECONDARY
    assert len(set(bench.SECONDARY)) == len(bench.SECONDARY) == 27 and bench.GRAPHED <= set(bench.WORKLOADS)
    assert bench.PEAK_FP8_TFLOPS == 5000.0



def test_mega_weight_transform_and_validation_on_the_host():
    """transform_weights_for_mega_moe is pure layout work (runs without a GPU); the fused operator validates before it needs a device."""
    import deepgemm_amd as dg
    groups, inter, k = 2, 256, 256
    w = torch.arange(groups * 2 * inter, dtype=torch.int32).remainder(251).to(torch.uint8).view(groups, 2 * inter, 1).expand(-1, -1, k).contiguous()
    sf = torch.arange(groups * 4 * 2, dtype=torch.float).view(groups, 4, 2)
    (w_t, sf_t), l2 = dg.transform_weights_for_mega_moe((w.view(torch.float8_e4m3fn), sf), ('l2', 'unchanged'))
    assert l2 == ('l2', 'unchanged')
    wt = w_t.view(torch.uint8)
    for blk in range(4):            # kernel tile blk: columns [0, 64) = gate rows 64 blk .., [64, 128) = the up rows of the same blk
        assert torch.equal(wt[:, 128 * blk:128 * blk + 64], w[:, 64 * blk:64 * blk + 64])
        assert torch.equal(wt[:, 128 * blk + 64:128 * blk + 128], w[:, inter + 64 * blk:inter + 64 * blk + 64])
    for blk in range(2):            # scale rows: [gate 0, up 0, gate 1, up 1]
        assert torch.equal(sf_t[:, 2 * blk], sf[:, blk]) and torch.equal(sf_t[:, 2 * blk + 1], sf[:, 2 + blk])
    q, q_sf = dg.empty_intermediate(3, 70, 512, 'cpu')
    assert q.shape == (3, 70, 512) and q_sf.shape == (3, 70, 4) and q_sf.stride() == (4 * 72, 1, 72)
    x = (torch.zeros((groups, 8, k), dtype=torch.float8_e4m3fn), torch.ones((groups, 8, 2)))
    masked = torch.zeros(groups, dtype=torch.int)
    with pytest.raises(RuntimeError, match='Assertion error'):        # out must be [G, m, n / 2]
        dg.m_grouped_fp8_gemm_nt_masked_swiglu(x, (w_t, sf_t), dg.empty_intermediate(groups, 8, 2 * inter, 'cpu'), masked, 1)
    with pytest.raises(RuntimeError, match='Assertion error'):        # expected_m > 0
        dg.m_grouped_fp8_gemm_nt_masked_swiglu(x, (w_t, sf_t), dg.empty_intermediate(groups, 8, inter, 'cpu'), masked, 0)
    with pytest.raises(RuntimeError, match='no CPU path'):   # valid arguments: fails loudly for want of a GPU, no CPU path
        dg.m_grouped_fp8_gemm_nt_masked_swiglu(x, (w_t, sf_t), dg.empty_intermediate(groups, 8, inter, 'cpu'), masked, 1)


def test_reciprocal_tile_mapping_division_is_exact():
    """fp8_gemm_kernels.hpp `div_small`: the tile mapping divides through ONE float reciprocal and a one-step correction instead of the
    integer-division sequence (round 4).  The same arithmetic in numpy float32 -- with the reciprocal pushed one ulp either way, which is
    more than `v_rcp_f32` may be off -- equals a // b for every (a, b) below 2^22, the range the kernels use it in."""
    import numpy as np
    rng = np.random.default_rng(7)
    a = np.concatenate([rng.integers(0, 1 << 22, 200000), np.arange(0, 4096), np.full(64, (1 << 22) - 1)]).astype(np.int64)
    b = np.concatenate([rng.integers(1, 1 << 22, 100000), rng.integers(1, 64, 100000), np.arange(1, 4097), np.arange(1, 65)]).astype(np.int64)
    # edge cases: exact multiples and one below them
    mult = (rng.integers(1, 1 << 11, 50000) * rng.integers(1, 1 << 11, 50000)).astype(np.int64)
    div = rng.integers(1, 1 << 11, 50000).astype(np.int64)
    a = np.concatenate([a, (mult // div) * div, np.maximum((mult // div) * div - 1, 0)])
    b = np.concatenate([b, div, div])
    for ulps in (-1, 0, 1):
        rcp = (np.float32(1.0) / b.astype(np.float32)).astype(np.float32)
        rcp = np.nextafter(rcp, np.float32(np.inf if ulps > 0 else -np.inf)) if ulps else rcp
        q = ((a.astype(np.float32) + np.float32(0.5)) * rcp).astype(np.float32).astype(np.int64)      # v_cvt_i32_f32 truncates
        rem = a - q * b
        q = q + (rem >= b) - (rem < 0)
        assert np.array_equal(q, a // b), f'reciprocal off by {ulps} ulp'
