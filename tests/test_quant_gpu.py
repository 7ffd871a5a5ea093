"""GPU tests of the fused per-token quantiser: bit-exact against the torch restatement of the reference's cast
(deepgemm_amd/utils/math.py, itself pinned to reference outputs by tests/test_oracle.py::test_quantisers_bit_exact_vs_reference)
and against the committed reference fixtures."""
import pytest
import torch

import deepgemm_amd as dg
from deepgemm_amd.utils import per_token_cast_to_fp8

pytestmark = pytest.mark.gpu


def _bits(t):
    return t.view(torch.uint8) if t.dtype == torch.float8_e4m3fn else t.view(torch.int32)


@pytest.mark.parametrize('m,n', [(1, 128), (5, 200), (130, 384), (64, 7168), (4096, 7168), (333, 1000), (7, 129), (3, 5)])
@pytest.mark.parametrize('use_ue8m0', [False, True])
def test_fused_per_token_cast_bit_exact(m, n, use_ue8m0):
    torch.manual_seed(m * 7 + n)
    x = torch.randn((m, n), device='cuda', dtype=torch.bfloat16) * 3
    x[0, :min(n, 7)] = 0
    if m > 2:
        x[2] = 0                                                   # all-zero blocks: the 1e-4 amax floor
        x[1] *= 1e4                                                # large magnitudes
    want_q, want_sf = per_token_cast_to_fp8(x.cpu(), use_ue8m0=use_ue8m0)
    for mn_major in (False, True):
        q, sf = dg.fused_per_token_cast_to_fp8(x, use_ue8m0=use_ue8m0, sf_mn_major=mn_major)
        assert q.shape == want_q.shape and sf.shape == want_sf.shape
        assert torch.equal(_bits(sf.cpu().contiguous()), _bits(want_sf))
        assert torch.equal(_bits(q.cpu()), _bits(want_q))
        if mn_major:
            assert sf.stride() == (1, dg.get_tma_aligned_size(m, 4))
            assert dg.get_mn_major_tma_aligned_tensor(sf).data_ptr() == sf.data_ptr()      # GEMM takes it as is


def test_fused_per_token_cast_strided_input_and_reference_fixtures(golden_quantisers):
    g = golden_quantisers
    for name in ('tok_5x200', 'tok_130x384', 'tok_64x512'):
        x = g.bf16(f'{name}_x').cuda()
        for ue in (False, True):
            q, sf = dg.fused_per_token_cast_to_fp8(x, use_ue8m0=ue)
            assert torch.equal(_bits(q.cpu()), _bits(g.fp8(f'{name}_ue{int(ue)}_q')))
            assert torch.equal(_bits(sf.cpu()), _bits(g.raw(f'{name}_ue{int(ue)}_sf')))
    wide = torch.randn((96, 1024), device='cuda', dtype=torch.bfloat16)
    view = wide[:, 128:640]                                        # row stride 1024, 512 columns
    q, sf = dg.fused_per_token_cast_to_fp8(view)
    want_q, want_sf = per_token_cast_to_fp8(view.cpu().contiguous(), use_ue8m0=False)
    assert torch.equal(_bits(q.cpu()), _bits(want_q)) and torch.equal(sf.cpu(), want_sf)


def test_fused_cast_feeds_gemm_without_transpose():
    from deepgemm_amd.utils import per_block_cast_to_fp8
    torch.manual_seed(0)
    m, n, k = 512, 768, 1024
    a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16)
    b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
    b_q = per_block_cast_to_fp8(b, use_ue8m0=False)
    d1, d2 = (torch.empty((m, n), device='cuda', dtype=torch.bfloat16) for _ in range(2))
    a_q, sfa = per_token_cast_to_fp8(a.cpu(), use_ue8m0=False)     # CPU torch: the arithmetic the fixtures pin
    dg.fp8_gemm_nt((a_q.cuda(), sfa.cuda()), b_q, d1)
    dg.fp8_gemm_nt(dg.fused_per_token_cast_to_fp8(a, sf_mn_major=True), b_q, d2)
    assert torch.equal(d1, d2)


@pytest.mark.parametrize('m,n', [(128, 128), (200, 300), (256, 384), (4096, 7168), (130, 1000), (5, 3)])
@pytest.mark.parametrize('use_ue8m0', [False, True])
def test_fused_per_block_cast_bit_exact(m, n, use_ue8m0):
    from deepgemm_amd.utils import per_block_cast_to_fp8
    torch.manual_seed(m + n)
    x = torch.randn((m, n), device='cuda', dtype=torch.bfloat16) * 0.5
    if m > 130:
        x[128:, :min(n, 128)] = 0                                   # an all-zero block: the 1e-4 floor
    want_q, want_sf = per_block_cast_to_fp8(x.cpu(), use_ue8m0=use_ue8m0)
    q, sf = dg.fused_per_block_cast_to_fp8(x, use_ue8m0=use_ue8m0)
    assert q.shape == want_q.shape and sf.shape == want_sf.shape
    assert torch.equal(_bits(sf.cpu()), _bits(want_sf))
    assert torch.equal(_bits(q.cpu()), _bits(want_q))


@pytest.mark.parametrize('k,n', [(128, 96), (256, 96), (1024, 4096), (384, 1001), (128, 5)])
@pytest.mark.parametrize('use_ue8m0', [False, True])
def test_fused_per_channel_cast_bit_exact(k, n, use_ue8m0):
    from deepgemm_amd.utils import per_channel_cast_to_fp8
    torch.manual_seed(k + n)
    x = torch.randn((k, n), device='cuda', dtype=torch.bfloat16)
    x[:, 0] = 0
    want_q, want_sf = per_channel_cast_to_fp8(x.cpu(), use_ue8m0=use_ue8m0)
    q, sf = dg.fused_per_channel_cast_to_fp8(x, use_ue8m0=use_ue8m0)
    assert q.shape == want_q.shape and sf.shape == want_sf.shape
    assert torch.equal(_bits(sf.cpu()), _bits(want_sf))
    assert torch.equal(_bits(q.cpu()), _bits(want_q))


def test_fused_block_casts_reference_fixtures(golden_quantisers):
    g = golden_quantisers
    for name in ('blk_200x300', 'blk_256x384'):
        x = g.bf16(f'{name}_x').cuda()
        for ue in (False, True):
            q, sf = dg.fused_per_block_cast_to_fp8(x, use_ue8m0=ue)
            assert torch.equal(_bits(q.cpu()), _bits(g.fp8(f'{name}_ue{int(ue)}_q')))
            assert torch.equal(_bits(sf.cpu()), _bits(g.raw(f'{name}_ue{int(ue)}_sf')))
    x = g.bf16('chn_256x96_x').cuda()
    q, sf = dg.fused_per_channel_cast_to_fp8(x)
    assert torch.equal(_bits(q.cpu()), _bits(g.fp8('chn_256x96_q'))) and torch.equal(_bits(sf.cpu()), _bits(g.raw('chn_256x96_sf')))
