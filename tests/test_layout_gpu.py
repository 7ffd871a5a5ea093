"""GPU tests of the SF layout kernel, modeled on the reference's tests/test_layout.py:45-80 (bit exact, exact strides)."""
import pytest
import torch

import deepgemm_amd as dg
import oracle
from deepgemm_amd.utils import ceil_div, get_tma_aligned_size, per_token_cast_to_fp8

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('mn', [1, 64, 4096, 4097, 8192])
@pytest.mark.parametrize('k', [128, 7168, 7296])
@pytest.mark.parametrize('num_groups', [1, 2, 4])
def test_transpose_kernel(mn, k, num_groups):
    torch.manual_seed(0)
    x = torch.randn((num_groups * mn, k), dtype=torch.bfloat16, device='cuda')
    _, sf = per_token_cast_to_fp8(x, use_ue8m0=False)
    sf = sf if num_groups == 1 else sf.view(num_groups, mn, -1)
    for pre_transposed in (False, True):
        src = sf.transpose(-1, -2).contiguous().transpose(-1, -2) if pre_transposed else sf
        out = dg.get_mn_major_tma_aligned_tensor(src)
        aligned, sf_k = get_tma_aligned_size(mn, 4), ceil_div(k, 128)
        if num_groups > 1:
            assert out.size(0) == num_groups and out.stride(0) == aligned * sf_k
        assert tuple(out.shape[-2:]) == (mn, sf_k) and tuple(out.stride()[-2:]) == (1, aligned)
        assert torch.equal(out, sf)
        want = oracle.transpose_sf(sf.cpu())
        assert torch.equal(out.cpu(), want) and out.stride() == want.stride()


@pytest.mark.parametrize('mn', [1, 33, 4096, 4097])
@pytest.mark.parametrize('k', [128, 896, 7168, 7296])
@pytest.mark.parametrize('num_groups', [1, 3])
def test_pack_ue8m0_kernel(mn, k, num_groups):
    """get_mn_major_tma_aligned_packed_ue8m0_tensor, modeled on the reference's tests/test_layout.py:45-60: bit exact
    against the oracle (itself pinned to the reference's torch statement by tests/test_oracle.py), exact strides."""
    torch.manual_seed(1)
    x = torch.randn((num_groups * mn, k), dtype=torch.bfloat16, device='cuda')
    _, sf = per_token_cast_to_fp8(x, use_ue8m0=True)
    sf = sf if num_groups == 1 else sf.view(num_groups, mn, -1)
    for pre_transposed in (False, True):
        src = sf.transpose(-1, -2).contiguous().transpose(-1, -2) if pre_transposed else sf
        out = dg.get_mn_major_tma_aligned_packed_ue8m0_tensor(src)
        want = oracle.pack_sf_ue8m0(src.cpu())
        assert out.dtype == torch.int and out.shape == want.shape and out.stride() == want.stride()
        assert torch.equal(out.cpu(), want)


def test_pack_ue8m0_kernel_golden(golden_sf_layout):
    g = golden_sf_layout
    for name in ('p33x7', 'p128x56', 'p3x20x9', 'p2x64x4'):
        sf = g.raw(f'{name}_sf').cuda()
        out = dg.get_mn_major_tma_aligned_packed_ue8m0_tensor(sf)
        assert out.stride() == tuple(int(x) for x in g.raw(f'{name}_strides'))
        assert torch.equal(out.cpu(), g.raw(f'{name}_packed'))


def test_packed_sf_feeds_the_scaled_mfma_gemm():
    """FP32 power-of-two scales -> pack kernel -> fp8_gemm_nt with int SFs == the FP32-scale path (same scale values)."""
    from deepgemm_amd.utils import per_block_cast_to_fp8
    torch.manual_seed(2)
    m, n, k = 384, 512, 1024
    a = torch.randn((m, k), device='cuda', dtype=torch.bfloat16)
    b = torch.randn((n, k), device='cuda', dtype=torch.bfloat16)
    a_q, sfa = per_token_cast_to_fp8(a, use_ue8m0=True)
    b_q, sfb = per_block_cast_to_fp8(b, use_ue8m0=True)
    sfb_rows = sfb.repeat_interleave(128, dim=0)[:n].contiguous()
    pa = dg.transform_sf_into_required_layout(dg.get_mn_major_tma_aligned_packed_ue8m0_tensor(sfa), m, k, (1, 1, 128), is_sfa=True)
    pb = dg.transform_sf_into_required_layout(dg.get_mn_major_tma_aligned_packed_ue8m0_tensor(sfb_rows), n, k, (1, 1, 128), is_sfa=False)
    d_hw = torch.empty((m, n), device='cuda', dtype=torch.bfloat16)
    dg.fp8_gemm_nt((a_q, pa), (b_q, pb), d_hw)
    assert dg.last_config().startswith('e8_')
    want = torch.empty((m, n), dtype=torch.bfloat16)
    oracle.fp8_gemm_nt(a_q.cpu(), sfa.cpu(), b_q.cpu(), sfb.cpu(), want)
    from gpu_helpers import assert_close_to_oracle
    assert_close_to_oracle(d_hw, want, 'packed scales from the pack kernel')
