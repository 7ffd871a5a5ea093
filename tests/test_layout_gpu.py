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
