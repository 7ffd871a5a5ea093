"""Fused producer-side quantisers (HIP): what a caller runs immediately before the FP8 GEMM.

The reference leaves the casts to the caller (README.md:72) and ships them as multi-pass torch expressions
(deep_gemm/utils/math.py:26-61, restated in ``deepgemm_amd/utils/math.py``); these are the single-pass HIP forms with
the same arithmetic, bit for bit.
"""
from typing import Tuple

import torch

from ._lib import lib, check, current_stream_ptr, require_device
from ._intmath import ceil_div
from .errors import host_assert
from .layout import get_tma_aligned_size


def fused_per_token_cast_to_fp8(x: torch.Tensor, use_ue8m0: bool = False,
                                sf_mn_major: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """BF16 ``x [m, n]`` -> ``(x_fp8 [m, n], sf [m, ceil(n / 128)] FP32)``, one pass over HBM.

    Same values as ``per_token_cast_to_fp8(x, use_ue8m0)`` (deep_gemm/utils/math.py:26-38).  With ``sf_mn_major`` the
    scale tensor comes back in the GEMM's SFA layout (strides ``(1, align(m, 4))``), which ``fp8_gemm_nt`` then takes
    without its transpose launch.
    """
    host_assert(x.dim() == 2, 'x.dim() == 2')
    host_assert(x.dtype == torch.bfloat16, 'x.scalar_type() == torch::kBFloat16')
    host_assert(x.stride(1) == 1, 'x.stride(-1) == 1')
    require_device(x)
    m, n = x.shape
    sf_k = ceil_div(n, 128)
    q = torch.empty((m, n), dtype=torch.float8_e4m3fn, device=x.device)
    if sf_mn_major:
        aligned = get_tma_aligned_size(m, 4)
        sf = torch.empty_strided((m, sf_k), (1, aligned), dtype=torch.float, device=x.device)
    else:
        sf = torch.empty((m, sf_k), dtype=torch.float, device=x.device)
    check(lib.dg_per_token_cast_to_fp8(x.data_ptr(), q.data_ptr(), sf.data_ptr(), m, n, x.stride(0), q.stride(0),
                                       sf.stride(0), sf.stride(1), int(use_ue8m0), current_stream_ptr()))
    return q, sf


def _block_cast(x: torch.Tensor, per_channel: bool, use_ue8m0: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    host_assert(x.dim() == 2, 'x.dim() == 2')
    host_assert(x.dtype == torch.bfloat16, 'x.scalar_type() == torch::kBFloat16')
    host_assert(x.stride(1) == 1, 'x.stride(-1) == 1')
    require_device(x)
    rows, cols = x.shape
    q = torch.empty((rows, cols), dtype=torch.float8_e4m3fn, device=x.device)
    sf = torch.empty((ceil_div(rows, 128), cols if per_channel else ceil_div(cols, 128)), dtype=torch.float, device=x.device)
    check(lib.dg_block_cast_to_fp8(x.data_ptr(), q.data_ptr(), sf.data_ptr(), rows, cols, x.stride(0), q.stride(0),
                                   sf.stride(0), sf.stride(1), int(per_channel), int(use_ue8m0), current_stream_ptr()))
    return q, sf


def fused_per_block_cast_to_fp8(x: torch.Tensor, use_ue8m0: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """BF16 ``x [m, n]`` -> ``(x_fp8, sf [ceil(m / 128), ceil(n / 128)])``: 128 x 128 block scales (the weight side), same
    values as ``per_block_cast_to_fp8`` (deep_gemm/utils/math.py:51-61), one pass."""
    return _block_cast(x, False, use_ue8m0)


def fused_per_channel_cast_to_fp8(x: torch.Tensor, use_ue8m0: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """BF16 ``x [k, n]`` (``k % 128 == 0``) -> ``(x_fp8, sf [k / 128, n])``: one scale per column per 128 rows (operands of
    the K-grouped GEMM), same values as ``per_channel_cast_to_fp8`` (deep_gemm/utils/math.py:41-48), one pass."""
    host_assert(x.size(0) % 128 == 0, 'x.size(0) % gran_k == 0')
    return _block_cast(x, True, use_ue8m0)
