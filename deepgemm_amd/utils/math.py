"""Host-side numerics helpers of the FP8 path: the quantisers that define what "blockwise scale" means.

Behavioural contract (reference: ``deep_gemm/utils/math.py:13-70``): an amax over each scaling block is
clamped at 1e-4 and divided by 448 (largest finite e4m3fn) to give the FP32 scaling factor, optionally
rounded UP to a power of two (UE8M0 semantics); the data is multiplied by the reciprocal and cast to
``torch.float8_e4m3fn``.  Ragged edges are zero-padded up to the block size before the amax.
"""
from typing import Tuple

import torch

from .._intmath import ceil_div, align     # noqa: F401  (re-exported, reference deep_gemm/utils/math.py:5-10)

_FP8_MAX = 448.0
_AMAX_FLOOR = 1e-4


def ceil_to_ue8m0(x: torch.Tensor) -> torch.Tensor:
    """Round |x| up to the next power of two (exponent-only FP32), clamped to the normal range."""
    raw = x.abs().float().view(torch.int32)
    exponent = (raw >> 23) & 0xFF
    has_fraction = (raw & 0x7FFFFF) != 0
    exponent = (exponent + has_fraction.to(torch.int32)).clamp(1, 254)
    return (exponent << 23).view(torch.float32)


def pack_ue8m0_to_int(x: torch.Tensor) -> torch.Tensor:
    """Four power-of-two FP32 scales -> one int32 of their biased exponents (little endian)."""
    assert x.dtype == torch.float32 and x.size(-1) % 4 == 0
    raw = x.view(torch.int32)
    assert bool((raw >= 0).all()) and bool(((raw & 0x7FFFFF) == 0).all()), 'scales must be positive powers of two'
    return (raw >> 23).to(torch.uint8).view(torch.int32)


def _scale_from_amax(amax: torch.Tensor, use_ue8m0: bool) -> torch.Tensor:
    sf = amax.clamp(_AMAX_FLOOR) / _FP8_MAX
    return ceil_to_ue8m0(sf) if use_ue8m0 else sf


def per_token_cast_to_fp8(x: torch.Tensor, use_ue8m0: bool, gran_k: int = 128,
                          use_packed_ue8m0: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """1 x gran_k blocks along the last dim.  Returns (x_fp8 [m, n], sf [m, ceil(n / gran_k)])."""
    assert x.dim() == 2
    rows, cols = x.shape
    cols_padded = align(cols, gran_k)
    padded = x.new_zeros((rows, cols_padded))
    padded[:, :cols] = x
    blocks = padded.view(rows, cols_padded // gran_k, gran_k)
    sf = _scale_from_amax(blocks.abs().float().amax(dim=2), use_ue8m0)
    quant = (blocks * (1.0 / sf.unsqueeze(2))).to(torch.float8_e4m3fn)
    quant = quant.view(rows, cols_padded)[:, :cols].contiguous()
    return quant, (pack_ue8m0_to_int(sf) if use_packed_ue8m0 else sf)


def per_channel_cast_to_fp8(x: torch.Tensor, use_ue8m0: bool, gran_k: int = 128) -> Tuple[torch.Tensor, torch.Tensor]:
    """gran_k x 1 blocks along the first dim.  Returns (x_fp8 [m, n], sf [m / gran_k, n])."""
    assert x.dim() == 2 and x.size(0) % gran_k == 0
    rows, cols = x.shape
    blocks = x.view(rows // gran_k, gran_k, cols)
    sf = _scale_from_amax(blocks.abs().float().amax(dim=1), use_ue8m0)
    quant = (blocks * (1.0 / sf.unsqueeze(1))).to(torch.float8_e4m3fn)
    return quant.view(rows, cols), sf


def per_block_cast_to_fp8(x: torch.Tensor, use_ue8m0: bool, gran_k: int = 128) -> Tuple[torch.Tensor, torch.Tensor]:
    """gran_k x gran_k blocks.  Returns (x_fp8 [m, n], sf [ceil(m / gran_k), ceil(n / gran_k)])."""
    assert x.dim() == 2
    rows, cols = x.shape
    padded = x.new_zeros((align(rows, gran_k), align(cols, gran_k)))
    padded[:rows, :cols] = x
    blocks = padded.view(padded.size(0) // gran_k, gran_k, padded.size(1) // gran_k, gran_k)
    sf = _scale_from_amax(blocks.abs().float().amax(dim=(1, 3), keepdim=True), use_ue8m0)
    quant = (blocks * (1.0 / sf)).to(torch.float8_e4m3fn).view_as(padded)[:rows, :cols].contiguous()
    return quant, sf.view(blocks.size(0), blocks.size(2))


def per_custom_dims_cast_to_fp8(x: torch.Tensor, dims: Tuple, use_ue8m0: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """One scale per index of ``dims`` (amax over all other dims)."""
    reduced = tuple(i for i in range(x.dim()) if i not in set(dims))
    sf = _scale_from_amax(x.abs().float().amax(dim=reduced, keepdim=True), use_ue8m0)
    return (x * (1.0 / sf)).to(torch.float8_e4m3fn), sf.squeeze()
