"""Expert-MLP hand-off (the single-GPU half of the reference's Mega-MoE): GEMM1 -> SwiGLU -> per-token FP8 re-quantisation -> GEMM2
without the BF16 intermediate going through memory.

Reference: ``deep_gemm/mega/__init__.py`` (``transform_weights_for_mega_moe`` :131-151, ``fp8_fp4_mega_moe`` :155-173), kernel
``deep_gemm/include/deep_gemm/impls/sm100_fp8_fp4_mega_moe.cuh`` (the L1 -> L2 hand-off in GEMM1's epilogue), host driver
``csrc/apis/mega.hpp:30-159``.  What is here: the fused L1 operator (``m_grouped_fp8_gemm_nt_masked_swiglu``), the weight transform
this library's kernel wants, and ``fp8_mega_moe_local`` = fused L1 + masked L2 on the tokens already resident on this GPU.  What is
NOT here: the reference kernel's in-kernel dispatch / combine over NVLink symmetric memory -- on MI355X the exchange is two RCCL
all-to-alls around these operators (``deepgemm_amd/ep.py``); fusing it needs xGMI peer access inside the kernel and a multi-GPU node to
measure it on.
"""
from typing import Optional, Tuple

import torch

from ._lib import lib, check, current_stream_ptr, require_device
from .errors import host_assert
from .layout import get_mn_major_tma_aligned_tensor, get_tma_aligned_size, is_k_major
from .gemm import m_grouped_fp8_gemm_nt_masked

TensorPair = Tuple[torch.Tensor, torch.Tensor]


def _interleave_blocks(t: torch.Tensor, block: int) -> torch.Tensor:
    """[G, 2 H, ...] with the first H rows = gate, the last H = up  ->  [gate blk 0, up blk 0, gate blk 1, up blk 1, ...] (blocks of
    ``block`` rows)."""
    g, n = t.shape[0], t.shape[1]
    half = n // 2
    host_assert(n % 2 == 0 and half % block == 0, 'n % 2 == 0 and (n / 2) % block == 0')
    gate = t[:, :half].reshape(g, half // block, block, *t.shape[2:])
    up = t[:, half:].reshape(g, half // block, block, *t.shape[2:])
    return torch.stack([gate, up], dim=2).reshape(t.shape).contiguous()


def transform_weights_for_mega_moe(l1_weights: TensorPair, l2_weights: TensorPair, activation: str = 'swiglu') -> Tuple[TensorPair, TensorPair]:
    """The weight layout the fused kernel wants (reference: deep_gemm/mega/__init__.py:131-151, which interleaves gate / up rows at
    granularity 8 and re-orders the scale factors for its UTCCP copy).  Here: ``l1_weights = (W1 [G, 2 I, K] e4m3, SF [G, 2 I / 128,
    K / 128] FP32)`` with gate rows first; gate and up rows are interleaved in BLOCKS OF 64 -- one 64 x 128 output tile of the kernel
    then holds 64 gate columns and the 64 up columns of the same intermediate columns (csrc/fp8_gemm_moe.hpp says why 64) -- and the
    scale ROWS are interleaved one by one ([gate 0, up 0, gate 1, up 1, ...]): every weight row keeps the 128 x 128 scale block it was
    quantised in (no re-quantisation).  ``l2_weights`` pass through unchanged."""
    host_assert(activation == 'swiglu', "activation == 'swiglu'")
    w1, sf1 = l1_weights
    host_assert(w1.dim() == 3 and sf1.dim() == 3 and w1.dtype == torch.float8_e4m3fn and sf1.dtype == torch.float, 'l1 = (fp8 [G, 2I, K], float [G, 2I/128, K/128])')
    host_assert(w1.size(1) % 256 == 0 and sf1.size(1) * 128 == w1.size(1), 'n % 256 == 0 and sf.size(1) == n / 128')
    return (_interleave_blocks(w1.view(torch.uint8), 64).view(torch.float8_e4m3fn), _interleave_blocks(sf1, 1)), l2_weights


def empty_intermediate(num_groups: int, m_max: int, intermediate: int, device) -> TensorPair:
    """GEMM2's operand pair as the fused kernel writes it: ``(A2 [G, m_max, I] e4m3, SFA2 [G, m_max, I / 128] FP32 in the MN-major,
    16-byte aligned layout`` (strides ``(I / 128 * aligned_m, 1, aligned_m)``) that the GEMMs take zero-copy)."""
    aligned = get_tma_aligned_size(m_max, 4)
    q = torch.empty((num_groups, m_max, intermediate), dtype=torch.float8_e4m3fn, device=device)
    sf = torch.empty_strided((num_groups, m_max, intermediate // 128), (intermediate // 128 * aligned, 1, aligned), dtype=torch.float, device=device)
    return q, sf


_workspaces = {}


def _exchange_workspace(num_groups: int, m: int, n: int, device: torch.device) -> torch.Tensor:
    """The kernel's amax exchange slots: zeroed once, left zeroed by every launch (include/deepgemm_amd.h); one per (device, stream)."""
    need = int(lib.dg_swiglu_workspace_bytes(num_groups, m, n))
    key = (device.index, current_stream_ptr())
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.zeros(need, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def m_grouped_fp8_gemm_nt_masked_swiglu(a: TensorPair, b: TensorPair, out: TensorPair, masked_m: torch.Tensor, expected_m: int,
                                        activation_clamp: Optional[float] = None, use_ue8m0: bool = False) -> None:
    """``out = per_token_cast_to_fp8( swiglu( a @ b^T ) )`` per expert, rows ``< masked_m[g]`` only: ``a = (A [G, M, K], SFA)``,
    ``b`` = the transformed W1 pair ``([G, 2 I, K], [G, 2 I / 128, K / 128])`` of :func:`transform_weights_for_mega_moe`,
    ``out`` = :func:`empty_intermediate` ``(G, M, I)``.  Bit-identical to ``m_grouped_fp8_gemm_nt_masked`` -> BF16 -> SwiGLU (``silu(g) *
    u`` in FP32 on the BF16 values, optional clamp ``g <= c``, ``|u| <= c``, result rounded to BF16) -> ``per_token_cast_to_fp8``."""
    (a_data, a_sf), (b_data, b_sf), (q, q_sf) = a, b, out
    host_assert(is_k_major(a_data) and is_k_major(b_data), 'major_a == cute::UMMA::Major::K and major_b == cute::UMMA::Major::K')
    host_assert(a_data.dim() == 3 and b_data.dim() == 3 and q.dim() == 3, 'a.dim() == 3 and b.dim() == 3 and out.dim() == 3')
    host_assert(a_data.dtype == torch.float8_e4m3fn and b_data.dtype == torch.float8_e4m3fn and q.dtype == torch.float8_e4m3fn,
                'ab.scalar_type() == torch::kFloat8_e4m3fn')
    num_groups, m, k = (int(x) for x in a_data.shape)
    num_groups_, n, k_ = (int(x) for x in b_data.shape)
    host_assert(num_groups == num_groups_ == q.size(0) == masked_m.numel(), 'num_groups == num_groups_ and num_groups == num_groups__')
    host_assert(k == k_ and n % 256 == 0 and k % 128 == 0, 'k == k_ and n % 256 == 0 and k % 128 == 0')
    host_assert(tuple(q.shape) == (num_groups, m, n // 2) and q.stride(-1) == 1, 'out.shape == (G, m, n / 2)')
    host_assert(tuple(q_sf.shape) == (num_groups, m, n // 256) and q_sf.dtype == torch.float and q_sf.stride(-2) == 1,
                'out_sf.shape == (G, m, n / 256) in the MN-major layout')
    host_assert(a_sf.dtype == torch.float and b_sf.dtype == torch.float and tuple(a_sf.shape) == (num_groups, m, k // 128) and
                tuple(b_sf.shape) == (num_groups, n // 128, k // 128), 'FP32 scales: sfa [G, m, k / 128], sfb [G, n / 128, k / 128]')
    host_assert(masked_m.dtype == torch.int and masked_m.is_contiguous() and expected_m > 0, 'masked_m int32, expected_m > 0')
    sfa = get_mn_major_tma_aligned_tensor(a_sf)
    require_device(a_data, b_data, sfa, b_sf, q, q_sf, masked_m)
    ws = _exchange_workspace(num_groups, m, n, a_data.device)
    check(lib.dg_m_grouped_fp8_gemm_nt_masked_swiglu(
        a_data.data_ptr(), sfa.data_ptr(), b_data.data_ptr(), b_sf.data_ptr(), q.data_ptr(), q_sf.data_ptr(), masked_m.data_ptr(),
        num_groups, m, n, k, int(expected_m), a_data.stride(0), a_data.stride(1), b_data.stride(0), b_data.stride(1),
        sfa.stride(0), sfa.stride(2), b_sf.stride(0), b_sf.stride(1), b_sf.stride(2), q.stride(0), q.stride(1), q_sf.stride(0), q_sf.stride(2),
        float(activation_clamp) if activation_clamp is not None else 0.0, int(use_ue8m0), ws.data_ptr(), ws.numel(), current_stream_ptr()))


def fp8_mega_moe_local(x: TensorPair, l1_weights: TensorPair, l2_weights: TensorPair, y: torch.Tensor, masked_m: torch.Tensor,
                       expected_m: int, activation_clamp: Optional[float] = None,
                       intermediate: Optional[TensorPair] = None) -> TensorPair:
    """The expert MLP of the tokens resident on this GPU in the masked layout: ``y[g, :masked_m[g]] = W2_g . swiglu(W1_g . x[g])`` --
    fused GEMM1 (:func:`m_grouped_fp8_gemm_nt_masked_swiglu`) + ``m_grouped_fp8_gemm_nt_masked``.  ``l1_weights`` / ``l2_weights`` as
    returned by :func:`transform_weights_for_mega_moe`.  Returns the intermediate pair (reusable as the ``intermediate`` argument)."""
    num_groups, m, _ = x[0].shape
    inter = l1_weights[0].size(1) // 2
    if intermediate is None:
        intermediate = empty_intermediate(num_groups, m, inter, x[0].device)
    m_grouped_fp8_gemm_nt_masked_swiglu(x, l1_weights, intermediate, masked_m, expected_m, activation_clamp)
    m_grouped_fp8_gemm_nt_masked(intermediate, l2_weights, y, masked_m, expected_m)
    return intermediate
