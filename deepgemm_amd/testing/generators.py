"""Synthetic input construction for the FP8 GEMM path, restating the reference's test generators
(``tests/generators.py:115-187`` shape sweeps, ``:301-408`` tensors) with an explicit ``device`` so that the same
inputs can be produced on the host (oracle / golden fixtures) and on the GPU.

Inputs are BF16 ``randn`` tensors; the kernel gets their FP8 quantisation (A: 1 x 128 per-token scales, B: 128 x 128
block scales, FP32, no UE8M0 rounding -- the reference's SM90 "1D2D" convention), the reference result is the FP32
matmul of the UNQUANTISED inputs cast to the output dtype.
"""
import random
from dataclasses import dataclass
from typing import Iterator, List, Optional, Tuple

import torch

from ..utils.math import align, ceil_div, per_block_cast_to_fp8, per_token_cast_to_fp8
from .. import runtime

FP8_MAX_DIFF = 1e-3         # reference gate: calc_diff vs the unquantised result (tests/generators.py:65-70)

# DeepSeek-V3 shape lists of the reference sweep (tests/generators.py:119-121, :159-160, :177-178)
DENSE_NK = [(2112, 7168), (576, 7168), (24576, 1536), (32768, 512), (7168, 16384), (4096, 7168), (7168, 2048)]
DENSE_M_FWD = [1, 128, 4096]
GROUPED_NK = [(6144, 7168), (7168, 3072), (4096, 4096), (4096, 2048)]
CONTIGUOUS_GROUPS = [(4, 8192), (8, 4096)]
MASKED_GROUPS = [(32, 192), (6, 1024), (32, 20), (6, 20)]
MASKED_MAX_M = 4096


def reset_seed(seed: int = 0) -> None:
    random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


def _with_major(data: torch.Tensor, sf: torch.Tensor, k_major: bool):
    """K-major keeps the row-major tensor; MN-major re-materialises it column-major behind the same logical shape."""
    return (data, sf) if k_major else (data.mT.contiguous().mT, sf)


def cast_a(x: torch.Tensor, k_major: bool = True, use_ue8m0: bool = False):
    return _with_major(*per_token_cast_to_fp8(x, use_ue8m0=use_ue8m0), k_major)


def cast_b(x: torch.Tensor, k_major: bool = True, per_token: bool = False, use_ue8m0: bool = False):
    quant = per_token_cast_to_fp8(x, use_ue8m0=use_ue8m0) if per_token else per_block_cast_to_fp8(x, use_ue8m0=use_ue8m0)
    return _with_major(*quant, k_major)


def grouped_cast(x: torch.Tensor, per_block: bool, k_major: bool = True, use_ue8m0: bool = False):
    groups, mn, k = x.shape
    data = torch.empty_like(x, dtype=torch.float8_e4m3fn)
    sf = torch.empty((groups, ceil_div(mn, 128) if per_block else mn, ceil_div(k, 128)), device=x.device, dtype=torch.float)
    for i in range(groups):
        data[i], sf[i] = per_block_cast_to_fp8(x[i], use_ue8m0=use_ue8m0) if per_block else per_token_cast_to_fp8(x[i], use_ue8m0=use_ue8m0)
    return (data, sf) if k_major else (data.mT.contiguous().mT, sf)


@dataclass
class DenseCase:
    a: tuple
    b: tuple
    c: Optional[torch.Tensor]
    d: torch.Tensor
    ref_d: torch.Tensor
    a_bf16: torch.Tensor
    b_bf16: torch.Tensor


def generate_normal(m: int, n: int, k: int, a_k_major: bool = True, b_k_major: bool = True, accumulate: bool = False,
                    out_dtype: torch.dtype = torch.bfloat16, per_token_b: bool = False, device: str = 'cuda',
                    use_ue8m0: bool = False) -> DenseCase:
    """tests/generators.py:301-324.  ``per_token_b`` selects the (1, 1, 128) recipe's per-column SFB; ``use_ue8m0`` rounds
    the scales up to powers of two (the reference's SM100 convention, deep_gemm/utils/math.py:13-16)."""
    a = torch.randn((m, k), device=device, dtype=torch.bfloat16)
    b = torch.randn((n, k), device=device, dtype=torch.bfloat16)
    d = torch.randn((m, n), device=device, dtype=out_dtype) * 32 if accumulate else \
        torch.empty((m, n), device=device, dtype=out_dtype)
    c = d if accumulate else None
    ref = (a.float() @ b.float().t() + (c if accumulate else 0)).to(out_dtype)
    return DenseCase(cast_a(a, a_k_major, use_ue8m0), cast_b(b, b_k_major, per_token_b, use_ue8m0), c, d, ref, a, b)


@dataclass
class ContiguousCase:
    m: int
    a: tuple
    b: tuple
    grouped_layout: torch.Tensor
    d: torch.Tensor
    ref_d: torch.Tensor
    actual_ms: List[int]
    aligned_ms: List[int]


def generate_m_grouped_contiguous(num_groups: int, expected_m_per_group: int, n: int, k: int, b_k_major: bool = True,
                                  use_psum_layout: bool = False, device: str = 'cuda',
                                  actual_ms: Optional[List[int]] = None, use_ue8m0: bool = False) -> ContiguousCase:
    """tests/generators.py:327-366: per-group M = int(expected * U(0.7, 1.3)) aligned up, padding rows zeroed / -1."""
    alignment = runtime.get_mk_alignment_for_contiguous_layout()
    if actual_ms is None:
        actual_ms = [int(expected_m_per_group * random.uniform(0.7, 1.3)) for _ in range(num_groups)]
    aligned_ms = [align(x, alignment) for x in actual_ms]
    m = sum(aligned_ms)
    a = torch.randn((m, k), device=device, dtype=torch.bfloat16)
    b = torch.randn((num_groups, n, k), device=device, dtype=torch.bfloat16)
    layout = torch.empty(num_groups if use_psum_layout else m, device=device, dtype=torch.int32)
    d = torch.empty((m, n), device=device, dtype=torch.bfloat16)
    ref = torch.randn((m, n), device=device, dtype=torch.bfloat16)
    start = 0
    for i, (actual, aligned) in enumerate(zip(actual_ms, aligned_ms)):
        if use_psum_layout:
            layout[i] = start + actual
        else:
            layout[start:start + actual] = i
            layout[start + actual:start + aligned] = -1
        a[start + actual:start + aligned] = 0
        ref[start:start + aligned] = (a[start:start + aligned].float() @ b[i].float().t()).to(torch.bfloat16)
        start += aligned
    return ContiguousCase(m, cast_a(a, use_ue8m0=use_ue8m0), grouped_cast(b, per_block=True, k_major=b_k_major, use_ue8m0=use_ue8m0),
                          layout, d, ref, actual_ms, aligned_ms)


@dataclass
class MaskedCase:
    a: tuple
    b: tuple
    masked_m: torch.Tensor
    d: torch.Tensor
    ref_d: torch.Tensor


def generate_m_grouped_masked(num_groups: int, max_m: int, expected_m_per_group: int, n: int, k: int,
                              device: str = 'cuda', masked_ms: Optional[List[int]] = None, use_ue8m0: bool = False) -> MaskedCase:
    """tests/generators.py:380-408."""
    a = torch.randn((num_groups, max_m, k), device=device, dtype=torch.bfloat16)
    b = torch.randn((num_groups, n, k), device=device, dtype=torch.bfloat16)
    d = torch.empty((num_groups, max_m, n), device=device, dtype=torch.bfloat16)
    ref = torch.einsum('gmk,gnk->gmn', a.float(), b.float()).to(torch.bfloat16)
    if masked_ms is None:
        masked_ms = [int(expected_m_per_group * random.uniform(0.7, 1.3)) for _ in range(num_groups)]
    assert max(masked_ms) <= max_m
    masked = torch.tensor(masked_ms, device=device, dtype=torch.int32)
    a_q = grouped_cast(a, per_block=False, use_ue8m0=use_ue8m0)
    if not use_ue8m0:           # (zero is not a power of two: the packed format keeps the scales of the masked-out rows)
        for j, rows in enumerate(masked_ms):
            a_q[1][j, rows:] = 0
    return MaskedCase(a_q, grouped_cast(b, per_block=True, use_ue8m0=use_ue8m0), masked, d, ref)


def packed_ue8m0_operand(data: torch.Tensor, sf: torch.Tensor, mn_rows: Optional[int] = None) -> tuple:
    """(fp8, FP32 power-of-two scales) -> (fp8, packed UE8M0 int32 words [.., mn, ceil(sf_k / 4)]), the reference's SM100 input
    format.  ``mn_rows``: the scales are per 128-row block and are broadcast to the ``mn_rows`` rows first (what the reference's
    transform does for recipe (1, 128, 128), csrc/apis/layout.hpp:52-56)."""
    from ..layout import get_mn_major_tma_aligned_packed_ue8m0_tensor
    if mn_rows is not None:
        sf = sf.repeat_interleave(128, dim=-2)[..., :mn_rows, :].contiguous()
    return data, get_mn_major_tma_aligned_packed_ue8m0_tensor(sf)


def enumerate_normal() -> Iterator[Tuple[int, int, int, bool, bool, bool, torch.dtype, bool]]:
    """(m, n, k, a_k_major, b_k_major, accumulate, out_dtype, per_token_b): forward shapes plus the backward dgrad /
    wgrad forms of tests/generators.py:115-154 (MN-major operands, FP32 accumulation with the (1,1,128) recipe)."""
    for m in DENSE_M_FWD:
        for n, k in DENSE_NK:
            yield m, n, k, True, True, False, torch.bfloat16, False
            yield m, n, k, True, True, True, torch.bfloat16, False
    for n, k in DENSE_NK:
        m = 4096
        yield m, k, n, True, False, False, torch.bfloat16, False         # dgrad
        yield n, m, k, False, False, True, torch.float, True             # wgrad, FP32 accumulate
        yield n, m, k, False, False, False, torch.bfloat16, False        # wgrad, BF16


def enumerate_m_grouped_contiguous() -> Iterator[Tuple[int, int, int, int, bool, bool]]:
    for use_psum in (True, False):
        for groups, expected in CONTIGUOUS_GROUPS:
            for n, k in GROUPED_NK:
                for b_k_major in (True, False):
                    yield groups, expected, n, k, b_k_major, use_psum


def enumerate_m_grouped_masked() -> Iterator[Tuple[int, int, int, int, int]]:
    for groups, expected in MASKED_GROUPS:
        for n, k in GROUPED_NK:
            yield groups, MASKED_MAX_M, expected, n, k


@dataclass
class KGroupedCase:
    a: tuple                    # operator-ready (fp8, sf) pair in the requested form
    b: tuple
    a_groups: List[tuple]       # per non-empty group: (A_g fp8 [m, k_g] K-major, sfa_g [m, k_g / 128]) for checkers
    b_groups: List[tuple]
    c: torch.Tensor
    d: torch.Tensor
    ref_d: torch.Tensor
    ks: List[int]
    grouped_layout: torch.Tensor


def build_psum_layout_from_ks(real_ks: List[int], k_alignment: int) -> List[int]:
    """tests/generators.py:480-487: per-group END offsets along K; every group starts at the previous end rounded up."""
    ends, prev_end = [], 0
    for k in real_ks:
        prev_end = align(prev_end, k_alignment) + k
        ends.append(prev_end)
    return ends


def generate_k_grouped_contiguous_psum(num_groups: int, m: int, n: int, real_ks: List[int], k_alignment: int = 128,
                                       device: str = 'cuda') -> KGroupedCase:
    """tests/generators.py:490-530 with FP32 scales and gran_k = 128: MN-major ``a [total_k, m]``, ``b [total_k, n]`` whose rows between
    a group's end and the next multiple of ``k_alignment`` are zeros, each group cast on its own padded copy (compact scale rows:
    ceil(k / 128) per non-empty group, counted from the group's own start -- k_grouped_per_channel_cast_to_fp8, :411-433);
    ``grouped_layout`` = the ends; ``ks`` = the aligned extents (what a caller may pass as ``ks_cpu``).  Any ``k_alignment`` that is a
    multiple of 32 (the reference's SM100 sweep: 32 / 128 / 160 / 192 / 224)."""
    assert len(real_ks) == num_groups and k_alignment % 32 == 0
    from ..utils.math import per_channel_cast_to_fp8
    ends = build_psum_layout_from_ks(real_ks, k_alignment)
    total_k = align(ends[-1] if ends else 0, k_alignment)
    a_q = torch.zeros((total_k, m), device=device, dtype=torch.float8_e4m3fn)
    b_q = torch.zeros((total_k, n), device=device, dtype=torch.float8_e4m3fn)
    c = torch.randn((num_groups, m, n), device=device, dtype=torch.float) * 32
    ref_d = torch.empty_like(c)
    a_groups, b_groups, sfa_rows, sfb_rows = [], [], [], []
    for g, (k, end) in enumerate(zip(real_ks, ends)):
        if k == 0:
            ref_d[g] = c[g]
            a_groups.append(None), b_groups.append(None)
            continue
        start, k_pad = end - k, align(k, 128)
        a_g = torch.zeros((k_pad, m), device=device, dtype=torch.bfloat16)
        b_g = torch.zeros((k_pad, n), device=device, dtype=torch.bfloat16)
        a_g[:k], b_g[:k] = torch.randn((k, m), device=device, dtype=torch.bfloat16), torch.randn((k, n), device=device, dtype=torch.bfloat16)
        ref_d[g] = c[g] + a_g.float().t() @ b_g.float()
        qa, sa = per_channel_cast_to_fp8(a_g, use_ue8m0=False)
        qb, sb = per_channel_cast_to_fp8(b_g, use_ue8m0=False)
        a_q[start:end], b_q[start:end] = qa[:k], qb[:k]
        sfa_rows.append(sa), sfb_rows.append(sb)
        a_groups.append((qa.t().contiguous(), sa.t().contiguous()))
        b_groups.append((qb.t().contiguous(), sb.t().contiguous()))
    if k_alignment == 128:
        # (one scale row per 128-row block of the operands, empty groups included: what the whole-block form indexes by k / 128)
        sfa = torch.ones((total_k // 128, m), device=device, dtype=torch.float)
        sfb = torch.ones((total_k // 128, n), device=device, dtype=torch.float)
        for (k, end), sa, sb in zip([(k, e) for k, e in zip(real_ks, ends) if k], sfa_rows, sfb_rows):
            start = end - k
            sfa[start // 128:start // 128 + sa.size(0)], sfb[start // 128:start // 128 + sb.size(0)] = sa, sb
    else:
        sfa = torch.cat(sfa_rows) if sfa_rows else torch.empty((0, m), device=device, dtype=torch.float)
        sfb = torch.cat(sfb_rows) if sfb_rows else torch.empty((0, n), device=device, dtype=torch.float)
    layout = torch.tensor(ends, device=device, dtype=torch.int32)
    return KGroupedCase((a_q, sfa), (b_q, sfb), a_groups, b_groups, c, c.clone(), ref_d, [align(k, k_alignment) for k in real_ks], layout)


def generate_k_grouped_contiguous_ue8m0(num_groups: int, m: int, n: int, real_ks: List[int], gran_k: int = 128, k_alignment: int = 128,
                                        use_psum_layout: bool = False, device: str = 'cuda') -> KGroupedCase:
    """The reference's SM100 K-grouped inputs (tests/generators.py:436-530 with use_ue8m0 = True): MN-major ``a [total_k, m]``, ``b [total_k, n]``;
    every group cast on its own copy padded to whole ``gran_k`` blocks (k_grouped_per_channel_cast_to_fp8, :411-433) with power-of-two FP32 scales
    ``[sum over groups of ceil(k_g / gran_k), mn]``, compact, in group order.  ``use_psum_layout``: ``grouped_layout`` holds the groups' ENDS, a group
    starts at the previous end rounded up to ``k_alignment`` and the rows in between hold zeros, ``real_ks`` may be anything; otherwise
    ``grouped_layout`` holds the extents, which must be multiples of ``k_alignment``.  ``a_groups`` / ``b_groups``: each group's K-major
    ``([mn, k_pad] fp8, [mn, k_pad / gran_k] scales)`` for the oracle; ``ks``: the aligned extents (``ks_cpu``)."""
    assert len(real_ks) == num_groups and k_alignment % 32 == 0 and gran_k in (32, 128)
    from ..utils.math import per_channel_cast_to_fp8
    if use_psum_layout:
        ends = build_psum_layout_from_ks(real_ks, k_alignment)
        total_k = align(ends[-1] if ends else 0, k_alignment)
    else:
        assert all(k % k_alignment == 0 for k in real_ks)
        ends, total = [], 0
        for k in real_ks:
            total += k
            ends.append(total)
        total_k = total
    a_q = torch.zeros((total_k, m), device=device, dtype=torch.float8_e4m3fn)
    b_q = torch.zeros((total_k, n), device=device, dtype=torch.float8_e4m3fn)
    c = torch.randn((num_groups, m, n), device=device, dtype=torch.float) * 32
    ref_d = torch.empty_like(c)
    a_groups, b_groups, sfa_rows, sfb_rows = [], [], [], []
    for g, (k, end) in enumerate(zip(real_ks, ends)):
        if k == 0:
            ref_d[g] = c[g]
            a_groups.append(None), b_groups.append(None)
            continue
        start, k_pad = end - k, align(k, gran_k)
        a_g = torch.zeros((k_pad, m), device=device, dtype=torch.bfloat16)
        b_g = torch.zeros((k_pad, n), device=device, dtype=torch.bfloat16)
        a_g[:k], b_g[:k] = torch.randn((k, m), device=device, dtype=torch.bfloat16), torch.randn((k, n), device=device, dtype=torch.bfloat16)
        ref_d[g] = c[g] + a_g.float().t() @ b_g.float()
        qa, sa = per_channel_cast_to_fp8(a_g, use_ue8m0=True, gran_k=gran_k)
        qb, sb = per_channel_cast_to_fp8(b_g, use_ue8m0=True, gran_k=gran_k)
        a_q[start:end], b_q[start:end] = qa[:k], qb[:k]
        sfa_rows.append(sa), sfb_rows.append(sb)
        a_groups.append((qa.t().contiguous(), sa.t().contiguous()))
        b_groups.append((qb.t().contiguous(), sb.t().contiguous()))
    sfa = torch.cat(sfa_rows) if sfa_rows else torch.empty((0, m), device=device, dtype=torch.float)
    sfb = torch.cat(sfb_rows) if sfb_rows else torch.empty((0, n), device=device, dtype=torch.float)
    layout = torch.tensor(ends if use_psum_layout else list(real_ks), device=device, dtype=torch.int32)
    return KGroupedCase((a_q, sfa), (b_q, sfb), a_groups, b_groups, c, c.clone(), ref_d, [align(k, k_alignment) for k in real_ks], layout)


def pack_k_grouped_ue8m0(sf: torch.Tensor, group_ks: List[int], gran_k: int) -> torch.Tensor:
    """FP32 power-of-two scales ``[sum of ceil(k_g / gran_k), mn]`` -> the packed int32 words ``[sum of ceil(k_g / (4 gran_k)), mn]`` of the
    reference's K-grouped layout (impls/smxx_layout.cuh:148-246): every group starts a new word row, byte j of its row r = its scale block
    4 r + j, zero beyond its last block.  Host-side statement for tests (the device path is dg_pack_sf_k_grouped_ue8m0)."""
    rows, start = [], 0
    for k in group_ks:
        blocks = -(-k // gran_k)
        exps = (sf[start:start + blocks].contiguous().view(torch.int32) >> 23) & 0xff
        padded = torch.zeros((-(-blocks // 4) * 4, sf.size(1)), dtype=torch.int32, device=sf.device)
        padded[:blocks] = exps
        q = padded.view(-1, 4, sf.size(1))
        rows.append(q[:, 0] | (q[:, 1] << 8) | (q[:, 2] << 16) | (q[:, 3] << 24))
        start += blocks
    return torch.cat(rows).contiguous() if rows else torch.empty((0, sf.size(1)), dtype=torch.int32, device=sf.device)


def generate_k_grouped_contiguous(num_groups: int, m: int, n: int, ks: List[int], k_major: bool,
                                  device: str = 'cuda') -> KGroupedCase:
    """tests/generators.py:436-477: ``a [sum_k, m]``, ``b [sum_k, n]`` BF16, per group ``ref_d[g] = c[g] + a_g^T @ b_g``;
    per-channel FP8 casts per group (128 x 1 blocks along K).  ``k_major`` selects the SM90 operand form (each group's
    ``[m, k_g]`` transposed block flattened one after another, scales as ``[m, sum_k / 128]`` views) instead of the
    MN-major ``[sum_k, m]`` form."""
    assert len(ks) == num_groups and all(k % 128 == 0 for k in ks)
    from ..utils.math import per_channel_cast_to_fp8
    sum_k = sum(ks)
    a = torch.randn((sum_k, m), device=device, dtype=torch.bfloat16)
    b = torch.randn((sum_k, n), device=device, dtype=torch.bfloat16)
    c = torch.randn((num_groups, m, n), device=device, dtype=torch.float) * 32
    ref_d = torch.empty_like(c)
    a_q = torch.empty((sum_k, m), device=device, dtype=torch.float8_e4m3fn)
    b_q = torch.empty((sum_k, n), device=device, dtype=torch.float8_e4m3fn)
    sfa = torch.empty((sum_k // 128, m), device=device, dtype=torch.float)
    sfb = torch.empty((sum_k // 128, n), device=device, dtype=torch.float)
    a_groups, b_groups, start = [], [], 0
    for g, k in enumerate(ks):
        end = start + k
        ref_d[g] = c[g] + a[start:end].float().t() @ b[start:end].float()
        if k > 0:
            a_q[start:end], sfa[start // 128:end // 128] = per_channel_cast_to_fp8(a[start:end], use_ue8m0=False)
            b_q[start:end], sfb[start // 128:end // 128] = per_channel_cast_to_fp8(b[start:end], use_ue8m0=False)
            a_groups.append((a_q[start:end].t().contiguous(), sfa[start // 128:end // 128].t().contiguous()))
            b_groups.append((b_q[start:end].t().contiguous(), sfb[start // 128:end // 128].t().contiguous()))
        else:
            a_groups.append(None), b_groups.append(None)
        start = end
    if k_major:
        flat_a = torch.cat([g[0].reshape(-1) for g in a_groups if g is not None]) if sum_k else a_q.reshape(-1)
        flat_b = torch.cat([g[0].reshape(-1) for g in b_groups if g is not None]) if sum_k else b_q.reshape(-1)
        a_op, b_op = (flat_a, sfa.t()), (flat_b, sfb.t())
    else:
        a_op, b_op = (a_q, sfa), (b_q, sfb)
    layout = torch.tensor(ks, device=device, dtype=torch.int32)
    return KGroupedCase(a_op, b_op, a_groups, b_groups, c, c.clone(), ref_d, list(ks), layout)
