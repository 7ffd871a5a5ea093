"""Scaling-factor layout handling of the FP8 path (host side).

Mirrors ``csrc/utils/layout.hpp:13-117``, ``csrc/apis/layout.hpp:14-90`` and ``csrc/jit_kernels/impls/smxx_layout.hpp:
120-153`` of the reference: SFA (1 x 128 granularity) is handed to the kernel MN-major with the MN extent padded to a
multiple of 4 floats ("TMA aligned" in the reference; on CDNA4 it makes the per-lane SFA reads of a wave contiguous),
SFB (128 x 128) is only checked.  Packed UE8M0 scales (int32 words of four exponent bytes, the reference's SM100 input
format) are brought to the same MN-major layout (``csrc/apis/layout.hpp:58-60``) and feed the hardware-scaled MFMA kernels;
``get_mn_major_tma_aligned_packed_ue8m0_tensor`` packs FP32 power-of-two scales on the device.
"""
from typing import Optional, Tuple, Union

import torch

from ._lib import lib, check, current_stream_ptr, require_device
from .errors import host_assert
from ._intmath import ceil_div, align

_TMA_ALIGNMENT_BYTES = 16


def get_tma_aligned_size(x: int, element_size: int) -> int:
    host_assert(_TMA_ALIGNMENT_BYTES % element_size == 0, 'kNumTMAAlignmentBytes % element_size == 0')
    return align(x, _TMA_ALIGNMENT_BYTES // element_size)


def major_check(t: torch.Tensor) -> None:
    host_assert(t.dim() in (2, 3), 'dim == 2 or dim == 3')
    if t.dim() == 3:
        host_assert(t.stride(0) == t.size(-2) * t.size(-1), 't.stride(0) == t.size(-2) * t.size(-1)')
    host_assert(t.stride(-2) == 1 or t.stride(-1) == 1, 't.stride(-2) == 1 or t.stride(-1) == 1')


def is_k_major(t: torch.Tensor) -> bool:
    """K-major (unit stride on the last dim) vs MN-major, decided from strides like ``get_major_type_ab``."""
    major_check(t)
    return t.stride(-1) == 1


def check_major_type_cd(t: torch.Tensor) -> None:
    major_check(t)
    host_assert(t.stride(-1) == 1, 't.stride(-1) == 1')


def get_default_recipe(sfa_dtype: torch.dtype, sfb_dtype: torch.dtype) -> Tuple[int, int, int]:
    """csrc/utils/layout.hpp:64-77: FP32 scales => (1, 128, 128) (gfx950 plays the role of the reference's FP32-scale
    architecture for them); packed UE8M0 (int) scales => (1, 1, 128), the reference's SM100 branch."""
    if sfa_dtype == torch.int and sfb_dtype == torch.int:
        return 1, 1, 128
    host_assert(sfa_dtype == torch.float and sfb_dtype == torch.float,
                'sfa_dtype == torch::kFloat and sfb_dtype == torch::kFloat')
    return 1, 128, 128


def check_sf_layout(sf: torch.Tensor, mn: int, k: int, gran_mn: int, gran_k: int, num_groups: Optional[int],
                    tma_stride_check: bool = False, sfb_check: bool = False,
                    type_check: Optional[torch.dtype] = None) -> torch.Tensor:
    if type_check is not None:
        host_assert(sf.dtype == type_check, 'sf.scalar_type() == type_check.value()')
    host_assert(sf.dtype in (torch.float, torch.int), 'sf_dtype == torch::kFloat or sf_dtype == torch::kInt')
    host_assert(sf.dim() == int(num_groups is not None) + 2, 'sf.dim() == static_cast<int>(num_groups.has_value()) + 2')
    if num_groups is not None:
        host_assert(sf.size(-3) == num_groups, 'sf.size(-3) == num_groups.value()')
    host_assert(sf.size(-2) == ceil_div(mn, gran_mn), 'sf.size(-2) == ceil_div(mn, gran_mn)')
    host_assert(sf.size(-1) == ceil_div(k, gran_k * (1 if sf.dtype == torch.float else 4)),
                'sf.size(-1) == ceil_div(k, gran_k * (sf_dtype == torch::kFloat ? 1 : 4))')
    if tma_stride_check:
        if num_groups is not None:
            host_assert(sf.stride(-3) == sf.stride(-1) * sf.size(-1), 'sf.stride(-3) == sf.stride(-1) * sf.size(-1)')
        host_assert(sf.stride(-2) == 1 or mn == 1, 'sf.stride(-2) == 1 or mn == 1')
        host_assert(sf.stride(-1) == get_tma_aligned_size(mn, sf.element_size()),
                    'sf.stride(-1) == get_tma_aligned_size(mn, sf.element_size())')
    if sfb_check:
        if num_groups is not None:
            host_assert(sf.stride(-3) == sf.size(-2) * sf.size(-1), 'sf.stride(-3) == sf.size(-2) * sf.size(-1)')
        host_assert((sf.stride(-1) == 1 and sf.stride(-2) == sf.size(-1)) or
                    (sf.stride(-1) == sf.size(-2) and sf.stride(-2) == 1),
                    'SFB must be contiguous, or contiguous after transposing the last two dimensions')
    return sf


def get_mn_major_tma_aligned_tensor(sf: torch.Tensor) -> torch.Tensor:
    """[..., mn, sf_k] FP32 -> same logical tensor with strides (aligned_mn * sf_k, 1, aligned_mn).

    Zero-copy when the input already has that layout (smxx_layout.hpp:124-125); otherwise one launch of the HIP
    transpose kernel (contiguous input) or a strided torch copy (anything else), as in the reference."""
    host_assert(sf.dim() in (2, 3), 'dim == 2 or dim == 3')
    host_assert(sf.dtype == torch.float, 'sf.scalar_type() == torch::kFloat')
    batched = sf.unsqueeze(0) if sf.dim() == 2 else sf
    nb, mn, sf_k = batched.shape
    aligned_mn = get_tma_aligned_size(mn, sf.element_size())
    if (batched.stride(0) == aligned_mn * sf_k or sf.dim() == 2) and batched.stride(1) == 1 and batched.stride(2) == aligned_mn:
        return sf
    out = torch.empty_strided((nb, mn, sf_k), (aligned_mn * sf_k, 1, aligned_mn), dtype=sf.dtype, device=sf.device)
    if batched.is_contiguous() and batched.is_cuda and nb <= 65535:
        check(lib.dg_transpose_sf_fp32(batched.data_ptr(), out.data_ptr(), nb, mn, sf_k, current_stream_ptr()))
    else:
        out.copy_(batched)
    return out.squeeze(0) if sf.dim() == 2 else out


def get_mn_major_tma_aligned_packed_ue8m0_tensor(sf: torch.Tensor, psum_layout: Optional[torch.Tensor] = None,
                                                 _gran_mn: int = 1, _mn: Optional[int] = None) -> torch.Tensor:
    """[..., mn, sf_k] FP32 power-of-two scales -> packed UE8M0 words ``[..., mn, ceil(sf_k / 4)]`` int32 with strides
    ``(packed_k * aligned_mn, 1, aligned_mn)``: four exponent bytes per word, MN-major (the layout the scaled-MFMA kernels
    read).  Reference: csrc/jit_kernels/impls/smxx_layout.hpp:181-246; mantissa bits are dropped as in its torch twin (``>> 23``).

    ``psum_layout`` (the cumulative row ends of the psum contiguous layout, int32 on the device): rows in the gaps between a group's
    end and the next group's aligned start hold uninitialised scales -- they are packed as zero words (smxx_layout.cuh:76-94).
    ``_gran_mn`` / ``_mn`` (host-layer internal): ``sf`` has one row per ``_gran_mn`` of the ``_mn`` output rows; the broadcast the
    reference materialises with ``index_select`` (csrc/apis/layout.hpp:52-53) happens inside the pack kernel."""
    host_assert(sf.dim() in (2, 3), 'dim == 2 or dim == 3')
    host_assert(sf.dtype == torch.float, 'sf.scalar_type() == torch::kFloat')
    batched = sf.unsqueeze(0) if sf.dim() == 2 else sf
    nb, src_rows, sf_k = batched.shape
    mn = src_rows if _mn is None else _mn
    host_assert(ceil_div(mn, _gran_mn) == src_rows, 'sf.size(-2) == ceil_div(mn, gran_mn)')
    require_device(sf)
    aligned_mn, packed_k = get_tma_aligned_size(mn, 4), ceil_div(sf_k, 4)
    layout_ptr, num_psum_groups, m_alignment = None, 0, 0
    if psum_layout is not None:
        from . import runtime
        host_assert(nb == 1 and batched.is_contiguous(), 'num_sf_batches == 1 and batched_sf.is_contiguous()')
        host_assert(psum_layout.dtype == torch.int and psum_layout.is_contiguous(),
                    'psum_layout->scalar_type() == torch::kInt and psum_layout->is_contiguous()')
        host_assert(psum_layout.numel() > 0, 'psum_layout->numel() > 0')
        require_device(psum_layout)
        layout_ptr, num_psum_groups = psum_layout.data_ptr(), psum_layout.numel()
        m_alignment = runtime.get_mk_alignment_for_contiguous_layout()
    out = torch.empty_strided((nb, mn, packed_k), (packed_k * aligned_mn, 1, aligned_mn), dtype=torch.int, device=sf.device)
    host_assert(nb <= 65535, 'num_sf_batches <= 65535')
    check(lib.dg_pack_sf_ue8m0_ex(batched.data_ptr(), out.data_ptr(), nb, mn, sf_k,
                                  batched.stride(0), batched.stride(1), batched.stride(2), _gran_mn,
                                  layout_ptr, num_psum_groups, m_alignment, current_stream_ptr()))
    return out.squeeze(0) if sf.dim() == 2 else out


def get_k_grouped_mn_major_tma_aligned_packed_ue8m0_tensor(sf: torch.Tensor, grouped_layout: torch.Tensor, ks_cpu, gran_k: int,
                                                           k_alignment: int, use_psum_layout: bool = False) -> torch.Tensor:
    """Per-channel FP32 power-of-two scales of a K-grouped operand, ``[sum_g ceil(k_g / gran_k), mn]`` (compact rows per group) ->
    packed UE8M0 words ``[sum_g ceil(k_g / (4 gran_k)), mn]`` int32: four consecutive scale rows OF ONE GROUP per word (byte j = row
    4 q + j, a group's last word zero-padded), MN-major.  Reference: csrc/jit_kernels/impls/smxx_layout.hpp:255-316 (kernel
    ``pack_fp32_into_ue8m0``, impls/smxx_layout.cuh:148); checks in its order.  The group extents are taken from ``ks_cpu`` (the reference
    reads the same numbers from ``grouped_layout`` on the device); one gather, works on any device.  The psum form packs from
    device-side ends and exists for SM100 only in the reference (tests/test_layout.py:108-111): it ends here as it ends there."""
    host_assert(gran_k in (32, 128), 'gran_k == 32 or gran_k == 128')
    host_assert(k_alignment % 32 == 0, 'k_alignment % 32 == 0')
    host_assert(sf.dim() == 2, 'sf.dim() == 2')
    sf_k, mn = (int(x) for x in sf.shape)
    num_groups = grouped_layout.numel()
    host_assert(sf.is_contiguous(), 'sf.is_contiguous()')
    host_assert(num_groups <= 128 and mn % 4 == 0, 'num_groups <= 128 and mn % 4 == 0')
    host_assert(grouped_layout.is_contiguous() and grouped_layout.dtype == torch.int,
                'grouped_layout.is_contiguous() and grouped_layout.scalar_type() == torch::kInt')
    has_synced_ks = ks_cpu is not None and len(ks_cpu) > 0
    if has_synced_ks:
        host_assert(len(ks_cpu) == num_groups, 'static_cast<int>(ks_cpu.value().size()) == num_groups')
    else:
        host_assert(use_psum_layout, 'use_psum_layout')
    if use_psum_layout:
        raise RuntimeError('Assertion error (layout.py): Unsupported architecture (the psum form of the K-grouped scale packing is '
                           'implemented by the reference for SM100 only)')
    host_assert(sf.dtype == torch.float, 'sf.scalar_type() == torch::kFloat')
    rows = [ceil_div(int(k), gran_k) for k in ks_cpu]
    host_assert(sum(rows) == sf_k, 'use_psum_layout or ref_sf_k == sf_k')
    index, first = [], 0
    for r in rows:
        for q in range(ceil_div(r, 4)):
            index.append([first + 4 * q + j if 4 * q + j < r else sf_k for j in range(4)])      # sf_k = the appended zero row
        first += r
    if not index:
        return torch.empty((0, mn), dtype=torch.int, device=sf.device)
    exps = ((sf.view(torch.int) >> 23) & 0xff).to(torch.uint8)
    exps = torch.cat([exps, torch.zeros((1, mn), dtype=torch.uint8, device=sf.device)])
    picked = exps[torch.tensor(index, dtype=torch.long, device=sf.device)]              # [packed rows, 4, mn]
    return picked.permute(0, 2, 1).contiguous().view(torch.int).squeeze(-1)


Recipe = Union[Tuple[int, int, int], Tuple[int, int]]


def transform_sf_into_required_layout(sf: torch.Tensor, mn: int, k: int, recipe: Recipe,
                                      num_groups: Optional[int] = None, is_sfa: Optional[bool] = None,
                                      disable_ue8m0_cast: bool = False,
                                      psum_layout: Optional[torch.Tensor] = None, keep_row_major: bool = False) -> torch.Tensor:
    recipe = tuple(recipe)
    if len(recipe) == 3:
        host_assert(is_sfa is not None, 'is_sfa.has_value()')
        gran_mn, gran_k = (recipe[0] if is_sfa else recipe[1]), recipe[2]
    elif len(recipe) == 2:
        host_assert(is_sfa is None, 'not is_sfa.has_value()')
        gran_mn, gran_k = recipe
    else:
        raise RuntimeError('Assertion error (layout.py): Invalid recipe')
    check_sf_layout(sf, mn, k, gran_mn, gran_k, num_groups)
    from . import runtime
    fp32_as_is = runtime.get_sf_cast_mode() == 'sm90' or disable_ue8m0_cast     # reference: arch_major == 9 or disable_ue8m0_cast

    # (FP32, 1, 128) consumed as FP32: MN-major, padded -- csrc/apis/layout.hpp:40-42
    if sf.dtype == torch.float and gran_mn == 1 and gran_k == 128 and fp32_as_is:
        # keep_row_major (round 4, gemm.py): the dense kernel the call will run reads a row-major SFA in place (duo_p_rm_256x256) -- checked
        # as above, handed over as it is: the transpose launch of smxx_layout.hpp:120-153 is not needed
        return sf if keep_row_major else get_mn_major_tma_aligned_tensor(sf)
    # (FP32, 128, 128) consumed as FP32: only checked -- csrc/apis/layout.hpp:44-46
    if sf.dtype == torch.float and gran_mn == 128 and gran_k == 128 and fp32_as_is:
        return check_sf_layout(sf, mn, k, gran_mn, gran_k, num_groups, False, True, torch.float)
    # (FP32, x, gran_k) in 'sm100' mode, gran_k 32 or 128: cast to (INT, 1, gran_k) -- broadcast to rows, packed (four consecutive scales along K
    # per word, whatever the granularity), MN-major -- csrc/apis/layout.hpp:48-54.  One fused kernel (dg_pack_sf_ue8m0_ex): no index_select temporary.
    if sf.dtype == torch.float and gran_k in (32, 128) and runtime.get_sf_cast_mode() == 'sm100':
        host_assert(not disable_ue8m0_cast, 'not disable_ue8m0_cast')
        return get_mn_major_tma_aligned_packed_ue8m0_tensor(sf, psum_layout, _gran_mn=gran_mn, _mn=mn)
    # (INT, 1, gran_k): packed UE8M0 words, only checked and brought to the MN-major layout -- csrc/apis/layout.hpp:56-58
    if sf.dtype == torch.int and gran_mn == 1 and gran_k in (32, 128):
        host_assert(sf.dim() == (2 if num_groups is None else 3), 'sf.dim() == static_cast<int>(num_groups.has_value()) + 2')
        host_assert(sf.size(-2) == mn and sf.size(-1) == ceil_div(k, gran_k * 4),
                    'sf.size(-2) == ceil_div(mn, gran_mn) and sf.size(-1) == ceil_div(k, gran_k * 4)')
        return get_mn_major_tma_aligned_tensor(sf.view(torch.float)).view(torch.int)
    raise RuntimeError('Assertion error (layout.py): Unknown SF transformation '
                       '(FP32 scales of granularity 32 along K are consumed as packed UE8M0 only: sf cast mode \'sm100\' or int scale tensors)')


def _cast_sf_pair_to_ue8m0(sfa, sfb, m, n, k, gran_m, gran_n, num_groups_a, num_groups_b, psum_layout, gran_k: int = 128):
    """The cast branch (csrc/apis/layout.hpp:48-54) for BOTH scale tensors of a call in one launch (dg_pack_sf_pair_ue8m0): checks as
    transform_sf_into_required_layout makes them for each tensor, one kernel boundary in front of the GEMM instead of two."""
    check_sf_layout(sfa, m, k, gran_m, gran_k, num_groups_a)
    check_sf_layout(sfb, n, k, gran_n, gran_k, num_groups_b)
    require_device(sfa, sfb)
    ba, bb = (sfa.unsqueeze(0) if sfa.dim() == 2 else sfa), (sfb.unsqueeze(0) if sfb.dim() == 2 else sfb)
    sf_k, packed_k = ceil_div(k, gran_k), ceil_div(ceil_div(k, gran_k), 4)
    host_assert(ba.size(0) <= 65535 and bb.size(0) <= 65535, 'num_sf_batches <= 65535')
    layout_ptr, num_psum_groups, m_alignment = None, 0, 0
    if psum_layout is not None:
        from . import runtime
        host_assert(ba.size(0) == 1 and ba.is_contiguous(), 'num_sf_batches == 1 and batched_sf.is_contiguous()')
        host_assert(psum_layout.dtype == torch.int and psum_layout.is_contiguous(),
                    'psum_layout->scalar_type() == torch::kInt and psum_layout->is_contiguous()')
        host_assert(psum_layout.numel() > 0, 'psum_layout->numel() > 0')
        require_device(psum_layout)
        layout_ptr, num_psum_groups, m_alignment = psum_layout.data_ptr(), psum_layout.numel(), runtime.get_mk_alignment_for_contiguous_layout()
    am, an = get_tma_aligned_size(m, 4), get_tma_aligned_size(n, 4)
    out_a = torch.empty_strided((ba.size(0), m, packed_k), (packed_k * am, 1, am), dtype=torch.int, device=sfa.device)
    out_b = torch.empty_strided((bb.size(0), n, packed_k), (packed_k * an, 1, an), dtype=torch.int, device=sfb.device)
    check(lib.dg_pack_sf_pair_ue8m0(ba.data_ptr(), out_a.data_ptr(), ba.size(0), m, ba.stride(0), ba.stride(1), ba.stride(2), gran_m,
                                    layout_ptr, num_psum_groups, m_alignment,
                                    bb.data_ptr(), out_b.data_ptr(), bb.size(0), n, bb.stride(0), bb.stride(1), bb.stride(2), gran_n,
                                    sf_k, current_stream_ptr()))
    return (out_a.squeeze(0) if sfa.dim() == 2 else out_a), (out_b.squeeze(0) if sfb.dim() == 2 else out_b)


def transform_sf_pair_into_required_layout(sfa, sfb, m, n, k, recipe, recipe_a, recipe_b,
                                           num_groups_a, num_groups_b, disable_ue8m0_cast=False, psum_layout=None,
                                           keep_sfa_row_major=False):
    """Returns (sfa', sfb', gran_n_of_sfb).  Recipe selection: csrc/apis/layout.hpp:74-80."""
    if recipe_a is None and recipe is None:
        recipe = get_default_recipe(sfa.dtype, sfb.dtype)
    host_assert((recipe_a is not None) == (recipe_b is not None), 'recipe_a.has_value() == recipe_b.has_value()')
    host_assert((recipe_a is not None) != (recipe is not None), 'recipe_a.has_value() != recipe.has_value()')
    if recipe is not None:
        recipe = tuple(recipe)
        host_assert(len(recipe) == 3, 'recipe must be (gran_m, gran_n, gran_k)')
        host_assert(recipe[0] == 1 and recipe[2] in (32, 128) and recipe[1] in (1, 32, 128),
                    'supported recipes: (1, 128, 128), (1, 1, 128) and -- packed UE8M0 only -- (1, 1, 32) / (1, 32, 32)')
        from . import runtime
        if (sfa.dtype == torch.float and sfb.dtype == torch.float and not disable_ue8m0_cast and runtime.get_sf_cast_mode() == 'sm100'):
            return _cast_sf_pair_to_ue8m0(sfa, sfb, m, n, k, recipe[0], recipe[1], num_groups_a, num_groups_b, psum_layout, recipe[2]) + (recipe[1],)
        host_assert(recipe[2] == 128 or (sfa.dtype == torch.int and sfb.dtype == torch.int),
                    "gran_k == 128 for FP32 scaling factors consumed as FP32 (gran_k == 32: packed UE8M0 words or sf cast mode 'sm100')")
        host_assert(recipe[1] in (1, 128) or recipe[2] == 32, 'gran_n in (1, 128)')
        t_sfa = transform_sf_into_required_layout(sfa, m, k, recipe, num_groups_a, True, disable_ue8m0_cast, psum_layout, keep_sfa_row_major)
        t_sfb = transform_sf_into_required_layout(sfb, n, k, recipe, num_groups_b, False, disable_ue8m0_cast)
        gran_n = recipe[1]
    else:
        recipe_a, recipe_b = tuple(recipe_a), tuple(recipe_b)
        host_assert(recipe_a[0] == 1 and recipe_a[1] in (32, 128) and recipe_b[1] == recipe_a[1] and recipe_b[0] in (1, recipe_b[1]),
                    'supported recipes: recipe_a = (1, gran_k), recipe_b in ((1, gran_k), (gran_k, gran_k)), gran_k 128 or -- packed UE8M0 only -- 32')
        from . import runtime
        if (recipe_a[1] == 32 and sfa.dtype == torch.float and sfb.dtype == torch.float and not disable_ue8m0_cast and
                runtime.get_sf_cast_mode() == 'sm100'):
            return _cast_sf_pair_to_ue8m0(sfa, sfb, m, n, k, 1, recipe_b[0], num_groups_a, num_groups_b, psum_layout, 32) + (recipe_b[0],)
        host_assert(recipe_a[1] == 128 or (sfa.dtype == torch.int and sfb.dtype == torch.int),
                    "gran_k == 128 for FP32 scaling factors consumed as FP32 (gran_k == 32: packed UE8M0 words or sf cast mode 'sm100')")
        t_sfa = transform_sf_into_required_layout(sfa, m, k, recipe_a, num_groups_a, None, disable_ue8m0_cast, psum_layout, keep_sfa_row_major)
        t_sfb = transform_sf_into_required_layout(sfb, n, k, recipe_b, num_groups_b, None, disable_ue8m0_cast)
        gran_n = recipe_b[0]
    return t_sfa, t_sfb, gran_n
