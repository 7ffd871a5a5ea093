"""The FP8 GEMM operator surface (host side): validation, trivial cases, SF layout step, C-ABI call.

Signatures, keyword names/defaults, assertion order and in-place semantics follow the reference's API layer
(``csrc/apis/gemm.hpp:19-297`` and the ``m.def`` table at ``:645-717``).  Differences, all supersets:
  * MN-major FP8 operands, ``c`` accumulation and FP32 outputs are accepted for every dense layout (the reference only
    accepts them on SM100);
  * ``compiled_dims`` is accepted and does not affect results (there is no JIT);
  * ``disable_ue8m0_cast`` is the reference's keyword (csrc/apis/layout.hpp:40-50): it matters in the ``'sm100'`` scaling-factor
    mode (``runtime.set_sf_cast_mode``), where FP32 scales are cast to packed UE8M0 words and the hardware-scaled MFMA kernels run
    unless a call disables the cast; in the default ``'sm90'`` mode FP32 scales are always consumed as FP32.
Every call is asynchronous on the current torch stream and never synchronises.
"""
from typing import Optional, Tuple

import torch

from ._lib import lib, check, current_stream_ptr, require_device
from .errors import host_assert
from .layout import (check_major_type_cd, get_mn_major_tma_aligned_tensor, is_k_major, major_check,
                     transform_sf_into_required_layout, transform_sf_pair_into_required_layout)
from . import runtime

_BF16, _FP32 = 0, 1
TensorPair = Tuple[torch.Tensor, torch.Tensor]


def _check_ab_fp8(t: torch.Tensor, dims: int):
    host_assert(t.dim() == dims, f't.dim() == {dims}')
    host_assert(t.dtype == torch.float8_e4m3fn, 'ab.scalar_type() == torch::kFloat8_e4m3fn')
    return tuple(int(s) for s in t.shape)


# MN-major FP8 operands of problems at least this large (in MACs) are re-majored into a K-major scratch buffer by
# dg_transpose_fp8 (2 bytes of HBM traffic per element) so that the LDS-DMA kernels can stream them; smaller ones go
# to the generic kernel in place.  0 disables the re-majoring (tests use it to reach the generic kernel's MN-major path).
REMAJOR_MIN_MACS = 1 << 27


def _as_k_major(t: torch.Tensor, macs: int) -> torch.Tensor:
    if t.stride(-1) == 1 or REMAJOR_MIN_MACS <= 0 or macs < REMAJOR_MIN_MACS:
        return t
    return _remajor(t)


def _remajor(t: torch.Tensor) -> torch.Tensor:
    """An MN-major FP8 operand view ``[.., mn, k]`` (stride 1 along mn) as a fresh K-major tensor (dg_transpose_fp8)."""
    require_device(t)
    mn, k = t.size(-2), t.size(-1)
    batches = t.size(0) if t.dim() == 3 else 1
    out = torch.empty(t.shape, dtype=t.dtype, device=t.device)          # contiguous: K-major
    check(lib.dg_transpose_fp8(t.data_ptr(), out.data_ptr(), batches, k, mn, t.stride(-1), k,
                               t.stride(0) if t.dim() == 3 else 0, mn * k, current_stream_ptr()))
    return out


def _operand_plan(gemm_type: int, a: torch.Tensor, b: torch.Tensor, sfa: torch.Tensor, gran_n: int, m: int, n: int, k: int,
                  m_alignment: int = 128) -> int:
    """dg_operand_plan (include/deepgemm_amd.h): which MN-major operands no kernel reads in place (bit 0: A, bit 1: B).  The C side
    decides with the predicates its own launch applies -- there is no Python copy of the alignment / tile rules to drift."""
    return int(lib.dg_operand_plan(gemm_type, a.data_ptr(), b.data_ptr(), m, n, k, a.stride(-2), a.stride(-1), b.stride(-2), b.stride(-1),
                                   b.stride(0) if b.dim() == 3 else 0, sfa.stride(-2), gran_n, m_alignment))


def _dense_operands(a_data: torch.Tensor, b_data: torch.Tensor, sfa: torch.Tensor, gran_n: int, m: int, n: int, k: int):
    """The FP8 operands as the C entry will read them: as they are wherever a kernel takes that majorness natively, re-majored
    into K-major scratch (dg_transpose_fp8) otherwise.  One place for the cached and the uncached path of fp8_gemm_nt."""
    if a_data.stride(-1) == 1 and b_data.stride(-1) == 1:
        return a_data, b_data
    plan, macs = _operand_plan(0, a_data, b_data, sfa, gran_n, m, n, k), m * n * k
    return (_as_k_major(a_data, macs) if plan & 1 else a_data), (_as_k_major(b_data, macs) if plan & 2 else b_data)


# Host-overhead diet for decode-sized calls: the reference's checks cost ~20 us of Python per call, more than the kernel of
# a small GEMM.  A call whose (shape, stride, dtype, device) signature has already passed every check once skips straight
# to the SF layout step and the C call; anything new, or any failing call, takes the full path below.
_VALIDATED_DENSE = {}
_VALIDATED_MASKED = {}


def _sig(t: Optional[torch.Tensor]):
    return None if t is None else (t.shape, t.stride(), t.dtype, t.device)


def _dtype_code(d: torch.Tensor) -> int:
    return _BF16 if d.dtype == torch.bfloat16 else _FP32


def _early_return(m: int, n: int, k: int, d: torch.Tensor, c: Optional[torch.Tensor]) -> bool:
    """csrc/apis/gemm.hpp:19-46."""
    if m == 0 or n == 0:
        return True
    is_cd_same = c is not None and c.data_ptr() == d.data_ptr()
    if is_cd_same:
        host_assert(c.shape == d.shape and c.stride() == d.stride(), 'c->sizes() == d.sizes() and c->strides() == d.strides()')
    host_assert(d.dtype in (torch.bfloat16, torch.float), 'd.scalar_type() == torch::kBFloat16 or d.scalar_type() == torch::kFloat')
    if c is not None:
        check_major_type_cd(c)
        host_assert(d.dtype == c.dtype, 'd.scalar_type() == c.value().scalar_type()')
    if k == 0:
        if not is_cd_same:
            d.copy_(c) if c is not None else d.zero_()
        return True
    if c is not None and not is_cd_same:
        d.copy_(c)
    return False


def _packed_sf_mn_major(sf: torch.Tensor, mn: int, k: int, gran_k: int = 128) -> torch.Tensor:
    """Packed UE8M0 scale words [mn, ceil(k / (4 gran_k))] int32 -> the MN-major, 16-byte aligned layout (zero-copy if already
    there): the int32 twin of get_mn_major_tma_aligned_tensor (csrc/jit_kernels/impls/smxx_layout.hpp:120-153)."""
    host_assert(sf.dtype == torch.int, 'sf.scalar_type() == torch::kInt')
    host_assert(sf.dim() == 2, 'sf.dim() == 2')
    host_assert(sf.size(0) == mn and sf.size(1) == -(-k // (4 * gran_k)), 'sf.size(-2) == mn and sf.size(-1) == ceil_div(k, gran_k * 4)')
    return get_mn_major_tma_aligned_tensor(sf.view(torch.float)).view(torch.int)


def _packed_gran_k(recipe, recipe_a, recipe_b) -> int:
    """Scale granularity along K of a packed-UE8M0 call: 128 (default recipe of int scales, csrc/utils/layout.hpp:64-77) or 32 (the SM100 MX
    recipe, csrc/apis/gemm.hpp:311-312); one granularity for both operands (the reference's mixed case, 128 / 32, is its FP8 x FP4 form)."""
    if recipe is not None:
        recipe = tuple(recipe)
        host_assert(len(recipe) == 3 and recipe[0] == 1 and recipe[2] in (32, 128), 'recipe == (1, gran_n, gran_k) with gran_k == 32 or gran_k == 128')
        return recipe[2]
    if recipe_a is not None and recipe_b is not None:
        host_assert(tuple(recipe_a)[1] == tuple(recipe_b)[1] and tuple(recipe_a)[1] in (32, 128),
                    'gran_k_a == gran_k_b and (gran_k == 32 or gran_k == 128): FP8 x FP8 operands share one K granularity')
        return tuple(recipe_a)[1]
    return 128


def _unpack_ue8m0(sf_packed: torch.Tensor, k: int) -> torch.Tensor:
    """Packed UE8M0 words [mn, ceil(k / 512)] (byte j of word q = exponent of K block 4 q + j, 127 = 1.0) -> FP32 scales
    [mn, ceil(k / 128)] = 2^(e - 127), exactly."""
    shifts = torch.tensor([0, 8, 16, 24], dtype=torch.int32, device=sf_packed.device)
    exps = (sf_packed.unsqueeze(-1) >> shifts) & 0xff                       # [mn, kq, 4]
    exps = exps.reshape(sf_packed.size(0), -1)[:, :-(-k // 128)]
    return (exps << 23).view(torch.float).contiguous()


def _casts_to_ue8m0(a_sf: torch.Tensor, b_sf: torch.Tensor, disable_ue8m0_cast: bool) -> bool:
    """FP32 scale tensors that the layout step will cast to packed UE8M0 words: the reference's default on SM100
    (csrc/apis/layout.hpp:48-54), here the ``'sm100'`` mode of ``runtime.set_sf_cast_mode``."""
    return (not disable_ue8m0_cast and a_sf.dtype == torch.float and b_sf.dtype == torch.float and
            runtime.get_sf_cast_mode() == 'sm100')


def _truncate_to_ue8m0(sf: torch.Tensor) -> torch.Tensor:
    """FP32 scales with everything but the exponent byte dropped: the value the cast branch's ``>> 23`` keeps (exactly 2^(e - 127))."""
    return (sf.view(torch.int) & 0x7f800000).view(torch.float)


_VALIDATED_PACKED = {}


def _fp8_gemm_nt_packed_ue8m0(a_data, a_sf, b_data, b_sf, d, c, recipe, recipe_a, recipe_b, cache_key=None) -> None:
    """Power-of-two scales handed over as packed exponent bytes (the reference's SM100 format, recipe (1, 1, 128)) -- or FP32 scales in
    the ``'sm100'`` mode, which the layout step casts to that format (csrc/apis/layout.hpp:48-54): hardware-scaled MFMA path, no FP32
    promotion."""
    fp32_in = a_sf.dtype == torch.float and b_sf.dtype == torch.float
    gran_k = _packed_gran_k(recipe, recipe_a, recipe_b)
    a_given, b_given = a_data, b_data
    if not fp32_in:
        host_assert(a_sf.dtype == torch.int and b_sf.dtype == torch.int, 'sfa.scalar_type() == torch::kInt and sfb.scalar_type() == torch::kInt')
        host_assert(recipe is None or tuple(recipe) == (1, 1, gran_k), 'recipe == (1, 1, gran_k) for packed UE8M0 scaling factors')
        host_assert((recipe_a is None) == (recipe_b is None) and (recipe_a is None or (tuple(recipe_a) == (1, gran_k) and tuple(recipe_b) == (1, gran_k))),
                    'recipe_a == (1, gran_k) and recipe_b == (1, gran_k) for packed UE8M0 scaling factors')
    major_check(a_data), major_check(b_data)
    check_major_type_cd(d)
    m, k = _check_ab_fp8(a_data, 2)
    n, k_ = _check_ab_fp8(b_data, 2)
    host_assert(d.dim() == 2, 'd.dim() == 2')
    host_assert((m, n) == tuple(d.shape) and k == k_, 'm == m_ and n == n_ and k == k_')
    if _early_return(m, n, k, d, c):
        return
    def _tail_operand_ok(t):        # K-major: 16-byte aligned rows; MN-major: re-majored into aligned scratch below
        return t.stride(-1) != 1 or (t.stride(0) % 16 == 0 and t.data_ptr() % 16 == 0)
    packed_tail_ok = not fp32_in and k % 16 == 0 and k > 128 and _tail_operand_ok(a_data) and _tail_operand_ok(b_data)
    # (granularity 32: whole 128-K blocks -- a packed word is one block's four exponents; the reference takes any K through TMA zero-fill)
    host_assert(gran_k == 128 or k % 128 == 0, 'k % 128 == 0 for scaling factors of granularity 32 along K')
    if k % 128 != 0 and not packed_tail_ok:
        # A partial last K block (the reference's SM100 kernels take any K: TMA zero-fills).  Packed words with K-major operands and whole
        # 16-byte chunks stay on the hardware-scaled path (e8_quad_kt_128x256: the partial block is zero-filled by the buffer range check).
        # Everything else: the exponents are expanded to FP32 scales -- exactly, they are powers of two -- and the FP32-scale path computes
        # the same sums (FP32 scales of the 'sm100' mode: truncated, then the duo kernels' tail stage; packed words of an MN-major or
        # unaligned operand: layout-agnostic kernel, correct, not fast).  `c` has been folded into `d` by _early_return.
        if fp32_in:
            fp8_gemm_nt((a_data, _truncate_to_ue8m0(a_sf)), (b_data, _truncate_to_ue8m0(b_sf)), d, d if c is not None else None,
                        recipe, recipe_a, recipe_b, disable_ue8m0_cast=True)
            return
        host_assert(a_sf.dim() == 2 and b_sf.dim() == 2 and a_sf.size(0) == m and b_sf.size(0) == n and
                    a_sf.size(1) == -(-k // 512) and b_sf.size(1) == -(-k // 512),
                    'sf.size(-2) == mn and sf.size(-1) == ceil_div(k, 128 * 4)')
        fp8_gemm_nt((a_data, _unpack_ue8m0(a_sf, k)), (b_data, _unpack_ue8m0(b_sf, k)), d, d if c is not None else None,
                    recipe=(1, 1, 128), disable_ue8m0_cast=True)
        return
    if fp32_in:
        sfa, sfb, _ = transform_sf_pair_into_required_layout(a_sf, b_sf, m, n, k, recipe, recipe_a, recipe_b, None, None, False)
    else:
        sfa, sfb = _packed_sf_mn_major(a_sf, m, k, gran_k), _packed_sf_mn_major(b_sf, n, k, gran_k)
    require_device(a_data, b_data, sfa, sfb, d)
    # MN-major operands (nn / tn / tt): read in place by the 8-wave hardware-scaled kernels where the library says that beats a re-majoring pass
    # (granularity 32: the four-wave kernels only -- every MN-major operand is re-majored)
    if a_data.stride(-1) != 1 or b_data.stride(-1) != 1:
        plan = 3 if gran_k == 32 else lib.dg_ue8m0_dense_operand_plan(a_data.data_ptr(), b_data.data_ptr(), m, n, k, a_data.stride(0), a_data.stride(1),
                                                                       b_data.stride(0), b_data.stride(1))
        if plan & 1 and a_data.stride(-1) != 1:
            a_data = _as_k_major(a_data, REMAJOR_MIN_MACS if REMAJOR_MIN_MACS > 0 else 1)
        if plan & 2 and b_data.stride(-1) != 1:
            b_data = _as_k_major(b_data, REMAJOR_MIN_MACS if REMAJOR_MIN_MACS > 0 else 1)
    # (under-filled launches with a long K loop are cut along K when the stream's scratch buffer is handed over: the library owns the rule)
    ws = None
    k_major = a_data.stride(-1) == 1 and b_data.stride(-1) == 1
    wants_ws = bool(k_major and lib.dg_ue8m0_dense_wants_workspace(m, n, k))
    if wants_ws:
        ws = _split_k_workspace(d.device, current_stream_ptr())
    if (cache_key is not None and k_major and a_data is a_given and b_data is b_given and sfa.data_ptr() == a_sf.data_ptr() and sfb.data_ptr() == b_sf.data_ptr() and
            sfa.stride() == a_sf.stride() and sfb.stride() == b_sf.stride() and len(_VALIDATED_PACKED) < 4096):
        # (the operands went through as they came: every integer argument of the C call is a function of the signature)
        _VALIDATED_PACKED[cache_key] = ((m, n, k, a_data.stride(0), a_data.stride(1), b_data.stride(0), b_data.stride(1), sfa.stride(0), sfa.stride(1),
                                         sfb.stride(0), sfb.stride(1), d.stride(0), _dtype_code(d), int(c is not None), gran_k),
                                        d.device.index if d.device.index is not None else -1, wants_ws)
    check(lib.dg_fp8_gemm_nt_ue8m0_ws(
        a_data.data_ptr(), sfa.data_ptr(), b_data.data_ptr(), sfb.data_ptr(), d.data_ptr(), m, n, k,
        a_data.stride(0), a_data.stride(1), b_data.stride(0), b_data.stride(1),
        sfa.stride(0), sfa.stride(1), sfb.stride(0), sfb.stride(1),
        d.stride(0), _dtype_code(d), int(c is not None), gran_k,
        ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, current_stream_ptr()))


_SPLIT_K_WORKSPACES = {}


def _split_k_workspace(device: torch.device, stream: int) -> Optional[torch.Tensor]:
    """Scratch buffer of the K-split tail (dg_m_grouped_fp8_gemm_nt_contiguous_ws): one zero-filled buffer per device and
    stream, created on first use and kept (its contents never matter between launches; launches of one stream are ordered, so
    they can share it).  The C ABI itself never allocates.  ``None`` (= no K split) on a stream that is being captured and
    has no buffer yet: an allocation made during capture belongs to the graph's private pool and must not outlive it.
    The buffer is keyed by the stream the call was ISSUED on: a graph captured on stream S keeps using S's buffer when it is replayed on
    another stream T -- replay such a graph concurrently with eager K-split calls on S and the two share scratch memory; keep them on one
    stream (or capture after warming the replay stream up) if that matters."""
    key = (device.index, stream)
    ws = _SPLIT_K_WORKSPACES.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        ws = torch.zeros(int(lib.dg_split_k_workspace_bytes()), dtype=torch.uint8, device=device)
        _SPLIT_K_WORKSPACES[key] = ws
    return ws


def _dense_split_k_workspace(m: int, n: int, k: int, gran_n: int, device: torch.device, a_mn_major: bool = False,
                             b_mn_major: bool = False) -> Optional[torch.Tensor]:
    """The K-split scratch buffer for a dense call the library would cut along K (dg_dense_wants_workspace: under-filled launches and
    partial last rounds of 128 x 256 tiles with long K loops, under-filled recipe (1, 1, 128) launches); None otherwise, so that ordinary
    calls neither create nor pass a buffer.  The C side owns the model and decides again, with the real pointers, at launch."""
    if not lib.dg_dense_wants_workspace(m, n, k, int(a_mn_major), int(b_mn_major), gran_n):
        return None
    return _split_k_workspace(device, current_stream_ptr())


def _call_dense(a_data, sfa, b_data, sfb, d, c, m, n, k, gran_n) -> None:
    ws = _dense_split_k_workspace(m, n, k, gran_n, d.device, a_data.stride(-1) != 1, b_data.stride(-1) != 1)
    check(lib.dg_fp8_gemm_nt_ws(
        a_data.data_ptr(), sfa.data_ptr(), b_data.data_ptr(), sfb.data_ptr(), d.data_ptr(), m, n, k,
        a_data.stride(0), a_data.stride(1), b_data.stride(0), b_data.stride(1),
        sfa.stride(0), sfa.stride(1), sfb.stride(0), sfb.stride(1), gran_n,
        d.stride(0), _dtype_code(d), int(c is not None),
        ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, current_stream_ptr()))


def _packed_sf_pair(a_sf, b_sf, m, n, k, recipe, recipe_a, recipe_b, num_groups_a, num_groups_b, psum_layout=None):
    """Both scale tensors as packed UE8M0 words in the MN-major layout (csrc/apis/layout.hpp:56-58: the (INT, 1, gran_k) branch of
    transform_sf_into_required_layout; default recipe for int scales is (1, 1, 128), csrc/utils/layout.hpp:64-77) -- or, for FP32
    scales in the 'sm100' mode, the cast branch (:48-54; ``psum_layout`` lets the SFA pack zero the psum layout's gap rows)."""
    host_assert(k % 128 == 0, 'k % 128 == 0')
    gran_k = _packed_gran_k(recipe, recipe_a, recipe_b)
    if a_sf.dtype == torch.float and b_sf.dtype == torch.float:
        sfa, sfb, _ = transform_sf_pair_into_required_layout(a_sf, b_sf, m, n, k, recipe, recipe_a, recipe_b, num_groups_a, num_groups_b,
                                                             False, psum_layout)
        return sfa, sfb, gran_k
    host_assert(a_sf.dtype == torch.int and b_sf.dtype == torch.int, 'sfa.scalar_type() == torch::kInt and sfb.scalar_type() == torch::kInt')
    host_assert(recipe is None or tuple(recipe) == (1, 1, gran_k), 'recipe == (1, 1, gran_k) for packed UE8M0 scaling factors')
    host_assert((recipe_a is None) == (recipe_b is None), 'recipe_a.has_value() == recipe_b.has_value()')
    host_assert(recipe_a is None or (tuple(recipe_a) == (1, gran_k) and tuple(recipe_b) == (1, gran_k)),
                'recipe_a == (1, gran_k) and recipe_b == (1, gran_k) for packed UE8M0 scaling factors')
    return (transform_sf_into_required_layout(a_sf, m, k, (1, gran_k), num_groups_a),
            transform_sf_into_required_layout(b_sf, n, k, (1, gran_k), num_groups_b), gran_k)


def _m_grouped_masked_packed_ue8m0(a_data, a_sf, b_data, b_sf, d, masked_m, expected_m, recipe, recipe_a, recipe_b) -> None:
    """m_grouped_fp8_gemm_nt_masked with packed UE8M0 scales (csrc/apis/gemm.hpp:250-297 with int scale tensors)."""
    host_assert(is_k_major(a_data) and is_k_major(b_data), 'major_a == cute::UMMA::Major::K and major_b == cute::UMMA::Major::K')
    host_assert(masked_m.is_contiguous(), 'masked_m.is_contiguous()')
    num_groups, m, k = _check_ab_fp8(a_data, 3)
    num_groups_, n, k_ = _check_ab_fp8(b_data, 3)
    host_assert(d.dim() == 3, 'd.dim() == 3')
    host_assert(num_groups == num_groups_ == d.size(0) == masked_m.numel(),
                'num_groups == num_groups_ and num_groups == num_groups__ and num_groups == num_groups___')
    host_assert((m, n) == tuple(d.shape[1:]) and k == k_, 'm == m_ and n == n_ and k == k_')
    host_assert(expected_m > 0 and m > 0 and n > 0 and k > 0 and num_groups > 0,
                'expected_m > 0 and m > 0 and n > 0 and k > 0 and num_groups > 0')
    host_assert(d.dtype == torch.bfloat16, 'd.scalar_type() == torch::kBFloat16')
    host_assert(masked_m.dtype == torch.int, 'masked_m.scalar_type() == torch::kInt')
    check_major_type_cd(d)
    sfa, sfb, gran_k = _packed_sf_pair(a_sf, b_sf, m, n, k, recipe, recipe_a, recipe_b, num_groups, num_groups)
    require_device(a_data, b_data, sfa, sfb, d, masked_m)
    check((lib.dg_m_grouped_fp8_gemm_nt_masked_ue8m0_g32 if gran_k == 32 else lib.dg_m_grouped_fp8_gemm_nt_masked_ue8m0)(
        a_data.data_ptr(), sfa.data_ptr(), b_data.data_ptr(), sfb.data_ptr(), d.data_ptr(), masked_m.data_ptr(),
        num_groups, m, n, k, int(expected_m),
        a_data.stride(0), a_data.stride(1), a_data.stride(2), b_data.stride(0), b_data.stride(1), b_data.stride(2),
        sfa.stride(0), sfa.stride(1), sfa.stride(2), sfb.stride(0), sfb.stride(1), sfb.stride(2),
        d.stride(0), d.stride(1), current_stream_ptr()))


def fp8_gemm_nt(a: TensorPair, b: TensorPair, d: torch.Tensor, c: Optional[torch.Tensor] = None,
                recipe: Optional[Tuple[int, int, int]] = None, recipe_a: Optional[Tuple[int, int]] = None,
                recipe_b: Optional[Tuple[int, int]] = None, compiled_dims: str = 'nk',
                disable_ue8m0_cast: bool = False) -> None:
    """D = C + A @ B^T with per-128-block FP32 scales; ``a = (A_fp8 [M,K], SFA)``, ``b = (B_fp8 [N,K], SFB)``."""
    (a_data, a_sf), (b_data, b_sf) = a, b
    if a_sf.dtype == torch.int and b_sf.dtype == torch.int:
        # packed UE8M0 words: a validated signature (K-major operands, scale words already in the kernels' layout) goes straight to the C call --
        # the host path of a decode-sized packed call was ~19 us against a 15 us kernel (tools/probes/packed_host_path_profile.py)
        same_cd = c is not None and c.data_ptr() == d.data_ptr()
        key = (_sig(a_data), _sig(a_sf), _sig(b_data), _sig(b_sf), _sig(d), _sig(c), same_cd,
               recipe if recipe is None else tuple(recipe), recipe_a if recipe_a is None else tuple(recipe_a),
               recipe_b if recipe_b is None else tuple(recipe_b))
        fast = _VALIDATED_PACKED.get(key)
        if fast is not None:
            if c is not None and not same_cd:
                d.copy_(c)
            stream = current_stream_ptr(fast[1])
            ws = _split_k_workspace(d.device, stream) if fast[2] else None
            rc = lib.dg_fp8_gemm_nt_ue8m0_ws(a_data.data_ptr(), a_sf.data_ptr(), b_data.data_ptr(), b_sf.data_ptr(), d.data_ptr(), *fast[0],
                                             ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, stream)
            if rc != 3:             # (3: these POINTERS -- not part of the signature -- are off the kernels' alignment: the full path has the fallbacks)
                check(rc)
                return
            return _fp8_gemm_nt_packed_ue8m0(a_data, a_sf, b_data, b_sf, d, d if c is not None else None, recipe, recipe_a, recipe_b)
        return _fp8_gemm_nt_packed_ue8m0(a_data, a_sf, b_data, b_sf, d, c, recipe, recipe_a, recipe_b, key)
    if a_sf.dtype == torch.int or b_sf.dtype == torch.int or _casts_to_ue8m0(a_sf, b_sf, disable_ue8m0_cast):
        return _fp8_gemm_nt_packed_ue8m0(a_data, a_sf, b_data, b_sf, d, c, recipe, recipe_a, recipe_b)
    same_cd = c is not None and c.data_ptr() == d.data_ptr()
    key = (_sig(a_data), _sig(a_sf), _sig(b_data), _sig(b_sf), _sig(d), _sig(c), same_cd,
           recipe if recipe is None else tuple(recipe), recipe_a if recipe_a is None else tuple(recipe_a),
           recipe_b if recipe_b is None else tuple(recipe_b))         # (FP32 scales consumed as FP32: the mode / keyword took the other exit)
    plan = _VALIDATED_DENSE.get(key)
    if plan is not None:
        m, n, k, gran_n, sfa_ready, fast_args = plan
        if c is not None and not same_cd:
            d.copy_(c)
        if fast_args is not None:
            # K-major operands, SFA already in the kernel's layout: every integer argument of the C call is a
            # function of the signature -- only the five pointers and the stream are read per call (host path of a decode-sized
            # call: 12.5 -> ~8 us; the kernel of m = 1, 4096 x 7168 takes 7.4 us)
            stream = current_stream_ptr(fast_args[1])
            # (round 6: shapes the library cuts along K -- fast_args[2] -- take the stream's scratch buffer on this path too: 18.4 -> 9 us of host
            #  time per call of a 192 x 4096 x 7168 problem whose kernel takes 24)
            ws = _split_k_workspace(d.device, stream) if fast_args[2] else None
            check(lib.dg_fp8_gemm_nt_ws(a_data.data_ptr(), a_sf.data_ptr(), b_data.data_ptr(), b_sf.data_ptr(), d.data_ptr(), *fast_args[0],
                                        ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, stream))
            return
        sfa = a_sf if sfa_ready else get_mn_major_tma_aligned_tensor(a_sf)
        a_data, b_data = _dense_operands(a_data, b_data, sfa, gran_n, m, n, k)
        _call_dense(a_data, sfa, b_data, b_sf, d, c, m, n, k, gran_n)
        return
    major_check(a_data), major_check(b_data)
    check_major_type_cd(d)
    m, k = _check_ab_fp8(a_data, 2)
    n, k_ = _check_ab_fp8(b_data, 2)
    host_assert(d.dim() == 2, 'd.dim() == 2')
    host_assert((m, n) == tuple(d.shape) and k == k_, 'm == m_ and n == n_ and k == k_')
    host_assert(d.dtype in (torch.bfloat16, torch.float), 'd.scalar_type() == torch::kBFloat16 or d.scalar_type() == torch::kFloat')
    if _early_return(m, n, k, d, c):
        return
    # A row-major SFA [m, k / 128] (how the reference's callers hold it, tests/test_fp8_fp4.py:45-55) stays as it is when the kernel this
    # call will run reads it in place (the C side decides: dg_dense_rowmajor_sfa_native) -- no transpose launch in front of the GEMM
    rm_native = bool(a_sf.dtype == torch.float and b_sf.dtype == torch.float and a_sf.dim() == 2 and a_sf.is_contiguous() and
                     a_sf.size(1) > 1 and a_data.stride(-1) == 1 and b_data.stride(-1) == 1 and
                     a_data.stride(0) % 16 == 0 and b_data.stride(0) % 16 == 0 and a_data.data_ptr() % 16 == 0 and b_data.data_ptr() % 16 == 0 and
                     lib.dg_dense_rowmajor_sfa_native(m, n, k))
    sfa, sfb, gran_n = transform_sf_pair_into_required_layout(a_sf, b_sf, m, n, k, recipe, recipe_a, recipe_b,
                                                              None, None, True, None, rm_native)     # (FP32 scales stay FP32 on this exit)
    if rm_native and gran_n != 128:         # (recipe (1, 1, 128): the per-column kernels want the MN-major layout)
        sfa = get_mn_major_tma_aligned_tensor(sfa)
    require_device(a_data, b_data, sfa, sfb, d)
    if sfb is b_sf and len(_VALIDATED_DENSE) < 4096:
        fast_args = None
        if sfa is a_sf and a_data.stride(-1) == 1 and b_data.stride(-1) == 1:
            fast_args = ((m, n, k, a_data.stride(0), a_data.stride(1), b_data.stride(0), b_data.stride(1), sfa.stride(0), sfa.stride(1),
                          sfb.stride(0), sfb.stride(1), gran_n, d.stride(0), _dtype_code(d), int(c is not None)),
                         d.device.index if d.device.index is not None else -1,
                         bool(lib.dg_dense_wants_workspace(m, n, k, 0, 0, gran_n)))
        _VALIDATED_DENSE[key] = (m, n, k, gran_n, sfa is a_sf, fast_args)
    a_data, b_data = _dense_operands(a_data, b_data, sfa, gran_n, m, n, k)
    _call_dense(a_data, sfa, b_data, sfb, d, c, m, n, k, gran_n)


def fp8_gemm_nn(a, b, d, c=None, recipe=None, recipe_a=None, recipe_b=None, compiled_dims='nk', disable_ue8m0_cast=False) -> None:
    """``b = (B [K,N], SFB [K/128, N/128])``: a transposed view of NT (csrc/apis/gemm.hpp:126-137)."""
    fp8_gemm_nt(a, (b[0].transpose(0, 1), b[1].transpose(0, 1)), d, c, recipe, recipe_a, recipe_b, compiled_dims, disable_ue8m0_cast)


def fp8_gemm_tn(a, b, d, c=None, recipe=None, recipe_a=None, recipe_b=None, compiled_dims='mn', disable_ue8m0_cast=False) -> None:
    fp8_gemm_nt((a[0].transpose(0, 1), a[1].transpose(0, 1)), (b[0].transpose(0, 1), b[1].transpose(0, 1)),
                d, c, recipe, recipe_a, recipe_b, compiled_dims, disable_ue8m0_cast)


def fp8_gemm_tt(a, b, d, c=None, recipe=None, recipe_a=None, recipe_b=None, compiled_dims='mn', disable_ue8m0_cast=False) -> None:
    fp8_gemm_nt((a[0].transpose(0, 1), a[1].transpose(0, 1)), b, d, c, recipe, recipe_a, recipe_b, compiled_dims, disable_ue8m0_cast)


def m_grouped_fp8_gemm_nt_contiguous(a: TensorPair, b: TensorPair, d: torch.Tensor, grouped_layout: torch.Tensor,
                                     recipe=None, recipe_a=None, recipe_b=None, compiled_dims: str = 'nk',
                                     disable_ue8m0_cast: bool = False, use_psum_layout: bool = False,
                                     ensure_zero_padding: bool = True,
                                     expected_m_for_psum_layout: Optional[int] = None) -> None:
    """Rows of ``a [M,K]`` are grouped contiguously (each group padded to the M alignment); ``b [G,N,K]``."""
    (a_data, a_sf), (b_data, b_sf) = a, b
    host_assert(is_k_major(a_data), 'major_a == cute::UMMA::Major::K')
    major_check(b_data)
    host_assert(grouped_layout.is_contiguous(), 'grouped_layout.is_contiguous()')
    m, k = _check_ab_fp8(a_data, 2)
    num_groups, n, k_ = _check_ab_fp8(b_data, 3)
    host_assert(d.dim() == 2, 'd.dim() == 2')
    host_assert((m, n) == tuple(d.shape) and k == k_, 'm == m_ and n == n_ and k == k_')
    host_assert(n > 0 and k > 0 and num_groups > 0, 'n > 0 and k > 0 and num_groups > 0')
    host_assert(d.dtype == torch.bfloat16, 'd.scalar_type() == torch::kBFloat16')
    host_assert(grouped_layout.dtype == torch.int, 'grouped_layout.scalar_type() == torch::kInt')
    host_assert(grouped_layout.dim() == 1, 'grouped_layout.dim() == 1')
    if use_psum_layout:
        host_assert(grouped_layout.numel() == num_groups, 'num_groups == num_groups_')
    else:
        host_assert(grouped_layout.numel() == m, 'm == m__')
        host_assert(expected_m_for_psum_layout is None, 'not expected_m_for_psum_layout.has_value()')
    check_major_type_cd(d)
    if m == 0:
        return
    if a_sf.dtype == torch.int or b_sf.dtype == torch.int or (_casts_to_ue8m0(a_sf, b_sf, disable_ue8m0_cast) and k % 128 == 0):
        # packed UE8M0 scales (SM100 format, recipe (1, 1, 128)) or FP32 scales cast to them ('sm100' mode): hardware-scaled MFMA kernels
        # (csrc/apis/gemm.hpp:213-216: the psum layout goes to the SFA pack so that it skips the gap rows)
        sfa, sfb, gran_k = _packed_sf_pair(a_sf, b_sf, m, n, k, recipe, recipe_a, recipe_b, None, num_groups,
                                           grouped_layout if use_psum_layout and a_sf.dtype == torch.float else None)
        require_device(a_data, b_data, sfa, sfb, d, grouped_layout)
        if gran_k == 32:
            # granularity 32 (the SM100 MX recipe): the four-wave G32 kernels, K-major weights (MN-major ones are re-majored), no K split
            b_km = _remajor(b_data) if b_data.stride(-1) != 1 else b_data
            check(lib.dg_m_grouped_fp8_gemm_nt_contiguous_ue8m0_g32(
                a_data.data_ptr(), sfa.data_ptr(), b_km.data_ptr(), sfb.data_ptr(), d.data_ptr(), grouped_layout.data_ptr(),
                num_groups, m, n, k, a_data.stride(0), a_data.stride(1), b_km.stride(0), b_km.stride(1), b_km.stride(2),
                sfa.stride(0), sfa.stride(1), sfb.stride(0), sfb.stride(1), sfb.stride(2), d.stride(0), int(use_psum_layout),
                runtime.get_mk_alignment_for_contiguous_layout(), 0, 0, current_stream_ptr()))
            return
        # MN-major weights ([G, K, N]: the nn form) stay as they are where the library reads them in place and that beats a pass over every
        # group's weights (dg_ue8m0_grouped_operand_plan: its predicates and model, no copy here)
        b_km = b_data
        if b_data.stride(-1) != 1 and lib.dg_ue8m0_grouped_operand_plan(
                a_data.data_ptr(), b_data.data_ptr(), num_groups, m, n, k, a_data.stride(0), b_data.stride(0), b_data.stride(1),
                b_data.stride(2), int(use_psum_layout), runtime.get_mk_alignment_for_contiguous_layout()):
            b_km = _remajor(b_data)
        # the K-split scratch buffer only where the library's group-relative tiling would cut its remainder tiles along K (it answers without
        # launching: dg_select_config) -- other packed-scale grouped calls neither create nor pass it
        stream = current_stream_ptr()
        picked = lib.dg_select_config(2 if use_psum_layout else 1, m, n, k, num_groups, 0, 0, int(b_km.stride(-1) != 1), 128,
                                      runtime.get_mk_alignment_for_contiguous_layout(), 1, 1)
        workspace = _split_k_workspace(d.device, stream) if b'_tab_' in picked and lib.dg_get_forced_config() == b'auto' else None
        check(lib.dg_m_grouped_fp8_gemm_nt_contiguous_ue8m0_ws(
            a_data.data_ptr(), sfa.data_ptr(), b_km.data_ptr(), sfb.data_ptr(), d.data_ptr(), grouped_layout.data_ptr(),
            num_groups, m, n, k, a_data.stride(0), a_data.stride(1), b_km.stride(0), b_km.stride(1), b_km.stride(2),
            sfa.stride(0), sfa.stride(1), sfb.stride(0), sfb.stride(1), sfb.stride(2), d.stride(0), int(use_psum_layout),
            runtime.get_mk_alignment_for_contiguous_layout(), workspace.data_ptr() if workspace is not None else 0,
            workspace.numel() if workspace is not None else 0, stream))
        return
    if _casts_to_ue8m0(a_sf, b_sf, disable_ue8m0_cast):
        # 'sm100' mode with a K tail: the FP32-scale kernels on TRUNCATED scales -- the values the cast branch keeps -- as the dense entry does
        # (one mode, one arithmetic, whatever the entry point and K)
        a_sf, b_sf = _truncate_to_ue8m0(a_sf), _truncate_to_ue8m0(b_sf)
    sfa, sfb, gran_n = transform_sf_pair_into_required_layout(a_sf, b_sf, m, n, k, recipe, recipe_a, recipe_b,
                                                              None, num_groups, True)
    host_assert(gran_n == 128, 'gran_n == 128 (the grouped kernels read one SFB value per 128 columns; per-column SFB takes packed UE8M0 scales)')
    require_device(a_data, b_data, sfa, sfb, d, grouped_layout)
    if b_data.stride(-1) != 1 and _operand_plan(2 if use_psum_layout else 1, a_data, b_data, sfa, gran_n, m, n, k,
                                                runtime.get_mk_alignment_for_contiguous_layout()) & 2:
        b_data = _as_k_major(b_data, m * n * k)
    stream = current_stream_ptr()
    # the K-split scratch buffer only where the library's own selection would cut this problem along K (it answers without launching:
    # dg_select_config with has_workspace = 1) -- ordinary grouped calls neither create nor pass the 64 MiB buffer
    picked = lib.dg_select_config(2 if use_psum_layout else 1, m, n, k, num_groups, 0, 0, int(b_data.stride(-1) != 1), 128,
                                  runtime.get_mk_alignment_for_contiguous_layout(), 1, 0)
    # (a K-split form forced by name -- tuning runs, tests -- gets the buffer too: dg_select_config answers for the automatic selection)
    workspace = _split_k_workspace(d.device, stream) if (b'_sk_' in picked or b'_tab_' in picked or b'_sk_' in lib.dg_get_forced_config()) else None
    check(lib.dg_m_grouped_fp8_gemm_nt_contiguous_ws(
        a_data.data_ptr(), sfa.data_ptr(), b_data.data_ptr(), sfb.data_ptr(), d.data_ptr(), grouped_layout.data_ptr(),
        num_groups, m, n, k, a_data.stride(0), a_data.stride(1),
        b_data.stride(0), b_data.stride(1), b_data.stride(2), sfa.stride(0), sfa.stride(1),
        sfb.stride(0), sfb.stride(1), sfb.stride(2), d.stride(0), int(use_psum_layout),
        runtime.get_mk_alignment_for_contiguous_layout(), workspace.data_ptr() if workspace is not None else 0,
        workspace.numel() if workspace is not None else 0, stream))


def m_grouped_fp8_gemm_nn_contiguous(a, b, d, grouped_layout, recipe=None, recipe_a=None, recipe_b=None,
                                     compiled_dims='nk', disable_ue8m0_cast=False, use_psum_layout=False,
                                     ensure_zero_padding=True) -> None:
    """``b = (B [G,K,N], SFB [G,K/128,N/128])`` (csrc/apis/gemm.hpp:234-248)."""
    m_grouped_fp8_gemm_nt_contiguous(a, (b[0].transpose(1, 2), b[1].transpose(1, 2)), d, grouped_layout, recipe, recipe_a,
                                     recipe_b, compiled_dims, disable_ue8m0_cast, use_psum_layout, ensure_zero_padding, None)


def m_grouped_fp8_gemm_nt_masked(a: TensorPair, b: TensorPair, d: torch.Tensor, masked_m: torch.Tensor, expected_m: int,
                                 recipe=None, recipe_a=None, recipe_b=None, compiled_dims: str = 'nk',
                                 disable_ue8m0_cast: bool = False) -> None:
    """``a [G,M,K]``, ``b [G,N,K]``, ``d [G,M,N]``; only ``d[g, :masked_m[g]]`` is written; ``masked_m`` stays on
    the device, ``expected_m`` is a tuning hint."""
    (a_data, a_sf), (b_data, b_sf) = a, b
    if a_sf.dtype == torch.int or b_sf.dtype == torch.int or (_casts_to_ue8m0(a_sf, b_sf, disable_ue8m0_cast) and a_data.size(-1) % 128 == 0):
        return _m_grouped_masked_packed_ue8m0(a_data, a_sf, b_data, b_sf, d, masked_m, expected_m, recipe, recipe_a, recipe_b)
    if _casts_to_ue8m0(a_sf, b_sf, disable_ue8m0_cast):         # 'sm100' mode, K tail: truncated scales on the FP32-scale kernels (see the contiguous entry)
        a_sf, b_sf = _truncate_to_ue8m0(a_sf), _truncate_to_ue8m0(b_sf)
    key = (_sig(a_data), _sig(a_sf), _sig(b_data), _sig(b_sf), _sig(d), _sig(masked_m), expected_m > 0,
           recipe if recipe is None else tuple(recipe), recipe_a if recipe_a is None else tuple(recipe_a),
           recipe_b if recipe_b is None else tuple(recipe_b))
    plan = _VALIDATED_MASKED.get(key)
    if plan is not None:
        num_groups, m, n, k, sfa_ready = plan
        sfa, sfb = (a_sf if sfa_ready else get_mn_major_tma_aligned_tensor(a_sf)), b_sf
    else:
        host_assert(is_k_major(a_data) and is_k_major(b_data), 'major_a == cute::UMMA::Major::K and major_b == cute::UMMA::Major::K')
        host_assert(masked_m.is_contiguous(), 'masked_m.is_contiguous()')
        num_groups, m, k = _check_ab_fp8(a_data, 3)
        num_groups_, n, k_ = _check_ab_fp8(b_data, 3)
        host_assert(d.dim() == 3, 'd.dim() == 3')
        host_assert(num_groups == num_groups_ == d.size(0) == masked_m.numel(),
                    'num_groups == num_groups_ and num_groups == num_groups__ and num_groups == num_groups___')
        host_assert((m, n) == tuple(d.shape[1:]) and k == k_, 'm == m_ and n == n_ and k == k_')
        host_assert(expected_m > 0 and m > 0 and n > 0 and k > 0 and num_groups > 0,
                    'expected_m > 0 and m > 0 and n > 0 and k > 0 and num_groups > 0')
        host_assert(d.dtype == torch.bfloat16, 'd.scalar_type() == torch::kBFloat16')
        host_assert(masked_m.dtype == torch.int, 'masked_m.scalar_type() == torch::kInt')
        check_major_type_cd(d)
        sfa, sfb, gran_n = transform_sf_pair_into_required_layout(a_sf, b_sf, m, n, k, recipe, recipe_a, recipe_b,
                                                                  num_groups, num_groups, True)
        host_assert(gran_n == 128, 'gran_n == 128 (the grouped kernels read one SFB value per 128 columns; per-column SFB takes packed UE8M0 scales)')
        require_device(a_data, b_data, sfa, sfb, d, masked_m)
        if sfb is b_sf and len(_VALIDATED_MASKED) < 4096:
            _VALIDATED_MASKED[key] = (num_groups, m, n, k, sfa is a_sf)
    check(lib.dg_m_grouped_fp8_gemm_nt_masked(
        a_data.data_ptr(), sfa.data_ptr(), b_data.data_ptr(), sfb.data_ptr(), d.data_ptr(), masked_m.data_ptr(),
        num_groups, m, n, k, int(expected_m),
        a_data.stride(0), a_data.stride(1), a_data.stride(2), b_data.stride(0), b_data.stride(1), b_data.stride(2),
        sfa.stride(0), sfa.stride(1), sfa.stride(2), sfb.stride(0), sfb.stride(1), sfb.stride(2),
        d.stride(0), d.stride(1), current_stream_ptr()))


# ---------------------------------------------------------------------------------------------------------------------
# K-grouped contiguous GEMM (MoE weight gradients): d[g] = c[g] + A_g^T-ish products over group g's K range, FP32.
# Reference: csrc/apis/gemm.hpp:299-400 (operators), :48-69 (argument checks), tests/generators.py:436-477 (layouts).
# ---------------------------------------------------------------------------------------------------------------------
_KGROUPED_BLOCKS, _KGROUPED_COLUMNS, _KGROUPED_ROWS = 0, 1, 2


def _check_k_grouped_args(ks_cpu, grouped_layout: torch.Tensor, num_groups: int, use_psum_layout: bool, k_alignment: int,
                          sum_k_if_ks_cpu_missing: int = 0) -> int:
    """csrc/apis/gemm.hpp:48-69, same order and condition text: the sum of the host-side K extents, or -- psum layout with ``ks_cpu``
    missing or empty, the K ranges then live on the device only -- the operands' own extent."""
    host_assert(grouped_layout.is_contiguous(), 'grouped_layout.is_contiguous()')
    host_assert(grouped_layout.dtype == torch.int, 'grouped_layout.scalar_type() == torch::kInt')
    host_assert(grouped_layout.numel() == num_groups, 'static_cast<int>(grouped_layout.numel()) == num_groups')
    if ks_cpu is not None and len(ks_cpu) > 0:
        host_assert(len(ks_cpu) == num_groups, 'static_cast<int>(ks_cpu.value().size()) == num_groups')
        for k in ks_cpu:
            host_assert(k % k_alignment == 0, 'k % k_alignment == 0')
        return int(sum(ks_cpu))
    host_assert(use_psum_layout, 'use_psum_layout')
    return sum_k_if_ks_cpu_missing


def _k_grouped_sf(sf: torch.Tensor, mn: int, sum_k: int) -> torch.Tensor:
    """FP32 per-channel scales of a K-grouped operand as ``[mn, sum_k / 128]`` -> MN-major, 16-byte aligned rows
    (transform_k_grouped_sf_into_required_layout, csrc/apis/layout.hpp:95-121, FP32 branch)."""
    host_assert(sf.dim() == 2 and sf.dtype == torch.float, 'sf.dim() == 2 and sf.scalar_type() == torch::kFloat')
    host_assert(tuple(sf.shape) == (mn, sum_k // 128), 'sf.size(0) == mn and sf.size(1) == sum_k / gran_k')
    return get_mn_major_tma_aligned_tensor(sf)


_NO_NATIVE_KERNEL = 3       # dg_k_grouped_fp8_gemm_nt_contiguous(DG_KGROUPED_ROWS): conditions not met, nothing was launched


def _k_grouped_launch(a_data, sfa, b_data, sfb, d, m, n, ks, layout, a_ld, b_ld, may_decline: bool = False) -> bool:
    """False (only with ``may_decline``) = the library has no kernel for this operand form and launched nothing."""
    import ctypes
    require_device(a_data, b_data, sfa, sfb, d)
    ks_arr = (ctypes.c_int32 * len(ks))(*[int(k) for k in ks])
    rc = lib.dg_k_grouped_fp8_gemm_nt_contiguous(
        a_data.data_ptr(), sfa.data_ptr(), b_data.data_ptr(), sfb.data_ptr(), d.data_ptr(), m, n,
        ctypes.cast(ks_arr, ctypes.c_void_p), len(ks), layout, a_ld, b_ld,
        sfa.stride(0), sfa.stride(1), sfb.stride(0), sfb.stride(1), current_stream_ptr())
    if may_decline and rc == _NO_NATIVE_KERNEL:
        return False
    check(rc)
    return True


def _k_grouped_psum_launch(a_data, sfa, b_data, sfb, d, m, n, total_k, psum_layout, layout, a_ld, b_ld, k_alignment: int = 128) -> bool:
    """dg_k_grouped_fp8_gemm_tn_psum[_aligned]; False = the library has no kernel for this operand form and launched nothing."""
    require_device(a_data, b_data, sfa, sfb, d, psum_layout)
    rc = lib.dg_k_grouped_fp8_gemm_tn_psum_aligned(
        a_data.data_ptr(), sfa.data_ptr(), b_data.data_ptr(), sfb.data_ptr(), d.data_ptr(), m, n, total_k, psum_layout.data_ptr(),
        psum_layout.numel(), layout, a_ld, b_ld, sfa.stride(0), sfa.stride(1), sfb.stride(0), sfb.stride(1), k_alignment, current_stream_ptr())
    if rc == _NO_NATIVE_KERNEL:
        return False
    check(rc)
    return True


def _k_grouped_packed_sf(sf: torch.Tensor, mn: int, ks, grouped_layout: torch.Tensor, num_groups: int, gran_k: int, k_alignment: int,
                         use_psum_layout: bool) -> torch.Tensor:
    """transform_k_grouped_sf_into_required_layout on SM100 (csrc/apis/layout.hpp:92-121): int tensors are the packed words already and are only
    checked (check_k_grouped_packed_ue8m0_tensor, csrc/jit_kernels/impls/smxx_layout.hpp:319-355); FP32 tensors ``[sf_k, mn]`` -- power-of-two
    values, per group ceil(k_g / gran_k) compact rows -- are packed on the device, four exponent bytes per word, every group starting a new word row
    (get_k_grouped_mn_major_tma_aligned_packed_ue8m0_tensor, :255-317 -> dg_pack_sf_k_grouped_ue8m0)."""
    host_assert(sf.dim() == 2, 'sf.dim() == 2')
    host_assert(sf.is_contiguous(), 'sf.is_contiguous()')
    host_assert(num_groups <= 128 and mn % 4 == 0, 'num_groups <= 128 and mn % 4 == 0')
    host_assert(sf.size(1) == mn, 'sf.size(1) == mn')
    has_ks = ks is not None and len(ks) > 0
    host_assert(has_ks or use_psum_layout, 'use_psum_layout')
    if sf.dtype == torch.int:
        host_assert(sf.size(0) > 0, 'packed_sf_k > 0')
        if has_ks and not use_psum_layout:
            host_assert(sf.size(0) >= sum(-(-int(k) // (gran_k * 4)) for k in ks), 'packed_sf_k >= aligned_packed_sf_k')
        return sf
    host_assert(sf.dtype == torch.float, 'sf.scalar_type() == torch::kFloat or sf.scalar_type() == torch::kInt')
    sf_k = int(sf.size(0))
    if has_ks:
        packed_sf_k = sum(-(-int(k) // (gran_k * 4)) for k in ks)
        host_assert(use_psum_layout or sum(-(-int(k) // gran_k) for k in ks) == sf_k, 'use_psum_layout or ref_sf_k == sf_k')
    else:
        packed_sf_k = (sf_k + num_groups * 3) // 4
    out = torch.empty((packed_sf_k, mn), dtype=torch.int, device=sf.device)
    if packed_sf_k == 0:
        return out
    require_device(sf, grouped_layout)
    check(lib.dg_pack_sf_k_grouped_ue8m0(sf.data_ptr(), out.data_ptr(), grouped_layout.data_ptr(), num_groups, mn, sf_k, packed_sf_k, gran_k,
                                         k_alignment, int(use_psum_layout), current_stream_ptr()))
    return out


def _k_grouped_tn_packed_ue8m0(a_data, a_sf, b_data, b_sf, d, ks, grouped_layout, m: int, n: int, sum_k: int, gran_k: int, k_alignment: int,
                               use_psum_layout: bool) -> None:
    """The reference's SM100 form of the K-grouped GEMM (csrc/apis/gemm.hpp:333-342 -> sm100_k_grouped_fp8_gemm_1d1d): UE8M0 scales of
    granularity 128 or 32 along K, one exponent per column of the MN-major operands -- the MX block format of the scaled MFMA, so the whole
    group's K range accumulates in the matrix core with no FP32 promotion (e8_quad_kg_*).  The operands are re-majored once ([sum_k, mn] ->
    [mn, sum_k]: the four-wave kernels read K-major rows)."""
    num_groups = int(d.size(0))
    sfa = _k_grouped_packed_sf(a_sf, m, ks, grouped_layout, num_groups, gran_k, k_alignment, use_psum_layout)
    sfb = _k_grouped_packed_sf(b_sf, n, ks, grouped_layout, num_groups, gran_k, k_alignment, use_psum_layout)
    host_assert(sum_k % 16 == 0, 'sum_k % 16 == 0 (K-major rows of whole 16-byte chunks)')
    require_device(a_data, b_data, sfa, sfb, d, grouped_layout)
    import ctypes
    if use_psum_layout:
        ks_arr, ks_ptr, layout_ptr = None, None, grouped_layout.data_ptr()
    else:
        host_assert(num_groups <= 64, 'num_groups <= 64 (K extents handed over by value)')
        ks_arr = (ctypes.c_int32 * num_groups)(*[int(k) for k in ks])
        ks_ptr, layout_ptr = ctypes.cast(ks_arr, ctypes.c_void_p), None

    def launch(a_op, b_op, layout):
        return lib.dg_k_grouped_fp8_gemm_ue8m0(a_op.data_ptr(), sfa.data_ptr(), b_op.data_ptr(), sfb.data_ptr(), d.data_ptr(), m, n, sum_k,
                                               ks_ptr, layout_ptr, num_groups, k_alignment, gran_k, layout, a_op.stride(0), b_op.stride(0),
                                               sfa.stride(0), sfb.stride(0), current_stream_ptr())
    # the MN-major tensors as they are wherever the library takes them (transposing fragment reads: m > 128, aligned rows); it declines without
    # launching otherwise and the operands are re-majored once ([sum_k, mn] -> [mn, sum_k])
    rc = launch(a_data, b_data, _KGROUPED_ROWS)
    if rc == _NO_NATIVE_KERNEL:
        a_km, b_km = _remajor(a_data.transpose(0, 1)), _remajor(b_data.transpose(0, 1))
        rc = launch(a_km, b_km, _KGROUPED_COLUMNS)
    check(rc)


def k_grouped_fp8_gemm_nt_contiguous(a: TensorPair, b: TensorPair, d: torch.Tensor, ks_cpu, grouped_layout: torch.Tensor,
                                     c: Optional[torch.Tensor] = None, recipe: Tuple[int, int, int] = (1, 1, 128),
                                     compiled_dims: str = 'mn', use_psum_layout: bool = False) -> None:
    """``a[0]``: the groups' K-major ``[M, ks_cpu[g]]`` matrices stored one after another (flat), ``a[1]``: ``[M, sum_k / 128]``;
    same for ``b`` with N; ``d [G, M, N]`` FP32 ``= c + A_g @ B_g^T`` (csrc/apis/gemm.hpp:348-400; keyword names of the
    ``m.def`` at ``:697-710``)."""
    (a_data, a_sf), (b_data, b_sf) = a, b
    ks = ks_cpu
    host_assert(tuple(recipe) == (1, 1, 128), 'recipe == std::make_tuple(1, 1, 128)')
    # No psum on FP8 NT (csrc/apis/gemm.hpp:360)
    host_assert(not use_psum_layout and ks is not None and len(ks) > 0,
                'not use_psum_layout and ks_cpu.has_value() and not ks_cpu.value().empty()')
    host_assert(d.dim() == 3, 'd.dim() == 3')
    num_groups, m, n = (int(x) for x in d.shape)
    sum_k = _check_k_grouped_args(ks, grouped_layout, num_groups, use_psum_layout, 128)
    host_assert(a_data.dtype == torch.float8_e4m3fn and b_data.dtype == torch.float8_e4m3fn,
                'ab.scalar_type() == torch::kFloat8_e4m3fn')
    host_assert(a_data.numel() == sum_k * m, 'sum_mk == static_cast<int64_t>(sum_k) * m')
    host_assert(b_data.numel() == sum_k * n, 'sum_nk == static_cast<int64_t>(sum_k) * n')
    host_assert(a_data.is_contiguous() and b_data.is_contiguous() and d.is_contiguous(),
                'a.first.is_contiguous() and b.first.is_contiguous() and d.is_contiguous()')
    host_assert(c is not None and c.is_contiguous(), 'c.has_value() and c.value().is_contiguous()')
    host_assert(d.dtype == torch.float, 'd.scalar_type() == torch::kFloat')
    if _early_return(m, n, sum_k, d, c):
        return
    sfa, sfb = _k_grouped_sf(a_sf, m, sum_k), _k_grouped_sf(b_sf, n, sum_k)
    _k_grouped_launch(a_data, sfa, b_data, sfb, d, m, n, ks, _KGROUPED_BLOCKS, 0, 0)


def k_grouped_fp8_gemm_tn_contiguous(a: TensorPair, b: TensorPair, d: torch.Tensor, ks_cpu, grouped_layout: torch.Tensor,
                                     c: Optional[torch.Tensor] = None, recipe: Tuple[int, int, int] = (1, 1, 128),
                                     compiled_dims: str = 'mn', use_psum_layout: bool = False) -> None:
    """MN-major operands: ``a[0] [sum_k, M]``, ``a[1] [sum_k / 128, M]`` (per-channel scales), ``b`` likewise with N;
    ``d [G, M, N]`` FP32 ``= c + A_g^T @ B_g`` (csrc/apis/gemm.hpp:299-346).  The operands go into the kernel as they are wherever the
    library takes them (hardware transpose reads), else they are re-majored once by ``dg_transpose_fp8``.

    ``use_psum_layout``: ``grouped_layout[g]`` is group g's END along K, groups start at the previous end rounded up to the K alignment,
    the rows in between hold zeros (tests/generators.py:480-530) and ``ks_cpu`` may be missing -- the K ranges are then read on the
    device, no host copy of the group sizes exists.  Implemented for FP32 per-channel scales with ``gran_k`` = 128 and any K alignment that
    is a multiple of 32 (round 6: 32 / 160 / 192 / 224 of the reference's SM100 sweep -- compact scale rows counted from each group's start,
    partial last blocks).  UE8M0 scales (round 6: int tensors of packed exponent words, FP32 tensors in the ``'sm100'`` scaling-factor mode,
    and every call with ``gran_k`` = 32 -- the reference's SM100 semantics, any K alignment that is a multiple of 32, with or without the psum
    layout) run on the hardware-scaled K-grouped kernels: see ``_k_grouped_tn_packed_ue8m0``."""
    (a_data, a_sf), (b_data, b_sf) = a, b
    ks = ks_cpu
    recipe = tuple(recipe)
    host_assert(recipe[0] == 1 and recipe[1] == 1, 'std::get<0>(recipe) == 1 and std::get<1>(recipe) == 1')
    gran_k = recipe[2]
    host_assert(gran_k == 32 or gran_k == 128, 'gran_k == 32 or gran_k == 128')
    k_alignment = runtime.get_mk_alignment_for_contiguous_layout()
    host_assert(k_alignment % 32 == 0, 'k_alignment % 32 == 0')
    host_assert(d.dim() == 3, 'd.dim() == 3')
    num_groups, m, n = (int(x) for x in d.shape)
    host_assert(a_data.dim() == 2 and b_data.dim() == 2, 'a.first.dim() == 2 and b.first.dim() == 2')
    # UE8M0 scales -- packed int words, FP32 tensors in the 'sm100' scaling-factor mode (the reference's cast, csrc/apis/layout.hpp:112-114), and
    # granularity 32 always (it only exists in that format): the hardware-scaled K-grouped kernels, the reference's SM100 semantics
    packed = (a_sf.dtype == torch.int and b_sf.dtype == torch.int) or gran_k == 32 or _casts_to_ue8m0(a_sf, b_sf, False)
    # (K extents in whole scale blocks -- the reference's `k % k_alignment == 0` at 128 -- except in the psum form, whose ranges are read on the
    #  device, and with UE8M0 scales: any alignment that is a multiple of 32, round 6)
    general = (use_psum_layout and k_alignment != 128) or packed
    sum_k = _check_k_grouped_args(ks, grouped_layout, num_groups, use_psum_layout, k_alignment if general else 128, int(a_data.size(0)))
    host_assert(a_data.dtype == torch.float8_e4m3fn and b_data.dtype == torch.float8_e4m3fn,
                'ab.scalar_type() == torch::kFloat8_e4m3fn')
    host_assert(tuple(a_data.shape) == (sum_k, m) and tuple(b_data.shape) == (sum_k, n),
                'm == m_ and n == n_ and sum_k == sum_k_ and sum_k == sum_k__')
    host_assert(a_data.is_contiguous() and b_data.is_contiguous() and d.is_contiguous(),
                'a.first.is_contiguous() and b.first.is_contiguous() and d.is_contiguous()')
    host_assert(c is not None and c.is_contiguous(), 'c.has_value() and c.value().is_contiguous()')
    host_assert(d.dtype == torch.float, 'd.scalar_type() == torch::kFloat')
    if _early_return(m, n, sum_k, d, c):
        return
    host_assert(a_sf.dim() == 2 and b_sf.dim() == 2, 'sf.dim() == 2')
    if packed:
        _k_grouped_tn_packed_ue8m0(a_data, a_sf, b_data, b_sf, d, ks, grouped_layout, m, n, sum_k, gran_k, k_alignment, use_psum_layout)
        return
    if general:
        # K alignment != 128 (the reference's SM100 sweep: 32 / 160 / 192 / 224 at gran_k = 128, tests/generators.py:192-194): groups start at
        # multiples of the alignment, their scale rows are compact and count from the group's own start (ceil(extent / 128) per non-empty group:
        # scheduler/gemm.cuh:247), the last block of a group is partial.  The ranges live on the device: the scale-row count cannot be checked
        # here beyond its upper bound.  MN-major operands in place only (the reference's own restriction, tests/generators.py:497).
        host_assert(sum_k % k_alignment == 0, 'sum_k % k_alignment == 0')
        for sf, mn in ((a_sf, m), (b_sf, n)):
            host_assert(sf.dtype == torch.float and sf.size(1) == mn and sf.size(0) <= sum_k // 128 + num_groups,
                        'sf.scalar_type() == torch::kFloat and sf.size(1) == mn and sf.size(0) <= sum over groups of ceil_div(k_g, gran_k)')
        sfa, sfb = get_mn_major_tma_aligned_tensor(a_sf.transpose(0, 1)), get_mn_major_tma_aligned_tensor(b_sf.transpose(0, 1))
        require_device(grouped_layout)
        if not _k_grouped_psum_launch(a_data, sfa, b_data, sfb, d, m, n, sum_k, grouped_layout, _KGROUPED_ROWS, a_data.stride(0), b_data.stride(0),
                                      k_alignment):
            raise RuntimeError('Assertion error (gemm.py): Unsupported architecture (the psum layout of the K-grouped GEMM with a K alignment of '
                               f'{k_alignment} needs m > 64 and MN-major operands and scales with 16-byte aligned rows: ' +
                               lib.dg_last_error().decode() + ')')
        return
    host_assert(sum_k % 128 == 0, 'sum_k % 128 == 0 (the operands end on a scale-block boundary)')
    sfa, sfb = _k_grouped_sf(a_sf.transpose(0, 1), m, sum_k), _k_grouped_sf(b_sf.transpose(0, 1), n, sum_k)
    if use_psum_layout:
        # K ranges from the device tensor (with or without ks_cpu: what the reference's kernel does, scheduler/gemm.cuh:74-85)
        require_device(grouped_layout)
        if _k_grouped_psum_launch(a_data, sfa, b_data, sfb, d, m, n, sum_k, grouped_layout, _KGROUPED_ROWS, a_data.stride(0), b_data.stride(0)):
            return
        a_km, b_km = _remajor(a_data.transpose(0, 1)), _remajor(b_data.transpose(0, 1))
        if _k_grouped_psum_launch(a_km, sfa, b_km, sfb, d, m, n, sum_k, grouped_layout, _KGROUPED_COLUMNS, a_km.stride(0), b_km.stride(0)):
            return
        # no single-launch kernel for this problem (m <= 64 ...): one launch per group needs the extents on the host
        host_assert(ks is not None and len(ks) > 0, 'ks_cpu.has_value() (this problem size has no device-side K-range kernel)')
        _k_grouped_launch(a_km, sfa, b_km, sfb, d, m, n, ks, _KGROUPED_COLUMNS, a_km.stride(0), b_km.stride(0))
        return
    # MN-major operands straight into the kernel (LDS-DMA of [k][m] rows, hardware transpose reads for the fragments) wherever the
    # library's own DG_KGROUPED_ROWS conditions hold -- it declines without launching otherwise, and the operands are re-majored
    if _k_grouped_launch(a_data, sfa, b_data, sfb, d, m, n, ks, _KGROUPED_ROWS, a_data.stride(0), b_data.stride(0), may_decline=True):
        return
    a_km, b_km = _remajor(a_data.transpose(0, 1)), _remajor(b_data.transpose(0, 1))       # [M, sum_k], [N, sum_k]: K-major
    _k_grouped_launch(a_km, sfa, b_km, sfb, d, m, n, ks, _KGROUPED_COLUMNS, a_km.stride(0), b_km.stride(0))


def fp8_gemm_nt_skip_head_mid(a: TensorPair, b: TensorPair, d: torch.Tensor, head_splits: Tuple[int, int, int],
                              recipe: Optional[Tuple[int, int, int]] = None, compiled_dims: str = 'nk',
                              disable_ue8m0_cast: bool = False) -> None:
    """``fp8_gemm_nt`` whose N columns are heads of ``left + right`` columns written into a ``d`` that reserves ``mid``
    untouched columns inside every head: ``d [M, N + N / (left + right) * mid]`` (csrc/apis/attention.hpp:19-73)."""
    (a_data, a_sf), (b_data, b_sf) = a, b
    host_assert(is_k_major(a_data) and is_k_major(b_data), 'major_a == cute::UMMA::Major::K and major_b == cute::UMMA::Major::K')
    check_major_type_cd(d)
    m, k = _check_ab_fp8(a_data, 2)
    n, k_ = _check_ab_fp8(b_data, 2)
    host_assert(d.dim() == 2, 'd.dim() == 2')
    host_assert(m == d.size(0) and k == k_, 'm == m_ and k == k_')
    host_assert(n > 0 and k > 0, 'n > 0 and k > 0')
    host_assert(d.dtype in (torch.bfloat16, torch.float), 'd.scalar_type() == torch::kBFloat16 or d.scalar_type() == torch::kFloat')
    left, mid, right = (int(x) for x in head_splits)
    host_assert(n % (left + right) == 0 and d.size(1) == n + n // (left + right) * mid,
                'n % (left + right) == 0 and n_ == n + n / (left + right) * mid')
    if m == 0:
        return
    # (the head-split epilogue lives on the FP32-scale kernels; in the 'sm100' scaling-factor mode they get the TRUNCATED scales -- the
    #  values the reference's cast branch keeps, csrc/apis/layout.hpp:48-54 -- so that the mode means one arithmetic at every entry point)
    if _casts_to_ue8m0(a_sf, b_sf, disable_ue8m0_cast):
        a_sf, b_sf = _truncate_to_ue8m0(a_sf), _truncate_to_ue8m0(b_sf)
    sfa, sfb, gran_n = transform_sf_pair_into_required_layout(a_sf, b_sf, m, n, k, recipe, None, None, None, None, True)
    require_device(a_data, b_data, sfa, sfb, d)
    check(lib.dg_fp8_gemm_nt_skip_head_mid(
        a_data.data_ptr(), sfa.data_ptr(), b_data.data_ptr(), sfb.data_ptr(), d.data_ptr(), m, n, k,
        a_data.stride(0), a_data.stride(1), b_data.stride(0), b_data.stride(1),
        sfa.stride(0), sfa.stride(1), sfb.stride(0), sfb.stride(1), gran_n,
        d.stride(0), _dtype_code(d), left, mid, right, current_stream_ptr()))
