"""Expert-MLP hand-off (the single-GPU half of the reference's Mega-MoE): GEMM1 -> SwiGLU -> per-token FP8 re-quantisation -> GEMM2
without the BF16 intermediate going through memory.

Reference: ``deep_gemm/mega/__init__.py`` (``transform_weights_for_mega_moe`` :131-151, ``fp8_fp4_mega_moe`` :155-173), kernel
``deep_gemm/include/deep_gemm/impls/sm100_fp8_fp4_mega_moe.cuh`` (the L1 -> L2 hand-off in GEMM1's epilogue), host driver
``csrc/apis/mega.hpp:30-159``.  What is here: the fused L1 operator (``m_grouped_fp8_gemm_nt_masked_swiglu``), the weight transform
this library's kernel wants, ``fp8_mega_moe_local`` = fused L1 + masked L2 on the tokens already resident on this GPU, and the
reference-shaped ``fp8_mega_moe(y, l1, l2, sym_buffer)`` for any group size.  With more than one rank the dispatch / combine legs are
either (round 6, ``SymmBuffer(..., p2p=True)``) IN-KERNEL over peer-mapped memory -- every rank maps every peer's symmetric region
(``hipIpcOpenMemHandle``), a dispatch kernel pushes each token row straight into the owner's masked layout, arrival flags with system-scope
release / acquire and bounded waits, results return as remote BF16 row writes, a local top-k sum: five launches per step, no collective
(csrc/fp8_gemm_moe.hpp; reference: sm100_fp8_fp4_mega_moe.cuh:357-405, 523-595, comm/barrier.cuh:47-83) -- or fixed-shape RCCL
all-to-alls (``deepgemm_amd/ep.py``: no host synchronisation, capturable), the fallback where peers cannot be mapped.  The routing weight
travels with each row and is applied before the re-quantisation exactly as at world size 1.  The protocol is exercised on ONE GPU by two
processes that map each other's regions (tests/test_mega_p2p_gpu.py); its xGMI timing is unmeasured (no multi-GPU node in any round).
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
    if not isinstance(l1_weights, tuple):
        raise RuntimeError('transform_weights_for_mega_moe: BF16 weights (bf16_mega_moe) are outside this library (FP8 GEMM path only)')
    w1, sf1 = l1_weights
    if sf1.dtype in (torch.int, torch.int32, torch.uint8):
        raise RuntimeError('transform_weights_for_mega_moe: packed UE8M0 / FP4 weight scales (the reference\'s SM100 fp8xfp4 format, recipe '
                           '(1, 1, 32)) are not supported on gfx950; pass FP8 e4m3 weights with FP32 128 x 128 block scales')
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


def swiglu_workspace(num_groups: int, m: int, n: int, device) -> torch.Tensor:
    """A zeroed exchange workspace for :func:`m_grouped_fp8_gemm_nt_masked_swiglu` (``workspace=`` argument): one per launch that may be
    in flight at the same time -- two streams, two hipGraphs replayed concurrently, a replay racing an eager call (launches that share
    one concurrently steal each other's slots).  256-byte header (word 0 = exchange waits that timed out) + one slot row per tile."""
    return torch.zeros(int(lib.dg_swiglu_workspace_bytes(num_groups, m, n)), dtype=torch.uint8, device=device)


def _exchange_workspace(num_groups: int, m: int, n: int, device: torch.device) -> torch.Tensor:
    """The default workspace: one per (device, stream the call is ISSUED on), zeroed once, left zeroed by every launch
    (include/deepgemm_amd.h).  As gemm._split_k_workspace: never allocated while the stream is being captured (the buffer would land in
    the graph's private pool and outlive it in this dict) -- warm the call up eagerly on the capture stream first, or pass ``workspace=``.
    A graph captured on stream S keeps using S's buffer wherever it is replayed; concurrent replays need caller-owned workspaces."""
    need = int(lib.dg_swiglu_workspace_bytes(num_groups, m, n))
    key = (device.index, current_stream_ptr())
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < need:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('m_grouped_fp8_gemm_nt_masked_swiglu: no exchange workspace for this stream yet and the stream is being captured; '
                               'run the call once eagerly on this stream before capturing, or pass workspace=swiglu_workspace(...)')
        ws = torch.zeros(need, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def exchange_timeouts(workspace: Optional[torch.Tensor] = None, reset: bool = True) -> int:
    """Number of partner waits that timed out since the workspace was last zeroed (synchronises).  The rows involved carry NaN scales.
    ``workspace=None``: summed over the library's per-stream default workspaces.  ``reset``: re-zero every workspace that reports one
    (an aborted exchange leaves valid bits behind, which would corrupt the next launch silently)."""
    total = 0
    for ws in ([workspace] if workspace is not None else list(_workspaces.values())):
        count = int(ws[:4].view(torch.int32).item())
        if count and reset:
            ws.zero_()
        total += count
    return total


def set_exchange_timeout_us(us: int) -> None:
    """Bound of the partner wait of the fused kernel (default 10 s; the reference's barriers give up after 60 s, comm/barrier.cuh:12)."""
    lib.dg_set_swiglu_exchange_timeout_us(int(us))


def _bf16_round(x: float) -> float:
    """The reference applies the activation clamp with BF16 operands (``__hmin2`` on bf16x2, sm100_fp8_fp4_mega_moe.cuh:993-1020): a bound
    that is not BF16-representable acts as its BF16 rounding."""
    return float(torch.tensor(x, dtype=torch.float32).to(torch.bfloat16).float())


def m_grouped_fp8_gemm_nt_masked_swiglu(a: TensorPair, b: TensorPair, out: TensorPair, masked_m: torch.Tensor, expected_m: int,
                                        activation_clamp: Optional[float] = None, use_ue8m0: bool = False,
                                        workspace: Optional[torch.Tensor] = None, row_weight: Optional[torch.Tensor] = None) -> None:
    """``out = per_token_cast_to_fp8( swiglu( a @ b^T ) )`` per expert, rows ``< masked_m[g]`` only: ``a = (A [G, M, K], SFA)``,
    ``b`` = the transformed W1 pair ``([G, 2 I, K], [G, 2 I / 128, K / 128])`` of :func:`transform_weights_for_mega_moe`,
    ``out`` = :func:`empty_intermediate` ``(G, M, I)``.  Bit-identical to ``m_grouped_fp8_gemm_nt_masked`` -> BF16 -> SwiGLU (``silu(g) *
    u`` in FP32 on the BF16 values, optional clamp ``g <= c'``, ``|u| <= c'`` with ``c'`` = ``activation_clamp`` ROUNDED TO BF16 -- the
    reference clamps with BF16 operands, sm100_fp8_fp4_mega_moe.cuh:1003-1008, so a bound that is not BF16-representable acts as its BF16
    rounding, e.g. 10.1 as 10.125 -- result rounded to BF16) -> ``per_token_cast_to_fp8``.  With ``row_weight`` the product
    ``silu(g) * u * w`` stays in FP32 up to the cast (the reference kernel's arithmetic; no BF16 intermediate exists in that pipeline)."""
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
    ws = workspace if workspace is not None else _exchange_workspace(num_groups, m, n, a_data.device)
    if row_weight is not None:
        # ``row_weight [G, >= align(m, 64)]`` FP32: the routing weight of every row slot: ``silu(g) * u * w`` in FP32 (g, u the BF16-rounded
        # GEMM outputs) straight into the amax and the FP8 cast -- the reference kernel's epilogue, sm100_fp8_fp4_mega_moe.cuh:1001-1020
        host_assert(row_weight.dtype == torch.float and row_weight.dim() == 2 and row_weight.size(0) == num_groups and row_weight.stride(1) == 1 and
                    row_weight.size(1) >= -(-m // 64) * 64, 'row_weight: float [G, >= align(m, 64)]')
        require_device(row_weight)
    check(lib.dg_m_grouped_fp8_gemm_nt_masked_swiglu_weighted(
        a_data.data_ptr(), sfa.data_ptr(), b_data.data_ptr(), b_sf.data_ptr(), q.data_ptr(), q_sf.data_ptr(), masked_m.data_ptr(),
        num_groups, m, n, k, int(expected_m), a_data.stride(0), a_data.stride(1), b_data.stride(0), b_data.stride(1),
        sfa.stride(0), sfa.stride(2), b_sf.stride(0), b_sf.stride(1), b_sf.stride(2), q.stride(0), q.stride(1), q_sf.stride(0), q_sf.stride(2),
        _bf16_round(activation_clamp) if activation_clamp is not None else 0.0, int(use_ue8m0),
        row_weight.data_ptr() if row_weight is not None else None, row_weight.stride(0) if row_weight is not None else 0,
        ws.data_ptr(), ws.numel(), current_stream_ptr()))


def fp8_mega_moe_local(x: TensorPair, l1_weights: TensorPair, l2_weights: TensorPair, y: torch.Tensor, masked_m: torch.Tensor,
                       expected_m: int, activation_clamp: Optional[float] = None,
                       intermediate: Optional[TensorPair] = None, workspace: Optional[torch.Tensor] = None) -> TensorPair:
    """The expert MLP of the tokens resident on this GPU in the masked layout: ``y[g, :masked_m[g]] = W2_g . swiglu(W1_g . x[g])`` --
    fused GEMM1 (:func:`m_grouped_fp8_gemm_nt_masked_swiglu`) + ``m_grouped_fp8_gemm_nt_masked``.  ``l1_weights`` / ``l2_weights`` as
    returned by :func:`transform_weights_for_mega_moe`.  Returns the intermediate pair (reusable as the ``intermediate`` argument)."""
    num_groups, m, _ = x[0].shape
    inter = l1_weights[0].size(1) // 2
    if intermediate is None:
        intermediate = empty_intermediate(num_groups, m, inter, x[0].device)
    m_grouped_fp8_gemm_nt_masked_swiglu(x, l1_weights, intermediate, masked_m, expected_m, activation_clamp, workspace=workspace)
    m_grouped_fp8_gemm_nt_masked(intermediate, l2_weights, y, masked_m, expected_m)
    return intermediate


# ---------------------------------------------------------------------------------------------------------------------------------
# The reference-shaped operator (deep_gemm/mega/__init__.py:18-58, 68-128, 155-173; host side csrc/apis/mega.hpp:30-159) at world
# size 1: buffer object with the reference's input views, one call = routing -> fused L1 -> L2 -> combine, every step a stream-ordered
# launch with device-resident counts (hipGraph-capturable).  With more than one rank the dispatch / combine legs are RCCL all-to-alls
# (deepgemm_amd/ep.py); the in-kernel xGMI peer-to-peer form is not built and asking for it says so.
# ---------------------------------------------------------------------------------------------------------------------------------
def get_token_alignment_for_mega_moe() -> int:
    """Rows per M tile of the fused kernel (reference: csrc/apis/mega.hpp, ``get_token_alignment_for_mega_moe``)."""
    return 64


class SymmBuffer:
    """The reference's ``SymmBuffer`` (deep_gemm/mega/__init__.py:18-58): the caller-visible input views ``x [T, H]`` e4m3,
    ``x_sf [T, H / 128]`` FP32 (per-token 1 x 128 scales), ``topk_idx [T, top_k]`` int64 GLOBAL expert indices (-1 = no expert),
    ``topk_weights [T, top_k]`` FP32, and the operator's private staging.

    ``group`` = ``None`` or a one-rank group: every expert is local; staging = the masked-layout activations of both layers
    (``[E, T, H]`` e4m3 + ``[E, T, I]`` e4m3 + ``[E, T, H]`` BF16), slot map, per-expert counts, routing weight per slot.
    A group of ``R > 1`` ranks: this rank owns experts ``[rank * E / R, (rank + 1) * E / R)`` (``l1_weights`` / ``l2_weights`` hold only
    those); rows are exchanged in fixed-shape blocks of ``exchange_capacity`` rows per (rank, expert) (default ``T``: a token names an
    expert at most once) and an expert takes at most ``expert_capacity`` rows (default ``R * T``, the worst case).  Footprint per rank
    with the defaults: ``(E / R) * R * T * (2 H + I + 2 H)`` bytes of staging + ``2 * E * T * (H + H / 32 + 4)`` bytes on the wire per
    call -- at E = 256, R = 8, T = 8192, H = 7168 that is ~70 GB: size ``expert_capacity`` / ``exchange_capacity`` for the routing you
    actually have (the reference's ring buffer is O(T * top_k * H); rows over a capacity are dropped and counted in ``errors[0]``)."""

    def __init__(self, group, num_experts: int, num_max_tokens_per_rank: int, num_topk: int, hidden: int, intermediate_hidden: int,
                 num_ring_tokens: int = 0, mma_type: str = 'fp8xfp8', activation: str = 'swiglu', device='cuda',
                 expert_capacity: Optional[int] = None, exchange_capacity: Optional[int] = None, force_exchange: bool = False,
                 p2p: bool = False):
        host_assert(activation == 'swiglu', "activation == 'swiglu'")
        self.world = 1 if group is None else group.size()
        self.p2p = bool(p2p)
        if self.p2p:
            self._init_p2p(group, num_experts, num_max_tokens_per_rank, num_topk, hidden, intermediate_hidden, mma_type, device, expert_capacity)
            return
        # force_exchange: take the multi-rank path (fixed-shape all-to-alls around the two GEMMs) even with one rank -- how the 1-GPU test box
        # runs the HIP side of that path over RCCL (tests/test_ep_gpu.py); needs an initialised process group
        self.exchange = self.world > 1 or force_exchange
        if mma_type not in ('fp8xfp8', 'fp8'):
            raise RuntimeError(f"SymmBuffer: mma_type '{mma_type}' is not supported on gfx950 (FP8 e4m3 activations x FP8 e4m3 weights only)")
        host_assert(hidden % 128 == 0 and intermediate_hidden % 128 == 0, 'hidden % 128 == 0 and intermediate_hidden % 128 == 0')
        host_assert(num_experts % self.world == 0, 'num_experts % num_ranks == 0')
        self.group, self.num_experts, self.num_topk = group, num_experts, num_topk
        self.num_local_experts = num_experts // self.world
        self.num_max_tokens_per_rank = -(-num_max_tokens_per_rank // 64) * 64
        self.hidden, self.intermediate_hidden, self.num_ring_tokens = hidden, intermediate_hidden, num_ring_tokens
        t, e = self.num_max_tokens_per_rank, self.num_local_experts
        self.x = torch.zeros((t, hidden), dtype=torch.float8_e4m3fn, device=device)
        self.x_sf = torch.zeros((t, hidden // 128), dtype=torch.float, device=device)
        self.topk_idx = torch.full((t, num_topk), -1, dtype=torch.int64, device=device)
        self.topk_weights = torch.zeros((t, num_topk), dtype=torch.float, device=device)
        self.errors = torch.zeros((4,), dtype=torch.int32, device=device)        # word 0: rows dropped by the routing (over a capacity)
        self.buffer = self.x                                                    # (reference attribute; no symmetric heap here)
        if not self.exchange:
            m = t                                                               # one token meets an expert at most once
            self.exchange_capacity = 0
        else:
            self.exchange_capacity = min(t, exchange_capacity) if exchange_capacity else t
            m = -(-min(self.world * self.exchange_capacity, expert_capacity or self.world * t) // 64) * 64
        self.expert_capacity = m
        aligned = get_tma_aligned_size(m, 4)
        self.masked_m = torch.zeros((e,), dtype=torch.int32, device=device)
        self.l2_out = torch.empty((e, m, hidden), dtype=torch.bfloat16, device=device)
        self.workspace = None
        self.l2_acts = self.l2_acts_sf = None
        if torch.device(device).type == 'cuda':
            self.l2_acts, self.l2_acts_sf = empty_intermediate(e, m, intermediate_hidden, device)
            self.workspace = swiglu_workspace(e, m, 2 * intermediate_hidden, device)
        if not self.exchange:
            self.l1_acts = torch.zeros((e, m, hidden), dtype=torch.float8_e4m3fn, device=device)
            self.l1_acts_sf = torch.zeros((e, hidden // 128, aligned), dtype=torch.float, device=device).transpose(1, 2)   # [E, m, H/128], MN-major
            self.row_weight = torch.zeros((e, aligned), dtype=torch.float, device=device)
            self.slot = torch.full((t * num_topk,), -1, dtype=torch.int32, device=device)
        else:
            self.l1_acts = self.l1_acts_sf = self.row_weight = self.slot = None     # (the exchange hands these over per call: ep.dispatch_fixed)

    # ---- the in-kernel (peer-to-peer) form: csrc/fp8_gemm_moe.hpp, "In-kernel dispatch / combine" ----
    def _init_p2p(self, group, num_experts, num_max_tokens_per_rank, num_topk, hidden, intermediate_hidden, mma_type, device, expert_capacity):
        """Every rank allocates one symmetric region (``dg_symm_alloc``), the ranks exchange its IPC handle (``all_gather_object`` over
        ``group`` -- any backend: gloo on one GPU shared by two processes, RCCL on a node) and map each other's regions.  The masked-layout
        input of the fused L1 kernel (``l1_acts``, ``l1_acts_sf``, ``row_weight``) are VIEWS of the own region: peers write straight into them."""
        import ctypes
        import torch.distributed as dist
        if mma_type not in ('fp8xfp8', 'fp8'):
            raise RuntimeError(f"SymmBuffer: mma_type '{mma_type}' is not supported on gfx950 (FP8 e4m3 activations x FP8 e4m3 weights only)")
        host_assert(hidden % 128 == 0 and intermediate_hidden % 128 == 0, 'hidden % 128 == 0 and intermediate_hidden % 128 == 0')
        host_assert(num_experts % self.world == 0 and self.world <= 16, 'num_experts % num_ranks == 0 and num_ranks <= 16')
        host_assert(torch.device(device).type == 'cuda', 'the in-kernel dispatch / combine has no CPU path')
        self.exchange = False
        self.rank = 0 if group is None else dist.get_rank(group)
        self.group, self.num_experts, self.num_topk = group, num_experts, num_topk
        self.num_local_experts = e = num_experts // self.world
        self.num_max_tokens_per_rank = t = -(-num_max_tokens_per_rank // 64) * 64
        self.hidden, self.intermediate_hidden, self.num_ring_tokens = hidden, intermediate_hidden, 0
        host_assert(t * num_topk < (1 << 24), 'num_max_tokens_per_rank * num_topk < 2^24 (return addresses are 24-bit)')
        # rows per local expert: the worst case (every token of every rank names it) unless the caller sizes it for its routing; rows over it
        # are dropped and counted in errors[0] ON THEIR SENDER
        m = -(-min(self.world * t, expert_capacity or self.world * t) // 64) * 64
        self.expert_capacity, self.exchange_capacity = m, 0
        offs = (ctypes.c_int64 * 10)()
        check(lib.dg_moe_p2p_layout(e, m, hidden, t, num_topk, self.world, offs))
        self._offsets = dict(zip(('counts', 'arrived', 'combined', 'done', 'l1_acts', 'l1_sf', 'row_w', 'src_info', 'y_rows', 'bytes'), list(offs)))
        local_bytes = e * m * (intermediate_hidden + intermediate_hidden // 32 + 2 * hidden) + t * (hidden + hidden // 32)
        free_bytes = torch.cuda.mem_get_info(torch.device(device))[0]
        if self._offsets['bytes'] + local_bytes > free_bytes:
            raise RuntimeError(f'SymmBuffer(p2p=True): {(self._offsets["bytes"] + local_bytes) / 2 ** 30:.1f} GiB of staging for {e} local experts x {m} rows '
                               f'(hidden {hidden}) exceed the {free_bytes / 2 ** 30:.1f} GiB free on the device; pass expert_capacity= sized for the '
                               f'routing you have (rows over it are dropped and counted in errors[0])')
        ptr, fine = ctypes.c_void_p(), ctypes.c_int()
        with torch.cuda.device(torch.device(device)):
            check(lib.dg_symm_alloc(self._offsets['bytes'], ctypes.byref(ptr), ctypes.byref(fine)))
        self._region, self.fine_grained = ptr.value, bool(fine.value)
        handle = (ctypes.c_char * 64)()
        self._peer_ptrs = [None] * self.world
        self._peer_ptrs[self.rank] = self._region
        if self.world > 1:
            # export, exchange, map -- and AGREE: a rank whose runtime cannot map a peer (another node, IPC disabled) must not leave the others
            # waiting; every rank learns the outcome of every rank and all take the same path
            problem = None
            try:
                check(lib.dg_ipc_get_handle(self._region, handle))
            except RuntimeError as err:
                problem = str(err)
            handles = [None] * self.world
            dist.all_gather_object(handles, None if problem else bytes(handle.raw), group=group)
            for r, h in enumerate(handles):
                if r == self.rank or problem or h is None:
                    continue
                opened = ctypes.c_void_p()
                try:
                    check(lib.dg_ipc_open_handle(ctypes.c_char_p(h), ctypes.byref(opened)))
                    self._peer_ptrs[r] = opened.value
                except RuntimeError as err:
                    problem = str(err)
            outcomes = [None] * self.world
            dist.all_gather_object(outcomes, problem or (None if all(h is not None for h in handles) else 'a peer could not export its region'), group=group)
            if any(o is not None for o in outcomes):
                for r, ptr in enumerate(self._peer_ptrs):
                    if r != self.rank and ptr:
                        lib.dg_ipc_close_handle(ptr)
                lib.dg_symm_free(self._region)
                self._region = None
                raise RuntimeError('HIP error: the ranks of this group cannot map each other\'s memory (' +
                                   '; '.join(f'rank {r}: {o}' for r, o in enumerate(outcomes) if o is not None) + ')')
        self._peers = (ctypes.c_void_p * self.world)(*self._peer_ptrs)
        self._epoch = 0
        dev = torch.device(device)
        self.x = torch.zeros((t, hidden), dtype=torch.float8_e4m3fn, device=dev)
        self.x_sf = torch.zeros((t, hidden // 128), dtype=torch.float, device=dev)
        self.topk_idx = torch.full((t, num_topk), -1, dtype=torch.int64, device=dev)
        self.topk_weights = torch.zeros((t, num_topk), dtype=torch.float, device=dev)
        self.errors = torch.zeros((4,), dtype=torch.int32, device=dev)
        self.buffer = self.x
        self.masked_m = torch.zeros((e,), dtype=torch.int32, device=dev)
        self.pair_ok = torch.zeros((t * num_topk,), dtype=torch.uint8, device=dev)
        self.l2_out = torch.empty((e, m, hidden), dtype=torch.bfloat16, device=dev)
        self.l2_acts, self.l2_acts_sf = empty_intermediate(e, m, intermediate_hidden, dev)
        self.workspace = swiglu_workspace(e, m, 2 * intermediate_hidden, dev)
        region = _raw_device_bytes(self._region, self._offsets['bytes'], dev)
        o = self._offsets
        self.l1_acts = region[o['l1_acts']:o['l1_acts'] + e * m * hidden].view(torch.float8_e4m3fn).view(e, m, hidden)
        self.l1_acts_sf = region[o['l1_sf']:o['l1_sf'] + 4 * e * (hidden // 128) * m].view(torch.float).view(e, hidden // 128, m).transpose(1, 2)
        self.row_weight = region[o['row_w']:o['row_w'] + 4 * e * m].view(torch.float).view(e, m)
        self.slot = None
        self._region_view = region
        if self.world > 1:
            dist.barrier(group=group)           # nobody dispatches into a region that is not zeroed and mapped everywhere yet

    def destroy(self):
        if getattr(self, 'p2p', False) and getattr(self, '_region', None):
            import torch.distributed as dist
            torch.cuda.synchronize()
            if self.world > 1:
                dist.barrier(group=self.group)  # every rank is done with every region before any mapping goes away
            for r, ptr in enumerate(self._peer_ptrs):
                if r != self.rank and ptr:
                    lib.dg_ipc_close_handle(ptr)
            self.l1_acts = self.l1_acts_sf = self.row_weight = self._region_view = None
            lib.dg_symm_free(self._region)
            self._region = None
        for name in ('x', 'x_sf', 'topk_idx', 'topk_weights', 'l1_acts', 'l1_acts_sf', 'l2_acts', 'l2_acts_sf', 'l2_out', 'row_weight', 'slot',
                     'masked_m', 'errors', 'workspace', 'buffer', 'group'):
            setattr(self, name, None)


def get_symm_buffer_for_mega_moe(group, num_experts: int, num_max_tokens_per_rank: int, num_topk: int, hidden: int, intermediate_hidden: int,
                                 use_fp8_dispatch: Optional[bool] = None, mma_type: str = 'fp8xfp8', activation: str = 'swiglu',
                                 expert_capacity: Optional[int] = None, exchange_capacity: Optional[int] = None,
                                 p2p: Optional[bool] = None) -> SymmBuffer:
    """deep_gemm/mega/__init__.py:68-128 (the ring-token sizing of the reference belongs to its NVLink pull pipeline).  Beyond the
    reference's arguments: ``expert_capacity`` / ``exchange_capacity`` size the staging for the routing the caller has (defaults: the worst
    case, O(E / R * R * T * H) bytes -- a clear sizing error is raised when that exceeds the device's free memory); ``p2p``: the in-kernel
    dispatch / combine over peer-mapped memory (default: taken when the group has more than one rank, its ranks share one node and peers can be
    mapped -- i.e. the ``DG_MEGA_P2P`` environment variable is not ``0`` --, RCCL all-to-alls otherwise)."""
    import os
    if p2p is None:
        p2p = group is not None and group.size() > 1 and os.environ.get('DG_MEGA_P2P', '1') != '0' and torch.cuda.is_available()
    if p2p:
        try:
            return SymmBuffer(group, num_experts, num_max_tokens_per_rank, num_topk, hidden, intermediate_hidden, 0, mma_type, activation,
                              expert_capacity=expert_capacity, p2p=True)
        except RuntimeError as e:
            if 'HIP error' not in str(e):       # (sizing and argument errors are the caller's to see; a runtime that cannot map peers falls back)
                raise
    return SymmBuffer(group, num_experts, num_max_tokens_per_rank, num_topk, hidden, intermediate_hidden, 0, mma_type, activation,
                      expert_capacity=expert_capacity, exchange_capacity=exchange_capacity)


def fp8_mega_moe(y: torch.Tensor, l1_weights: TensorPair, l2_weights: TensorPair, sym_buffer: SymmBuffer,
                 cumulative_local_expert_recv_stats: Optional[torch.Tensor] = None, recipe: Optional[Tuple[int, int, int]] = None,
                 activation: str = 'swiglu', activation_clamp: Optional[float] = None, fast_math: bool = True,
                 local_ops=None) -> None:
    """``y[t] = sum_j W2[e_j] . fp8( swiglu( W1[e_j] . x[t] ) * w_j )`` over the token's top-k experts ``e_j = topk_idx[t, j] >= 0`` --
    the reference's ``fp8_fp4_mega_moe(y, l1_weights, l2_weights, sym_buffer, ...)`` (deep_gemm/mega/__init__.py:155-173) with FP8 e4m3
    weights and FP32 128 x 128 block scales (``transform_weights_for_mega_moe``).  With a group of more than one rank the weights are this
    rank's experts and the rows travel through two fixed-shape RCCL all-to-alls (:func:`_mega_moe_ep`; same arithmetic, same bits as
    one rank holding every expert: tests/test_mega_gloo.py).  ``local_ops``: test hook -- ``(l1, l2)`` callables replacing the two HIP
    operators (the gloo CPU tests put the oracle there, as ``ep.py``'s ``local_gemm``); the product path never sets it.  Inputs are read from the buffer's views
    (``x``, ``x_sf``, ``topk_idx``, ``topk_weights``; rows ``[0, y.size(0))``), as the reference's test fills them
    (tests/test_mega_moe.py:103-121).  Four stream-ordered launches + one memset, nothing returns to the host: scatter into the masked
    layout, fused L1 (GEMM + SwiGLU + routing weight + per-token FP8 re-quantisation), masked L2, gather-sum in top-k order (FP32).
    Bit-identical to the unfused pipeline of the same operators (tests/test_mega_gpu.py).  ``fast_math`` is accepted and ignored: this
    kernel has one SwiGLU form (exact ``expf``).  ``sym_buffer.errors`` (device int32 [4], never read by this call): word 0 = rows dropped
    by the routing (over a capacity; cannot happen with the default capacities when a token lists an expert at most once), word 1 =
    partner waits of the fused L1 kernel that timed out (those rows carry NaN; re-zero ``sym_buffer.workspace`` after one)."""
    host_assert(activation == 'swiglu', "activation == 'swiglu'")
    host_assert(recipe is None or tuple(recipe) == (1, 128, 128), 'recipe == (1, 128, 128): FP32 block scales (the (1, 1, 32) UE8M0 / FP4 recipe is SM100-only)')
    b = sym_buffer
    tokens = int(y.size(0))
    host_assert(y.dim() == 2 and y.dtype == torch.bfloat16 and y.size(1) == b.hidden and y.stride(1) == 1 and tokens <= b.num_max_tokens_per_rank,
                'y: bfloat16 [num_tokens <= num_max_tokens_per_rank, hidden]')
    host_assert(l1_weights[0].size(0) == b.num_local_experts and l1_weights[0].size(1) == 2 * b.intermediate_hidden and l1_weights[0].size(2) == b.hidden,
                'l1_weights[0].shape == (num_experts / num_ranks, 2 * intermediate_hidden, hidden)')
    host_assert(l2_weights[0].size(0) == b.num_local_experts and l2_weights[0].size(1) == b.hidden and l2_weights[0].size(2) == b.intermediate_hidden,
                'l2_weights[0].shape == (num_experts / num_ranks, hidden, intermediate_hidden)')
    host_assert(l1_weights[0].size(0) == b.num_local_experts, 'l1_weights[0].size(0) == num_experts / num_ranks')
    if getattr(b, 'p2p', False):
        _mega_moe_p2p(y, l1_weights, l2_weights, b, cumulative_local_expert_recv_stats, activation_clamp)
        return
    if b.exchange:
        _mega_moe_ep(y, l1_weights, l2_weights, b, cumulative_local_expert_recv_stats, activation_clamp, local_ops)
        return
    require_device(y, b.x, l1_weights[0], l2_weights[0])
    stream = current_stream_ptr()
    m = b.num_max_tokens_per_rank
    check(lib.dg_moe_scatter_to_masked(
        b.x.data_ptr(), b.x_sf.data_ptr(), b.topk_idx.data_ptr(), int(b.topk_idx.dtype == torch.int64), b.topk_weights.data_ptr(),
        tokens, b.hidden, b.num_topk, b.num_experts, m, b.x.stride(0), b.x_sf.stride(0),
        b.l1_acts.data_ptr(), b.l1_acts_sf.data_ptr(), b.row_weight.data_ptr(), b.slot.data_ptr(), b.masked_m.data_ptr(), b.errors.data_ptr(),
        b.l1_acts.stride(0), b.l1_acts.stride(1), b.l1_acts_sf.stride(0), b.l1_acts_sf.stride(2), b.row_weight.stride(0), stream))
    expected_m = max(1, min(m, -(-tokens * b.num_topk // b.num_experts)))
    m_grouped_fp8_gemm_nt_masked_swiglu((b.l1_acts, b.l1_acts_sf), l1_weights, (b.l2_acts, b.l2_acts_sf), b.masked_m, expected_m,
                                        activation_clamp, workspace=b.workspace, row_weight=b.row_weight)
    b.errors[1:2].copy_(b.workspace[:4].view(torch.int32))   # word 1: partner waits of the fused kernel that timed out since the workspace was zeroed
    m_grouped_fp8_gemm_nt_masked((b.l2_acts, b.l2_acts_sf), l2_weights, b.l2_out, b.masked_m, expected_m)
    check(lib.dg_moe_combine_from_masked(b.l2_out.data_ptr(), b.slot.data_ptr(), tokens, b.num_topk, b.hidden, b.l2_out.stride(1),
                                         y.data_ptr(), y.stride(0), stream))
    if cumulative_local_expert_recv_stats is not None:
        cumulative_local_expert_recv_stats.add_(b.masked_m.to(cumulative_local_expert_recv_stats.dtype))


class _RawDeviceMemory:
    """``__cuda_array_interface__`` over a raw device pointer (memory owned by the C side: dg_symm_alloc)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {'shape': (nbytes,), 'typestr': '|u1', 'data': (ptr, False), 'version': 2, 'strides': None}


def _raw_device_bytes(ptr: int, nbytes: int, device: torch.device) -> torch.Tensor:
    return torch.as_tensor(_RawDeviceMemory(ptr, nbytes), device=device)


def set_p2p_timeout_us(us: int) -> None:
    """Bound of the flag waits of the in-kernel dispatch / combine (default 10 s; reference: comm/barrier.cuh:12, 60 s)."""
    lib.dg_set_moe_p2p_timeout_us(int(us))


def _mega_moe_p2p(y, l1_weights, l2_weights, b: SymmBuffer, stats, activation_clamp) -> None:
    """``fp8_mega_moe`` with the in-kernel dispatch / combine (csrc/fp8_gemm_moe.hpp; reference: sm100_fp8_fp4_mega_moe.cuh:357-405, 523-595):
    FIVE stream-ordered launches -- push dispatch (+ arrival wait + counts), fused L1, masked L2, push combine, wait + top-k sum -- no
    collective, no host round trip.  Every rank of the group must make the call (the flags are an all-to-all), with the same epoch: one
    ``fp8_mega_moe`` per rank per step on this buffer."""
    require_device(y, b.x, l1_weights[0], l2_weights[0])
    tokens = int(y.size(0))
    stream = current_stream_ptr()
    if torch.cuda.is_current_stream_capturing():
        # the step's epoch is a host value in the kernels' arguments: a replayed graph would present the SAME epoch again and its waits would
        # be satisfied by the previous step's flags
        raise RuntimeError('fp8_mega_moe over peer-mapped memory cannot be captured in a hipGraph (the exchange epoch is a launch argument); '
                           'capture the RCCL form (SymmBuffer(p2p=False) / DG_MEGA_P2P=0) or call it eagerly')
    b._epoch += 1
    geometry = (b.world, b.rank, b.num_local_experts, b.expert_capacity, b.hidden, b.num_max_tokens_per_rank, b.num_topk)
    check(lib.dg_moe_p2p_dispatch(b._peers, *geometry, b.x.data_ptr(), b.x_sf.data_ptr(), b.topk_idx.data_ptr(),
                                  int(b.topk_idx.dtype == torch.int64), b.topk_weights.data_ptr(), tokens, b.x.stride(0), b.x_sf.stride(0),
                                  b._epoch, b.masked_m.data_ptr(), b.pair_ok.data_ptr(), b.errors.data_ptr(), stream))
    expected_m = max(1, min(b.expert_capacity, -(-tokens * b.num_topk // b.num_local_experts)))     # (a tuning hint: this rank's share as a proxy)
    m_grouped_fp8_gemm_nt_masked_swiglu((b.l1_acts, b.l1_acts_sf), l1_weights, (b.l2_acts, b.l2_acts_sf), b.masked_m, expected_m,
                                        activation_clamp, workspace=b.workspace, row_weight=b.row_weight)
    m_grouped_fp8_gemm_nt_masked((b.l2_acts, b.l2_acts_sf), l2_weights, b.l2_out, b.masked_m, expected_m)
    check(lib.dg_moe_p2p_combine(b._peers, *geometry, b.l2_out.data_ptr(), b.l2_out.stride(0), b.l2_out.stride(1), b.masked_m.data_ptr(),
                                 b._epoch, b.errors.data_ptr(), stream))
    check(lib.dg_moe_p2p_reduce(b._peers, *geometry, b.pair_ok.data_ptr(), tokens, y.data_ptr(), y.stride(0) if tokens else 0,
                                b.workspace.data_ptr(), b._epoch, b.errors.data_ptr(), stream))
    if stats is not None:
        stats.add_(b.masked_m.to(stats.dtype))


def _mega_moe_ep(y, l1_weights, l2_weights, b: SymmBuffer, stats, activation_clamp, local_ops) -> None:
    """``fp8_mega_moe`` over a group of R > 1 ranks (reference: the dispatch / L1 / L2 / combine stages of
    sm100_fp8_fp4_mega_moe.cuh:357-405, 523-595, 1019; baseline with the same stages: tests/test_mega_moe.py:149-214):

      1. fixed-shape dispatch (``ep.dispatch_fixed``): every (token, expert) pair's FP8 row + its 1 x 128 scales + its routing weight go to
         the expert's owner -- two equal-split all-to-alls (counts, payload), receiver-side compaction into the masked layout on the
         device, no host synchronisation (hipGraph-capturable; rows over a capacity are dropped and counted in ``b.errors[0]``);
      2. fused L1 on the local experts (GEMM + SwiGLU + routing weight + per-token FP8 re-quantisation), masked L2 -- the launches of
         the one-rank path on ``E / R`` groups;
      3. fixed-shape combine: BF16 result rows return in the same blocks; the token's owner sums its ``top_k`` rows in top-k order in FP32
         (what ``dg_moe_combine_from_masked`` does at one rank), pairs without an expert contribute zeros."""
    from . import ep
    tokens = int(y.size(0))
    x = (b.x[:tokens], b.x_sf[:tokens])
    (a, a_sf), plan = ep.dispatch_fixed(x, b.topk_idx[:tokens], b.num_experts, b.expert_capacity, b.exchange_capacity, b.group,
                                        row_extra=b.topk_weights[:tokens])
    b.masked_m.copy_(plan.masked_m)
    # word 0 counts ROWS, as at one rank: this rank's pairs over the exchange capacity + rows it received over an expert's capacity (the
    # latter belong to other ranks' tokens: the count tells that rows were zeroed somewhere in the group, the sum over ranks how many)
    b.errors[0] += plan.dropped.to(torch.int32)
    expected_m = max(1, min(b.expert_capacity, -(-tokens * b.num_topk // b.num_local_experts)))
    if local_ops is None:
        m_grouped_fp8_gemm_nt_masked_swiglu((a, a_sf), l1_weights, (b.l2_acts, b.l2_acts_sf), b.masked_m, expected_m, activation_clamp,
                                            workspace=b.workspace, row_weight=plan.row_extra)
        b.errors[1:2].copy_(b.workspace[:4].view(torch.int32))
        m_grouped_fp8_gemm_nt_masked((b.l2_acts, b.l2_acts_sf), l2_weights, b.l2_out, b.masked_m, expected_m)
    else:
        inter = local_ops[0]((a, a_sf), l1_weights, b.masked_m, activation_clamp, plan.row_extra)
        local_ops[1](inter, l2_weights, b.l2_out, b.masked_m)
    rows = ep.combine_fixed(b.l2_out, plan, tokens, b.num_topk, b.world, b.exchange_capacity, b.group)        # [T, top_k, H]
    acc = rows[:, 0].float()
    for j in range(1, b.num_topk):
        acc += rows[:, j].float()
    y.copy_(acc.to(torch.bfloat16))
    if stats is not None:
        stats.add_(b.masked_m.to(stats.dtype))


fp8_fp4_mega_moe = fp8_mega_moe      # the reference's name; FP4 weights (its SM100 default) are rejected by the weight transform


def bf16_mega_moe(*args, **kwargs):
    raise RuntimeError('bf16_mega_moe is outside this library (FP8 GEMM path only; SURVEY.md section 8 scope)')
