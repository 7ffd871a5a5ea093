"""Expert-parallel execution of the M-grouped masked FP8 GEMM: experts sharded across the GPUs of one node, token rows
exchanged with RCCL all-to-all over xGMI, one local ``m_grouped_fp8_gemm_nt_masked`` per rank.

This is the only place where the FP8 GEMM path has a real exchange step: the groups of a grouped
GEMM are independent problems that share nothing but the token stream, so rank ``r`` owns experts
``[r * E / R, (r + 1) * E / R)`` -- their weights ``B[g]`` / ``SFB[g]`` stay resident in its HBM and never move -- and
one step is

  1. *dispatch*: every (token row, expert) pair travels to the expert's owner: ``K`` FP8 bytes + ``K / 128`` FP32 scales
     per row in ONE payload all-to-all, variable split sizes exchanged first with a tiny int64 all-to-all (this replaces the in-kernel count
     exchange of the reference's fused kernel, deep_gemm/include/deep_gemm/impls/sm100_fp8_fp4_mega_moe.cuh:357-405);
  2. *local GEMM*: received rows are packed per local expert into the masked layout ``[G_local, M_max, K]`` with the
     row counts in a device tensor ``masked_m`` (the kernel reads them on the device) and
     ``m_grouped_fp8_gemm_nt_masked`` runs once (reference semantics: csrc/apis/gemm.hpp:250-297);
  3. *combine*: the BF16 result rows (``2 N`` bytes each) return to their source rank in the order they were sent.

xGMI is a point-to-point mesh, so an all-to-all puts only each pair's own traffic on each of the 7 links; at decode
sizes (<= 64 rows per expert) the messages are a few MB per GPU and latency dominated.

The collective layer is ``torch.distributed`` (backend ``nccl`` = RCCL on ROCm; ``gloo`` in the CPU tests, where the
local GEMM is replaced by the test's checker through the ``local_gemm`` argument).  The exact-size exchange (``dispatch`` /
``combine``) needs the split sizes as Python ints -- one small device-to-host copy per step, as in any unfused EP dispatch; the
fixed-capacity exchange (``dispatch_fixed`` / ``combine_fixed``) pads every (rank, expert) block to ``capacity`` rows instead
and never touches the host: the form for latency-bound, graph-captured decode steps.
"""
from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist

TensorPair = Tuple[torch.Tensor, torch.Tensor]


def expert_range(num_experts: int, rank: int, world: int) -> Tuple[int, int]:
    """Experts owned by ``rank``: a contiguous, equally sized block (``num_experts`` must divide by ``world``)."""
    if num_experts % world != 0:
        raise RuntimeError(f'num_experts ({num_experts}) must be a multiple of the world size ({world})')
    per_rank = num_experts // world
    return rank * per_rank, (rank + 1) * per_rank


@dataclass
class DispatchPlan:
    """Bookkeeping of one dispatch, needed to route the results back."""
    send_order: torch.Tensor          # [P] permutation that sorts this rank's (row, expert) pairs by destination expert
    send_splits: List[int]            # rows sent to each rank
    recv_splits: List[int]            # rows received from each rank
    recv_expert: torch.Tensor         # [P_recv] local expert index of every received row
    recv_slot: torch.Tensor           # [P_recv] row slot inside that expert's masked block
    masked_m: torch.Tensor            # [G_local] int32 rows per local expert (device tensor)


def _all_to_all_rows(rows: torch.Tensor, send_splits: List[int], recv_splits: List[int], group) -> torch.Tensor:
    out = rows.new_empty((sum(recv_splits),) + tuple(rows.shape[1:]))
    dist.all_to_all_single(out, rows.contiguous(), output_split_sizes=recv_splits, input_split_sizes=send_splits, group=group)
    return out


def dispatch(x: TensorPair, expert_ids: torch.Tensor, num_experts: int, max_m: int,
             group=None) -> Tuple[TensorPair, DispatchPlan]:
    """Sends every (row, expert) pair of this rank to the expert's owner and packs what arrives into the masked layout.

    ``x = (x_fp8 [T, K], sf [T, K / 128])`` are this rank's token rows (already quantised per token, reference
    ``per_token_cast_to_fp8``); ``expert_ids [T, top_k]`` holds global expert indices.  Returns the local masked
    operand ``(a [G_local, max_m, K], sfa [G_local, max_m, K / 128])`` and the plan for :func:`combine`.
    Raises ``RuntimeError`` if a local expert receives more than ``max_m`` rows.
    """
    x_fp8, x_sf = x
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    first, last = expert_range(num_experts, rank, world)
    per_rank = last - first
    tokens, top_k = expert_ids.shape
    device = x_fp8.device

    flat_expert = expert_ids.reshape(-1).to(torch.int64)
    flat_row = torch.arange(tokens, device=device).repeat_interleave(top_k)
    order = torch.argsort(flat_expert, stable=True)                      # grouped by destination expert (=> by rank)
    sorted_expert = flat_expert[order]
    per_expert = torch.bincount(sorted_expert, minlength=num_experts)    # rows this rank sends to each expert
    send_counts = per_expert.view(world, per_rank)

    # counts first: recv_counts[s, g] = rows rank s sends to my local expert g
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts.contiguous(), group=group)
    send_splits = send_counts.sum(dim=1).tolist()
    recv_splits = recv_counts.sum(dim=1).tolist()

    # one payload per row: K FP8 bytes followed by the row's K / 128 FP32 scales as bytes -- a single all-to-all instead of one for
    # the data and one for the scales (at decode sizes every collective is latency, not bandwidth)
    rows = flat_row[order]
    k, sf_bytes = x_fp8.size(1), 4 * x_sf.size(1)
    packed = torch.empty((rows.numel(), k + sf_bytes), dtype=torch.uint8, device=device)
    packed[:, :k] = x_fp8.view(torch.uint8)[rows]
    packed[:, k:] = x_sf.contiguous().view(torch.uint8).view(tokens, sf_bytes)[rows]
    received = _all_to_all_rows(packed, send_splits, recv_splits, group)
    payload = received[:, :k].contiguous().view(torch.float8_e4m3fn)
    payload_sf = received[:, k:].contiguous().view(torch.float)

    # received rows arrive rank by rank, inside a rank expert by expert: recover (local expert, slot) of every row
    recv_expert = torch.repeat_interleave(torch.arange(per_rank, device=device).repeat(world), recv_counts.reshape(-1))
    masked_m = recv_counts.sum(dim=0)
    busiest = int(masked_m.max().item()) if masked_m.numel() else 0
    if busiest > max_m:
        raise RuntimeError(f'an expert received {busiest} rows, more than max_m = {max_m}')
    # slot = rank-major running index inside the expert: offset of (source rank, expert) + position inside that run
    run_offsets = torch.cumsum(recv_counts, dim=0) - recv_counts                       # [world, per_rank]
    run_start = torch.repeat_interleave(run_offsets.reshape(-1), recv_counts.reshape(-1))
    flat_run_begin = torch.cumsum(recv_counts.reshape(-1), dim=0) - recv_counts.reshape(-1)
    pos_in_run = torch.arange(payload.size(0), device=device) - torch.repeat_interleave(flat_run_begin, recv_counts.reshape(-1))
    recv_slot = run_start + pos_in_run

    a = torch.zeros((per_rank, max_m, k), dtype=torch.float8_e4m3fn, device=device)
    sfa = torch.zeros((per_rank, max_m, x_sf.size(1)), dtype=torch.float, device=device)
    a.view(torch.uint8)[recv_expert, recv_slot] = payload.view(torch.uint8)
    sfa[recv_expert, recv_slot] = payload_sf
    plan = DispatchPlan(order, send_splits, recv_splits, recv_expert, recv_slot, masked_m.to(torch.int32))
    return (a, sfa), plan


def combine(d: torch.Tensor, plan: DispatchPlan, tokens: int, top_k: int, group=None,
            topk_weights: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Returns the result rows to their source ranks.  Without weights: ``[tokens, top_k, N]`` in the order of ``expert_ids``;
    with ``topk_weights [tokens, top_k]`` the owner of a token reduces its ``top_k`` rows: ``[tokens, N]`` BF16 =
    ``sum_j topk_weights[t, j] * row(t, j)`` accumulated in FP32 (the reference's fused kernel does
    the same reduction in its combine stage, deep_gemm/include/deep_gemm/impls/sm100_fp8_fp4_mega_moe.cuh:523-595)."""
    rows_out = d[plan.recv_expert, plan.recv_slot]                       # [P_recv, N] in arrival order
    back = _all_to_all_rows(rows_out, plan.recv_splits, plan.send_splits, group)
    out = torch.empty((tokens * top_k, d.size(-1)), dtype=d.dtype, device=d.device)
    out[plan.send_order] = back
    out = out.view(tokens, top_k, d.size(-1))
    if topk_weights is None:
        return out
    return (out.float() * topk_weights.to(torch.float).unsqueeze(-1)).sum(dim=1).to(d.dtype)


@dataclass
class FixedPlan:
    """Bookkeeping of one fixed-capacity dispatch (device tensors only: nothing here ever reached the host)."""
    send_order: torch.Tensor          # [P] permutation that sorts this rank's (row, expert) pairs by destination expert
    pair_dest: torch.Tensor           # [P] flat slot (rank, local expert, position) of every sorted pair in the exchanged blocks;
                                      #     world * per_rank * capacity = the dump slot (no expert, or over capacity: the pair's result is zeros)
    recv_expert: torch.Tensor         # [world * per_rank * capacity] local expert of every received slot
    recv_slot: torch.Tensor           # same shape: row inside that expert's masked block (invalid slots: clamped)
    recv_valid: torch.Tensor          # same shape, bool: the slot holds a row this rank kept (not padding, not over ``max_m``)
    masked_m: torch.Tensor            # [G_local] int32 rows per local expert
    overflow: torch.Tensor            # 0-dim bool: a block or an expert was over capacity (rows were dropped)
    dropped: Optional[torch.Tensor] = None     # 0-dim int64: rows dropped -- over ``capacity`` on this rank as a sender + over ``max_m`` on this rank as a receiver
    row_extra: Optional[torch.Tensor] = None   # [G_local, max_m] FP32: the per-pair value that travelled with the rows (routing weight)


def dispatch_fixed(x: TensorPair, expert_ids: torch.Tensor, num_experts: int, max_m: int, capacity: int,
                   group=None, row_extra: Optional[torch.Tensor] = None) -> Tuple[TensorPair, FixedPlan]:
    """The dispatch without any host synchronisation (a decode step that is captured into a hipGraph cannot read split sizes back):
    every rank sends every peer a block of fixed shape ``[experts per rank, capacity, K + K / 32 bytes]`` plus the row counts of its
    blocks -- two equal-split all-to-alls -- and the receiver compacts the valid rows into the masked layout with index arithmetic
    on the device.  ``capacity`` = the most rows one rank may send to one expert (``tokens`` is always enough: a token names an
    expert at most once); the price is the padding on the wire, ``world * experts_per_rank * capacity`` rows per rank instead of
    ``tokens * top_k``.  Rows over capacity (or over ``max_m`` on the receiver) are dropped and ``plan.overflow`` is set -- check it
    where a synchronisation is affordable.  Entries ``expert_ids[t, j] < 0`` name no expert (the reference's ``-1``,
    tests/test_mega_moe.py:103-121): nothing travels for them and :func:`combine_fixed` returns zeros in their place.
    ``row_extra [T, top_k]`` FP32 rides with the rows (4 more bytes per row: the routing weight the fused expert MLP applies before its
    re-quantisation) and arrives as ``plan.row_extra [G_local, max_m]``.  Returns the local masked operand and the plan."""
    x_fp8, x_sf = x
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    first, last = expert_range(num_experts, rank, world)
    per_rank = last - first
    tokens, top_k = expert_ids.shape
    device = x_fp8.device
    k, sf_bytes = x_fp8.size(1), 4 * x_sf.size(1)
    extra_bytes = 4 if row_extra is not None else 0
    row_bytes = k + sf_bytes + extra_bytes

    flat_expert = expert_ids.reshape(-1).to(torch.int64)
    key = torch.where(flat_expert < 0, torch.full_like(flat_expert, num_experts), flat_expert)      # "no expert" sorts behind every expert
    flat_row = torch.arange(tokens, device=device).repeat_interleave(top_k)
    order = torch.argsort(key, stable=True)
    sorted_key = key[order]
    # (not torch.bincount: it reads the maximum back to size its output -- a device-to-host synchronisation, illegal under stream capture)
    per_expert = torch.zeros(num_experts + 1, dtype=torch.int64, device=device).scatter_add_(0, key, torch.ones_like(key))
    run_begin = torch.cumsum(per_expert, dim=0) - per_expert
    pos = torch.arange(sorted_key.numel(), device=device) - run_begin[sorted_key]
    per_expert = per_expert[:num_experts]
    overflow = (per_expert > capacity).any()
    dropped = (per_expert - capacity).clamp(min=0).sum()
    dump = world * per_rank * capacity
    # flat slot of a pair: ((rank * per_rank + local expert) * capacity + position) = sorted_key * capacity + pos
    pair_dest = torch.where((sorted_key < num_experts) & (pos < capacity), sorted_key * capacity + pos, torch.full_like(pos, dump))

    rows = flat_row[order]
    packed = torch.empty((rows.numel(), row_bytes), dtype=torch.uint8, device=device)
    packed[:, :k] = x_fp8.view(torch.uint8)[rows]
    packed[:, k:k + sf_bytes] = x_sf.contiguous().view(torch.uint8).view(tokens, sf_bytes)[rows]
    if row_extra is not None:
        packed[:, k + sf_bytes:] = row_extra.to(torch.float).reshape(-1, 1).contiguous().view(torch.uint8)[order]
    send_flat = torch.zeros((dump + 1, row_bytes), dtype=torch.uint8, device=device)
    send_flat[pair_dest] = packed
    send = send_flat[:dump].view(world, per_rank, capacity, row_bytes)
    send_counts = per_expert.clamp(max=capacity).view(world, per_rank).to(torch.int32)

    recv = torch.empty_like(send)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts.contiguous(), group=group)
    dist.all_to_all_single(recv, send, group=group)

    # slot of entry (source s, expert g, c): rows of earlier sources first; invalid entries go to the dump row max_m
    offsets = torch.cumsum(recv_counts, dim=0) - recv_counts                        # [world, per_rank]
    c = torch.arange(capacity, device=device).view(1, 1, capacity)
    slot = offsets.unsqueeze(-1) + c
    valid = (c < recv_counts.unsqueeze(-1)) & (slot < max_m)
    overflow = overflow | (recv_counts.sum(dim=0) > max_m).any()
    dropped = dropped + (recv_counts.sum(dim=0).to(torch.int64) - max_m).clamp(min=0).sum()
    recv_expert = torch.arange(per_rank, device=device).view(1, per_rank, 1).expand(world, per_rank, capacity).reshape(-1)
    recv_slot = slot.clamp(max=max_m - 1).reshape(-1).to(torch.int64)
    # row of the flat [per_rank * max_m (+ 1 dump row)] stores: invalid entries all land in the dump row behind the last block
    flat_row = torch.where(valid.reshape(-1), recv_expert * max_m + recv_slot, torch.full_like(recv_slot, per_rank * max_m))
    masked_m = recv_counts.sum(dim=0).clamp(max=max_m).to(torch.int32)

    a_store = torch.zeros((per_rank * max_m + 1, k), dtype=torch.uint8, device=device)
    sf_store = torch.zeros((per_rank * max_m + 1, x_sf.size(1)), dtype=torch.float, device=device)
    flat = recv.view(-1, row_bytes)
    a_store[flat_row] = flat[:, :k]
    sf_store[flat_row] = flat[:, k:k + sf_bytes].contiguous().view(torch.float)
    extra = None
    if row_extra is not None:
        extra_store = torch.zeros((per_rank * max_m + 1,), dtype=torch.float, device=device)
        extra_store[flat_row] = flat[:, k + sf_bytes:].contiguous().view(torch.float).reshape(-1)
        extra = extra_store[:per_rank * max_m].view(per_rank, max_m)
    plan = FixedPlan(order, pair_dest, recv_expert, recv_slot, valid.reshape(-1), masked_m, overflow, dropped=dropped, row_extra=extra)
    a = a_store[:per_rank * max_m].view(torch.float8_e4m3fn).view(per_rank, max_m, k)
    return (a, sf_store[:per_rank * max_m].view(per_rank, max_m, x_sf.size(1))), plan


def combine_fixed(d: torch.Tensor, plan: FixedPlan, tokens: int, top_k: int, world: int, capacity: int, group=None,
                  topk_weights: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The return path of :func:`dispatch_fixed`: result rows travel back in the same fixed-shape blocks (one equal-split all-to-all),
    every rank picks its pairs' rows out of the blocks it gets back (zeros for pairs without an expert, over ``capacity`` on the sender
    or over ``max_m`` on the receiver -- a dropped pair never returns somebody else's row).  Same outputs as :func:`combine`."""
    per_rank, max_m, n = d.size(0), d.size(1), d.size(2)
    picked = d[plan.recv_expert, plan.recv_slot]
    blocks = torch.where(plan.recv_valid.unsqueeze(-1), picked, torch.zeros_like(picked)).view(world, per_rank, capacity, n).contiguous()
    back = torch.zeros((world * per_rank * capacity + 1, n), dtype=d.dtype, device=d.device)           # last row: the dump slot's zeros
    dist.all_to_all_single(back[:-1].view(world, per_rank, capacity, n), blocks, group=group)
    out = torch.empty((tokens * top_k, n), dtype=d.dtype, device=d.device)
    out[plan.send_order] = back[plan.pair_dest]
    out = out.view(tokens, top_k, n)
    if topk_weights is None:
        return out
    return (out.float() * topk_weights.to(torch.float).unsqueeze(-1)).sum(dim=1).to(d.dtype)


def _default_local_gemm(a: TensorPair, b: TensorPair, d: torch.Tensor, masked_m: torch.Tensor, expected_m: int) -> None:
    from .gemm import m_grouped_fp8_gemm_nt_masked          # the HIP path; there is no CPU fallback
    m_grouped_fp8_gemm_nt_masked(a, b, d, masked_m, expected_m)


def ep_m_grouped_fp8_gemm_nt_masked(x: TensorPair, expert_ids: torch.Tensor, b_local: TensorPair, num_experts: int,
                                    max_m: int, expected_m: Optional[int] = None, group=None,
                                    local_gemm: Callable = _default_local_gemm,
                                    topk_weights: Optional[torch.Tensor] = None,
                                    phase_events: Optional[list] = None, capacity: Optional[int] = None) -> torch.Tensor:
    """One expert-parallel step: dispatch -> local masked grouped GEMM -> combine (-> top-k weighted reduce).

    ``b_local = (B [G_local, N, K] fp8, SFB [G_local, N / 128, K / 128])`` are this rank's resident expert weights.
    Returns ``[T, top_k, N]`` BF16 (row ``(t, j)`` is ``x[t] @ B[expert_ids[t, j]]^T``) or, with ``topk_weights``, ``[T, N]``.
    ``phase_events``: a list that receives one ``(e0, e1, e2, e3)`` tuple of recorded CUDA events per call -- dispatch is
    ``e0..e1``, the local GEMM ``e1..e2``, combine ``e2..e3`` (bench.py's time split; only on CUDA tensors).
    ``capacity``: rows one rank may send to one expert -- selects the fixed-shape exchange (:func:`dispatch_fixed`), which never
    synchronises with the host (capturable); ``None`` = exact split sizes (one small device-to-host copy per step).
    """
    tokens, top_k = expert_ids.shape
    marks = []

    def mark():
        if phase_events is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append(ev)
    mark()
    if capacity is None:
        (a, sfa), plan = dispatch(x, expert_ids, num_experts, max_m, group)
    else:
        (a, sfa), plan = dispatch_fixed(x, expert_ids, num_experts, max_m, capacity, group)
    mark()
    groups, n = b_local[0].size(0), b_local[0].size(1)
    d = torch.empty((groups, max_m, n), dtype=torch.bfloat16, device=a.device)
    local_gemm((a, sfa), b_local, d, plan.masked_m, expected_m if expected_m is not None else max(1, max_m // 2))
    mark()
    if capacity is None:
        out = combine(d, plan, tokens, top_k, group, topk_weights)
    else:
        out = combine_fixed(d, plan, tokens, top_k, dist.get_world_size(group), capacity, group, topk_weights)
    mark()
    if phase_events is not None:
        phase_events.append(tuple(marks))
    return out
