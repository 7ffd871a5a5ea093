"""In-kernel dispatch / combine of ``fp8_mega_moe`` over peer-mapped memory (csrc/fp8_gemm_moe.hpp; reference: the dispatch / combine stages of
sm100_fp8_fp4_mega_moe.cuh:357-405, 523-595 and the NVLink barrier comm/barrier.cuh:47-83) on the ONE GPU of the test box:

* one process, one rank (every "peer" is the own region): the whole five-launch step against the one-rank scatter / gather path, bit for bit;
* TWO processes on device 0 (gloo for the control plane only) that map each other's symmetric region through ``hipIpcGetMemHandle`` /
  ``hipIpcOpenMemHandle``: rank r owns experts [r E / 2, (r + 1) E / 2); every rank's ``y`` must equal, bit for bit, what ONE rank holding every
  expert computes for the same tokens (the expectation of tests/test_mega_gloo.py, evaluated here by the one-rank HIP path), over several
  steps (epochs), with uneven and empty token sets; rows over a capacity are dropped and counted on their sender; a missing peer ends in
  a timed-out, flagged wait -- not in a hung device.

What this cannot show: xGMI timing and cross-DEVICE cache behaviour (two GPUs) -- stated in DESIGN.md section 7."""
import os
import sys
import traceback

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _expert(e: int, hidden: int, inter: int):
    """Expert e's quantised (W1 [2I, H], W2 [H, I]) pairs, reproducible in any process (as tests/test_mega_gloo.py)."""
    from deepgemm_amd.utils.math import per_block_cast_to_fp8
    g = torch.Generator().manual_seed(7000 + e)
    w1 = torch.randn((2 * inter, hidden), dtype=torch.bfloat16, generator=g) / hidden ** 0.5
    w2 = torch.randn((hidden, inter), dtype=torch.bfloat16, generator=g) / inter ** 0.5
    return per_block_cast_to_fp8(w1, use_ue8m0=False), per_block_cast_to_fp8(w2, use_ue8m0=False)


def _weights(first: int, last: int, hidden: int, inter: int):
    import deepgemm_amd as dg
    local = [_expert(e, hidden, inter) for e in range(first, last)]
    l1 = (torch.stack([q[0][0] for q in local]).cuda(), torch.stack([q[0][1] for q in local]).cuda())
    l2 = (torch.stack([q[1][0] for q in local]).cuda(), torch.stack([q[1][1] for q in local]).cuda())
    return dg.transform_weights_for_mega_moe(l1, l2)


def _inputs(rank: int, step: int, tokens: int, num_experts: int, top_k: int, hidden: int):
    from deepgemm_amd.utils.math import per_token_cast_to_fp8
    g = torch.Generator().manual_seed(300 + 17 * rank + 1000 * step)
    x = per_token_cast_to_fp8(torch.randn((max(tokens, 1), hidden), dtype=torch.bfloat16, generator=g), use_ue8m0=False)
    scores = torch.rand((max(tokens, 1), num_experts), generator=g)
    w, idx = torch.topk(scores, top_k, dim=1)
    idx = idx.to(torch.int64)
    idx[0, -1] = -1                                                  # an entry without an expert
    if tokens > 2:
        idx[2, 0] = -1
    return (x[0][:tokens], x[1][:tokens]), idx[:tokens], w.float()[:tokens]


def _fill(buf, x, idx, w):
    t = x[0].size(0)
    buf.x[:t].copy_(x[0]); buf.x_sf[:t].copy_(x[1])
    buf.topk_idx[:t].copy_(idx); buf.topk_weights[:t].copy_(w)


def _one_rank(ref_buf, all_l1, all_l2, x, idx, w, hidden, clamp):
    """The expectation: ONE rank holding every expert (the scatter / gather path of fp8_mega_moe at world size 1)."""
    import deepgemm_amd as dg
    t = x[0].size(0)
    y = torch.full((t, hidden), float('nan'), dtype=torch.bfloat16, device='cuda')
    if t == 0:
        return y
    _fill(ref_buf, x, idx, w)
    dg.fp8_mega_moe(y, all_l1, all_l2, ref_buf, activation_clamp=clamp)
    return y


def test_p2p_protocol_with_one_rank_is_the_scatter_gather_path():
    import deepgemm_amd as dg
    from deepgemm_amd import mega
    num_experts, top_k, hidden, inter, max_tokens = 8, 4, 512, 256, 64
    l1, l2 = _weights(0, num_experts, hidden, inter)
    ref_buf = mega.SymmBuffer(None, num_experts, max_tokens, top_k, hidden, inter)
    buf = mega.SymmBuffer(None, num_experts, max_tokens, top_k, hidden, inter, p2p=True)
    assert buf.p2p and buf.world == 1 and buf.l1_acts.data_ptr() == buf._region + buf._offsets['l1_acts']
    try:
        for step, tokens in enumerate((37, 64, 0, 5, 64)):
            x, idx, w = _inputs(0, step, tokens, num_experts, top_k, hidden)
            x, idx, w = (x[0].cuda(), x[1].cuda()), idx.cuda(), w.cuda()
            want = _one_rank(ref_buf, l1, l2, x, idx, w, hidden, 10.0)
            _fill(buf, x, idx, w)
            y = torch.full((tokens, hidden), float('nan'), dtype=torch.bfloat16, device='cuda')
            stats = torch.zeros((num_experts,), dtype=torch.int, device='cuda')
            dg.fp8_mega_moe(y, l1, l2, buf, cumulative_local_expert_recv_stats=stats, activation_clamp=10.0)
            torch.cuda.synchronize()
            assert buf.errors.tolist() == [0, 0, 0, 0], (step, buf.errors.tolist())
            assert int(stats.sum()) == int((idx >= 0).sum()) and (tokens == 0 or torch.equal(stats, ref_buf.masked_m))
            assert torch.equal(y.view(torch.int16), want.view(torch.int16)), (step, tokens)
            assert int(buf._region_view[buf._offsets['counts']:buf._offsets['counts'] + 4 * num_experts].view(torch.int32).abs().sum()) == 0
    finally:
        buf.destroy(); ref_buf.destroy()


@pytest.mark.parametrize('num_experts,top_k,hidden,inter,tokens', [(16, 9, 2304, 256, 33), (8, 2, 4096, 384, 64), (32, 8, 1024, 128, 7)])
def test_p2p_one_rank_other_geometries(num_experts, top_k, hidden, inter, tokens):
    """More than eight entries per token (the reduce kernel's second batch of rows in flight), a hidden size over several 2048-column blocks with a
    partial last one, few tokens: the five-launch step against the scatter / gather path, bit for bit; both against themselves on a second step."""
    import deepgemm_amd as dg
    from deepgemm_amd import mega
    max_tokens = 64
    l1, l2 = _weights(0, num_experts, hidden, inter)
    ref_buf = mega.SymmBuffer(None, num_experts, max_tokens, top_k, hidden, inter)
    buf = mega.SymmBuffer(None, num_experts, max_tokens, top_k, hidden, inter, p2p=True)
    try:
        for step in range(2):
            x, idx, w = _inputs(3, step, tokens, num_experts, top_k, hidden)
            x, idx, w = (x[0].cuda(), x[1].cuda()), idx.cuda(), w.cuda()
            want = _one_rank(ref_buf, l1, l2, x, idx, w, hidden, 10.0)
            _fill(buf, x, idx, w)
            y = torch.full((tokens, hidden), float('nan'), dtype=torch.bfloat16, device='cuda')
            dg.fp8_mega_moe(y, l1, l2, buf, activation_clamp=10.0)
            torch.cuda.synchronize()
            assert buf.errors.tolist() == [0, 0, 0, 0], (step, buf.errors.tolist())
            assert torch.equal(y.view(torch.int16), want.view(torch.int16)), step
    finally:
        buf.destroy(); ref_buf.destroy()


def _worker(rank: int, world: int, port: int, queue, scenario: str):
    try:
        import torch.distributed as dist
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        torch.cuda.set_device(0)                                    # BOTH ranks on the one GPU of the box
        dist.init_process_group('gloo', rank=rank, world_size=world)
        import deepgemm_amd as dg
        from deepgemm_amd import ep, mega
        group = dist.group.WORLD
        num_experts, top_k, hidden, inter, max_tokens = 8, 4, 512, 256, 64
        first, last = ep.expert_range(num_experts, rank, world)
        l1, l2 = _weights(first, last, hidden, inter)
        all_l1, all_l2 = _weights(0, num_experts, hidden, inter)
        ref_buf = mega.SymmBuffer(None, num_experts, max_tokens, top_k, hidden, inter)
        capacity = 64 if scenario == 'overflow' else None
        buf = mega.get_symm_buffer_for_mega_moe(group, num_experts, max_tokens, top_k, hidden, inter, expert_capacity=capacity)
        assert buf.p2p and buf.world == world and buf.rank == rank and buf.num_local_experts == num_experts // world, 'the peers of one GPU must be mappable'
        report = {'rank': rank, 'fine_grained': buf.fine_grained}
        token_plan = {'steps': [(37, 64), (64, 3), (0, 41), (5, 0), (64, 64)], 'overflow': [(64, 64)], 'timeout': [(16, 16), (16, 16)]}[scenario]
        mega.set_p2p_timeout_us(200_000 if scenario == 'timeout' else 10_000_000)
        for step, per_rank in enumerate(token_plan):
            tokens = per_rank[rank]
            x, idx, w = _inputs(rank, step, tokens, num_experts, top_k, hidden)
            if scenario == 'overflow':
                idx = torch.stack([torch.arange(top_k) for _ in range(tokens)]).to(torch.int64)      # every token of both ranks names experts 0 .. top_k - 1
            x, idx, w = (x[0].cuda(), x[1].cuda()), idx.cuda(), w.cuda()
            want = _one_rank(ref_buf, all_l1, all_l2, x, idx, w, hidden, 10.0)
            _fill(buf, x, idx, w)
            y = torch.full((tokens, hidden), float('nan'), dtype=torch.bfloat16, device='cuda')
            if scenario == 'timeout' and step == 1 and rank == 1:
                break                                               # rank 1 is "lost": it never makes the second call
            dg.fp8_mega_moe(y, l1, l2, buf, activation_clamp=10.0)
            torch.cuda.synchronize()                                # (returns: every wait is bounded)
            errors = buf.errors.tolist()
            if scenario == 'steps':
                assert errors == [0, 0, 0, 0], (step, errors)
                assert torch.equal(y.view(torch.int16), want.view(torch.int16)), (rank, step, tokens)
            elif scenario == 'overflow':
                # 128 rows for each of the experts 0 .. 3 (all on rank 0), 64 fit: every expert drops 64, counted on the SENDERS; a delivered
                # pair contributes exactly what it contributes in the one-rank result, a dropped one nothing -- y stays finite
                counts = [None] * world
                dist.all_gather_object(counts, errors[0])
                assert sum(counts) == top_k * (2 * 64 - 64), counts
                assert errors[1:] == [0, 0, 0] and bool(torch.isfinite(y.float()).all())
                report['masked_m'] = buf.masked_m.tolist()
                assert buf.masked_m.tolist() == ([64] * top_k if rank == 0 else [0] * (num_experts // world))
            elif scenario == 'timeout' and step == 1:
                assert errors[2] > 0 and errors[3] > 0, errors      # rank 0: neither rank 1's rows nor its returns ever came -- counted, not hung
                report['errors'] = errors
        buf.destroy()                                               # (a barrier inside: nobody unmaps a region a peer may still write to)
        ref_buf.destroy()
        queue.put((rank, 'ok', report))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:                                               # noqa: BLE001
        queue.put((rank, 'fail', traceback.format_exc()))


def _run(scenario: str, timeout_s: int = 240):
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    queue = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, queue, scenario)) for r in range(2)]
    for p in procs:
        p.start()
    results = []
    try:
        for _ in procs:
            results.append(queue.get(timeout=timeout_s))
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()                                            # (the exact processes this test started)
    bad = [r for r in results if r[1] != 'ok']
    assert not bad and len(results) == 2, bad
    return {r[0]: r[2] for r in results}


def test_two_processes_on_one_gpu_map_each_other_and_match_one_rank():
    reports = _run('steps')
    assert set(reports) == {0, 1}


def test_rows_over_a_capacity_are_dropped_and_counted_on_their_sender():
    _run('overflow')


def test_a_lost_peer_ends_in_a_flagged_timeout_not_a_hang():
    reports = _run('timeout', timeout_s=120)
    assert reports[0]['errors'][2] > 0
