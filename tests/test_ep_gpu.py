"""Expert-parallel harness on the GPU: the real path (RCCL all-to-all + HIP masked grouped GEMM) at world_size 1 on the
1-GPU test box, checked against the oracle; the multi-rank logic is covered on CPU by tests/test_ep_gloo.py."""
import os

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def test_ep_step_single_rank_nccl():
    import oracle
    from deepgemm_amd import ep
    from deepgemm_amd.utils.math import per_block_cast_to_fp8, per_token_cast_to_fp8
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29711')
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        torch.manual_seed(7)
        num_experts, tokens, top_k, n, k, max_m = 8, 40, 4, 512, 1024, 64
        w = torch.randn((num_experts, n, k), dtype=torch.bfloat16, device='cuda')
        bq = [per_block_cast_to_fp8(w[e], use_ue8m0=False) for e in range(num_experts)]
        b_local = (torch.stack([q[0] for q in bq]), torch.stack([q[1] for q in bq]))
        x = per_token_cast_to_fp8(torch.randn((tokens, k), dtype=torch.bfloat16, device='cuda'), use_ue8m0=False)
        ids = torch.stack([torch.randperm(num_experts, device='cuda')[:top_k] for _ in range(tokens)])
        out = ep.ep_m_grouped_fp8_gemm_nt_masked(x, ids, b_local, num_experts, max_m)
        torch.cuda.synchronize()
        assert out.shape == (tokens, top_k, n)
        out_cpu, ids_cpu = out.cpu(), ids.cpu()
        for t in range(0, tokens, 7):
            for j in range(top_k):
                e = int(ids_cpu[t, j])
                want = torch.empty((1, n), dtype=torch.bfloat16)
                oracle.fp8_gemm_nt(x[0][t:t + 1].cpu(), x[1][t:t + 1].cpu(), bq[e][0].cpu(), bq[e][1].cpu(), want)
                diff = (out_cpu[t, j].float() - want[0].float()).abs().max().item()
                scale = want[0].float().abs().max().item()
                assert diff <= 2 ** -7 * scale, (t, j, e, diff, scale)        # one BF16 ulp of the row's largest value
        # the fixed-capacity exchange (no host synchronisation): same bits
        fixed = ep.ep_m_grouped_fp8_gemm_nt_masked(x, ids, b_local, num_experts, max_m, capacity=tokens)
        torch.cuda.synchronize()
        assert torch.equal(fixed, out)
        # ... and it really is free of host synchronisation: torch's sync debug mode turns every implicit device-to-host wait
        # (an .item(), a nonzero(), torch.bincount sizing its output ...) into an error for the duration of the step -- the property a
        # hipGraph capture of a decode step needs.  (Capturing the RCCL exchange itself at world size 1 is not attempted here.)
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode('error')
        try:
            (a_fix, sfa_fix), plan = ep.dispatch_fixed(x, ids, num_experts, max_m, tokens)
            d_fix = torch.empty((num_experts, max_m, n), dtype=torch.bfloat16, device='cuda')
            import deepgemm_amd as dg
            dg.m_grouped_fp8_gemm_nt_masked((a_fix, sfa_fix), b_local, d_fix, plan.masked_m, max(1, tokens * top_k // num_experts))
            again = ep.combine_fixed(d_fix, plan, tokens, top_k, 1, tokens)
        finally:
            torch.cuda.set_sync_debug_mode('default')
        torch.cuda.synchronize()
        assert torch.equal(again, out)
    finally:
        if created:
            dist.destroy_process_group()


def test_mega_moe_exchange_path_single_rank_nccl():
    """The multi-rank form of fp8_mega_moe (fixed-shape all-to-alls + routing weight travelling with the rows + the two HIP GEMM launches
    + FP32 top-k sum; deepgemm_amd/mega.py::_mega_moe_ep) on the GPU: one rank over RCCL, forced onto that path -- the same bits as the
    one-rank scatter / combine kernels, no host synchronisation.  (More than one rank: tests/test_mega_gloo.py on CPU, oracle as the
    local operators; an N > 1 GPU run needs a multi-GPU node.)"""
    import deepgemm_amd as dg
    from deepgemm_amd import mega
    from deepgemm_amd.utils.math import per_block_cast_to_fp8, per_token_cast_to_fp8
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29711')
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        torch.manual_seed(11)
        tokens, experts, topk, hidden, inter = 37, 8, 3, 512, 256
        x = per_token_cast_to_fp8(torch.randn((tokens, hidden), device='cuda', dtype=torch.bfloat16), use_ue8m0=False)
        cast = lambda w: tuple(torch.stack(t) for t in zip(*[per_block_cast_to_fp8(w[g], use_ue8m0=False) for g in range(experts)]))   # noqa: E731
        w1 = cast(torch.randn((experts, 2 * inter, hidden), device='cuda', dtype=torch.bfloat16) / hidden ** 0.5)
        w2 = cast(torch.randn((experts, hidden, inter), device='cuda', dtype=torch.bfloat16) / inter ** 0.5)
        topk_w, topk_idx = torch.topk(torch.rand((tokens, experts), device='cuda'), topk, dim=1)
        topk_idx = topk_idx.to(torch.int64)
        topk_idx[0, -1] = -1
        l1_t, l2_t = dg.transform_weights_for_mega_moe(w1, w2)
        outs = []
        for force in (False, True):
            buf = mega.SymmBuffer(dist.group.WORLD, experts, tokens, topk, hidden, inter, force_exchange=force)
            assert buf.exchange == force
            buf.x[:tokens].copy_(x[0]); buf.x_sf[:tokens].copy_(x[1])
            buf.topk_idx[:tokens].copy_(topk_idx); buf.topk_weights[:tokens].copy_(topk_w.float())
            stats = torch.zeros((experts,), dtype=torch.int, device='cuda')
            y = torch.full((tokens, hidden), float('nan'), device='cuda', dtype=torch.bfloat16)
            dg.fp8_mega_moe(y, l1_t, l2_t, buf, cumulative_local_expert_recv_stats=stats, activation_clamp=10.0)     # warm-up (workspaces)
            torch.cuda.synchronize()
            if force:
                stats.zero_()
                torch.cuda.set_sync_debug_mode('error')
                try:
                    dg.fp8_mega_moe(y, l1_t, l2_t, buf, cumulative_local_expert_recv_stats=stats, activation_clamp=10.0)
                finally:
                    torch.cuda.set_sync_debug_mode('default')
                torch.cuda.synchronize()
            assert int(buf.errors[0]) == 0 and int(buf.errors[1]) == 0
            counts = torch.bincount(topk_idx[topk_idx >= 0].flatten(), minlength=experts).to(torch.int)
            assert torch.equal(stats, counts)
            outs.append(y)
        assert torch.equal(outs[0], outs[1]), f'max |diff| {(outs[0].float() - outs[1].float()).abs().max().item():.3e}'
    finally:
        if created:
            dist.destroy_process_group()
