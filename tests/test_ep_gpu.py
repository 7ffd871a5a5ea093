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
