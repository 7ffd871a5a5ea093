"""The fused expert-MLP hand-off (single-GPU half of the reference's Mega-MoE, deep_gemm/mega/__init__.py:155,
impls/sm100_fp8_fp4_mega_moe.cuh): GEMM1 with SwiGLU + per-token FP8 re-quantisation in its epilogue against the UNFUSED pipeline
``m_grouped_fp8_gemm_nt_masked -> BF16 -> torch SwiGLU -> reference per_token_cast_to_fp8`` -- bit-exact on the re-quantised bytes and
scales -- and the two-GEMM expert MLP end to end."""
import os

import pytest
import torch

import deepgemm_amd as dg
from deepgemm_amd._lib import lib
from deepgemm_amd.testing import calc_diff, generators as gen
from deepgemm_amd.utils.math import per_block_cast_to_fp8, per_token_cast_to_fp8

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _auto_config():
    dg.set_forced_config('auto')
    dg.set_sf_cast_mode('sm90')
    yield
    dg.set_forced_config('auto')


def _weights(groups, n, k):
    w = torch.randn((groups, n, k), device='cuda', dtype=torch.bfloat16)
    q = [per_block_cast_to_fp8(w[g], use_ue8m0=False) for g in range(groups)]
    return torch.stack([x[0] for x in q]), torch.stack([x[1] for x in q])


def _unfused(x, w1, masked_ms, clamp, use_ue8m0):
    """The oracle pipeline on the ORIGINAL (gate rows first, up rows last) weights: masked GEMM -> BF16 -> SwiGLU -> per-token cast."""
    groups, m, _ = x[0].shape
    inter = w1[0].size(1) // 2
    h = torch.empty((groups, m, 2 * inter), device='cuda', dtype=torch.bfloat16)
    dg.m_grouped_fp8_gemm_nt_masked(x, w1, h, torch.tensor(masked_ms, dtype=torch.int, device='cuda'), max(1, max(masked_ms)))
    out = []
    for g, rows in enumerate(masked_ms):
        gate, up = h[g, :rows, :inter].float(), h[g, :rows, inter:].float()
        if clamp is not None:
            c = float(torch.tensor(clamp).to(torch.bfloat16))      # the reference clamps with BF16 operands: 10.1 acts as 10.125
            gate, up = gate.clamp(max=c), up.clamp(-c, c)
        y = (torch.nn.functional.silu(gate) * up).to(torch.bfloat16)
        out.append(per_token_cast_to_fp8(y, use_ue8m0=use_ue8m0) if rows else None)
    return out


@pytest.mark.parametrize('masked_ms,m_max,inter,k', [([5, 0, 64, 33], 64, 256, 512), ([200, 1, 129], 256, 384, 1024),
                                                      ([48, 64, 17, 0, 64, 33, 2, 60], 64, 2048, 7168)])
@pytest.mark.parametrize('use_ue8m0', [False, True])
@pytest.mark.parametrize('clamp', [None, 10.0, 0.3])          # 0.3 is not BF16-representable (acts as 0.30078125) and actually clips
@pytest.mark.parametrize('one_per_cu', [False, True])         # the 3-stage ring with two workgroups per CU (default) / the 6-stage ring, one per CU
def test_fused_swiglu_requant_is_the_unfused_pipeline(masked_ms, m_max, inter, k, use_ue8m0, clamp, one_per_cu, monkeypatch):
    if one_per_cu:
        if clamp == 10.0:
            pytest.skip('the 6-stage form: two of the three clamp cases are enough')
        monkeypatch.setenv('DG_SWIGLU_ONE_PER_CU', '1')
    lib.dg_reload_env()
    gen.reset_seed(len(masked_ms) + inter)
    groups = len(masked_ms)
    a = torch.randn((groups, m_max, k), device='cuda', dtype=torch.bfloat16)
    xq = [per_token_cast_to_fp8(a[g], use_ue8m0=False) for g in range(groups)]
    x = (torch.stack([q[0] for q in xq]), torch.stack([q[1] for q in xq]))
    w1 = _weights(groups, 2 * inter, k)
    (w1_t, _) = dg.transform_weights_for_mega_moe(w1, w1)
    # the transform: 64-row block j of the interleaved rows = gate block j / 2 or up block j / 2; scale rows alternate gate / up
    assert torch.equal(w1_t[0].view(torch.uint8)[:, 64:128], w1[0].view(torch.uint8)[:, inter:inter + 64])
    assert torch.equal(w1_t[0].view(torch.uint8)[:, 128:192], w1[0].view(torch.uint8)[:, 64:128])
    assert torch.equal(w1_t[1][:, 2], w1[1][:, 1]) and torch.equal(w1_t[1][:, 1], w1[1][:, inter // 128])
    masked = torch.tensor(masked_ms, dtype=torch.int, device='cuda')
    q, q_sf = dg.empty_intermediate(groups, m_max, inter, 'cuda')
    q.view(torch.uint8).fill_(0x7f)                       # NaN poison: rows >= masked_m must stay untouched
    q_sf.fill_(float('nan'))
    dg.m_grouped_fp8_gemm_nt_masked_swiglu(x, w1_t, (q, q_sf), masked, max(1, max(masked_ms)), activation_clamp=clamp, use_ue8m0=use_ue8m0)
    assert dg.last_config() == ('stream_swiglu_64x128' if one_per_cu else 'stream_swiglu2_64x128')
    monkeypatch.delenv('DG_SWIGLU_ONE_PER_CU', raising=False)
    lib.dg_reload_env()
    want = _unfused(x, w1, masked_ms, clamp, use_ue8m0)
    for g, rows in enumerate(masked_ms):
        if rows:
            wq, wsf = want[g]
            assert torch.equal(q[g, :rows].view(torch.uint8), wq.view(torch.uint8)), f'group {g}: re-quantised bytes differ from the unfused pipeline'
            assert torch.equal(q_sf[g, :rows], wsf), f'group {g}: scales differ'
        assert bool((q[g, rows:].view(torch.uint8) == 0x7f).all()), f'group {g}: rows >= masked_m were written'
        assert bool(torch.isnan(q_sf[g, rows:]).all())


def test_expert_mlp_end_to_end():
    """fp8_mega_moe_local (fused GEMM1 + masked GEMM2) == the unfused two-GEMM pipeline, bit for bit, and close to the BF16 MLP."""
    gen.reset_seed(3)
    groups, m_max, hidden, inter = 8, 64, 1024, 512
    masked_ms = [64, 3, 0, 40, 17, 64, 1, 30]
    a = torch.randn((groups, m_max, hidden), device='cuda', dtype=torch.bfloat16)
    xq = [per_token_cast_to_fp8(a[g], use_ue8m0=False) for g in range(groups)]
    x = (torch.stack([q[0] for q in xq]), torch.stack([q[1] for q in xq]))
    w1_bf16 = torch.randn((groups, 2 * inter, hidden), device='cuda', dtype=torch.bfloat16) / hidden ** 0.5
    w2_bf16 = torch.randn((groups, hidden, inter), device='cuda', dtype=torch.bfloat16) / inter ** 0.5
    cast = lambda w: tuple(torch.stack(t) for t in zip(*[per_block_cast_to_fp8(w[g], use_ue8m0=False) for g in range(groups)]))   # noqa: E731
    w1, w2 = cast(w1_bf16), cast(w2_bf16)
    w1_t, w2_t = dg.transform_weights_for_mega_moe(w1, w2)
    masked = torch.tensor(masked_ms, dtype=torch.int, device='cuda')
    y = torch.full((groups, m_max, hidden), float('nan'), device='cuda', dtype=torch.bfloat16)
    dg.fp8_mega_moe_local(x, w1_t, w2_t, y, masked, 32)
    mids = _unfused(x, w1, masked_ms, None, False)
    for g, rows in enumerate(masked_ms):
        if not rows:
            continue
        mid = (mids[g][0].unsqueeze(0), mids[g][1].unsqueeze(0))
        y_ref = torch.empty((1, rows, hidden), device='cuda', dtype=torch.bfloat16)
        dg.m_grouped_fp8_gemm_nt_masked(mid, (w2[0][g:g + 1], w2[1][g:g + 1]), y_ref, torch.tensor([rows], dtype=torch.int, device='cuda'), rows)
        assert calc_diff(y[g, :rows], y_ref[0]) < 2e-6
        hbf = a[g, :rows].float() @ w1_bf16[g].float().t()
        mlp = (torch.nn.functional.silu(hbf[:, :inter]) * hbf[:, inter:]) @ w2_bf16[g].float().t()
        assert calc_diff(y[g, :rows], mlp) < 3e-3, (g, calc_diff(y[g, :rows], mlp))      # two FP8 quantisations deep
        assert bool(torch.isnan(y[g, rows:]).all())


def test_fused_kernel_under_a_small_cu_limit():
    """set_num_sms(1) / (3): the persistent walk still runs its tiles in partner pairs (an even grid of at least two workgroups)."""
    gen.reset_seed(11)
    groups, m_max, inter, k = 3, 64, 256, 512
    masked_ms = [64, 7, 33]
    a = torch.randn((groups, m_max, k), device='cuda', dtype=torch.bfloat16)
    xq = [per_token_cast_to_fp8(a[g], use_ue8m0=False) for g in range(groups)]
    x = (torch.stack([q[0] for q in xq]), torch.stack([q[1] for q in xq]))
    w1 = _weights(groups, 2 * inter, k)
    w1_t, _ = dg.transform_weights_for_mega_moe(w1, w1)
    masked = torch.tensor(masked_ms, dtype=torch.int, device='cuda')
    want = dg.empty_intermediate(groups, m_max, inter, 'cuda')
    dg.m_grouped_fp8_gemm_nt_masked_swiglu(x, w1_t, want, masked, 40)
    saved = dg.get_num_sms()
    try:
        for limit in (1, 3):
            dg.set_num_sms(limit)
            got = dg.empty_intermediate(groups, m_max, inter, 'cuda')
            dg.m_grouped_fp8_gemm_nt_masked_swiglu(x, w1_t, got, masked, 40)
            for g, rows in enumerate(masked_ms):
                assert torch.equal(got[0][g, :rows].view(torch.uint8), want[0][g, :rows].view(torch.uint8)) and torch.equal(got[1][g, :rows], want[1][g, :rows])
    finally:
        dg.set_num_sms(saved)


def _quantised_tokens(tokens, hidden):
    a = torch.randn((tokens, hidden), device='cuda', dtype=torch.bfloat16)
    return per_token_cast_to_fp8(a, use_ue8m0=False)


def _unfused_moe(x, topk_idx, topk_w, w1, w2, clamp):
    """The reference-shaped operator spelled out with the plain operators, token by token and expert by expert (the reference's own
    baseline has the same stages: dispatch -> L1 GEMM -> SwiGLU * weight -> FP8 -> L2 GEMM -> combine, tests/test_mega_moe.py:149-214):
    masked GEMM -> BF16 -> SwiGLU * routing weight in FP32 -> per_token_cast_to_fp8 -> masked GEMM -> sum in top-k order (FP32)."""
    tokens, hidden = x[0].shape
    experts, inter = w1[0].size(0), w1[0].size(1) // 2
    y = torch.zeros((tokens, hidden), dtype=torch.float, device='cuda')
    one = torch.tensor([1], dtype=torch.int, device='cuda')
    for j in range(topk_idx.size(1)):
        for t in range(tokens):
            e = int(topk_idx[t, j])
            if e < 0:
                continue
            xa = (x[0][t:t + 1].unsqueeze(0).contiguous(), x[1][t:t + 1].unsqueeze(0).contiguous())
            h = torch.empty((1, 1, 2 * inter), device='cuda', dtype=torch.bfloat16)
            dg.m_grouped_fp8_gemm_nt_masked(xa, (w1[0][e:e + 1], w1[1][e:e + 1]), h, one, 1)
            gate, up = h[0, :, :inter].float(), h[0, :, inter:].float()
            if clamp is not None:
                c = float(torch.tensor(clamp).to(torch.bfloat16))
                gate, up = gate.clamp(max=c), up.clamp(-c, c)
            act = torch.nn.functional.silu(gate) * up * topk_w[t, j]          # FP32 up to the cast (sm100_fp8_fp4_mega_moe.cuh:1001-1020)
            q, q_sf = per_token_cast_to_fp8(act, use_ue8m0=False)
            o = torch.empty((1, 1, hidden), device='cuda', dtype=torch.bfloat16)
            dg.m_grouped_fp8_gemm_nt_masked((q.unsqueeze(0), q_sf.unsqueeze(0)), (w2[0][e:e + 1], w2[1][e:e + 1]), o, one, 1)
            y[t] += o[0, 0].float()
    return y.to(torch.bfloat16)


@pytest.mark.parametrize('tokens,experts,topk,hidden,inter,clamp', [(37, 8, 2, 512, 256, None), (64, 4, 3, 1024, 512, 10.0), (5, 16, 4, 256, 128, None),
                                                                         (21, 16, 9, 2304, 128, None)])     # (round 6: > 8 entries per token, two column blocks of the combine)
def test_reference_shaped_mega_moe_entry(tokens, experts, topk, hidden, inter, clamp):
    """fp8_mega_moe(y, l1, l2, sym_buffer) (deep_gemm/mega/__init__.py:155-173) at world size 1: routing -> fused L1 -> L2 -> combine ==
    the unfused pipeline bit for bit; entries without an expert (-1) are skipped; the per-expert counts land in the stats tensor; the
    whole call replays as a hipGraph with different inputs.
    The comparator (`_unfused_moe`) is THIS LIBRARY's own operators called row by row plus torch -- not the oracle: it proves the
    orchestration (scatter, slots, weights, combine order) and the fusion, not the GEMM arithmetic.  Each stage is pinned to the C oracle
    elsewhere: the masked GEMM in tests/test_gemm_gpu.py / test_full_output_parity_gpu.py (every element of C5), the fused SwiGLU +
    re-quantisation in test_fused_swiglu_against_the_c_oracle below, the casts in tests/test_quant_gpu.py; the multi-rank form of the
    same operator is checked against the oracle directly in tests/test_mega_gloo.py."""
    gen.reset_seed(tokens + experts)
    x = _quantised_tokens(tokens, hidden)
    cast = lambda w: tuple(torch.stack(t) for t in zip(*[per_block_cast_to_fp8(w[g], use_ue8m0=False) for g in range(experts)]))   # noqa: E731
    w1 = cast(torch.randn((experts, 2 * inter, hidden), device='cuda', dtype=torch.bfloat16) / hidden ** 0.5)
    w2 = cast(torch.randn((experts, hidden, inter), device='cuda', dtype=torch.bfloat16) / inter ** 0.5)
    scores = torch.rand((tokens, experts), device='cuda')
    topk_w, topk_idx = torch.topk(scores, topk, dim=1)
    topk_idx = topk_idx.to(torch.int64)
    topk_idx[0, -1] = -1                                   # an entry without an expert
    topk_w = topk_w.float()
    buf = dg.get_symm_buffer_for_mega_moe(None, experts, tokens, topk, hidden, inter)
    assert buf.num_max_tokens_per_rank % dg.get_token_alignment_for_mega_moe() == 0
    l1_t, l2_t = dg.transform_weights_for_mega_moe(w1, w2)

    def fill(xp, idx, wts):
        buf.x[:tokens].copy_(xp[0]); buf.x_sf[:tokens].copy_(xp[1])
        buf.topk_idx[:tokens].copy_(idx); buf.topk_weights[:tokens].copy_(wts)

    fill(x, topk_idx, topk_w)
    stats = torch.zeros((experts,), dtype=torch.int, device='cuda')
    y = torch.full((tokens, hidden), float('nan'), device='cuda', dtype=torch.bfloat16)
    dg.fp8_mega_moe(y, l1_t, l2_t, buf, cumulative_local_expert_recv_stats=stats, activation_clamp=clamp)
    want = _unfused_moe(x, topk_idx, topk_w, w1, w2, clamp)
    assert torch.equal(y, want), f'max |diff| {(y.float() - want.float()).abs().max().item():.3e}'
    counts = torch.bincount(topk_idx[topk_idx >= 0].flatten(), minlength=experts).to(torch.int)
    assert torch.equal(stats, counts) and int(buf.errors[0]) == 0
    # hipGraph: capture once (the call above warmed every workspace up), replay on new inputs
    graph = torch.cuda.CUDAGraph()
    y2 = torch.empty_like(y)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dg.fp8_mega_moe(y2, l1_t, l2_t, buf, activation_clamp=clamp)          # (the capture stream's own workspaces)
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(graph, stream=side):
        dg.fp8_mega_moe(y2, l1_t, l2_t, buf, activation_clamp=clamp)
    x_b = _quantised_tokens(tokens, hidden)
    idx_b = torch.topk(torch.rand((tokens, experts), device='cuda'), topk, dim=1)[1].to(torch.int64)
    fill(x_b, idx_b, topk_w)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(y2, _unfused_moe(x_b, idx_b, topk_w, w1, w2, clamp))
    # the reference's names resolve, what is outside this library says so
    assert dg.fp8_fp4_mega_moe is dg.fp8_mega_moe
    with pytest.raises(RuntimeError, match='outside this library'):
        dg.bf16_mega_moe()
    with pytest.raises(RuntimeError, match='not supported on gfx950'):
        dg.transform_weights_for_mega_moe((w1[0], torch.zeros((experts, 2 * inter, hidden // 512), dtype=torch.int, device='cuda')), w2)
    buf.destroy()


def test_exchange_wait_is_bounded_and_loud():
    """A lost partner (fault injection: odd tiles never publish) ends in NaN scales and a host-visible error count within the configured
    bound -- not in a hung device (reference: comm/barrier.cuh:12,36-40) -- and the workspace is re-zeroed for the next launch."""
    from deepgemm_amd import mega
    from deepgemm_amd._lib import lib
    gen.reset_seed(5)
    groups, m_max, inter, k = 2, 64, 256, 512
    masked_ms = [64, 9]
    a = torch.randn((groups, m_max, k), device='cuda', dtype=torch.bfloat16)
    xq = [per_token_cast_to_fp8(a[g], use_ue8m0=False) for g in range(groups)]
    x = (torch.stack([q[0] for q in xq]), torch.stack([q[1] for q in xq]))
    w1_t, _ = dg.transform_weights_for_mega_moe(_weights(groups, 2 * inter, k), _weights(groups, 2 * inter, k))
    masked = torch.tensor(masked_ms, dtype=torch.int, device='cuda')
    ws = mega.swiglu_workspace(groups, m_max, 2 * inter, 'cuda')
    good = dg.empty_intermediate(groups, m_max, inter, 'cuda')
    dg.m_grouped_fp8_gemm_nt_masked_swiglu(x, w1_t, good, masked, 40, workspace=ws)
    assert mega.exchange_timeouts(ws) == 0 and bool(torch.isfinite(good[1][0, :64]).all())
    mega.set_exchange_timeout_us(20000)
    os.environ['DG_TEST_SWIGLU_FAULT'] = '1'                            # read at dg_reload_env only: no entry point arms the fault
    ignored = dg.empty_intermediate(groups, m_max, inter, 'cuda')
    dg.m_grouped_fp8_gemm_nt_masked_swiglu(x, w1_t, ignored, masked, 40, workspace=ws)
    assert mega.exchange_timeouts(ws) == 0 and torch.equal(ignored[1][0, :64], good[1][0, :64])
    lib.dg_reload_env()
    try:
        bad = dg.empty_intermediate(groups, m_max, inter, 'cuda')
        dg.m_grouped_fp8_gemm_nt_masked_swiglu(x, w1_t, bad, masked, 40, workspace=ws)
        torch.cuda.synchronize()                                    # returns: every wait gave up after 20 ms
    finally:
        mega.set_exchange_timeout_us(10_000_000)
        del os.environ['DG_TEST_SWIGLU_FAULT']
        lib.dg_reload_env()
    assert bool(torch.isnan(bad[1][0, :64]).any()), 'rows whose partner never published must carry NaN scales'
    assert mega.exchange_timeouts(ws, reset=False) > 0
    assert mega.exchange_timeouts(ws) > 0 and int(ws.view(torch.int32).abs().sum()) == 0      # counted, then re-zeroed
    again = dg.empty_intermediate(groups, m_max, inter, 'cuda')
    dg.m_grouped_fp8_gemm_nt_masked_swiglu(x, w1_t, again, masked, 40, workspace=ws)
    for g, rows in enumerate(masked_ms):
        assert torch.equal(again[0][g, :rows].view(torch.uint8), good[0][g, :rows].view(torch.uint8)) and torch.equal(again[1][g, :rows], good[1][g, :rows])


def test_fused_swiglu_against_the_c_oracle():
    """Closes the loop on the fused kernel's oracle: C oracle GEMM (oracle/fp8_gemm_oracle.c) -> SwiGLU -> the reference cast on the CPU;
    the fused kernel's bytes, dequantised, sit within one FP8 step (2^-3 relative: e4m3 has 3 mantissa bits) + the BF16 step of the
    intermediate of that pipeline's values, and its scales within the BF16 step of the row amax."""
    import oracle
    gen.reset_seed(21)
    groups, m_max, inter, k = 2, 64, 256, 1024
    masked_ms = [64, 21]
    a = torch.randn((groups, m_max, k), device='cuda', dtype=torch.bfloat16)
    xq = [per_token_cast_to_fp8(a[g], use_ue8m0=False) for g in range(groups)]
    x = (torch.stack([q[0] for q in xq]), torch.stack([q[1] for q in xq]))
    w1 = _weights(groups, 2 * inter, k)
    w1_t, _ = dg.transform_weights_for_mega_moe(w1, w1)
    masked = torch.tensor(masked_ms, dtype=torch.int, device='cuda')
    q, q_sf = dg.empty_intermediate(groups, m_max, inter, 'cuda')
    dg.m_grouped_fp8_gemm_nt_masked_swiglu(x, w1_t, (q, q_sf), masked, 40)
    for g, rows in enumerate(masked_ms):
        h = torch.empty((rows, 2 * inter), dtype=torch.bfloat16)
        oracle.fp8_gemm_nt(x[0][g, :rows].cpu(), x[1][g, :rows].cpu(), w1[0][g].cpu(), w1[1][g].cpu(), h)
        y = (torch.nn.functional.silu(h[:, :inter].float()) * h[:, inter:].float()).to(torch.bfloat16)
        want_q, want_sf = per_token_cast_to_fp8(y, use_ue8m0=False)
        got = q[g, :rows].cpu().float() * q_sf[g, :rows].cpu().repeat_interleave(128, dim=1)
        want = want_q.float() * want_sf.repeat_interleave(128, dim=1)
        block_amax = y.float().abs().view(rows, inter // 128, 128).amax(dim=2).clamp(min=1e-4).repeat_interleave(128, dim=1)
        # one e4m3 step at the element's own magnitude (2^-3 relative), floored by the block's smallest step (amax / 448 * 2^-9... subnormals),
        # plus one BF16 step of the intermediate
        bound = want.abs() * 2.0 ** -3 + block_amax / 448.0 * 2.0 ** -6 + y.float().abs() * 2.0 ** -7
        assert bool(((got - want).abs() <= bound).all()), f'group {g}: dequantised bytes off by {((got - want).abs() - bound).max().item():.3e}'
        assert bool(((q_sf[g, :rows].cpu() - want_sf).abs() <= want_sf * 2.0 ** -7).all()), f'group {g}: scales'
        assert calc_diff(got, want) < 1e-4
