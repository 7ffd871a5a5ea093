"""The fused expert-MLP hand-off (single-GPU half of the reference's Mega-MoE, deep_gemm/mega/__init__.py:155,
impls/sm100_fp8_fp4_mega_moe.cuh): GEMM1 with SwiGLU + per-token FP8 re-quantisation in its epilogue against the UNFUSED pipeline
``m_grouped_fp8_gemm_nt_masked -> BF16 -> torch SwiGLU -> reference per_token_cast_to_fp8`` -- bit-exact on the re-quantised bytes and
scales -- and the two-GEMM expert MLP end to end."""
import pytest
import torch

import deepgemm_amd as dg
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
            gate, up = gate.clamp(max=clamp), up.clamp(-clamp, clamp)
        y = (torch.nn.functional.silu(gate) * up).to(torch.bfloat16)
        out.append(per_token_cast_to_fp8(y, use_ue8m0=use_ue8m0) if rows else None)
    return out


@pytest.mark.parametrize('masked_ms,m_max,inter,k', [([5, 0, 64, 33], 64, 256, 512), ([200, 1, 129], 256, 384, 1024),
                                                      ([48, 64, 17, 0, 64, 33, 2, 60], 64, 2048, 7168)])
@pytest.mark.parametrize('use_ue8m0', [False, True])
@pytest.mark.parametrize('clamp', [None, 10.0])
def test_fused_swiglu_requant_is_the_unfused_pipeline(masked_ms, m_max, inter, k, use_ue8m0, clamp):
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
    assert dg.last_config() == 'stream_swiglu_64x128'
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
