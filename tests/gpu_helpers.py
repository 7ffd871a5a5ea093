"""Shared pieces of the GPU parity tests: run the HIP path through the public operators, run the CPU oracle on copies
of the same tensors, and compare under the tolerances stated here.

Tolerances (floating-point path; north star allows 1e-2 rel-err, these are far tighter):
  * vs the oracle ("BF16-simulated FP8 GEMM", same FP32 block-promotion arithmetic):
      BF16 out: calc_diff <= 2e-6, rel-Frobenius <= 1e-3 (4e-3 below 256 output elements), and every element within
                |x - y| <= 2^-7 |y| (one BF16 ulp) + 2e-4 rms(y) (matrix-core accumulation noise),
      FP32 out: rel-Frobenius <= 5e-5;
    the sources of difference are the MFMA's internal accumulation of the 128 products of one K block (measured on
    MI355X: about 3e-5 rms(y) at K = 7168, i.e. the matrix core does not round the block sum correctly to FP32 the way
    the oracle's float64 block sum does) and FMA contraction of the promotion.
  * vs the reference's own test expression on the unquantised inputs: calc_diff < 1e-3 (tests/generators.py:65-70).
"""
import torch

import oracle
from deepgemm_amd.testing import calc_diff, rel_frobenius


def cpu_pair(pair):
    return pair[0].cpu(), pair[1].cpu()


def strided_cpu(t: torch.Tensor) -> torch.Tensor:
    """CPU copy that keeps the logical values (strides may differ; the oracle takes any strides)."""
    return t.cpu()


def assert_close_to_oracle(got: torch.Tensor, want: torch.Tensor, label: str = '', addend: torch.Tensor = None):
    """``addend`` (the C operand of an accumulating call): the reduce-add rounds the GEMM result to BF16 before adding,
    so the one-ulp term is taken at the magnitude of the larger of |result|, |C| and |result - C|."""
    got, want = got.float().cpu(), want.float().cpu()
    assert torch.isfinite(got).all(), f'{label}: non-finite output'
    if want.numel() == 0:
        return
    diff = calc_diff(got, want)
    rel = rel_frobenius(got, want)
    assert diff <= 2e-6, f'{label}: calc_diff vs oracle {diff:.3e}'
    # (outputs of a few dozen elements: ONE element that rounds to the neighbouring BF16 value -- legitimate, see the element-wise bound
    # below -- is already 2^-8 / sqrt(numel) of the norm: 1e-3 at 16 elements.  The Frobenius gate is a whole-matrix check; tiny outputs
    # are held to the element-wise bound and to a gate that admits a couple of such flips.)
    rel_bound = 1e-3 if want.numel() >= 256 else 4e-3
    assert rel <= rel_bound, f'{label}: rel-Frobenius vs oracle {rel:.3e}'
    mag = want.abs()
    if addend is not None:
        addend = addend.float().cpu()
        mag = torch.maximum(torch.maximum(mag, addend.abs()), (want - addend).abs()) * 2
    bound = mag * 2.0 ** -7 + 2e-4 * want.pow(2).mean().sqrt() + 1e-30
    worst = ((got - want).abs() - bound).max().item()
    assert worst <= 0, f'{label}: element error exceeds 1 BF16 ulp + 2e-4 rms by {worst:.3e}'


def assert_close_fp32(got: torch.Tensor, want: torch.Tensor, label: str = ''):
    rel = rel_frobenius(got.cpu(), want.cpu())
    assert rel <= 5e-5, f'{label}: rel-Frobenius vs oracle {rel:.3e}'


def oracle_dense(case, gran_n=128, c_cpu=None):
    a, sfa = cpu_pair(case.a)
    b, sfb = cpu_pair(case.b)
    d = torch.empty(case.d.shape, dtype=case.d.dtype)
    return oracle.fp8_gemm_nt(a, sfa, b, sfb, d, c=c_cpu, gran_n=gran_n)
