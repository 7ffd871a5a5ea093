"""Parity metric and byte accounting (contract: reference ``deep_gemm/testing/numeric.py:5-21``)."""
from typing import Iterable

import torch


def calc_diff(x: torch.Tensor, y: torch.Tensor) -> float:
    """1 - 2<x,y> / (|x|^2 + |y|^2) in float64; 0 for two all-zero tensors."""
    xd, yd = x.double(), y.double()
    norm = (xd * xd + yd * yd).sum()
    if norm == 0:
        return 0.0
    return float(1 - 2 * (xd * yd).sum() / norm)


def rel_frobenius(x: torch.Tensor, ref: torch.Tensor) -> float:
    """|x - ref|_F / |ref|_F in float64 (the "rel-err" of BASELINE.json's north star)."""
    xd, rd = x.double(), ref.double()
    denom = rd.norm()
    return float((xd - rd).norm() / denom) if denom > 0 else float(xd.norm())


def count_bytes(*tensors) -> int:
    total = 0
    for t in tensors:
        if isinstance(t, (tuple, list)):
            total += count_bytes(*t)
        elif t is not None:
            total += t.numel() * t.element_size()
    return total


def assert_bitwise_equal(x: torch.Tensor, y: torch.Tensor, label: str = '') -> None:
    assert x.shape == y.shape, f'{label}: shape {tuple(x.shape)} vs {tuple(y.shape)}'
    assert x.dtype == y.dtype, f'{label}: dtype {x.dtype} vs {y.dtype}'
    xb, yb = x.contiguous().view(torch.uint8).flatten(), y.contiguous().view(torch.uint8).flatten()
    bad = (xb != yb).nonzero()
    if bad.numel() == 0:
        return
    first = int(bad[0])
    elem = first // x.element_size()
    raise AssertionError(f'bitwise mismatch ({label}): {bad.numel()} of {xb.numel()} bytes differ; first at element {elem}: '
                         f'{x.flatten()[elem].item()} vs {y.flatten()[elem].item()}')
