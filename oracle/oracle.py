"""ctypes front-end of ``fp8_gemm_oracle.c`` (CPU restatement of the reference arithmetic).

TEST INFRASTRUCTURE ONLY -- never imported by ``deepgemm_amd``.  All tensors are CPU torch tensors;
FP8 operands may have any strides (K-major or MN-major views), exactly like the reference's operators
(``/root/reference/csrc/apis/gemm.hpp:73-297``).
"""
import ctypes
import os
import subprocess
import threading

import torch

_DIR = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_DIR, 'fp8_gemm_oracle.c')
_OUT_DIR = os.path.join(_DIR, '_build')
_SO = os.path.join(_OUT_DIR, 'libdg_oracle.so')
_lock = threading.Lock()
_lib = None

_i64, _i32, _vp = ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (OpenMP for the GPU box's host cores)."""
    os.makedirs(_OUT_DIR, exist_ok=True)
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        tmp = _SO + f'.{os.getpid()}.tmp'
        subprocess.check_call(['gcc', '-O2', '-fPIC', '-shared', '-fopenmp', '-std=c99', _SRC, '-o', tmp, '-lm'])
        os.replace(tmp, _SO)
    return _SO


def lib() -> ctypes.CDLL:
    global _lib
    with _lock:
        if _lib is None:
            handle = ctypes.CDLL(build())
            handle.dgo_e4m3_to_f32.restype = ctypes.c_float
            handle.dgo_e4m3_to_f32.argtypes = [ctypes.c_uint8]
            handle.dgo_f32_to_bf16.restype = ctypes.c_uint16
            handle.dgo_f32_to_bf16.argtypes = [ctypes.c_float]
            handle.dgo_fp8_gemm.restype = ctypes.c_int
            handle.dgo_fp8_gemm.argtypes = [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64,
                                            _i32, _vp, _i64, _i32, _i32, _i32, _i32, _i32]
            handle.dgo_fp8_gemm_m_grouped_contiguous.restype = ctypes.c_int
            handle.dgo_fp8_gemm_m_grouped_contiguous.argtypes = [
                _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64,
                _vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _i32]
            handle.dgo_fp8_gemm_m_grouped_masked.restype = ctypes.c_int
            handle.dgo_fp8_gemm_m_grouped_masked.argtypes = [
                _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64,
                _vp, _i64, _i64, _vp, _i32, _i32, _i32, _i32]
            handle.dgo_transpose_sf.restype = ctypes.c_int
            handle.dgo_transpose_sf.argtypes = [_vp, _vp, _i32, _i32, _i32]
            handle.dgo_pack_sf_ue8m0.restype = ctypes.c_int
            handle.dgo_pack_sf_ue8m0.argtypes = [_vp, _vp, _i32, _i32, _i32, _i64, _i64, _i64]
            _lib = handle
    return _lib


def e4m3_lut() -> torch.Tensor:
    """All 256 e4m3fn code points decoded by the C oracle (NaN for 0x7f/0xff)."""
    handle = lib()
    return torch.tensor([handle.dgo_e4m3_to_f32(i) for i in range(256)], dtype=torch.float32)


def f32_to_bf16_bits(x: float) -> int:
    return int(lib().dgo_f32_to_bf16(x))


def _cpu(t: torch.Tensor, dtype=None) -> torch.Tensor:
    assert t.device.type == 'cpu', 'the oracle is a CPU checker'
    if dtype is not None:
        assert t.dtype == dtype, f'{t.dtype=} != {dtype}'
    return t


def _sf(t: torch.Tensor) -> torch.Tensor:
    return _cpu(t, torch.float32)


def _d_dtype(d: torch.Tensor) -> int:
    assert d.dtype in (torch.bfloat16, torch.float32)
    return 0 if d.dtype == torch.bfloat16 else 1


class _gran_k:
    """Scale granularity along K of the calls inside the block (32: the SM100 MX recipe; default 128)."""

    def __init__(self, gran_k: int, grouped_gran_n: int = 128):
        self.gran_k, self.grouped_gran_n = gran_k, grouped_gran_n

    def __enter__(self):
        assert lib().dgo_set_gran_k(self.gran_k) == 0, f'gran_k {self.gran_k}'
        assert lib().dgo_set_grouped_gran_n(self.grouped_gran_n) == 0, f'gran_n {self.grouped_gran_n}'

    def __exit__(self, *exc):
        lib().dgo_set_gran_k(128)
        lib().dgo_set_grouped_gran_n(128)


def fp8_gemm_nt(a, sfa, b, sfb, d, c=None, gran_n: int = 128, gran_k: int = 128) -> torch.Tensor:
    """D = (C +) A @ B^T.  a:[M,K] e4m3, sfa:[M,ceil(K/gran_k)], b:[N,K], sfb:[ceil(N/gran_n),ceil(K/gran_k)]."""
    if gran_k != 128:
        with _gran_k(gran_k):
            return fp8_gemm_nt(a, sfa, b, sfb, d, c, gran_n)
    a, b, sfa, sfb = _cpu(a, torch.float8_e4m3fn), _cpu(b, torch.float8_e4m3fn), _sf(sfa), _sf(sfb)
    m, k = a.shape
    n, k_ = b.shape
    assert k == k_ and d.shape == (m, n) and d.stride(1) == 1
    if c is not None and c.data_ptr() != d.data_ptr():
        d.copy_(c)                                  # csrc/apis/gemm.hpp:43-44
    if m == 0 or n == 0:
        return d
    if k == 0:
        if c is None:
            d.zero_()
        return d
    au, bu = a.view(torch.uint8), b.view(torch.uint8)
    rc = lib().dgo_fp8_gemm(au.data_ptr(), au.stride(0), au.stride(1), sfa.data_ptr(), sfa.stride(0), sfa.stride(1),
                            bu.data_ptr(), bu.stride(0), bu.stride(1), sfb.data_ptr(), sfb.stride(0), sfb.stride(1),
                            gran_n, d.data_ptr(), d.stride(0), _d_dtype(d), int(c is not None), m, n, k)
    assert rc == 0, f'oracle error {rc}'
    return d


def m_grouped_fp8_gemm_nt_contiguous(a, sfa, b, sfb, d, grouped_layout, use_psum_layout=False,
                                     m_alignment: int = 128, gran_k: int = 128, gran_n: int = 128) -> torch.Tensor:
    """a:[M,K], b:[G,N,K], sfb:[G,ceil(N/gran_n),ceil(K/gran_k)], d:[M,N] bf16, grouped_layout int32 [M] or [G]."""
    if gran_k != 128 or gran_n != 128:
        with _gran_k(gran_k, gran_n):
            return m_grouped_fp8_gemm_nt_contiguous(a, sfa, b, sfb, d, grouped_layout, use_psum_layout, m_alignment)
    a, b, sfa, sfb = _cpu(a, torch.float8_e4m3fn), _cpu(b, torch.float8_e4m3fn), _sf(sfa), _sf(sfb)
    layout = _cpu(grouped_layout, torch.int32).contiguous()
    m, k = a.shape
    g, n, _ = b.shape
    assert d.dtype == torch.bfloat16 and d.shape == (m, n) and d.stride(1) == 1
    if m == 0:
        return d
    au, bu = a.view(torch.uint8), b.view(torch.uint8)
    rc = lib().dgo_fp8_gemm_m_grouped_contiguous(
        au.data_ptr(), au.stride(0), au.stride(1), sfa.data_ptr(), sfa.stride(0), sfa.stride(1),
        bu.data_ptr(), bu.stride(0), bu.stride(1), bu.stride(2), sfb.data_ptr(), sfb.stride(0), sfb.stride(1), sfb.stride(2),
        d.data_ptr(), d.stride(0), layout.data_ptr(), g, int(use_psum_layout), m_alignment, m, n, k)
    assert rc == 0, f'oracle error {rc}'
    return d


def m_grouped_fp8_gemm_nt_masked(a, sfa, b, sfb, d, masked_m, gran_k: int = 128, gran_n: int = 128) -> torch.Tensor:
    """a:[G,Mmax,K], sfa:[G,Mmax,ceil(K/gran_k)], b:[G,N,K], sfb:[G,ceil(N/gran_n),ceil(K/gran_k)], d:[G,Mmax,N] bf16; rows >= masked_m[g]
    left untouched."""
    if gran_k != 128 or gran_n != 128:
        with _gran_k(gran_k, gran_n):
            return m_grouped_fp8_gemm_nt_masked(a, sfa, b, sfb, d, masked_m)
    a, b, sfa, sfb = _cpu(a, torch.float8_e4m3fn), _cpu(b, torch.float8_e4m3fn), _sf(sfa), _sf(sfb)
    masked = _cpu(masked_m, torch.int32).contiguous()
    g, m_max, k = a.shape
    _, n, _ = b.shape
    assert d.dtype == torch.bfloat16 and d.shape == (g, m_max, n) and d.stride(2) == 1
    au, bu = a.view(torch.uint8), b.view(torch.uint8)
    rc = lib().dgo_fp8_gemm_m_grouped_masked(
        au.data_ptr(), au.stride(0), au.stride(1), au.stride(2), sfa.data_ptr(), sfa.stride(0), sfa.stride(1), sfa.stride(2),
        bu.data_ptr(), bu.stride(0), bu.stride(1), bu.stride(2), sfb.data_ptr(), sfb.stride(0), sfb.stride(1), sfb.stride(2),
        d.data_ptr(), d.stride(0), d.stride(1), masked.data_ptr(), g, m_max, n, k)
    assert rc == 0, f'oracle error {rc}'
    return d


def transpose_sf(sf: torch.Tensor) -> torch.Tensor:
    """[..., mn, sf_k] FP32 -> MN-major tensor with strides (aligned_mn*sf_k, 1, aligned_mn); padding zero-filled here."""
    sf = _sf(sf).contiguous()
    squeeze = sf.dim() == 2
    batched = sf.unsqueeze(0) if squeeze else sf
    nb, mn, sf_k = batched.shape
    aligned = (mn + 3) // 4 * 4
    storage = torch.zeros(nb * aligned * sf_k, dtype=torch.float32)
    lib().dgo_transpose_sf(batched.data_ptr(), storage.data_ptr(), nb, mn, sf_k)
    out = storage.as_strided((nb, mn, sf_k), (aligned * sf_k, 1, aligned))
    return out.squeeze(0) if squeeze else out


def pack_sf_ue8m0(sf: torch.Tensor) -> torch.Tensor:
    """[..., mn, sf_k] FP32 (any strides) -> packed UE8M0 words [..., mn, ceil(sf_k / 4)] int32, MN-major with strides
    (packed_k * aligned_mn, 1, aligned_mn); padding rows zero-filled here."""
    sf = _cpu(sf, torch.float32)
    squeeze = sf.dim() == 2
    batched = sf.unsqueeze(0) if squeeze else sf
    nb, mn, sf_k = batched.shape
    aligned, packed_k = (mn + 3) // 4 * 4, (sf_k + 3) // 4
    storage = torch.zeros(nb * aligned * packed_k, dtype=torch.int32)
    rc = lib().dgo_pack_sf_ue8m0(batched.data_ptr(), storage.data_ptr(), nb, mn, sf_k,
                                 batched.stride(0), batched.stride(1), batched.stride(2))
    assert rc == 0
    out = storage.as_strided((nb, mn, packed_k), (aligned * packed_k, 1, aligned))
    return out.squeeze(0) if squeeze else out


def fp8_gemm_nt_blockwise_torch(a, sfa, b, sfb, gran_n: int = 128, out_dtype=torch.bfloat16, c=None) -> torch.Tensor:
    """Same arithmetic as ``dgo_rows`` expressed with torch ops on the operands' own device (float64 block products of FP8 values are
    exact, so the block sum is the correctly rounded one whatever order the GEMM behind ``@`` uses), used where the C loop nest would
    take too long: full BASELINE.json sizes on the CPU for row subsets, and -- on the GPU box -- EVERY row of every BASELINE config
    (tests/test_full_output_parity_gpu.py; MI355X runs FP64 GEMMs at tens of TFLOPS).  ``c``: the accumulation operand (added in the
    output dtype's memory arithmetic: FP32 add for FP32 outputs, BF16 round-then-add for BF16, as the reduce-add epilogue does)."""
    m, k = a.shape
    n = b.shape[0]
    dev = a.device
    total = torch.zeros((m, n), dtype=torch.float32, device=dev)
    col_block = torch.arange(n, device=dev) // gran_n
    for kb in range((k + 127) // 128):
        ks = slice(kb * 128, min(k, kb * 128 + 128))
        block = (a[:, ks].to(torch.float64) @ b[:, ks].to(torch.float64).t()).to(torch.float32)
        scale = sfa[:, kb].unsqueeze(1) * sfb[col_block, kb].unsqueeze(0)
        total += scale * block
    if c is None:
        return total.to(out_dtype)
    return (total + c.float()) if out_dtype == torch.float else (total.to(out_dtype).float() + c.float()).to(out_dtype)


def dequant_matmul_f64(a, sfa, b, sfb, gran_n: int = 128) -> torch.Tensor:
    """Independent "BF16-simulated FP8 GEMM" check: dequantise both operands, one float64 matmul."""
    m, k = a.shape
    n = b.shape[0]
    ka = torch.arange(k) // 128
    a_deq = a.to(torch.float64) * sfa.to(torch.float64)[:, ka]
    b_deq = b.to(torch.float64) * sfb.to(torch.float64)[torch.arange(n) // gran_n][:, ka]
    return a_deq @ b_deq.t()
