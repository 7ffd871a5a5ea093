"""ctypes binding of the C ABI in ``include/deepgemm_amd.h`` (the analogue of the reference's pybind ``_C`` module,
csrc/python_api.cpp:17-28).  Importing this module requires the compiled HIP extension; there is NO fallback path:
a missing library is an ImportError, a failing call is a RuntimeError carrying ``dg_last_error()``."""
import ctypes
import os

import torch  # noqa: F401  (loads torch's libamdhip64.so.7 first so the extension shares the same HIP runtime)

from .build import LIB_PATH

_i32, _i64, _vp, _cp, _u32 = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint32

# name -> (restype, argtypes); must list every symbol declared in include/deepgemm_amd.h
SIGNATURES = {
    'dg_fp8_gemm_nt': (_i32, [_vp] * 5 + [_i32] * 3 + [_i64] * 8 + [_i32, _i64, _i32, _i32, _vp]),
    'dg_fp8_gemm_nt_ws': (_i32, [_vp] * 5 + [_i32] * 3 + [_i64] * 8 + [_i32, _i64, _i32, _i32, _vp, _i64, _vp]),
    'dg_fp8_gemm_nt_skip_head_mid': (_i32, [_vp] * 5 + [_i32] * 3 + [_i64] * 8 + [_i32, _i64, _i32, _i32, _i32, _i32, _vp]),
    'dg_fp8_gemm_nt_ue8m0': (_i32, [_vp] * 5 + [_i32] * 3 + [_i64] * 8 + [_i64, _i32, _i32, _vp]),
    'dg_ue8m0_dense_operand_plan': (_i32, [_vp, _vp, _i32, _i32, _i32, _i64, _i64, _i64, _i64]),
    'dg_ue8m0_grouped_operand_plan': (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _i32, _i32]),
    'dg_m_grouped_fp8_gemm_nt_contiguous_ue8m0': (_i32, [_vp] * 6 + [_i32] * 4 + [_i64] * 11 + [_i32, _i32, _vp]),
    'dg_m_grouped_fp8_gemm_nt_contiguous_ue8m0_ws': (_i32, [_vp] * 6 + [_i32] * 4 + [_i64] * 11 + [_i32, _i32, _vp, _i64, _vp]),
    'dg_m_grouped_fp8_gemm_nt_masked_ue8m0': (_i32, [_vp] * 6 + [_i32] * 5 + [_i64] * 14 + [_vp]),
    'dg_fp8_gemm_nt_ue8m0_ws': (_i32, [_vp] * 5 + [_i32] * 3 + [_i64] * 8 + [_i64, _i32, _i32, _i32, _vp, _i64, _vp]),
    'dg_ue8m0_dense_wants_workspace': (_i32, [_i32] * 3),
    'dg_fp8_gemm_nt_ue8m0_g32': (_i32, [_vp] * 5 + [_i32] * 3 + [_i64] * 8 + [_i64, _i32, _i32, _vp]),
    'dg_m_grouped_fp8_gemm_nt_contiguous_ue8m0_g32': (_i32, [_vp] * 6 + [_i32] * 4 + [_i64] * 11 + [_i32, _i32, _vp, _i64, _vp]),
    'dg_m_grouped_fp8_gemm_nt_masked_ue8m0_g32': (_i32, [_vp] * 6 + [_i32] * 5 + [_i64] * 14 + [_vp]),
    'dg_m_grouped_fp8_gemm_nt_masked_swiglu': (_i32, [_vp] * 7 + [_i32] * 5 + [_i64] * 13 + [ctypes.c_float, _i32, _vp, _i64, _vp]),
    'dg_swiglu_workspace_bytes': (_i64, [_i32, _i32, _i32]),
    'dg_set_swiglu_exchange_timeout_us': (None, [_i64]),
    'dg_m_grouped_fp8_gemm_nt_masked_swiglu_weighted': (_i32, [_vp] * 7 + [_i32] * 5 + [_i64] * 13 + [ctypes.c_float, _i32, _vp, _i64, _vp, _i64, _vp]),
    'dg_moe_scatter_to_masked': (_i32, [_vp, _vp, _vp, _i32, _vp] + [_i32] * 5 + [_i64, _i64] + [_vp] * 6 + [_i64] * 5 + [_vp]),
    'dg_moe_combine_from_masked': (_i32, [_vp, _vp, _i32, _i32, _i32, _i64, _vp, _i64, _vp]),
    'dg_k_grouped_fp8_gemm_nt_contiguous': (_i32, [_vp] * 5 + [_i32, _i32, _vp, _i32, _i32] + [_i64] * 6 + [_vp]),
    'dg_k_grouped_fp8_gemm_tn_psum': (_i32, [_vp] * 5 + [_i32, _i32, _i32, _vp, _i32, _i32] + [_i64] * 6 + [_vp]),
    'dg_k_grouped_fp8_gemm_ue8m0': (_i32, [_vp] * 5 + [_i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _i32] + [_i64] * 4 + [_vp]),
    'dg_pack_sf_k_grouped_ue8m0': (_i32, [_vp, _vp, _vp] + [_i32] * 7 + [_vp]),
    'dg_k_grouped_fp8_gemm_tn_psum_aligned': (_i32, [_vp] * 5 + [_i32, _i32, _i32, _vp, _i32, _i32] + [_i64] * 6 + [_i32, _vp]),
    'dg_m_grouped_fp8_gemm_nt_contiguous': (_i32, [_vp] * 6 + [_i32] * 4 + [_i64] * 11 + [_i32, _i32, _vp]),
    'dg_m_grouped_fp8_gemm_nt_contiguous_ws': (_i32, [_vp] * 6 + [_i32] * 4 + [_i64] * 11 + [_i32, _i32, _vp, _i64, _vp]),
    'dg_split_k_workspace_bytes': (_i64, []),
    'dg_m_grouped_fp8_gemm_nt_masked': (_i32, [_vp] * 6 + [_i32] * 5 + [_i64] * 14 + [_vp]),
    'dg_transpose_sf_fp32': (_i32, [_vp, _vp, _i32, _i32, _i32, _vp]),
    'dg_pack_sf_ue8m0': (_i32, [_vp, _vp, _i32, _i32, _i32, _i64, _i64, _i64, _vp]),
    'dg_pack_sf_ue8m0_ex': (_i32, [_vp, _vp, _i32, _i32, _i32, _i64, _i64, _i64, _i32, _vp, _i32, _i32, _vp]),
    'dg_pack_sf_pair_ue8m0': (_i32, [_vp, _vp, _i32, _i32, _i64, _i64, _i64, _i32, _vp, _i32, _i32,
                              _vp, _vp, _i32, _i32, _i64, _i64, _i64, _i32, _i32, _vp]),
    'dg_per_token_cast_to_fp8': (_i32, [_vp, _vp, _vp, _i32, _i32, _i64, _i64, _i64, _i64, _i32, _vp]),
    'dg_block_cast_to_fp8': (_i32, [_vp, _vp, _vp, _i32, _i32, _i64, _i64, _i64, _i64, _i32, _i32, _vp]),
    'dg_transpose_fp8': (_i32, [_vp, _vp, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _vp]),
    'dg_set_num_cus': (_i32, [_i32]),
    'dg_get_num_cus': (_i32, []),
    'dg_set_forced_config': (_i32, [_cp]),
    'dg_reload_env': (None, []),
    'dg_set_moe_p2p_timeout_us': (None, [_i64]),
    'dg_symm_alloc': (_i32, [_i64, _vp, _vp]),
    'dg_symm_free': (_i32, [_vp]),
    'dg_ipc_get_handle': (_i32, [_vp, _vp]),
    'dg_ipc_open_handle': (_i32, [_vp, _vp]),
    'dg_ipc_close_handle': (_i32, [_vp]),
    'dg_moe_p2p_layout': (_i32, [_i32] * 6 + [_vp]),
    'dg_moe_p2p_dispatch': (_i32, [_vp] + [_i32] * 7 + [_vp, _vp, _vp, _i32, _vp, _i32, _i64, _i64, _u32, _vp, _vp, _vp, _vp]),
    'dg_moe_p2p_combine': (_i32, [_vp] + [_i32] * 7 + [_vp, _i64, _i64, _vp, _u32, _vp, _vp]),
    'dg_moe_p2p_reduce': (_i32, [_vp] + [_i32] * 7 + [_vp, _i32, _vp, _i64, _vp, _u32, _vp, _vp]),
    'dg_get_forced_config': (_cp, []),
    'dg_set_debug_buffer': (_i32, [_vp]),
    'dg_list_configs': (_cp, []),
    'dg_last_config': (_cp, []),
    'dg_select_config': (_cp, [_i32] * 12),
    'dg_dense_wants_workspace': (_i32, [_i32] * 6),
    'dg_dense_rowmajor_sfa_native': (_i32, [_i32] * 3),
    'dg_operand_plan': (_i32, [_i32, _vp, _vp, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _i32]),
    'dg_last_error': (_cp, []),
    'dg_version': (_cp, []),
}


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f'deepgemm_amd: the HIP extension {LIB_PATH} has not been built. '
            'Run `python -m deepgemm_amd.build` (needs hipcc, cross-compiles gfx950 without a GPU).')
    handle = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(handle, name)     # AttributeError here means the header and the library disagree
        fn.restype, fn.argtypes = restype, argtypes
    return handle


lib = _load()


def check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError(lib.dg_last_error().decode() or f'deepgemm_amd call failed with code {rc}')


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def current_stream_ptr(device_index: int = -1) -> int:
    """hipStream_t of torch's current stream.  The raw getter costs ~0.3 us; torch.cuda.current_stream() builds a Stream object through
    several Python layers (~3 us -- a quarter of the whole host path of a decode-sized call)."""
    if _raw_stream is not None:
        return _raw_stream(device_index if device_index >= 0 else torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def require_device(*tensors) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('deepgemm_amd: operands must live on the GPU (there is no CPU path); '
                               f'got a tensor on {t.device}')
