"""CPU oracle for the FP8 blockwise-scaled GEMM path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; the product (``deepgemm_amd``) never does.  See ``oracle/README.md`` for how it is pinned.
"""
from .oracle import (  # noqa: F401
    build, lib,
    e4m3_lut, f32_to_bf16_bits,
    fp8_gemm_nt, m_grouped_fp8_gemm_nt_contiguous, m_grouped_fp8_gemm_nt_masked,
    transpose_sf, pack_sf_ue8m0, fp8_gemm_nt_blockwise_torch, dequant_matmul_f64,
)
