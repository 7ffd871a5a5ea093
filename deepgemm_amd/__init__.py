"""deepgemm_amd -- MI355X (gfx950) native FP8 blockwise-scaled GEMM with DeepGEMM's operator surface.

Drop-in for the reference's FP8 GEMM path (``deep_gemm/__init__.py:17-78``): same function names, argument order,
keyword names and defaults.  ``import deep_gemm`` (the alias package at the repository root) resolves to this package.
"""
from . import _lib                                    # loads the HIP extension; ImportError if it is not built
from .runtime import (                                # noqa: F401
    set_num_sms, get_num_sms, set_tc_util, get_tc_util, set_pdl, get_pdl,
    set_ignore_compile_dims, set_block_size_multiple_of,
    set_forced_config, list_configs, last_config, set_sf_cast_mode, get_sf_cast_mode,
)
from .gemm import (                                   # noqa: F401
    fp8_gemm_nt, fp8_gemm_nn, fp8_gemm_tn, fp8_gemm_tt,
    m_grouped_fp8_gemm_nt_contiguous, m_grouped_fp8_gemm_nn_contiguous, m_grouped_fp8_gemm_nt_masked,
    k_grouped_fp8_gemm_nt_contiguous, k_grouped_fp8_gemm_tn_contiguous, fp8_gemm_nt_skip_head_mid,
)
from .layout import transform_sf_into_required_layout                 # noqa: F401
from .quant import (fused_per_token_cast_to_fp8, fused_per_block_cast_to_fp8,      # noqa: F401
                    fused_per_channel_cast_to_fp8)
from .mega import (transform_weights_for_mega_moe, m_grouped_fp8_gemm_nt_masked_swiglu,       # noqa: F401
                   fp8_mega_moe_local, empty_intermediate, SymmBuffer, get_symm_buffer_for_mega_moe, fp8_mega_moe, fp8_fp4_mega_moe,
                   bf16_mega_moe, get_token_alignment_for_mega_moe)
from . import testing, utils                                          # noqa: F401
from .utils import *                                                  # noqa: F401,F403

# Names the reference exports for the same entry points (csrc/apis/gemm.hpp:711-717, deep_gemm/__init__.py:77)
fp8_fp4_gemm_nt, fp8_fp4_gemm_nn, fp8_fp4_gemm_tn, fp8_fp4_gemm_tt = fp8_gemm_nt, fp8_gemm_nn, fp8_gemm_tn, fp8_gemm_tt
m_grouped_fp8_fp4_gemm_nt_contiguous = m_grouped_fp8_gemm_nt_contiguous
m_grouped_fp8_fp4_gemm_nn_contiguous = m_grouped_fp8_gemm_nn_contiguous
m_grouped_fp8_fp4_gemm_nt_masked = m_grouped_fp8_gemm_nt_masked
fp8_m_grouped_gemm_nt_masked = m_grouped_fp8_gemm_nt_masked

__version__ = '0.1.0'
