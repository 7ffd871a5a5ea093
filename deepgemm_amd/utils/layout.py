"""Layout helpers re-exported under the reference's ``deep_gemm.utils.layout`` names (deep_gemm/utils/layout.py:1-21)."""
from ..layout import (get_tma_aligned_size, get_mn_major_tma_aligned_tensor,           # noqa: F401
                      get_mn_major_tma_aligned_packed_ue8m0_tensor,
                      get_k_grouped_mn_major_tma_aligned_packed_ue8m0_tensor)
from ..runtime import (                                                            # noqa: F401
    set_mk_alignment_for_contiguous_layout,
    get_mk_alignment_for_contiguous_layout,
    get_theoretical_mk_alignment_for_contiguous_layout,
)

get_m_alignment_for_contiguous_layout = get_mk_alignment_for_contiguous_layout
get_k_alignment_for_contiguous_layout = get_mk_alignment_for_contiguous_layout
