"""Process-global knobs with the reference's names (csrc/apis/runtime.hpp:12-49, csrc/apis/layout.hpp:142-150).

``num_sms`` maps to the CU budget of a persistent launch (0 / unset = all 256 CUs).  ``tc_util``, ``pdl``,
``ignore_compile_dims`` and ``block_size_multiple_of`` only influence NVIDIA codegen/launch in the reference; they are
accepted and stored so that callers keep working, and never change results."""
from typing import Optional, Tuple, Union

from ._lib import lib, check

_LEGACY_MK_ALIGNMENT = 128      # reference csrc/jit_kernels/heuristics/runtime.hpp:10
_state = {'tc_util': 100, 'pdl': False, 'ignore_compile_dims': False, 'block_size_multiple_of': (1, 1),
          'mk_alignment': _LEGACY_MK_ALIGNMENT, 'sf_cast_mode': 'sm90'}


def set_sf_cast_mode(mode: str) -> None:
    """Which architecture's scaling-factor convention FP32 scale tensors follow in ``transform_sf_into_required_layout`` -- the role
    ``device_runtime->get_arch_major()`` plays in the reference (csrc/apis/layout.hpp:22,40-58).  Process-wide, like the knobs below.

    ``'sm90'`` (default): FP32 scales are consumed as FP32 -- (1, 128) MN-major, (128, 128) checked only; results are the reference's
    SM90 results for ANY positive scales (BASELINE's headline semantics).
    ``'sm100'``: unless a call passes ``disable_ue8m0_cast=True`` (the reference's keyword, default False), FP32 scales are cast to
    UE8M0 -- per-128-row scales broadcast to rows, exponent bytes packed four to a word (mantissa bits dropped, as the reference's
    ``>> 23``; its producers hand over powers of two) -- and the GEMM runs on the hardware-scaled MFMA kernels, the path the reference
    takes by default on SM100."""
    if mode not in ('sm90', 'sm100'):
        raise ValueError(f"sf cast mode must be 'sm90' or 'sm100', got {mode!r}")
    _state['sf_cast_mode'] = mode


def get_sf_cast_mode() -> str:
    return _state['sf_cast_mode']


def set_num_sms(new_num_sms: int) -> None:
    check(lib.dg_set_num_cus(int(new_num_sms)))


def get_num_sms() -> int:
    return int(lib.dg_get_num_cus())


def set_tc_util(new_tc_util: int) -> None:
    """Stored and otherwise ignored: in the reference it throttles tensor-core utilisation in NVIDIA codegen
    (csrc/apis/runtime.hpp:22-29); the ahead-of-time gfx950 kernels have no such knob.  Never changes results."""
    _state['tc_util'] = int(new_tc_util)


def get_tc_util() -> int:
    return _state['tc_util']


def set_pdl(new_enable_pdl: bool) -> None:
    """Stored and otherwise ignored: programmatic dependent launch is a CUDA launch attribute
    (csrc/jit/kernel_runtime.hpp:146); HIP has no equivalent.  Never changes results."""
    _state['pdl'] = bool(new_enable_pdl)


def get_pdl() -> bool:
    return _state['pdl']


def set_ignore_compile_dims(new_value: bool) -> None:
    """Stored and otherwise ignored: it steers which shape dimensions the reference bakes into JIT-compiled code
    (csrc/jit_kernels/impls/runtime_utils.hpp:22-31); the kernels here are compiled ahead of time with runtime shapes."""
    _state['ignore_compile_dims'] = bool(new_value)


def set_block_size_multiple_of(new_value: Union[int, Tuple[int, int]]) -> None:
    """Stored and otherwise ignored: a constraint on the reference's heuristic tile search (csrc/apis/runtime.hpp:36-41); the
    gfx950 tile shapes are a fixed set (dg_list_configs())."""
    _state['block_size_multiple_of'] = (new_value, new_value) if isinstance(new_value, int) else tuple(new_value)


def set_mk_alignment_for_contiguous_layout(new_value: int) -> None:
    _state['mk_alignment'] = int(new_value)


def get_mk_alignment_for_contiguous_layout() -> int:
    return _state['mk_alignment']


def get_theoretical_mk_alignment_for_contiguous_layout(expected_m: Optional[int] = None) -> int:
    """The reference returns 128 off SM100 (heuristics/runtime.hpp:47-49); the CDNA4 kernels tile M in 128-row
    blocks for the contiguous layout as well."""
    return _LEGACY_MK_ALIGNMENT


def set_forced_config(name: str) -> None:
    """Tuning hook: force a kernel configuration ('auto' restores the heuristic).  Process-wide, like the knobs above.  The cached call
    plans of the dense / masked operators are dropped: whether a call gets the K-split workspace depends on the configuration."""
    check(lib.dg_set_forced_config(name.encode()))
    from . import gemm
    gemm._VALIDATED_DENSE.clear()
    gemm._VALIDATED_MASKED.clear()
    gemm._VALIDATED_PACKED.clear()


def last_forced_config() -> str:
    return lib.dg_get_forced_config().decode()


def list_configs():
    return lib.dg_list_configs().decode().split(',')


def last_config() -> str:
    return lib.dg_last_config().decode()
