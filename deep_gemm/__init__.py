"""Alias package: ``import deep_gemm`` resolves to ``deepgemm_amd`` so that callers of the reference's FP8 GEMM
operators switch over without touching their imports (``deep_gemm.fp8_gemm_nt``, ``deep_gemm.utils``, ``deep_gemm.testing``)."""
import sys

import deepgemm_amd as _impl

for _name in ('utils', 'utils.math', 'utils.layout', 'utils.dist', 'testing', 'testing.bench', 'testing.numeric', 'testing.utils', 'mega'):
    sys.modules[f'{__name__}.{_name}'] = sys.modules[f'deepgemm_amd.{_name}']
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
__version__ = _impl.__version__
