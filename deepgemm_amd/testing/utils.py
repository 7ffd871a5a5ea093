"""Small test-harness helpers (contract: reference ``deep_gemm/testing/utils.py:6-38``)."""
import functools
import os
from typing import Callable

import torch


def get_arch_major() -> int:
    """CUDA-style major of the current device.  gfx950 reports 9 through torch-ROCm; the FP32-scale
    "1D2D" conventions of the reference's SM90 path are the ones this library implements."""
    major, _ = torch.cuda.get_device_capability()
    return major


def test_filter(condition: Callable):
    def decorator(func):
        @functools.wraps(func)
        def wrapper(*args, **kwargs):
            if condition():
                return func(*args, **kwargs)
            print(f'{func.__name__}:\n > Filtered by {condition}\n')
        return wrapper
    return decorator


def ignore_env(name: str, condition: Callable):
    def decorator(func):
        @functools.wraps(func)
        def wrapper(*args, **kwargs):
            saved = os.environ.pop(name, None) if condition() else None
            try:
                return func(*args, **kwargs)
            finally:
                if saved is not None:
                    os.environ[name] = saved
        return wrapper
    return decorator
