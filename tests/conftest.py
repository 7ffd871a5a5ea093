import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _ensure_built():
    # The package has no fallback path: build the HIP extension (hipcc cross-compiles gfx950 without a GPU) before any
    # test module imports it.  On the GPU box the prebuilt library travels with the snapshot and this is a no-op.
    import importlib.util
    spec = importlib.util.spec_from_file_location('_dg_build', os.path.join(ROOT, 'deepgemm_amd', 'build.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build_extension()


_ensure_built()

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with `-m gpu` on the GPU box)')


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def _from_bits(arr: np.ndarray, dtype: torch.dtype) -> torch.Tensor:
    t = torch.from_numpy(np.ascontiguousarray(arr))
    return t.view(dtype) if dtype in (torch.bfloat16, torch.float8_e4m3fn) else t


class Golden:
    def __init__(self, name: str):
        self.data = np.load(os.path.join(GOLDEN_DIR, name))

    def bf16(self, key): return _from_bits(self.data[key], torch.bfloat16)
    def fp8(self, key): return _from_bits(self.data[key], torch.float8_e4m3fn)
    def raw(self, key): return torch.from_numpy(np.ascontiguousarray(self.data[key]))
    def scalar(self, key): return float(self.data[key])


@pytest.fixture(scope='session')
def golden_quantisers():
    return Golden('quantisers.npz')


@pytest.fixture(scope='session')
def golden_gran32():
    return Golden('gran32.npz')


@pytest.fixture(scope='session')
def golden_gemm():
    return Golden('gemm_cases.npz')


@pytest.fixture(scope='session')
def golden_sf_layout():
    return Golden('sf_layout.npz')
