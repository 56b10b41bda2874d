import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("sdsl-lite_amd")


@pytest.fixture(scope="session")
def gpu(pkg):
    """The engine on cuda:0.  GPU tests fail (not skip) if the HIP library or the device is missing."""
    import torch
    assert torch.cuda.is_available(), "GPU test selected but no HIP device is visible"
    assert pkg.capi.lib().sdsl_hip_device_count() >= 1, "no gfx950 device visible to libsdsl_hip"
    return pkg


def unpack_bits(words: np.ndarray, n_bits: int) -> np.ndarray:
    return np.unpackbits(np.ascontiguousarray(words).view(np.uint8), bitorder="little")[:n_bits]
