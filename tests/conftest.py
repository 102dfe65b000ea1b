import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def emu_backend():
    """TEST-ONLY: the kernel sources compiled with g++ against tests/emu/hip_emu.h, bound to the same ctypes
    signatures as the HIP library, installed as the backend for CPU tensors."""
    import ctypes
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    from build_emu import build_emu
    import install as emu_install
    from vae_lagging_encoder_amd import _lib
    lib = _lib.bind(ctypes.CDLL(build_emu()), "tests/emu/liblvae_emu.so")
    emu_install.install(lib)
    yield lib
    emu_install.install(None)


@pytest.fixture(scope="session")
def hip_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
