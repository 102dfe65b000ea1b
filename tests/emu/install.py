"""TEST-ONLY: route CPU tensors of the package's host code to an emulator build of the kernel sources.

The package refuses anything but ROCm 'cuda' tensors (engine.backend_for).  The GPU-less CI replaces that one function from
out here, so that the same engines / trainers can be driven against tests/emu/liblvae_emu.so; nothing in the package knows
about it."""
import torch

from vae_lagging_encoder_amd import engine

_product_backend_for = engine.backend_for
_installed = None


def install(lib):
    """lib: a _lib.Lib bound to the emulator build, or None to restore the product behaviour.  Returns the previous one."""
    global _installed
    prev = _installed
    _installed = lib
    if lib is None:
        engine.backend_for = _product_backend_for
        return prev

    def backend_for(device):
        if torch.device(device).type == "cpu":
            return lib
        return _product_backend_for(device)
    engine.backend_for = backend_for
    return prev


def installed():
    return _installed
