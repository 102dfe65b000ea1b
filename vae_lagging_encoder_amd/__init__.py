"""vae_lagging_encoder_amd -- MI355X-native (gfx950) aggressive inference-network VAE training hot path.

Drop-in for the class surface of jxhe/vae-lagging-encoder's `modules` package on the path
text.py:366-424 / image.py:295-348: `from vae_lagging_encoder_amd.modules import VAE, LSTMEncoder, LSTMDecoder`.
All arithmetic runs in hand-written HIP kernels behind the C ABI declared in include/lvae.h
(csrc/liblvae_hip.so); there is no CPU fallback.
"""
from . import _lib  # noqa: F401
from .engine import backend_for  # noqa: F401

__all__ = ["modules", "config", "trainer", "dist"]
