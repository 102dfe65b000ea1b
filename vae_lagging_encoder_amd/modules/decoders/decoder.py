"""Decoder interface of the reference (modules/decoders/decoder.py:5-70): the method names the VAE wrapper and the
trainers call.  Training-path methods (`reconstruct_error`, `log_probability`) are implemented by the HIP-backed
subclasses; the generation methods are declared so a missing one fails with the reference's NotImplementedError."""
import torch.nn as nn


def _unimplemented(name, doc):
    def method(self, *args, **kwargs):
        raise NotImplementedError("%s.%s" % (type(self).__name__, name))
    method.__name__ = name
    method.__doc__ = doc
    return method


class DecoderBase(nn.Module):
    """Abstract p(x|z)."""

    reconstruct_error = _unimplemented("reconstruct_error", "x (batch, *), z (batch, n_sample, nz) -> loss (batch, n_sample)")
    log_probability = _unimplemented("log_probability", "log p(x|z) -> (batch, n_sample)")
    decode = _unimplemented("decode", "teacher-forced / ancestral decode of (x, z)")
    beam_search_decode = _unimplemented("beam_search_decode", "(z, K) -> decoded sentences")
    greedy_decode = _unimplemented("greedy_decode", "z -> decoded sentences")
    sample_decode = _unimplemented("sample_decode", "z -> decoded sentences")
