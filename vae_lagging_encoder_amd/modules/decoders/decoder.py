"""Decoder interface of the reference (modules/decoders/decoder.py:5-70)."""
import torch.nn as nn


class DecoderBase(nn.Module):
    """Abstract decoder: subclasses implement reconstruct_error / log_probability (training path) and,
    optionally, the generation methods."""

    def __init__(self):
        super(DecoderBase, self).__init__()

    def decode(self, x, z):
        raise NotImplementedError

    def reconstruct_error(self, x, z):
        """x (batch, *), z (batch, n_sample, nz) -> loss (batch, n_sample)."""
        raise NotImplementedError

    def beam_search_decode(self, z, K):
        raise NotImplementedError

    def sample_decode(self, z):
        raise NotImplementedError

    def greedy_decode(self, z):
        raise NotImplementedError

    def log_probability(self, x, z):
        """log p(x|z): (batch, n_sample)."""
        raise NotImplementedError
