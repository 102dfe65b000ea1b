"""Mirror of the reference's `modules` package surface (modules/__init__.py) for the hot path."""
from .utils import log_sum_exp, generate_grid
from .encoders import GaussianEncoderBase, LSTMEncoder, ResNetEncoderV2
from .decoders import DecoderBase, LSTMDecoder, PixelCNNDecoderV2
from .vae import VAE

__all__ = ["VAE", "GaussianEncoderBase", "LSTMEncoder", "DecoderBase", "LSTMDecoder", "ResNetEncoderV2",
           "PixelCNNDecoderV2", "log_sum_exp", "generate_grid"]
