from .encoder import GaussianEncoderBase
from .enc_lstm import LSTMEncoder

__all__ = ["GaussianEncoderBase", "LSTMEncoder"]
