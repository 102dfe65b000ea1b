from .decoder import DecoderBase
from .dec_lstm import LSTMDecoder

__all__ = ["DecoderBase", "LSTMDecoder"]
