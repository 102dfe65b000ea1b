from .decoder import DecoderBase
from .dec_lstm import LSTMDecoder
from .dec_pixelcnn_v2 import PixelCNNDecoderV2

__all__ = ["DecoderBase", "LSTMDecoder", "PixelCNNDecoderV2"]
