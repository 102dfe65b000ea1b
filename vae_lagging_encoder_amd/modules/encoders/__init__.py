from .encoder import GaussianEncoderBase
from .enc_lstm import LSTMEncoder
from .enc_resnet_v2 import ResNetEncoderV2

__all__ = ["GaussianEncoderBase", "LSTMEncoder", "ResNetEncoderV2"]
