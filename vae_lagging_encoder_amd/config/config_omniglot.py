"""Omniglot 28x28 (reference config/config_omniglot.py): 3-block ResNet encoder, 12-layer gated PixelCNN."""
from ._contract import resnet_pixelcnn_image

params = resnet_pixelcnn_image("omniglot", img_size=(1, 28, 28), nz=32, enc_width=64, enc_blocks=3,
                               dec_kernels=[(9, 3), (7, 3), (5, 3), (3, 3)], dec_width=32, latent_feature_map=4,
                               batch_size=50, epochs=1000, test_nepoch=5)
