"""ResNetEncoderV2 -- drop-in for the reference's modules/encoders/enc_resnet_v2.py:27-126 on MI355X.

The module TREE (attribute names, construction order, initialisers) mirrors the reference so that seeded
initialisation and state_dict keys (`main.0.main.{0,1,2}.{conv1,bn1,conv2,bn2,downsample.{0,1}}.*`, `main.1.weight`,
`main.2.*`, `linear.*`) are interchangeable; the containers only hold parameters -- `forward` runs the HIP path
(NHWC im2col + MFMA GEMM convolutions, fused BatchNorm(train)+residual+ELU kernels) through image_engine with a
hand-written backward behind torch.autograd.Function.
"""
import math

import torch
import torch.nn as nn

from ... import image_engine as _ie
from .encoder import GaussianEncoderBase


def _conv(c_in, c_out, k, stride=1, pad=0):
    return nn.Conv2d(c_in, c_out, kernel_size=k, stride=stride, padding=pad, bias=False)


def conv3x3(in_planes, out_planes, stride=1):
    """Public helper name of the reference (enc_resnet_v2.py:9-12)."""
    return _conv(in_planes, out_planes, 3, stride, 1)


def _reinit(root):
    """He-normal(fan_out) convolutions, BatchNorm gamma=1 beta=0, in registration order so a seeded
    construction consumes the RNG stream exactly as the reference's does."""
    for m in root.modules():
        if isinstance(m, nn.Conv2d):
            kh, kw = m.kernel_size
            nn.init.normal_(m.weight, mean=0.0, std=math.sqrt(2.0 / (kh * kw * m.out_channels)))
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)


class ResNetBlock(nn.Module):
    """conv3x3(stride) BN ELU conv3x3 BN, plus a 1x1(stride) conv + BN shortcut when the shape changes; ELU after
    the sum (reference enc_resnet_v2.py:27-70).  Parameter container: the arithmetic is image_engine.resnet_block,
    which reads conv1/bn1/conv2/bn2/downsample/stride."""

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.stride = stride
        self.conv1, self.bn1 = conv3x3(inplanes, planes, stride), nn.BatchNorm2d(planes)
        self.activation = nn.ELU()
        self.conv2, self.bn2 = conv3x3(planes, planes), nn.BatchNorm2d(planes)
        reshapes = (stride != 1) or (inplanes != planes)
        self.downsample = nn.Sequential(_conv(inplanes, planes, 1, stride), nn.BatchNorm2d(planes)) if reshapes else None
        self.reset_parameters()

    def reset_parameters(self):
        _reinit(self)


class ResNet(nn.Module):
    """Chain of ResNetBlocks; `planes[i]` output channels at `strides[i]`."""

    def __init__(self, inplanes, planes, strides):
        super().__init__()
        if len(planes) != len(strides):
            raise ValueError("planes and strides differ in length")
        widths = [inplanes] + list(planes)
        self.main = nn.Sequential(*[ResNetBlock(widths[i], widths[i + 1], stride=s) for i, s in enumerate(strides)])


class _ImageEncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, x, *params):
        mulv = eng.forward(x)
        ctx.eng = eng
        ctx.gen = eng.gen
        return mulv.clone()

    @staticmethod
    def backward(ctx, dmulv):
        eng = ctx.eng
        eng.backward(dmulv, ctx.gen)
        return (None, None) + tuple(eng.flat.gviews[n].clone() for n in eng.flat.names)


_WIDTHS, _STRIDES, _HIDDEN = (64, 64, 64), (2, 2, 2), 512


class ResNetEncoderV2(GaussianEncoderBase):
    """q(z|x) for 1x28x28 images (reference enc_resnet_v2.py:84-126): three stride-2 ResNet blocks 28 -> 14 -> 7 -> 4,
    a 4x4 valid convolution to 512 features, BN, ELU, Linear -> (mu, logvar)."""

    def __init__(self, args, ngpu=1):
        super().__init__()
        self.ngpu, self.nz, self.nc = ngpu, args.nz, 1
        trunk = ResNet(self.nc, list(_WIDTHS), list(_STRIDES))
        self.main = nn.Sequential(trunk, _conv(_WIDTHS[-1], _HIDDEN, 4), nn.BatchNorm2d(_HIDDEN), nn.ELU())
        self.linear = nn.Linear(_HIDDEN, 2 * self.nz)
        self.reset_parameters()
        self._hip = _ie.ImageEncoderEngine(self)

    def reset_parameters(self):
        _reinit(self.main)
        nn.init.xavier_uniform_(self.linear.weight)
        nn.init.zeros_(self.linear.bias)

    def _forward_mulv(self, input):
        self._hip.ensure(input.device)
        return _ImageEncoderFn.apply(self._hip, input, *self.parameters())

    def forward(self, input):
        """input (batch, 1, 28, 28) -> mean (batch, nz), logvar (batch, nz)."""
        mulv = self._forward_mulv(input)
        return mulv[:, :self.nz], mulv[:, self.nz:]
