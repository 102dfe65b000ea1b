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


def conv3x3(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def _he_normal_convs_unit_bn(root):
    """He-normal(fan_out) convolutions, BatchNorm gamma=1 beta=0."""
    for m in root.modules():
        if isinstance(m, nn.Conv2d):
            n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
            m.weight.data.normal_(0, math.sqrt(2. / n))
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.data.fill_(1)
            m.bias.data.zero_()


class ResNetBlock(nn.Module):
    """conv3x3(stride) - BN - ELU - conv3x3 - BN, plus a 1x1(stride) conv + BN shortcut when the shape changes;
    ELU after the sum.  Parameter container: the arithmetic runs in image_engine.resnet_block."""

    def __init__(self, inplanes, planes, stride=1):
        super(ResNetBlock, self).__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.activation = nn.ELU()
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        downsample = None
        if stride != 1 or inplanes != planes:
            downsample = nn.Sequential(
                nn.Conv2d(inplanes, planes, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes),
            )
        self.downsample = downsample
        self.stride = stride
        self.reset_parameters()

    def reset_parameters(self):
        _he_normal_convs_unit_bn(self)


class ResNet(nn.Module):
    def __init__(self, inplanes, planes, strides):
        super(ResNet, self).__init__()
        assert len(planes) == len(strides)
        blocks = []
        for plane, stride in zip(planes, strides):
            blocks.append(ResNetBlock(inplanes, plane, stride=stride))
            inplanes = plane
        self.main = nn.Sequential(*blocks)


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


class ResNetEncoderV2(GaussianEncoderBase):
    """q(z|x) for 1x28x28 images: 3 stride-2 ResNet blocks (1->64->64->64), 4x4 conv -> 512, BN, ELU, Linear -> 2nz."""

    def __init__(self, args, ngpu=1):
        super(ResNetEncoderV2, self).__init__()
        self.ngpu = ngpu
        self.nz = args.nz
        self.nc = 1
        hidden_units = 512
        self.main = nn.Sequential(
            ResNet(self.nc, [64, 64, 64], [2, 2, 2]),
            nn.Conv2d(64, hidden_units, 4, 1, 0, bias=False),
            nn.BatchNorm2d(hidden_units),
            nn.ELU(),
        )
        self.linear = nn.Linear(hidden_units, 2 * self.nz)
        self.reset_parameters()
        self._hip = _ie.ImageEncoderEngine(self)

    def reset_parameters(self):
        _he_normal_convs_unit_bn(self.main)
        nn.init.xavier_uniform_(self.linear.weight)
        nn.init.constant_(self.linear.bias, 0.0)

    def _forward_mulv(self, input):
        self._hip.ensure(input.device)
        return _ImageEncoderFn.apply(self._hip, input, *self.parameters())

    def forward(self, input):
        """input (batch, 1, 28, 28) -> mean (batch, nz), logvar (batch, nz)."""
        mulv = self._forward_mulv(input)
        return mulv[:, :self.nz], mulv[:, self.nz:]
