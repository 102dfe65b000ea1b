"""PixelCNNDecoderV2 -- drop-in for the reference's modules/decoders/dec_pixelcnn_v2.py:12-195 on MI355X (training path).

The module TREE mirrors the reference (MaskedConv2d with its `mask` buffer, PixelCNNBlock, MaskABlock, PixelCNN with
`main` + `direct_connects` ModuleLists, `z_transform`, tail `main.{1,2,4}`) so seeded init and state_dict keys are
interchangeable.  `reconstruct_error` runs the HIP path through image_engine: z_transform GEMM, NHWC input assembly,
masked convolutions as tap-skipping im2col + MFMA GEMM (type-B k x k uses (k*k+1)/2 taps), 1x1 convolutions as plain
GEMMs, fused BatchNorm(train)+residual+ELU, fused sigmoid+BCE; hand-written backward.  MaskedConv2d's in-place
`weight.data.mul_(mask)` on every forward (reference line 29) is reproduced by lv_mul_inplace_f32.
Ancestral sampling `decode` (784 sequential forwards) is generation: out of the hot path's scope.
"""
import math

import torch
import torch.nn as nn

from ... import image_engine as _ie
from .decoder import DecoderBase


class MaskedConv2d(nn.Conv2d):
    """Conv2d whose weight is multiplied by a causal mask: type 'A' hides the centre tap and everything after it in
    raster order, type 'B' keeps the centre; only the first `masked_channels` input channels are masked."""

    def __init__(self, mask_type, masked_channels, *args, **kwargs):
        super(MaskedConv2d, self).__init__(*args, **kwargs)
        assert mask_type in {'A', 'B'}
        self.register_buffer('mask', self.weight.data.clone())
        _, _, kH, kW = self.weight.size()
        self.mask.fill_(1)
        self.mask[:, :masked_channels, kH // 2, kW // 2 + (mask_type == 'B'):] = 0
        self.mask[:, :masked_channels, kH // 2 + 1:] = 0

    def reset_parameters(self):
        n = self.kernel_size[0] * self.kernel_size[1] * self.out_channels
        self.weight.data.normal_(0, math.sqrt(2. / n))
        if self.bias is not None:
            self.bias.data.zero_()


def _he_normal_convs_unit_bn(root):
    for m in root.modules():
        if isinstance(m, nn.Conv2d):
            n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
            m.weight.data.normal_(0, math.sqrt(2. / n))
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.data.fill_(1)
            m.bias.data.zero_()


class PixelCNNBlock(nn.Module):
    """1x1 (C -> C/2) BN ELU, masked-B kxk (C/2 -> C/2) BN ELU, 1x1 (C/2 -> C) BN, residual, ELU.
    Parameter container: the arithmetic runs in image_engine.pixelcnn_block."""

    def __init__(self, in_channels, kernel_size):
        super(PixelCNNBlock, self).__init__()
        self.mask_type = 'B'
        padding = kernel_size // 2
        out_channels = in_channels // 2
        self.main = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, 1, bias=False),
            nn.BatchNorm2d(out_channels),
            nn.ELU(),
            MaskedConv2d(self.mask_type, out_channels, out_channels, out_channels, kernel_size, padding=padding, bias=False),
            nn.BatchNorm2d(out_channels),
            nn.ELU(),
            nn.Conv2d(out_channels, in_channels, 1, bias=False),
            nn.BatchNorm2d(in_channels),
        )
        self.activation = nn.ELU()
        self.reset_parameters()

    def reset_parameters(self):
        _he_normal_convs_unit_bn(self)


class MaskABlock(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, masked_channels):
        super(MaskABlock, self).__init__()
        self.mask_type = 'A'
        padding = kernel_size // 2
        self.main = nn.Sequential(
            MaskedConv2d(self.mask_type, masked_channels, in_channels, out_channels, kernel_size, padding=padding, bias=False),
            nn.BatchNorm2d(out_channels),
            nn.ELU(),
        )
        self.reset_parameters()

    def reset_parameters(self):
        m = self.main[1]
        assert isinstance(m, nn.BatchNorm2d)
        m.weight.data.fill_(1)
        m.bias.data.zero_()


class PixelCNN(nn.Module):
    """A MaskA block followed by PixelCNN blocks, with a 'direct connection' block feeding the output of block i-3
    into the input of block i (and one more after the last block)."""

    def __init__(self, in_channels, out_channels, num_blocks, kernel_sizes, masked_channels):
        super(PixelCNN, self).__init__()
        assert num_blocks == len(kernel_sizes)
        self.blocks = []
        for i in range(num_blocks):
            if i == 0:
                block = MaskABlock(in_channels, out_channels, kernel_sizes[i], masked_channels)
            else:
                block = PixelCNNBlock(out_channels, kernel_sizes[i])
            self.blocks.append(block)
        self.main = nn.ModuleList(self.blocks)
        self.direct_connects = []
        for i in range(1, num_blocks - 1):
            self.direct_connects.append(PixelCNNBlock(out_channels, kernel_sizes[i]))
        self.direct_connects = nn.ModuleList(self.direct_connects)


class _ImageDecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, x, z2d, *params):
        rec = eng.forward(x, z2d)
        ctx.eng = eng
        ctx.gen = eng.gen
        return rec.clone()

    @staticmethod
    def backward(ctx, drec):
        eng = ctx.eng
        dz = eng.backward(drec, ctx.gen)
        return (None, None, dz.clone()) + tuple(eng.flat.gviews[n].clone() for n in eng.flat.names)


class PixelCNNDecoderV2(DecoderBase):
    def __init__(self, args, ngpu=1, mode='large'):
        super(PixelCNNDecoderV2, self).__init__()
        self.ngpu = ngpu
        self.nz = args.nz
        self.nc = 1
        self.fm_latent = args.latent_feature_map
        self.img_latent = 28 * 28 * self.fm_latent
        if self.nz != 0:
            self.z_transform = nn.Sequential(
                nn.Linear(self.nz, self.img_latent),
            )
        if mode == 'small':
            kernal_sizes = [7, 7, 7, 5, 5, 3, 3]
        elif mode == 'large':
            kernal_sizes = [7, 7, 7, 7, 7, 5, 5, 5, 5, 3, 3, 3, 3]
        else:
            raise ValueError('unknown mode: %s' % mode)
        hidden_channels = 64
        self.main = nn.Sequential(
            PixelCNN(self.nc + self.fm_latent, hidden_channels, len(kernal_sizes), kernal_sizes, self.nc),
            nn.Conv2d(hidden_channels, hidden_channels, 1, bias=False),
            nn.BatchNorm2d(hidden_channels),
            nn.ELU(),
            nn.Conv2d(hidden_channels, self.nc, 1, bias=False),
            nn.Sigmoid(),
        )
        self.reset_parameters()
        self._hip = _ie.ImageDecoderEngine(self)

    def reset_parameters(self):
        if self.nz != 0:
            nn.init.xavier_uniform_(self.z_transform[0].weight)
            nn.init.constant_(self.z_transform[0].bias, 0)
        m = self.main[2]
        assert isinstance(m, nn.BatchNorm2d)
        m.weight.data.fill_(1)
        m.bias.data.zero_()

    def reconstruct_error(self, x, z, masks=None):
        """Binary cross entropy summed over pixels.  x (batch, 1, 28, 28) in {0,1}, z (batch, n_sample, nz)
        -> (batch, n_sample)."""
        if z is None:
            raise NotImplementedError("the nz == 0 variant (z is None) is not on the hot path")
        B, ns, nz = z.size()
        self._hip.ensure(x.device)
        if ns > 1:
            x = x.repeat_interleave(ns, dim=0)
        z2d = z.reshape(B * ns, nz)
        rec = _ImageDecoderFn.apply(self._hip, x, z2d, *self.parameters())
        return rec.view(B, ns)

    def log_probability(self, x, z):
        return -self.reconstruct_error(x, z)

    def decode(self, z, deterministic=False):
        raise NotImplementedError("ancestral sampling is outside the MI355X hot path (SURVEY.md section 8: out of scope)")
