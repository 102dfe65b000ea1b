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


def _fanout_normal_(conv):
    """weight ~ N(0, 2 / (kH * kW * C_out)): the fan-out He initialisation both conv flavours use."""
    kh, kw = conv.kernel_size
    nn.init.normal_(conv.weight, mean=0.0, std=math.sqrt(2.0 / (kh * kw * conv.out_channels)))


def _unit_affine_(bn):
    nn.init.ones_(bn.weight)
    nn.init.zeros_(bn.bias)


def _reinit(root):
    """Walk `root` in registration order (the order fixes the RNG draw sequence of a seeded construction)."""
    for m in root.modules():
        if isinstance(m, nn.Conv2d):
            _fanout_normal_(m)
        elif isinstance(m, nn.BatchNorm2d):
            _unit_affine_(m)


def raster_causal_mask(weight_shape, masked_channels, centre_visible):
    """1 for the taps that precede the centre in raster order (plus the centre itself when `centre_visible`),
    0 after; applies to the first `masked_channels` input channels, all others stay fully visible."""
    _, _, kh, kw = weight_shape
    tap = torch.arange(kh * kw).view(kh, kw)
    last_visible = (kh // 2) * kw + kw // 2 - (0 if centre_visible else 1)
    mask = torch.ones(weight_shape)
    mask[:, :masked_channels] = (tap <= last_visible).to(mask.dtype)
    return mask


class MaskedConv2d(nn.Conv2d):
    """Conv2d whose weight is multiplied by a causal mask before use: type 'A' hides the centre tap and everything
    after it in raster order, type 'B' keeps the centre (reference dec_pixelcnn_v2.py:12-38).  Signature and the
    `mask` buffer name follow the reference so checkpoints interchange."""

    def __init__(self, mask_type, masked_channels, *conv_args, **conv_kwargs):
        if mask_type not in ('A', 'B'):
            raise ValueError("mask_type must be 'A' or 'B', got %r" % (mask_type,))
        super().__init__(*conv_args, **conv_kwargs)
        mask = raster_causal_mask(self.weight.shape, masked_channels, centre_visible=(mask_type == 'B'))
        self.register_buffer('mask', mask.to(self.weight.dtype))

    def reset_parameters(self):       # called by nn.Conv2d.__init__
        _fanout_normal_(self)
        if self.bias is not None:
            nn.init.zeros_(self.bias)


def _pointwise(c_in, c_out):
    return nn.Conv2d(c_in, c_out, kernel_size=1, bias=False)


def _causal(kind, masked, c_in, c_out, k):
    return MaskedConv2d(kind, masked, c_in, c_out, k, padding=k // 2, bias=False)


def _conv_bn_chain(*stages):
    """stages: (conv, followed_by_elu).  Yields conv, BatchNorm2d[, ELU] per stage -- the Sequential index layout
    (0,1,2 | 3,4,5 | 6,7) the reference's checkpoints use."""
    layers = []
    for conv, elu in stages:
        layers.append(conv)
        layers.append(nn.BatchNorm2d(conv.out_channels))
        if elu:
            layers.append(nn.ELU())
    return nn.Sequential(*layers)


class PixelCNNBlock(nn.Module):
    """Bottleneck residual block: 1x1 (C -> C/2) BN ELU, masked-B kxk (C/2 -> C/2) BN ELU, 1x1 (C/2 -> C) BN, + input,
    ELU (reference dec_pixelcnn_v2.py:40-74).  Parameter container: the arithmetic is image_engine.pixelcnn_block."""
    mask_type = 'B'

    def __init__(self, in_channels, kernel_size):
        super().__init__()
        half = in_channels // 2
        squeeze = _pointwise(in_channels, half)
        causal = _causal(self.mask_type, half, half, half, kernel_size)
        expand = _pointwise(half, in_channels)
        self.main = _conv_bn_chain((squeeze, True), (causal, True), (expand, False))
        self.activation = nn.ELU()
        self.reset_parameters()

    def reset_parameters(self):
        _reinit(self)


class MaskABlock(nn.Module):
    """First layer: masked-A kxk conv that never sees the pixel it predicts, BN, ELU (reference :77-98)."""
    mask_type = 'A'

    def __init__(self, in_channels, out_channels, kernel_size, masked_channels):
        super().__init__()
        self.main = _conv_bn_chain((_causal(self.mask_type, masked_channels, in_channels, out_channels, kernel_size), True))
        self.reset_parameters()

    def reset_parameters(self):
        _unit_affine_(self.main[1])


class PixelCNN(nn.Module):
    """MaskA block then PixelCNN blocks; from block 3 on, block i also receives the output of block i-3 passed
    through a 'direct connection' block, and one more closes the stack (reference :101-146; walked by
    image_engine.pixelcnn_forward)."""

    def __init__(self, in_channels, out_channels, num_blocks, kernel_sizes, masked_channels):
        super().__init__()
        kernel_sizes = list(kernel_sizes)
        if num_blocks != len(kernel_sizes):
            raise ValueError("num_blocks=%d but %d kernel sizes" % (num_blocks, len(kernel_sizes)))
        stem = MaskABlock(in_channels, out_channels, kernel_sizes[0], masked_channels)
        self.main = nn.ModuleList([stem] + [PixelCNNBlock(out_channels, k) for k in kernel_sizes[1:]])
        self.direct_connects = nn.ModuleList([PixelCNNBlock(out_channels, k) for k in kernel_sizes[1:-1]])
        self.blocks = list(self.main)       # plain-list alias the reference also exposes


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


_KERNEL_PLANS = {'small': (7, 7, 7, 5, 5, 3, 3), 'large': (7,) * 5 + (5,) * 4 + (3,) * 4}
_HIDDEN = 64
_SIDE = 28


class PixelCNNDecoderV2(DecoderBase):
    """p(x|z) for 28x28 binary images (reference dec_pixelcnn_v2.py:149-232): z -> Linear -> fm_latent feature maps
    concatenated under the image, gated PixelCNN trunk, 1x1 conv BN ELU 1x1 conv sigmoid."""

    def __init__(self, args, ngpu=1, mode='large'):
        super().__init__()
        if mode not in _KERNEL_PLANS:
            raise ValueError('unknown mode: %s' % mode)
        self.ngpu, self.nz, self.nc = ngpu, args.nz, 1
        self.fm_latent = args.latent_feature_map
        self.img_latent = _SIDE * _SIDE * self.fm_latent
        if self.nz != 0:
            self.z_transform = nn.Sequential(nn.Linear(self.nz, self.img_latent))
        plan = list(_KERNEL_PLANS[mode])
        trunk = PixelCNN(self.nc + self.fm_latent, _HIDDEN, len(plan), plan, self.nc)
        head = [_pointwise(_HIDDEN, _HIDDEN), nn.BatchNorm2d(_HIDDEN), nn.ELU(), _pointwise(_HIDDEN, self.nc), nn.Sigmoid()]
        self.main = nn.Sequential(trunk, *head)
        self.reset_parameters()
        self._hip = _ie.ImageDecoderEngine(self)

    def reset_parameters(self):
        if self.nz != 0:
            lin = self.z_transform[0]
            nn.init.xavier_uniform_(lin.weight)
            nn.init.zeros_(lin.bias)
        _unit_affine_(self.main[2])

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

    def forward_probs(self, x_img, z2d):
        """Pixel probabilities sigma(logit) (batch, 1, 28, 28) of the decoder on image x_img given z2d (batch, nz), through the
        HIP forward (reference PixelCNNDecoderV2.forward, dec_pixelcnn_v2.py:165-170).  Inference only."""
        with torch.no_grad():
            self._hip.ensure(x_img.device)
            self._hip.forward(x_img.contiguous().float(), z2d.contiguous().float())
            B = x_img.shape[0]
            return torch.sigmoid(self._hip.logit.t.view(B, 1, _SIDE, _SIDE))

    def decode(self, z, deterministic=False, generator=None, incremental=True):
        """Ancestral sampling (reference dec_pixelcnn_v2.py:201-232): the image is filled pixel by pixel in
        raster order -- thresholded at 0.5 when `deterministic`, else a Bernoulli draw -- and a last full pass gives the
        probabilities.  -> (img (batch, 1, 28, 28), probs).

        incremental (default): each pixel's probability comes from ONE launch that evaluates every layer at that position
        only (image_engine.PixelCNNSampler, lv_pixelcnn_sample.hip), bit-equal to the full forward; incremental=False runs
        the reference's literal procedure, one full decoder pass per pixel (784 passes)."""
        batch_size = z.size(0)
        z2d = z.reshape(batch_size, -1)
        if not incremental or self.training:
            img = torch.zeros(batch_size, self.nc, _SIDE, _SIDE, device=z.device)
            for i in range(_SIDE):
                for j in range(_SIDE):
                    p = self.forward_probs(img, z2d)[:, :, i, j]
                    if deterministic:
                        img[:, :, i, j] = (p >= 0.5).float()
                    else:
                        img[:, :, i, j] = (torch.rand(p.shape, device=p.device, generator=generator) < p).float()
            return img, self.forward_probs(img, z2d)
        with torch.no_grad():
            smp = _ie.PixelCNNSampler(self).start(z2d)
            for i in range(_SIDE):
                for j in range(_SIDE):
                    p = torch.sigmoid(smp.step(i, j)).view(batch_size, 1)
                    if deterministic:
                        smp.set_pixel(i, j, (p >= 0.5).float())
                    else:
                        smp.set_pixel(i, j, (torch.rand(p.shape, device=p.device, generator=generator) < p).float())
            img = smp.img.clone()
        return img, self.forward_probs(img, z2d)
