"""CPU oracle for the Omniglot VAE aggressive inner step.  TEST INFRASTRUCTURE -- not a product path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Restates, with plain torch CPU functional ops on a dict of tensors keyed by the reference's state_dict names:
    ResNetEncoderV2.forward / ResNetBlock.forward        /root/reference/modules/encoders/enc_resnet_v2.py:55-71,120-126
    PixelCNNDecoderV2.reconstruct_error / forward        /root/reference/modules/decoders/dec_pixelcnn_v2.py:165-195
    MaskedConv2d.forward, PixelCNNBlock, MaskABlock, PixelCNN.forward   dec_pixelcnn_v2.py:12-121
    VAE.loss                                              /root/reference/modules/vae.py:79-98
    the inner-loop body with Adam                         /root/reference/image.py:300-314
eps (the reparameterisation noise) and the binarised image are explicit inputs.  BatchNorm runs in train mode (batch
statistics; running statistics updated with momentum 0.1 and the unbiased variance -- SURVEY.md G11).
Pinned by tests/golden/make_golden_image.py against the imported reference (fixtures tests/golden/image_*.npz).
"""
import math

import torch
import torch.nn.functional as F

LARGE_KERNELS = [7, 7, 7, 7, 7, 5, 5, 5, 5, 3, 3, 3, 3]


class Ctx(object):
    """Carries the parameter dict, the leaf tensors created for autograd and the updated BN running statistics."""

    def __init__(self, P, train=True):
        self.P = P
        self.train = train
        self.leaf = {}
        self.new_stats = {}

    def w(self, key, mask_key=None):
        if key not in self.leaf:
            v = self.P[key]
            if mask_key is not None:
                v = v * self.P[mask_key]          # MaskedConv2d.forward: weight.data.mul_(mask) (in place, outside autograd)
            self.leaf[key] = v.detach().clone().requires_grad_(True)
        return self.leaf[key]

    def bn(self, x, pre):
        w, b = self.w(pre + ".weight"), self.w(pre + ".bias")
        if not self.train:
            return F.batch_norm(x, self.P[pre + ".running_mean"], self.P[pre + ".running_var"], w, b, False, 0.1, 1e-5)
        rm, rv = self.P[pre + ".running_mean"].clone(), self.P[pre + ".running_var"].clone()
        y = F.batch_norm(x, rm, rv, w, b, True, 0.1, 1e-5)
        self.new_stats[pre + ".running_mean"], self.new_stats[pre + ".running_var"] = rm, rv
        return y


def resnet_block(c, pre, x, stride, has_ds=True):
    residual = x
    if has_ds:
        residual = c.bn(F.conv2d(x, c.w(pre + ".downsample.0.weight"), stride=stride), pre + ".downsample.1")
    out = F.elu(c.bn(F.conv2d(x, c.w(pre + ".conv1.weight"), stride=stride, padding=1), pre + ".bn1"))
    out = c.bn(F.conv2d(out, c.w(pre + ".conv2.weight"), padding=1), pre + ".bn2")
    return F.elu(out + residual)


def encoder_forward(c, x):
    """x (B,1,28,28) -> mu, logvar (B,nz)."""
    h = x
    for i in range(3):
        h = resnet_block(c, "encoder.main.0.main.%d" % i, h, 2)
    h = F.elu(c.bn(F.conv2d(h, c.w("encoder.main.1.weight")), "encoder.main.2"))
    o = h.view(h.size(0), -1) @ c.w("encoder.linear.weight").t() + c.w("encoder.linear.bias")
    nz = o.shape[1] // 2
    return o[:, :nz], o[:, nz:]


def pixelcnn_block(c, pre, x, k):
    h = F.elu(c.bn(F.conv2d(x, c.w(pre + ".main.0.weight")), pre + ".main.1"))
    h = F.elu(c.bn(F.conv2d(h, c.w(pre + ".main.3.weight", pre + ".main.3.mask"), padding=k // 2), pre + ".main.4"))
    h = c.bn(F.conv2d(h, c.w(pre + ".main.6.weight")), pre + ".main.7")
    return F.elu(h + x)


def decoder_reconstruct_error(c, x, z, kernels=LARGE_KERNELS):
    """x (B,1,28,28) in {0,1}; z (B,ns,nz) -> BCE (B,ns)."""
    B, ns, nz = z.shape
    fm = c.P["decoder.z_transform.0.weight"].shape[0] // 784
    zt = (z @ c.w("decoder.z_transform.0.weight").t() + c.w("decoder.z_transform.0.bias")).view(B, ns, fm, 28, 28)
    img = x.unsqueeze(1).expand(B, ns, *x.shape[1:])
    inp = torch.cat([img, zt], dim=2).reshape(B * ns, 1 + fm, 28, 28)
    pre = "decoder.main.0"
    kA = kernels[0]
    h = F.elu(c.bn(F.conv2d(inp, c.w(pre + ".main.0.main.0.weight", pre + ".main.0.main.0.mask"), padding=kA // 2),
                   pre + ".main.0.main.1"))
    direct = [h]
    for i in range(1, len(kernels)):
        if i > 2:
            h = h + pixelcnn_block(c, pre + ".direct_connects.%d" % (i - 3), direct.pop(0), kernels[i - 2])
        h = pixelcnn_block(c, pre + ".main.%d" % i, h, kernels[i])
        direct.append(h)
    h = h + pixelcnn_block(c, pre + ".direct_connects.%d" % (len(kernels) - 3), direct.pop(0), kernels[len(kernels) - 2])
    h = F.elu(c.bn(F.conv2d(h, c.w("decoder.main.1.weight")), "decoder.main.2"))
    p = torch.sigmoid(F.conv2d(h, c.w("decoder.main.4.weight"))).view(B, ns, -1)
    xf = x.view(B, -1).unsqueeze(1)
    bce = (p + 1e-12).log() * xf + (1.0 - p + 1e-12).log() * (1.0 - xf)
    return -bce.sum(dim=2)


def vae_loss(c, x, kl_weight, eps):
    mu, logvar = encoder_forward(c, x)
    z = mu.unsqueeze(1) + eps * (0.5 * logvar).exp().unsqueeze(1)
    kl = 0.5 * (mu.pow(2) + logvar.exp() - logvar - 1).sum(dim=1)
    rec = decoder_reconstruct_error(c, x, z).mean(dim=1)
    return rec + kl_weight * kl, rec, kl


def param_keys(P):
    return [k for k in P if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked")
                                 or k.endswith(".mask"))]


def inner_step_adam(P, x, kl_weight, eps, adam=None, lr=1e-3, clip=5.0, betas=(0.9, 0.999), adam_eps=1e-8):
    """image.py:300-314: grads of mean_b loss_b, clip over ALL parameters, Adam step on the encoder.
    adam: dict(step, m{key}, v{key}) or None (fresh).  Returns dict(...)."""
    c = Ctx(P, train=True)
    loss, rec, kl = vae_loss(c, x, kl_weight, eps)
    loss.mean(dim=-1).backward()
    keys = param_keys(P)
    grads = {k: (c.leaf[k].grad if k in c.leaf and c.leaf[k].grad is not None else torch.zeros_like(P[k])) for k in keys}
    total = math.sqrt(sum(float(g.double().pow(2).sum()) for g in grads.values()))
    coef = min(1.0, clip / (total + 1e-6))
    enc = [k for k in keys if k.startswith("encoder.")]
    if adam is None:
        adam = dict(step=0, m={k: torch.zeros_like(P[k]) for k in enc}, v={k: torch.zeros_like(P[k]) for k in enc})
    t = adam["step"] + 1
    new = {}
    m2, v2 = {}, {}
    for k in enc:
        g = grads[k] * coef
        m2[k] = adam["m"][k] + (g - adam["m"][k]) * (1 - betas[0])
        v2[k] = adam["v"][k] * betas[1] + (1 - betas[1]) * g * g
        denom = v2[k].sqrt() / math.sqrt(1 - betas[1] ** t) + adam_eps
        new[k] = P[k] - (lr / (1 - betas[0] ** t)) * (m2[k] / denom)
    # MaskedConv2d weights after this forward's in-place weight.data.mul_(mask) (G5)
    masked = {k: c.leaf[k].detach() for k in c.leaf if k.endswith(".weight") and (k[:-len(".weight")] + ".mask") in P}
    return dict(loss=loss.detach(), rec=rec.detach(), kl=kl.detach(), grads=grads, total_norm=total, coef=coef,
                new_params=new, new_stats=c.new_stats, masked_weights=masked, adam=dict(step=t, m=m2, v=v2))
