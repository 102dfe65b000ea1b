"""GaussianEncoderBase -- drop-in for the reference's modules/encoders/encoder.py.

Hot path: `encode` and `reparameterize` run the fused HIP kernel
lv_reparam_kl_{fwd,bwd}_f32 (z = mu + eps*exp(0.5 logvar); KL = 0.5*sum(mu^2 + e^lv - lv - 1), wave64 shuffle
reduction over nz).  eps is drawn with torch's device generator exactly where the reference draws it
(encoder.py:77), or injected through the optional `eps=` argument (parity tests).
The evaluation helpers (eval_inference_dist, calc_mi: SURVEY.md 8f row 1) run the kernels of lv_eval.hip.
"""
import torch
import torch.nn as nn

from ... import engine as _eng


class _ReparamKLFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mulv, eps):
        z, kl = _eng.reparam_kl_forward(mulv, eps)
        ctx.save_for_backward(mulv, eps)
        return z, kl

    @staticmethod
    def backward(ctx, dz, dkl):
        mulv, eps = ctx.saved_tensors
        if dz is None:
            dz = torch.zeros_like(eps)
        if dkl is None:
            dkl = torch.zeros(mulv.shape[0], dtype=mulv.dtype, device=mulv.device)
        return _eng.reparam_kl_backward(mulv, eps, dz, dkl), None


class GaussianEncoderBase(nn.Module):
    """q(z|x) = N(mu(x), diag(exp(logvar(x))))."""

    def __init__(self):
        super(GaussianEncoderBase, self).__init__()

    def forward(self, x):
        """x (batch, *) -> mu (batch, nz), logvar (batch, nz)."""
        raise NotImplementedError

    def _forward_mulv(self, x):
        """[mu | logvar] as one (batch, 2nz) tensor; subclasses with a fused head override this."""
        mu, logvar = self.forward(x)
        return torch.cat((mu, logvar), dim=-1)

    def _draw_eps(self, batch, nsamples, nz, device, eps=None):
        if eps is not None:
            assert tuple(eps.shape) == (batch, nsamples, nz)
            return eps.to(device=device, dtype=torch.float32)
        return torch.zeros(batch, nsamples, nz, device=device).normal_()

    def sample(self, input, nsamples):
        """-> z (batch, nsamples, nz), (mu, logvar)."""
        mulv = self._forward_mulv(input)
        nz = mulv.shape[1] // 2
        eps = self._draw_eps(mulv.shape[0], nsamples, nz, mulv.device)
        z, _ = _ReparamKLFn.apply(mulv, eps)
        return z, (mulv[:, :nz], mulv[:, nz:])

    def encode(self, input, nsamples, eps=None):
        """-> z (batch, nsamples, nz), KL (batch,)   (reference encoder.py:40-57)."""
        mulv = self._forward_mulv(input)
        nz = mulv.shape[1] // 2
        eps = self._draw_eps(mulv.shape[0], nsamples, nz, mulv.device, eps)
        return _ReparamKLFn.apply(mulv, eps)

    def reparameterize(self, mu, logvar, nsamples=1, eps=None):
        """mu, logvar (batch, nz) -> z (batch, nsamples, nz)   (reference encoder.py:59-79)."""
        mulv = torch.cat((mu, logvar), dim=-1)
        eps = self._draw_eps(mu.shape[0], nsamples, mu.shape[1], mu.device, eps)
        z, _ = _ReparamKLFn.apply(mulv, eps)
        return z

    def eval_inference_dist(self, x, z, param=None):
        """log q(z|x) for z (batch, nsamples, nz) -> (batch, nsamples)   (reference encoder.py:81-109; lv_gauss_logpdf_f32)."""
        mu, logvar = param if param else self.forward(x)
        return _eng.gauss_logpdf(z, mu, logvar)

    def calc_mi(self, x, eps=None):
        """I(x;z) under q: E_x E_q log q(z|x) - E_x E_q log q(z), aggregate posterior from the same batch (reference
        encoder.py:111-145).  One HIP launch pair on top of the encoder forward: no (z_batch, x_batch, nz) temporary."""
        mu, logvar = self.forward(x)
        z = self.reparameterize(mu, logvar, 1, eps=eps)            # (z_batch, 1, nz)
        return float(_eng.calc_mi(mu, logvar, z.reshape(z.shape[0], -1))[0].item())
