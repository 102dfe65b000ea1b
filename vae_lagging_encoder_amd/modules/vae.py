"""VAE -- drop-in for the reference's modules/vae.py (normal prior) on MI355X.

`loss`, `encode`, `encode_stats` are the hot-path surface; they compose the HIP-backed
encoder / reparam+KL / decoder autograd Functions.  The evaluation helpers keep the reference's names and
semantics on top of the same HIP forward (nll_iw, eval_*, calc_mi_q: SURVEY.md 8f "next" rows).
"""
import math

import torch
import torch.nn as nn

from .utils import log_sum_exp


class VAE(nn.Module):
    """VAE with a standard normal prior."""

    def __init__(self, encoder, decoder, args):
        super(VAE, self).__init__()
        self.encoder = encoder
        self.decoder = decoder
        self.args = args
        self.nz = args.nz
        loc = torch.zeros(self.nz, device=args.device)
        scale = torch.ones(self.nz, device=args.device)
        self.prior = torch.distributions.normal.Normal(loc, scale)

    def encode(self, x, nsamples=1, eps=None):
        """-> z (batch, nsamples, nz), KL (batch,)."""
        if eps is None:
            return self.encoder.encode(x, nsamples)
        return self.encoder.encode(x, nsamples, eps=eps)

    def encode_stats(self, x):
        """-> mu (batch, nz), logvar (batch, nz)."""
        return self.encoder(x)

    def decode(self, z, strategy, K=5):
        if strategy == "beam":
            return self.decoder.beam_search_decode(z, K)
        elif strategy == "greedy":
            return self.decoder.greedy_decode(z)
        elif strategy == "sample":
            return self.decoder.sample_decode(z)
        raise ValueError("the decoding strategy is not supported")

    def reconstruct(self, x, decoding_strategy="greedy", K=5):
        z = self.sample_from_inference(x).squeeze(1)
        return self.decode(z, decoding_strategy, K)

    def loss(self, x, kl_weight, nsamples=1, noise=None):
        """-> (rec + kl_weight*KL, rec, KL), each (batch,)   (reference vae.py:79-98).

        noise = (eps, mask_in, mask_out) injects the random draws of this call (parity tests); the default draws
        them from torch's device generator at the same three places the reference does."""
        if noise is None:
            z, KL = self.encode(x, nsamples)
            reconstruct_err = self.decoder.reconstruct_error(x, z).mean(dim=1)
        else:
            eps, m_in, m_out = noise
            z, KL = self.encode(x, nsamples, eps=eps)
            reconstruct_err = self.decoder.reconstruct_error(x, z, masks=(m_in, m_out)).mean(dim=1)
        return reconstruct_err + kl_weight * KL, reconstruct_err, KL

    def nll_iw(self, x, nsamples, ns=100):
        """Importance-weighted estimate of -log p(x), `ns` samples at a time -> (batch,)."""
        tmp = []
        for _ in range(int(nsamples / ns)):
            z, param = self.encoder.sample(x, ns)
            log_comp_ll = self.eval_complete_ll(x, z)
            log_infer_ll = self.eval_inference_dist(x, z, param)
            tmp.append(log_comp_ll - log_infer_ll)
        ll_iw = log_sum_exp(torch.cat(tmp, dim=-1), dim=-1) - math.log(nsamples)
        return -ll_iw

    def KL(self, x):
        _, KL = self.encode(x, 1)
        return KL

    def eval_prior_dist(self, zrange):
        return self.prior.log_prob(zrange).sum(dim=-1)

    def eval_complete_ll(self, x, z):
        """log p(z, x) for z (batch, nsamples, nz) -> (batch, nsamples)."""
        return self.eval_prior_dist(z) + self.eval_cond_ll(x, z)

    def eval_cond_ll(self, x, z):
        return self.decoder.log_probability(x, z)

    def eval_log_model_posterior(self, x, grid_z):
        batch_size = x.size(0) if torch.is_tensor(x) else x[0].size(0)
        grid_z = grid_z.unsqueeze(0).expand(batch_size, *grid_z.size()).contiguous()
        log_comp = self.eval_complete_ll(x, grid_z)
        return log_comp - log_sum_exp(log_comp, dim=1, keepdim=True)

    def sample_from_prior(self, nsamples):
        return self.prior.sample((nsamples,))

    def sample_from_inference(self, x, nsamples=1):
        z, _ = self.encoder.sample(x, nsamples)
        return z

    def calc_model_posterior_mean(self, x, grid_z):
        posterior = self.eval_log_model_posterior(x, grid_z).exp()
        return torch.mul(posterior.unsqueeze(2), grid_z.unsqueeze(0)).sum(1)

    def calc_infer_mean(self, x):
        mean, _ = self.encoder.forward(x)
        return mean

    def eval_inference_dist(self, x, z, param=None):
        return self.encoder.eval_inference_dist(x, z, param)

    def calc_mi_q(self, x):
        return self.encoder.calc_mi(x)
