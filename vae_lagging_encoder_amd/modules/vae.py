"""VAE -- drop-in for the reference's modules/vae.py (normal prior) on MI355X.

`loss`, `encode`, `encode_stats` are the hot-path surface; they compose the HIP-backed
encoder / reparam+KL / decoder autograd Functions.  The evaluation helpers keep the reference's method names and
semantics on top of the same HIP forward (nll_iw, eval_*, calc_mi_q: SURVEY.md 8f "next" rows).
"""
import math

import torch
import torch.nn as nn

from .. import engine as _eng
from .utils import log_sum_exp

_GENERATORS = {"beam": "beam_search_decode", "greedy": "greedy_decode", "sample": "sample_decode"}


class VAE(nn.Module):
    """Encoder q(z|x), decoder p(x|z), prior N(0, I)."""

    def __init__(self, encoder, decoder, args):
        super().__init__()
        self.encoder, self.decoder = encoder, decoder
        self.args, self.nz = args, args.nz
        dev = args.device
        self.prior = torch.distributions.normal.Normal(torch.zeros(self.nz, device=dev), torch.ones(self.nz, device=dev))

    def set_precision(self, precision, encoder_forward=None, forward_operands=None):
        """Arithmetic of the HIP path behind this model's `loss` / `backward` (no counterpart in the reference, whose precision is
        the tensors' dtype): "f32" (default; exact-f32 MFMA, north_star's 1e-4 parity path) or "bf16" (BASELINE.json's GPU
        configuration: bf16 matrix pipe with f32 accumulation for the large products and the recurrent operands; master weights,
        state, gradients and reductions stay f32).  encoder_forward="f32" with "bf16": the encoder's forward f32-accurate (the KL
        then meets 1e-4 too; see trainer.AggressiveTextTrainer).  Image models also take "bf16x3".  Returns self."""
        for m in (self.encoder, self.decoder):
            eng = getattr(m, "_hip", None)
            if eng is None:
                raise TypeError("%s has no HIP engine" % type(m).__name__)
            eng.precision = precision
        if hasattr(self.encoder._hip, "exact_forward"):
            self.encoder._hip.exact_forward = ("gx", "rec") if (encoder_forward == "f32" and precision == "bf16") else ()
        if forward_operands is not None:
            self.set_forward_operands(forward_operands)
        return self

    def set_forward_operands(self, fmt):
        """Number format of the ENCODER FORWARD's matrix-pipe operands under precision "bf16": "f16" (default: IEEE binary16 images
        of X, W_ih, W_hh and the h hand-off -- 11-bit significands on the same instructions and rates as bf16, which puts the KL
        within north_star's 1e-4; range assumption: |weights|, |embeddings|, |h| << 65504 (values beyond saturate at +-65504, values
        below 6e-5 are kept as binary16 subnormals: tests/test_gpu_kernels.py::test_binary16_subnormal_*) or "bf16" (the arithmetic of
        rounds 1-4: KL 2e-4..5e-4).  Gradient products and the BPTT are bf16 either way.  Returns self."""
        if fmt not in ("f16", "bf16"):
            raise ValueError("forward operands: 'f16' or 'bf16'")
        eng = getattr(self.encoder, "_hip", None)
        if eng is None or not hasattr(eng, "fwd_operands"):
            raise TypeError("%s has no binary16 forward" % type(self.encoder).__name__)
        eng.fwd_operands = fmt
        return self

    def use_flat_grads(self, flag=True):
        """Drop-in autograd path: hand the parameters' gradients out as VIEWS of the engines' flat gradient buffers after a plain
        `loss.backward()` (default; zero-copy, engine.FlatBuffer.deliver_grads -- a .grad obtained this way is overwritten by the next
        backward) or always as fresh tensors (False: stock autograd aliasing semantics, 215 MB of copies per step at the Yahoo
        shape).  Returns self."""
        for m in (self.encoder, self.decoder):
            eng = getattr(m, "_hip", None)
            if eng is not None and getattr(eng, "flat", None) is not None:
                eng.flat.flat_grads = bool(flag)
            elif eng is not None:
                eng._flat_grads_pref = bool(flag)
        return self

    def arithmetic(self):
        """The run-config record of the HIP path's arithmetic (what a checkpoint's sidecar / a log line should carry)."""
        e = getattr(self.encoder, "_hip", None)
        return {"precision": getattr(e, "precision", None), "encoder_forward": "f32" if getattr(e, "exact_forward", ()) else "operands",
                "forward_operands": getattr(e, "fwd_operands", None)}

    # ---- training path (reference vae.py:35-98) ---------------------------------------------------------
    def encode(self, x, nsamples=1, eps=None):
        """-> z (batch, nsamples, nz), KL (batch,).  `eps` injects the reparameterisation noise."""
        extra = {} if eps is None else {"eps": eps}
        return self.encoder.encode(x, nsamples, **extra)

    def encode_stats(self, x):
        """-> mu (batch, nz), logvar (batch, nz)."""
        return self.encoder(x)

    def loss(self, x, kl_weight, nsamples=1, noise=None):
        """-> (rec + kl_weight*KL, rec, KL), each (batch,).

        noise = (eps, mask_in, mask_out) injects the random draws of this call (parity tests); the default draws
        them from torch's device generator at the same three places the reference does."""
        eps, masks = (None, None) if noise is None else (noise[0], (noise[1], noise[2]))
        z, kl = self.encode(x, nsamples, eps=eps)
        dec_extra = {} if masks is None else {"masks": masks}
        rec = self.decoder.reconstruct_error(x, z, **dec_extra).mean(dim=1)
        return rec + kl_weight * kl, rec, kl

    def KL(self, x):
        return self.encode(x, 1)[1]

    # ---- generation (delegated; generation itself is outside the hot path) -----------------------------
    def decode(self, z, strategy, K=5):
        if strategy not in _GENERATORS:
            raise ValueError("the decoding strategy is not supported")
        fn = getattr(self.decoder, _GENERATORS[strategy])
        return fn(z, K) if strategy == "beam" else fn(z)

    def reconstruct(self, x, decoding_strategy="greedy", K=5):
        return self.decode(self.sample_from_inference(x).squeeze(1), decoding_strategy, K)

    def sample_from_prior(self, nsamples):
        return self.prior.sample((nsamples,))

    def sample_from_inference(self, x, nsamples=1):
        return self.encoder.sample(x, nsamples)[0]

    # ---- evaluation (reference vae.py:100-227) -------------------------------------------------------------
    def nll_iw(self, x, nsamples, ns=100):
        """Importance-weighted estimate of -log p(x) from `nsamples` draws, `ns` at a time -> (batch,)   (reference
        vae.py:100-129).  The decoder pass over batch*ns sequences is the hot path's HIP forward; log p(z), log q(z|x) and
        the final log_sum_exp are lv_eval.hip kernels."""
        log_w = []
        for _ in range(int(nsamples / ns)):
            z, stats = self.encoder.sample(x, ns)
            log_w.append(self.eval_complete_ll(x, z) - self.eval_inference_dist(x, z, stats))
        return -_eng.logsumexp_rows(torch.cat(log_w, dim=-1), -math.log(nsamples))

    def eval_prior_dist(self, zrange):
        """log N(z; 0, I) summed over the latent dimension (reference vae.py:135-145)."""
        if zrange.dim() == 3:
            return _eng.gauss_logpdf(zrange, None, None)
        return self.prior.log_prob(zrange).sum(dim=-1)

    def eval_cond_ll(self, x, z):
        return self.decoder.log_probability(x, z)

    def eval_complete_ll(self, x, z):
        """log p(z, x) for z (batch, nsamples, nz) -> (batch, nsamples)."""
        return self.eval_prior_dist(z) + self.eval_cond_ll(x, z)

    def eval_inference_dist(self, x, z, param=None):
        return self.encoder.eval_inference_dist(x, z, param)

    def eval_log_model_posterior(self, x, grid_z):
        """log p(z|x) on a grid of K points (K, nz) -> (batch, K), normalised over the grid."""
        n = x.size(0) if torch.is_tensor(x) else x[0].size(0)
        joint = self.eval_complete_ll(x, grid_z.unsqueeze(0).expand(n, *grid_z.size()).contiguous())
        return joint - log_sum_exp(joint, dim=1, keepdim=True)

    def calc_model_posterior_mean(self, x, grid_z):
        w = self.eval_log_model_posterior(x, grid_z).exp()
        return (w.unsqueeze(2) * grid_z.unsqueeze(0)).sum(1)

    def calc_infer_mean(self, x):
        return self.encoder.forward(x)[0]

    def calc_mi_q(self, x):
        return self.encoder.calc_mi(x)
