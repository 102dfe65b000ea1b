"""Shared helpers for the parity tests (fixture loading, model construction, comparisons)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

ENC_KEYS = ["encoder.embed.weight", "encoder.lstm.weight_ih_l0", "encoder.lstm.weight_hh_l0",
            "encoder.lstm.bias_ih_l0", "encoder.lstm.bias_hh_l0", "encoder.linear.weight"]
DEC_KEYS = ["decoder.embed.weight", "decoder.trans_linear.weight", "decoder.lstm.weight_ih_l0",
            "decoder.lstm.weight_hh_l0", "decoder.lstm.bias_ih_l0", "decoder.lstm.bias_hh_l0",
            "decoder.pred_linear.weight"]
ALL_KEYS = ENC_KEYS + DEC_KEYS


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def fixture_params(fx, prefix="param/"):
    return {k: torch.from_numpy(fx[prefix + k]) for k in ALL_KEYS if (prefix + k) in fx}


def build_vae(V, ni, H, nz, device, seed=0, model_scale=0.01, emb_scale=0.1, params=None):
    """Our drop-in modules, constructed exactly as text.py:265-279 constructs the reference's."""
    from vae_lagging_encoder_amd.factory import build_text_vae
    return build_text_vae(V, ni, H, nz, device, seed=seed, model_scale=model_scale, emb_scale=emb_scale, params=params)


def rel_err(a, b, floor=0.0):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + floor + 1e-300))
