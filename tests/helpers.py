"""Shared helpers for the parity tests (fixture loading, model construction, comparisons)."""
import argparse
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


class Vocab(object):
    def __init__(self, n):
        self.n = n
        self.w2i = {"<pad>": 0, "<s>": 1, "</s>": 2, "<unk>": 3}

    def __len__(self):
        return self.n

    def __getitem__(self, w):
        return self.w2i[w]

    def id2word(self, i):
        return "w%d" % i


class uniform_initializer(object):
    def __init__(self, stdv):
        self.stdv = stdv

    def __call__(self, tensor):
        torch.nn.init.uniform_(tensor, -self.stdv, self.stdv)


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def fixture_params(fx, prefix="param/"):
    return {k: torch.from_numpy(fx[prefix + k]) for k in ALL_KEYS if (prefix + k) in fx}


def build_vae(V, ni, H, nz, device, seed=0, model_scale=0.01, emb_scale=0.1, params=None):
    """Our drop-in modules, constructed exactly as text.py:265-279 constructs the reference's."""
    from vae_lagging_encoder_amd.modules import VAE, LSTMEncoder, LSTMDecoder
    args = argparse.Namespace(ni=ni, enc_nh=H, dec_nh=H, nz=nz, dec_dropout_in=0.5, dec_dropout_out=0.5,
                              device=torch.device(device))
    torch.manual_seed(seed)
    enc = LSTMEncoder(args, V, uniform_initializer(model_scale), uniform_initializer(emb_scale))
    dec = LSTMDecoder(args, Vocab(V), uniform_initializer(model_scale), uniform_initializer(emb_scale))
    vae = VAE(enc, dec, args)
    if params is not None:
        missing, unexpected = vae.load_state_dict(params, strict=False)
        assert set(missing) <= {"decoder.loss.weight"} and not unexpected   # CrossEntropyLoss's ones() buffer
    vae = vae.to(device)
    vae.train()
    return vae


def rel_err(a, b, floor=0.0):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + floor + 1e-300))
