"""Model construction the way the reference's entry scripts do it, and the synthetic inputs of the benchmark.

`build_text_vae` follows text.py:265-279 (uniform initialisers, LSTMEncoder / LSTMDecoder / VAE, `.to(device)`),
`build_image_vae` follows image.py:225-241; `synthetic_batch` is SURVEY.md 8d's input distribution.  Used by
bench.py, `__graft_entry__.smoke()` and the tests, so that all of them build exactly the same modules.
"""
import argparse

import torch


class SizedVocab(object):
    """The slice of the reference's VocabEntry (data/text_data.py:11-47) that LSTMDecoder touches."""

    def __init__(self, n):
        self.n = n
        self.word2id = {"<pad>": 0, "<s>": 1, "</s>": 2, "<unk>": 3}

    def __len__(self):
        return self.n

    def __getitem__(self, word):
        return self.word2id[word]

    def id2word(self, i):
        return "w%d" % i


class UniformInit(object):
    """text.py:23-30: uniform(-stdv, stdv)."""

    def __init__(self, stdv):
        self.stdv = stdv

    def __call__(self, tensor):
        torch.nn.init.uniform_(tensor, -self.stdv, self.stdv)


def build_text_vae(V, ni, H, nz, device, seed=0, model_scale=0.01, emb_scale=0.1, params=None,
                   dropout_in=0.5, dropout_out=0.5, vocab=None):
    """vocab: a data.VocabEntry (what text.py:251,277 hands the decoder); default a sized stand-in with the same special ids."""
    from .modules import VAE, LSTMDecoder, LSTMEncoder
    args = argparse.Namespace(ni=ni, enc_nh=H, dec_nh=H, nz=nz, dec_dropout_in=dropout_in, dec_dropout_out=dropout_out,
                              device=torch.device(device))
    torch.manual_seed(seed)
    enc = LSTMEncoder(args, V, UniformInit(model_scale), UniformInit(emb_scale))
    dec = LSTMDecoder(args, vocab if vocab is not None else SizedVocab(V), UniformInit(model_scale), UniformInit(emb_scale))
    vae = VAE(enc, dec, args)
    if params is not None:
        missing, unexpected = vae.load_state_dict(params, strict=False)
        if not set(missing) <= {"decoder.loss.weight"} or unexpected:      # CrossEntropyLoss's ones() buffer may be absent
            raise KeyError("state dict mismatch: missing %s unexpected %s" % (sorted(missing), sorted(unexpected)))
    vae = vae.to(device)
    vae.train()
    return vae


def build_image_vae(device, seed, nz=32, latent_feature_map=4):
    from .modules import VAE, PixelCNNDecoderV2, ResNetEncoderV2
    args = argparse.Namespace(nz=nz, latent_feature_map=latent_feature_map, device=torch.device(device))
    torch.manual_seed(seed)
    vae = VAE(ResNetEncoderV2(args), PixelCNNDecoderV2(args), args).to(device)
    vae.train()
    return vae


def synthetic_batch(B, T, V, seed=0, dist="uniform"):
    """ids ~ U{4..V-1} (dist="zipf": rank-frequency 1/r over the same range, like natural text -- the most frequent
    token then occurs a few hundred times in a 6000-token batch), column 0 = <s> (1), last column = </s> (2); int64 [B][T] on the CPU."""
    g = torch.Generator().manual_seed(seed)
    if dist == "zipf":
        w = 1.0 / torch.arange(1, V - 3, dtype=torch.float64)
        x = (torch.multinomial(w, B * T, replacement=True, generator=g) + 4).view(B, T).to(torch.int64)
    else:
        x = torch.randint(4, V, (B, T), generator=g, dtype=torch.int64)
    x[:, 0] = 1
    x[:, -1] = 2
    return x
