"""CPU oracle for the LSTM-VAE aggressive inner step.  TEST INFRASTRUCTURE -- not a product path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the package
`vae_lagging_encoder_amd` never does and fails loudly when its HIP extension is missing.

What it is: a restatement, in plain torch CPU tensor math, of the reference's algorithm for
    VAE.loss                         /root/reference/modules/vae.py:79-98
    GaussianEncoderBase.encode       /root/reference/modules/encoders/encoder.py:40-57
    GaussianEncoderBase.reparameterize  encoder.py:59-79
    LSTMEncoder.forward              /root/reference/modules/encoders/enc_lstm.py:47-64
    LSTMDecoder.decode               /root/reference/modules/decoders/dec_lstm.py:66-111
    LSTMDecoder.reconstruct_error    dec_lstm.py:113-148
    the inner-loop body              /root/reference/text.py:371-400 (zero_grad, loss, backward,
                                     clip_grad_norm_(all params, 5.0), encoder SGD step)
with the random draws (eps, the two dropout keep-masks) taken as explicit INPUTS.

Pinning: the arithmetic of the reference lives in PyTorch (unpinned by the reference, which has no tests).
tests/golden/make_golden.py imports the reference from /root/reference in the build container, replays its
RNG draws, and checks this oracle against it (loss/rec/KL, every gradient, clip coefficient, post-step
encoder weights) before writing the fixtures under tests/golden/; tests/test_oracle_golden.py re-checks
the oracle against those fixtures on every run.

Two implementations of the same function:
  * `*_explicit` : hand-written LSTM cell loop, logsumexp, etc. -- dtype-generic (run it in float64 for a
                   tight reference); gradients via autograd over these primitive ops.
  * `*_aten`     : the same graph expressed with the ATen composite ops the reference dispatches to
                   (torch._VF.lstm -> oneDNN on CPU, F.embedding, F.cross_entropy); used as the timed
                   `cpu_baseline` ("port") because it is what the reference's CPU path executes.
"""
import math

import torch
import torch.nn.functional as F

ENC_KEYS = ["encoder.embed.weight", "encoder.lstm.weight_ih_l0", "encoder.lstm.weight_hh_l0",
            "encoder.lstm.bias_ih_l0", "encoder.lstm.bias_hh_l0", "encoder.linear.weight"]
DEC_KEYS = ["decoder.embed.weight", "decoder.trans_linear.weight", "decoder.lstm.weight_ih_l0",
            "decoder.lstm.weight_hh_l0", "decoder.lstm.bias_ih_l0", "decoder.lstm.bias_hh_l0",
            "decoder.pred_linear.weight"]
ALL_KEYS = ENC_KEYS + DEC_KEYS


def lstm_explicit(x, w_ih, w_hh, b_ih, b_hh, h0, c0):
    """x (B,T,in) batch-first. PyTorch gate order i|f|g|o, two biases.

    Returns outputs (B,T,H), (h_T, c_T)."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h, c = h0, c0
    outs = []
    gx = x @ w_ih.t() + (b_ih + b_hh)
    for t in range(T):
        a = gx[:, t] + h @ w_hh.t()
        i = torch.sigmoid(a[:, 0 * H:1 * H])
        f = torch.sigmoid(a[:, 1 * H:2 * H])
        g = torch.tanh(a[:, 2 * H:3 * H])
        o = torch.sigmoid(a[:, 3 * H:4 * H])
        c = f * c + i * g
        h = o * torch.tanh(c)
        outs.append(h)
    out = torch.stack(outs, dim=1) if T > 0 else x.new_zeros(B, 0, H)
    return out, (h, c)


def lstm_aten(x, w_ih, w_hh, b_ih, b_hh, h0, c0):
    """Same as lstm_explicit through the ATen op nn.LSTM uses (enc_lstm.py:60, dec_lstm.py:104)."""
    out, hT, cT = torch._VF.lstm(x, (h0.unsqueeze(0), c0.unsqueeze(0)), [w_ih, w_hh, b_ih, b_hh],
                                 True, 1, 0.0, False, False, True)
    return out, (hT[0], cT[0])


def encoder_forward(P, x, impl="explicit"):
    """LSTMEncoder.forward (enc_lstm.py:47-64): embed ALL of x (G2), LSTM from zeros, Linear(no bias), chunk."""
    emb = P["encoder.embed.weight"][x]                      # (B,T,ni)
    B = x.shape[0]
    H = P["encoder.lstm.weight_hh_l0"].shape[1]
    z0 = emb.new_zeros(B, H)
    lstm = lstm_explicit if impl == "explicit" else lstm_aten
    _, (hT, _) = lstm(emb, P["encoder.lstm.weight_ih_l0"], P["encoder.lstm.weight_hh_l0"],
                      P["encoder.lstm.bias_ih_l0"], P["encoder.lstm.bias_hh_l0"], z0, z0)
    mulv = hT @ P["encoder.linear.weight"].t()
    nz = mulv.shape[1] // 2
    return mulv[:, :nz], mulv[:, nz:]


def reparam_kl(mu, logvar, eps):
    """encoder.py:53-55,71-79.  eps (B,ns,nz).  Returns z (B,ns,nz), KL (B,)."""
    std = (0.5 * logvar).exp()
    z = mu.unsqueeze(1) + eps * std.unsqueeze(1)
    kl = 0.5 * (mu.pow(2) + logvar.exp() - logvar - 1).sum(dim=1)
    return z, kl


def decoder_reconstruct_error(P, x, z, mask_in=None, mask_out=None, p_in=0.5, p_out=0.5, impl="explicit"):
    """LSTMDecoder.reconstruct_error + decode (dec_lstm.py:66-148).

    x (B,T) int64; z (B,ns,nz); mask_in (B*ns? no: (B,T-1,ni)) keep-mask of dropout_in, mask_out (B*ns,T-1,H)
    keep-mask of dropout_out; None = eval mode (no dropout).  Returns rec (B,ns)."""
    src, tgt = x[:, :-1], x[:, 1:]
    B, Td = src.shape
    ns, nz = z.shape[1], z.shape[2]
    E = P["decoder.embed.weight"]
    V = E.shape[0]
    # F.embedding with padding_idx=V-1 ("padding_idx=-1", dec_lstm.py:28, G3): forward identical, grad row zero
    we = F.embedding(src, E, padding_idx=V - 1)
    if mask_in is not None:
        we = we * mask_in.to(we.dtype) / (1.0 - p_in)
    if ns == 1:
        z_ = z.expand(B, Td, nz)
    else:
        we = we.unsqueeze(1).expand(B, ns, Td, we.shape[-1]).reshape(B * ns, Td, -1)
        z_ = z.unsqueeze(2).expand(B, ns, Td, nz).reshape(B * ns, Td, nz)
    inp = torch.cat((we, z_), -1)
    zf = z.reshape(B * ns, nz)
    c0 = zf @ P["decoder.trans_linear.weight"].t()
    h0 = torch.tanh(c0)                                     # G7
    lstm = lstm_explicit if impl == "explicit" else lstm_aten
    out, _ = lstm(inp, P["decoder.lstm.weight_ih_l0"], P["decoder.lstm.weight_hh_l0"],
                  P["decoder.lstm.bias_ih_l0"], P["decoder.lstm.bias_hh_l0"], h0, c0)
    if mask_out is not None:
        out = out * mask_out.to(out.dtype) / (1.0 - p_out)
    logits = out @ P["decoder.pred_linear.weight"].t()     # (B*ns,Td,V)
    if ns == 1:
        tg = tgt.reshape(-1)
    else:
        tg = tgt.unsqueeze(1).expand(B, ns, Td).reshape(-1)
    if impl == "explicit":
        flat = logits.reshape(-1, V)
        nll = torch.logsumexp(flat, dim=-1) - flat.gather(1, tg.unsqueeze(1)).squeeze(1)
    else:
        nll = F.cross_entropy(logits.reshape(-1, V), tg, reduction="none")
    return nll.view(B, ns, -1).sum(-1)


def vae_loss(P, x, kl_weight, eps, mask_in=None, mask_out=None, p_in=0.5, p_out=0.5, impl="explicit"):
    """VAE.loss (vae.py:79-98): returns (loss, rec, KL), each (B,)."""
    mu, logvar = encoder_forward(P, x, impl)
    z, kl = reparam_kl(mu, logvar, eps)
    rec = decoder_reconstruct_error(P, x, z, mask_in, mask_out, p_in, p_out, impl).mean(dim=1)
    return rec + kl_weight * kl, rec, kl


def clip_coef(total_norm, max_norm=5.0):
    """torch.nn.utils.clip_grad_norm_: coef = clamp(max_norm / (norm + 1e-6), max=1)."""
    return min(1.0, max_norm / (float(total_norm) + 1e-6))


def inner_step(P, x, kl_weight, eps, mask_in, mask_out, lr=1.0, clip=5.0, p_in=0.5, p_out=0.5,
               impl="explicit", update="encoder"):
    """One body of the aggressive loop (text.py:373-387): grads of mean_b loss_b wrt ALL params, global-norm
    clip over encoder+decoder grads (G1), SGD on the encoder ('encoder'), decoder ('decoder', the joint step
    under aggressive mode text.py:407-424) or both ('both').

    P: dict name->tensor (not modified).  Returns dict(loss, rec, kl, grads (unclipped), total_norm, coef,
    new_params (dict of updated tensors for the stepped side))."""
    Q = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    loss, rec, kl = vae_loss(Q, x, kl_weight, eps, mask_in, mask_out, p_in, p_out, impl)
    loss.mean(dim=-1).backward()
    grads = {k: (Q[k].grad if Q[k].grad is not None else torch.zeros_like(Q[k])) for k in ALL_KEYS}
    total = math.sqrt(sum(float(g.double().pow(2).sum()) for g in grads.values()))
    coef = clip_coef(total, clip)
    keys = {"encoder": ENC_KEYS, "decoder": DEC_KEYS, "both": ALL_KEYS}[update]
    new_params = {k: (P[k] - lr * (grads[k] * coef)).detach() for k in keys}
    return dict(loss=loss.detach(), rec=rec.detach(), kl=kl.detach(), grads=grads, total_norm=total, coef=coef,
                new_params=new_params)


def random_params(V, ni, H, nz, seed=0, scale=0.01, emb_scale=0.1, dtype=torch.float32, head_scale=None):
    """Parameters with the reference's shapes and init distribution (text.py:265-266: U(-0.01,0.01), embeddings
    U(-0.1,0.1)); NOT the reference's RNG stream (fixtures carry exact state_dicts where that matters).
    head_scale widens encoder.linear so that KL is O(1) instead of ~1e-5 (SURVEY.md 8c conditioning warning)."""
    g = torch.Generator().manual_seed(seed)

    def u(*shape, s=scale):
        return ((torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1) * s).to(dtype)
    P = {
        "encoder.embed.weight": u(V, ni, s=emb_scale),
        "encoder.lstm.weight_ih_l0": u(4 * H, ni),
        "encoder.lstm.weight_hh_l0": u(4 * H, H),
        "encoder.lstm.bias_ih_l0": u(4 * H),
        "encoder.lstm.bias_hh_l0": u(4 * H),
        "encoder.linear.weight": u(2 * nz, H, s=head_scale if head_scale else scale),
        "decoder.embed.weight": u(V, ni, s=emb_scale),
        "decoder.trans_linear.weight": u(H, nz),
        "decoder.lstm.weight_ih_l0": u(4 * H, ni + nz),
        "decoder.lstm.weight_hh_l0": u(4 * H, H),
        "decoder.lstm.bias_ih_l0": u(4 * H),
        "decoder.lstm.bias_hh_l0": u(4 * H),
        "decoder.pred_linear.weight": u(V, H),
    }
    return P


def synthetic_batch(B, T, V, seed=0):
    """ids ~ U{4..V-1}, column 0 = <s> (1), last column = </s> (2)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randint(4, V, (B, T), generator=g, dtype=torch.int64)
    x[:, 0] = 1
    x[:, -1] = 2
    return x


def draw_noise(B, T, ni, H, nz, ns=1, p_in=0.5, p_out=0.5, seed=0):
    """eps + the two keep-masks in the reference's shapes/order (App. B), from a private generator."""
    g = torch.Generator().manual_seed(seed)
    eps = torch.randn(B, ns, nz, generator=g)
    mask_in = (torch.rand(B, T - 1, ni, generator=g) >= p_in)
    mask_out = (torch.rand(B * ns, T - 1, H, generator=g) >= p_out)
    return eps, mask_in, mask_out
