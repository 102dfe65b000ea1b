"""LSTMDecoder -- drop-in for the reference's modules/decoders/dec_lstm.py:17-161 on MI355X (training path).

Same constructor, containers, construction order and state_dict keys (`embed.weight` with padding_idx=-1 => V-1
(SURVEY.md G3), `trans_linear.weight`, `lstm.*_l0`, `pred_linear.weight`).  `reconstruct_error` runs the HIP path:
embedding gather fused with dropout_in, z-projection folded into the input GEMM epilogue (the cat((embed, z)) is
never materialised), fused LSTM step kernels with dropout_out in the epilogue, f32 MFMA vocabulary projection,
row softmax-NLL; hand-written backward.  Generation (beam/greedy/sample, reference lines 163-367) is outside the
hot path (SURVEY.md section 8 scope table) and not provided.
"""
import torch
import torch.nn as nn

from ... import engine as _eng
from .decoder import DecoderBase


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, x, z, mask_in, mask_out, p_in, p_out, *params):
        rec = eng.forward(x, z, mask_in, mask_out, p_in, p_out)
        ctx.eng = eng
        ctx.gen = eng.gen
        ctx.zshape = z.shape
        return rec.clone()

    @staticmethod
    def backward(ctx, drec):
        eng = ctx.eng
        dz = eng.backward(drec, ctx.gen).clone().view(ctx.zshape)
        eng.join()
        grads = tuple(eng.flat.gviews[n].clone() for n in eng.flat.names)
        return (None, None, dz, None, None, None, None) + grads


class LSTMDecoder(DecoderBase):
    """LSTM decoder with constant-length batching (reference dec_lstm.py:17-161)."""

    def __init__(self, args, vocab, model_init, emb_init):
        super(LSTMDecoder, self).__init__()
        self.ni = args.ni
        self.nh = args.dec_nh
        self.nz = args.nz
        self.vocab = vocab
        self.device = args.device

        # "no padding" in the reference's comment, but padding_idx=-1 resolves to row V-1 (G3)
        self.embed = nn.Embedding(len(vocab), args.ni, padding_idx=-1)
        self.dropout_in = nn.Dropout(args.dec_dropout_in)
        self.dropout_out = nn.Dropout(args.dec_dropout_out)
        # initial cell state from z
        self.trans_linear = nn.Linear(args.nz, args.dec_nh, bias=False)
        # z is concatenated to every step input
        self.lstm = nn.LSTM(input_size=args.ni + args.nz, hidden_size=args.dec_nh, num_layers=1, batch_first=True)
        self.pred_linear = nn.Linear(args.dec_nh, len(vocab), bias=False)

        vocab_mask = torch.ones(len(vocab))
        self.loss = nn.CrossEntropyLoss(weight=vocab_mask, reduction="none")

        self.reset_parameters(model_init, emb_init)
        self._hip = _eng.LSTMDecoderEngine(self)

    def reset_parameters(self, model_init, emb_init):
        for param in self.parameters():
            model_init(param)
        emb_init(self.embed.weight)

    def _params(self):
        return (self.embed.weight, self.trans_linear.weight, self.lstm.weight_ih_l0, self.lstm.weight_hh_l0,
                self.lstm.bias_ih_l0, self.lstm.bias_hh_l0, self.pred_linear.weight)

    def _draw_masks(self, B, Td, device):
        """nn.Dropout keep-masks for dropout_in (B,Td,ni) / dropout_out (B,Td,nh); None when not training."""
        p_in, p_out = self.dropout_in.p, self.dropout_out.p
        m_in = m_out = None
        if self.training and p_in > 0:
            m_in = torch.empty(B, Td, self.ni, device=device).bernoulli_(1 - p_in).to(torch.uint8)
        if self.training and p_out > 0:
            m_out = torch.empty(B, Td, self.nh, device=device).bernoulli_(1 - p_out).to(torch.uint8)
        return m_in, m_out

    def reconstruct_error(self, x, z, masks=None):
        """Token cross entropy summed over time.  x (batch, seq_len) int64, z (batch, n_sample, nz)
        -> (batch, n_sample).  `masks=(mask_in, mask_out)` injects the dropout keep-masks (parity tests)."""
        B, T = x.shape
        ns = z.size(1)
        self._hip.ensure(x.device)
        if masks is None:
            m_in, m_out = self._draw_masks(B, T - 1, x.device)
            if ns > 1 and m_out is not None:
                m_out = torch.empty(B * ns, T - 1, self.nh, device=x.device).bernoulli_(1 - self.dropout_out.p).to(torch.uint8)
        else:
            m_in, m_out = masks
            m_in = None if m_in is None else m_in.to(device=x.device, dtype=torch.uint8).contiguous()
            m_out = None if m_out is None else m_out.to(device=x.device, dtype=torch.uint8).contiguous()
        if ns > 1:
            # dec_lstm.py:86-94: the same (dropped) word embeddings for every sample of a sentence
            x = x.repeat_interleave(ns, dim=0)
            if m_in is not None:
                m_in = m_in.repeat_interleave(ns, dim=0).contiguous()
            z = z.reshape(B * ns, 1, self.nz)
        rec = _DecoderFn.apply(self._hip, x, z, m_in, m_out, self.dropout_in.p, self.dropout_out.p, *self._params())
        return rec.view(B, ns)

    def log_probability(self, x, z):
        """log p(x|z): (batch, n_sample)."""
        return -self.reconstruct_error(x, z)

    def decode(self, input, z):
        """Logits (batch*n_sample, seq_len, vocab) of the teacher-forced decoder (reference dec_lstm.py:66-111).
        Inference-only view of the HIP path's logits buffer (not differentiable; training goes through
        reconstruct_error, which never hands logits back to Python)."""
        B, T = input.shape
        ns = z.size(1)
        with torch.no_grad():
            # run the teacher-forced path on `input` as the source sequence: append a dummy target column
            x = torch.cat((input, input[:, -1:]), dim=1)
            self.reconstruct_error(x, z)
            w = self._hip._ws(B * ns, T)
            V = len(self.vocab)
            return w.logits[:, :V].reshape(T, B * ns, V).transpose(0, 1).contiguous()

    def beam_search_decode(self, z, K=5):
        raise NotImplementedError("generation is outside the MI355X hot path (SURVEY.md section 8: out of scope)")

    def greedy_decode(self, z):
        raise NotImplementedError("generation is outside the MI355X hot path (SURVEY.md section 8: out of scope)")

    def sample_decode(self, z):
        raise NotImplementedError("generation is outside the MI355X hot path (SURVEY.md section 8: out of scope)")
