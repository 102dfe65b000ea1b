"""LSTMDecoder -- drop-in for the reference's modules/decoders/dec_lstm.py:17-161 on MI355X (training path).

Same constructor, containers, construction order and state_dict keys (`embed.weight` with padding_idx=-1 => V-1
(SURVEY.md G3), `trans_linear.weight`, `lstm.*_l0`, `pred_linear.weight`).  `reconstruct_error` runs the HIP path:
embedding gather fused with dropout_in, z-projection folded into the input GEMM epilogue (the cat((embed, z)) is
never materialised), fused LSTM step kernels with dropout_out in the epilogue, f32 MFMA vocabulary projection,
row softmax-NLL; hand-written backward.  Generation (beam / greedy / sample decoding, reference lines 163-367;
SURVEY.md 8f row 4) steps the same kernels one token at a time through engine.LSTMDecodeStepper.
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
        eng.flat.before_autograd_backward()
        dz = eng.backward(drec, ctx.gen).clone().view(ctx.zshape)
        eng.join()
        return (None, None, dz, None, None, None, None) + eng.flat.deliver_grads()


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
            fused, self._hip.fused_nll = self._hip.fused_nll, False       # this view needs the f32 logits image
            try:
                self.reconstruct_error(x, z)
            finally:
                self._hip.fused_nll = fused
            w = self._hip._ws(B * ns, T)
            V = len(self.vocab)
            return w.logits[:, :V].reshape(T, B * ns, V).transpose(0, 1).contiguous()

    # ---- generation (reference dec_lstm.py:163-367) ------------------------------------------------------
    def _stepper(self, device):
        st = getattr(self, "_gen", None)
        if st is None or st.device != torch.device(device):
            st = self._gen = _eng.LSTMDecodeStepper(self._hip, device)
        return st

    def _roll_out(self, z, pick):
        """Shared loop of greedy_decode / sample_decode (dec_lstm.py:270-367): every sentence starts at <s>, a step feeds the
        picked word back, a sentence stops after emitting </s>, at most 99 words."""
        batch_size = z.size(0)
        dev = z.device
        st = self._stepper(dev)
        z2 = z.reshape(batch_size, -1).float()
        with torch.no_grad():
            h, c = st.init_state(z2)
            tok = torch.full((batch_size,), self.vocab["<s>"], dtype=torch.int64, device=dev)
            end = self.vocab["</s>"]
            alive = torch.ones(batch_size, dtype=torch.bool, device=dev)
            picked, masks = [], []
            length_c = 1
            while length_c < 100:
                logits, h, c = st.step(tok, z2, h, c)
                tok = pick(st, logits)
                picked.append(tok)
                masks.append(alive)
                alive = alive & (tok != end)
                length_c += 1
                if not bool(alive.any().item()):            # the reference's mask.sum().item() != 0 test (one host read per step)
                    break
        ids = torch.stack(picked, dim=1).cpu().tolist()
        keep = torch.stack(masks, dim=1).cpu().tolist()
        return [[self.vocab.id2word(w) for w, k in zip(row, krow) if k] for row, krow in zip(ids, keep)]

    def greedy_decode(self, z):
        """Greedy decoding from z (batch_size, nz) -> list of word lists (reference dec_lstm.py:270-318)."""
        return self._roll_out(z, lambda st, logits: st.argmax(logits))

    def sample_decode(self, z, generator=None):
        """Ancestral sampling from z (reference dec_lstm.py:320-367).  The categorical draw is an inverse-CDF pick from a device
        uniform (torch.rand; `generator` makes it reproducible) instead of torch.multinomial's sampler: same distribution,
        different stream."""
        def pick(st, logits):
            u = torch.rand(logits.shape[0], device=logits.device, generator=generator)
            return st.sample(logits, u)
        return self._roll_out(z, pick)

    def beam_search_decode(self, z, K=5):
        """Beam search, sentence by sentence (reference dec_lstm.py:163-268): live hypotheses are expanded together, the K -
        len(completed) best continuations over (hypothesis, word) survive, a hypothesis completes when it emits </s>, the
        best by total log-probability is returned with <s> in front."""
        batch_size = z.size(0)
        dev = z.device
        st = self._stepper(dev)
        z2 = z.reshape(batch_size, -1).float()
        V = len(self.vocab)
        end = self.vocab["</s>"]
        decoded = []
        with torch.no_grad():
            h_init, c_init = st.init_state(z2)
            for idx in range(batch_size):
                # a hypothesis: (word ids so far, log-probability); states live in (h, c) rows aligned with `live`
                live = [([self.vocab["<s>"]], 0.0)]
                h, c = h_init[idx:idx + 1].clone(), c_init[idx:idx + 1].clone()
                completed = []
                t = 0
                while len(completed) < K and t < 100:
                    t += 1
                    n = len(live)
                    tok = torch.tensor([hyp[0][-1] for hyp in live], dtype=torch.int64, device=dev)
                    logits, h, c = st.step(tok, z2[idx:idx + 1].expand(n, -1), h, c)
                    prev = torch.tensor([hyp[1] for hyp in live], dtype=torch.float32, device=dev)
                    scores = st.log_softmax(logits, prev).reshape(-1)
                    log_prob, indexes = torch.topk(scores, K - len(completed))
                    live_ids = (indexes // V).tolist()
                    word_ids = (indexes % V).tolist()
                    new_live, keep_rows = [], []
                    for live_id, word_id, lp in zip(live_ids, word_ids, log_prob.tolist()):
                        hyp = (live[live_id][0] + [word_id], lp)
                        if word_id == end:
                            completed.append(hyp)
                        else:
                            new_live.append(hyp)
                            keep_rows.append(live_id)
                    live = new_live
                    if len(completed) == K or not live:
                        break
                    rows = torch.tensor(keep_rows, dtype=torch.int64, device=dev)
                    h, c = h.index_select(0, rows), c.index_select(0, rows)
                completed.extend(live)
                best = max(completed, key=lambda hyp: hyp[1]) if completed else ([self.vocab["<s>"]], 0.0)
                decoded.append([self.vocab.id2word(w) for w in best[0]])
        return decoded
