"""LSTMEncoder -- drop-in for the reference's modules/encoders/enc_lstm.py:11-64 on MI355X.

Same constructor signature, same nn.Module containers in the same construction order (so seeded initialisation
and state_dict keys `embed.weight, lstm.{weight,bias}_{ih,hh}_l0, linear.weight` are interchangeable with the
reference's checkpoints), but `forward` runs the hand-written HIP path (embedding gather -> f32 MFMA input
projection -> per-timestep fused LSTM kernels -> head GEMM) through engine.LSTMEncoderEngine, with a
hand-written backward behind torch.autograd.Function.
"""
import torch
import torch.nn as nn

from ... import engine as _eng
from .encoder import GaussianEncoderBase


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, x, *params):
        mulv = eng.forward(x)
        ctx.eng = eng
        ctx.gen = eng.gen
        return mulv.clone()

    @staticmethod
    def backward(ctx, dmulv):
        eng = ctx.eng
        eng.flat.before_autograd_backward()
        eng.backward(dmulv, ctx.gen)
        return (None, None) + eng.flat.deliver_grads()


class LSTMEncoder(GaussianEncoderBase):
    """Gaussian LSTM encoder with constant-length batching (reference enc_lstm.py:11-64)."""

    def __init__(self, args, vocab_size, model_init, emb_init):
        super(LSTMEncoder, self).__init__()
        self.ni = args.ni
        self.nh = args.enc_nh
        self.nz = args.nz

        self.embed = nn.Embedding(vocab_size, args.ni)
        self.lstm = nn.LSTM(input_size=args.ni, hidden_size=args.enc_nh, num_layers=1, batch_first=True, dropout=0)
        # mean and logvar head
        self.linear = nn.Linear(args.enc_nh, 2 * args.nz, bias=False)

        self.reset_parameters(model_init, emb_init)
        self._hip = _eng.LSTMEncoderEngine(self)

    def reset_parameters(self, model_init, emb_init):
        # every parameter (LSTM biases included), then the embedding again
        for param in self.parameters():
            model_init(param)
        emb_init(self.embed.weight)

    def _params(self):
        return (self.embed.weight, self.lstm.weight_ih_l0, self.lstm.weight_hh_l0, self.lstm.bias_ih_l0,
                self.lstm.bias_hh_l0, self.linear.weight)

    def _forward_mulv(self, input):
        self._hip.ensure(input.device)
        return _EncoderFn.apply(self._hip, input, *self._params())

    def forward(self, input):
        """input (batch, seq_len) int64 -> mean (batch, nz), logvar (batch, nz).
        The whole of `input` is embedded, <s> and </s> included."""
        mulv = self._forward_mulv(input)
        return mulv[:, :self.nz], mulv[:, self.nz:]
