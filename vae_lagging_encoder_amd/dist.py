"""Data-parallel sharding of the aggressive inner loop: one process per GPU, RCCL over xGMI.

The reference has no distributed code; this is new design.  Every rank holds a full replica and
takes its own rows of the global batch; because the objective is mean_b(loss_b), the global gradient is the mean
of the per-rank gradients.  Per inner step there is exactly one exchange: a sum all-reduce of the flat gradient
buffers followed by a 1/P scale (the clip norm of text.py:385 spans encoder AND decoder grads -- SURVEY.md G1 --
so in `strict` mode both flat buffers are reduced; `encoder_only` reduces the 66 MB encoder buffer and uses the
local decoder-grad norm, a documented deviation).  The decoder buffer is reduced on a side stream so that it
overlaps the encoder BPTT that is still running on the compute stream.

Backend "nccl" IS RCCL on ROCm.  The CPU tests run the same class over gloo with world_size 2.
"""
import os

import torch
import torch.distributed as dist

from . import engine as _eng
from .engine import P


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's env (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("LVAE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            # one process per GPU; LVAE_DIST_BACKEND=gloo lets several ranks share a device (single-GPU smoke tests)
            local = local % torch.cuda.device_count()
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


class GradSync(object):
    """Mean all-reduce of the flat gradient buffers of (encoder, decoder)."""

    def __init__(self, group=None, mode="strict"):
        assert mode in ("strict", "encoder_only")
        self.group = group
        self.mode = mode
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._inv = None

    def _scale(self, flat):
        lib = _eng.backend_for(flat.device)
        if self._inv is None or self._inv.device != flat.device:
            self._inv = torch.full((1,), 1.0 / self.world, dtype=torch.float32, device=flat.device)
        lib.lv_scale_f32(P(flat.grad), flat.numel, P(self._inv), _eng.stream_ptr(flat.device))

    def start_decoder(self, dec_flat):
        """Issue the decoder-gradient all-reduce as soon as the decoder's backward has been queued: RCCL runs it on
        its own stream underneath the encoder's BPTT (strict mode only).  Completed by sync()."""
        if self.world == 1 or self.mode != "strict":
            return
        self._h_dec = dist.all_reduce(dec_flat.grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def sync(self, enc_flat, dec_flat):
        if self.world == 1:
            return
        h_dec = getattr(self, "_h_dec", None)
        self._h_dec = None
        if self.mode == "strict" and h_dec is None:
            h_dec = dist.all_reduce(dec_flat.grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        h_enc = dist.all_reduce(enc_flat.grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if h_dec is not None:
            h_dec.wait()
            self._scale(dec_flat)
        h_enc.wait()
        self._scale(enc_flat)

    def window_mean(self, loss_sum, num_words):
        """Global mean loss per word of one exit window (text.py:393-396): sum of the ranks' loss sums over the sum of
        their word counts, identical on every rank, so all ranks take the same data-dependent `break`."""
        if self.world == 1:
            return loss_sum / num_words
        t = torch.tensor([loss_sum, float(num_words)], dtype=torch.float64)
        if dist.get_backend(self.group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        v = t.tolist()
        return v[0] / v[1]
