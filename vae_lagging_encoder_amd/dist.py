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
    """Gradient exchange of the data-parallel inner loop: mean over ranks of the flat gradient buffers of (encoder, decoder).

    mode "strict": the clip norm spans encoder AND decoder gradients (text.py:385), so both enter the exchange.
      decoder="allreduce": both flat buffers are mean-all-reduced (66 MB + 149 MB fp32 at the Yahoo shape).
      decoder="norm" (default where the backend has reduce-scatter): in an ENCODER-ONLY step (text.py:387) the decoder
        gradient is needed for one number only -- its contribution to the clip norm; the encoder-only step discards it.
        The decoder buffer is then reduce-scattered (each rank receives 1/P of the summed vector: half the bytes of an
        all-reduce), every rank takes the sum of squares of its shard of the MEAN gradient, and a scalar all-reduce adds
        them up: exactly the reference's norm of the global mean gradient, 149 MB -> 75 MB of decoder traffic per step.  The
        decoder's .grad buffers keep the LOCAL gradient (nobody reads them before the next zero_grad).  Steps that update
        the decoder (text.py:418-424) all-reduce it in full.
    mode "encoder_only": only the encoder buffer is exchanged and the local decoder-gradient norm enters the clip norm -- a
      documented deviation from the reference (replicas stay identical, the clip coefficient differs slightly per step).
    payload "f32" exchanges the fp32 gradients; "bf16" rounds them to bf16 for the wire (RNE, lv_cvt_bf16_f32), sums in bf16
      and unpacks with the 1/world mean folded in: half the bytes of every exchange, at a per-element relative error of 2^-9
      on the mean gradient -- the size of the operand rounding the bf16 throughput configuration already accepts inside its
      GEMMs.  "auto" (default) follows the trainer's arithmetic: bf16 wire for the bf16 configuration, exact fp32 for the
      fp32 parity path (`resolve_payload`, called by the trainer).  Replicas stay bit-identical to each other either way.

    The encoder exchange -- the one the update has to wait for -- can be issued in BUCKETS as the encoder's backward produces
    them (`start_encoder_bucket`: the 41 MB embedding gradient right after the embedding scatter, under the LSTM
    weight-gradient GEMMs; the rest is picked up by `sync`).  `profile = True` brackets the phases of `sync` with HIP events
    on the compute stream (`breakdown()`): what a step spends WAITING for each collective, i.e. the exposed communication.
    """

    def __init__(self, group=None, mode="strict", decoder="auto", payload="auto"):
        assert mode in ("strict", "encoder_only")
        assert decoder in ("auto", "norm", "allreduce")
        assert payload in ("auto", "f32", "bf16")
        self.payload = payload
        self._buckets = []        # encoder buckets already in flight: (lo, hi, handle, wire tensor or None)
        self.profile = False
        self._prof = []           # per step: list of (name, start event, end event)
        self._b16 = {}            # bf16 wire images of the flat gradient buffers (payload "bf16")
        self.group = group
        self.mode = mode
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.decoder = decoder
        self._inv = None
        self._h_dec = None
        self._h_rs = None         # reduce-scatter issued early by start_decoder()
        self._shard16 = None
        self._shard = None
        self._ss = None           # device scalar: sum of squares of the mean decoder gradient (decoder="norm")

    def resolve_payload(self, precision):
        """payload "auto" -> the wire format that matches the trainer's arithmetic (called once by the trainer)."""
        if self.payload == "auto":
            self.payload = "bf16" if precision == "bf16" else "f32"
        return self.payload

    def _scale(self, flat, lo=0, hi=None):
        lib = _eng.backend_for(flat.device)
        if self._inv is None or self._inv.device != flat.device:
            self._inv = torch.full((1,), 1.0 / self.world, dtype=torch.float32, device=flat.device)
        hi = flat.grad_padded.numel() if hi is None else min(hi, flat.grad_padded.numel())
        if hi <= lo:
            return
        # over the padded buffer: the guard element in the tail padding (trainer's transaction gate) only has to stay non-zero
        lib.lv_scale_f32(P(flat.grad_padded, lo), hi - lo, P(self._inv), _eng.stream_ptr(flat.device))

    def begin_step(self):
        """Called by the trainer before a step queues anything: collectives a previous step left in flight -- an exception between
        the encoder backward and sync() -- are waited for and dropped, so that this step's buckets start from a clean slate on
        every rank (a stale handle would otherwise trip the overlap check of start_encoder_bucket for good)."""
        stale = [b[0] for b in self._buckets] + [h[0] for h in (self._h_dec,) if h is not None] + [h for h in (self._h_rs,) if h is not None]
        self._buckets, self._h_dec, self._h_rs = [], None, None
        for h in stale:
            h.wait()

    # ---- measurement ------------------------------------------------------------------------------------------------
    class _Phase(object):
        def __init__(self, gs, name, device):
            self.on = gs.profile and torch.device(device).type == "cuda"
            self.gs, self.name = gs, name

        def __enter__(self):
            if self.on:
                self.e0 = torch.cuda.Event(enable_timing=True)
                self.e1 = torch.cuda.Event(enable_timing=True)
                self.e0.record()

        def __exit__(self, *a):
            if self.on:
                self.e1.record()
                self.gs._prof.append((self.name, self.e0, self.e1))

    def breakdown(self, reset=True):
        """Mean milliseconds per step the COMPUTE stream spent inside each phase of the exchange since the last call
        (profile = True; call after a device synchronise): {phase: ms, ..., "steps": n}.  A phase's time is what the step
        waited for that collective (plus its pack / unpack kernels) -- communication hidden under compute does not show."""
        out, n = {}, {}
        for name, e0, e1 in self._prof:
            out[name] = out.get(name, 0.0) + e0.elapsed_time(e1)
            n[name] = n.get(name, 0) + 1
        steps = max(n.values()) if n else 0
        res = {k: v / max(1, steps) for k, v in out.items()}
        res["steps"] = steps
        if reset:
            self._prof = []
        return res

    # ---- bf16 wire format -------------------------------------------------------------------------------------------
    def _wire(self, flat, lo=0, hi=None):
        """bf16 image of elements [lo, hi) of the (padded) flat gradient buffer (lo, hi multiples of 1024; default: all of it);
        returns the int16 view of that range (viewed as bfloat16 for the collective)."""
        src = flat.grad_padded
        n = src.numel()
        hi = n if hi is None else hi
        t = self._b16.get(id(flat))
        if t is None or t.numel() != n or t.device != src.device:
            t = torch.empty(n, dtype=torch.int16, device=src.device)
            self._b16[id(flat)] = t
        lib = _eng.backend_for(src.device)
        lib.lv_cvt_bf16_f32(P(src, lo), 1024, (hi - lo) // 1024, 1024, P(t, lo), 1024, None, 0, _eng.stream_ptr(src.device))
        return t[lo:hi]

    def _unwire(self, flat, t16, scale, lo=0, hi=None):
        lib = _eng.backend_for(t16.device)
        pad = flat.grad_padded.numel()
        hi = pad if hi is None else min(hi, pad)
        if hi <= lo:
            return
        # into the padded buffer (the tail padding carries the transaction guard element)
        lib.lv_cvt_f32_bf16_scaled(P(t16), hi - lo, scale, P(flat.grad_padded, lo), _eng.stream_ptr(t16.device))

    def _all_reduce_mean_start(self, flat, lo=0, hi=None):
        """Start the mean all-reduce of (a 1024-aligned range of) a flat gradient buffer; returns what
        _all_reduce_mean_finish needs."""
        hi = flat.grad_padded.numel() if hi is None else hi
        if self.payload == "bf16":
            t16 = self._wire(flat, lo, hi)
            return (dist.all_reduce(t16.view(torch.bfloat16), op=dist.ReduceOp.SUM, group=self.group, async_op=True), t16, lo, hi)
        return (dist.all_reduce(flat.grad_padded[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True), None, lo, hi)

    def _all_reduce_mean_finish(self, flat, started):
        h, t16, lo, hi = started
        h.wait()
        if t16 is not None:
            self._unwire(flat, t16, 1.0 / self.world, lo, hi)
        else:
            self._scale(flat, lo, hi)

    def start_encoder_bucket(self, enc_flat, lo, hi):
        """Issue the mean all-reduce of encoder-gradient elements [lo, hi) now (lo, hi multiples of 1024 elements, or hi = the
        padded end): called from inside the encoder's backward as soon as that part of the gradient is final, so that the
        collective runs under the backward work still queued behind it.  sync() completes it and exchanges what is left."""
        if self.world == 1:
            return
        pad = enc_flat.grad_padded.numel()
        hi = min(hi, pad)
        assert lo < hi
        if self.payload == "bf16":          # the wire conversion works on rows of 1024 elements
            assert lo % 1024 == 0 and (hi % 1024 == 0 or hi == pad), (lo, hi)
        assert all(hi <= b[2] or lo >= b[3] for b in self._buckets), "encoder buckets must not overlap"
        self._buckets.append(self._all_reduce_mean_start(enc_flat, lo, hi))

    def _norm_only(self, dec_flat, update):
        return (self.mode == "strict" and update == "encoder" and self.decoder in ("auto", "norm")
                and dec_flat.grad_padded.numel() % self.world == 0)

    def start_decoder(self, dec_flat, update="encoder"):
        """Issue the decoder-gradient exchange (strict mode) once the decoder's backward has been queued: RCCL runs it on its
        own stream, behind everything queued on the compute stream so far and beside what is queued afterwards (the
        encoder's backward, or its weight-gradient GEMMs when the BPTT is a persistent launch).  Completed by sync()."""
        if self.world == 1 or self.mode != "strict":
            return
        if self._norm_only(dec_flat, update):
            self._h_rs = self._reduce_scatter(dec_flat, update, async_op=True)
            return
        self._h_dec = self._all_reduce_mean_start(dec_flat)

    def _reduce_scatter(self, dec_flat, update, async_op):
        """Reduce-scatter of the (padded) decoder gradient into this rank's shard; returns a waitable handle or None."""
        src = dec_flat.grad_padded
        n = src.numel() // self.world
        self.ss_handle(dec_flat, update)
        bf = self.payload == "bf16"
        if bf:
            src = self._wire(dec_flat).view(torch.bfloat16)
            if self._shard16 is None or self._shard16.numel() != n or self._shard16.device != src.device:
                self._shard16 = torch.empty(n, dtype=torch.int16, device=src.device)
        # the same call on RCCL and on gloo (torch >= 2.10's gloo has reduce_scatter_tensor, bf16 included): the CPU tests
        # exercise the product's collective, not a substitute
        dst = self._shard16.view(torch.bfloat16) if bf else self._shard
        return dist.reduce_scatter_tensor(dst, src, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def sync(self, enc_flat, dec_flat, update="encoder"):
        """Exchange the gradients of one step.  Returns None when both buffers now hold the global mean gradient, or a
        device scalar with the sum of squares of the mean decoder gradient when only that was exchanged (the trainer adds
        it to the encoder's sum of squares for the clip coefficient)."""
        if self.world == 1:
            return None
        h_dec, self._h_dec = self._h_dec, None
        lib, s = _eng.backend_for(enc_flat.device), _eng.stream_ptr(enc_flat.device)
        ss = None
        dev = enc_flat.device
        if self._norm_only(dec_flat, update):
            n = dec_flat.grad_padded.numel() // self.world
            h_rs, self._h_rs = self._h_rs, None
            if h_rs is None:
                h_rs = self._reduce_scatter(dec_flat, update, async_op=True)      # not started early (hipGraph split, direct callers)
            with self._Phase(self, "encoder_allreduce_issue", dev):
                enc_rest = self._start_encoder_rest(enc_flat)
            with self._Phase(self, "decoder_reduce_scatter_wait", dev):
                h_rs.wait()
                if self.payload == "bf16":
                    lib.lv_cvt_f32_bf16_scaled(P(self._shard16), n, 1.0, P(self._shard), s)
                lib.lv_sumsq_f32(P(self._shard), n, P(self._ws), P(self._ss), 0, s)
                lib.lv_scale_f32(P(self._ss), 1, P(self._inv2), s)          # shard of the SUM -> shard of the mean
            with self._Phase(self, "scalar_allreduce", dev):
                dist.all_reduce(self._ss, op=dist.ReduceOp.SUM, group=self.group)
            ss = self._ss
        else:
            if self.mode == "strict" and h_dec is None:
                h_dec = self._all_reduce_mean_start(dec_flat)
            with self._Phase(self, "encoder_allreduce_issue", dev):
                enc_rest = self._start_encoder_rest(enc_flat)
            if h_dec is not None:
                with self._Phase(self, "decoder_allreduce_wait", dev):
                    self._all_reduce_mean_finish(dec_flat, h_dec)
        with self._Phase(self, "encoder_allreduce_wait", dev):
            for st in enc_rest:
                self._all_reduce_mean_finish(enc_flat, st)
        return ss

    def _start_encoder_rest(self, enc_flat):
        """Issue the all-reduce of every encoder range no bucket has covered; returns all encoder handles in flight, lowest
        range first (finished by sync in that order)."""
        started, self._buckets = sorted(self._buckets, key=lambda b: b[2]), []
        pad = enc_flat.grad_padded.numel()
        gaps, pos = [], 0
        for b in started:
            if b[2] > pos:
                gaps.append((pos, b[2]))
            pos = b[3]
        if pos < pad:
            gaps.append((pos, pad))
        return started + [self._all_reduce_mean_start(enc_flat, a, b) for a, b in gaps]

    def ss_handle(self, dec_flat, update="encoder"):
        """The device scalar sync() will return for this kind of step (None when it returns None); no communication."""
        if self.world == 1 or not self._norm_only(dec_flat, update):
            return None
        src = dec_flat.grad_padded
        n = src.numel() // self.world
        if self._shard is None or self._shard.numel() != n or self._shard.device != src.device:
            lib = _eng.backend_for(src.device)
            self._shard = torch.empty(n, dtype=torch.float32, device=src.device)
            self._ss = torch.zeros(1, dtype=torch.float32, device=src.device)
            self._ws = torch.empty(lib.lv_sumsq_workspace_floats(), dtype=torch.float32, device=src.device)
            self._inv2 = torch.full((1,), 1.0 / (self.world * self.world), dtype=torch.float32, device=src.device)
        return self._ss

    def bytes_per_step(self, enc_flat, dec_flat, update="encoder"):
        """fp32 bytes each rank sends per step (ring algorithms: 2(P-1)/P x buffer for an all-reduce, (P-1)/P for a
        reduce-scatter) -- for the schedule table in DESIGN.md."""
        f = (self.world - 1) / max(1, self.world) * (0.5 if self.payload == "bf16" else 1.0)      # "auto" counts as fp32 until resolved
        enc = 2 * f * 4 * enc_flat.numel
        if self.mode != "strict":
            return enc
        return enc + (f if self._norm_only(dec_flat, update) else 2 * f) * 4 * dec_flat.numel

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
