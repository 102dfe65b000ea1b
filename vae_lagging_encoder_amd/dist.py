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


def init_from_env(backend=None, timeout_s=None, force=False, banner=False):
    """Initialise torch.distributed from torchrun's env (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*).

    timeout_s (default LVAE_DIST_TIMEOUT or 180): the process group's collective timeout -- a rank that never arrives turns into
    an exception on the others instead of a wait without end (the default of 10 / 30 minutes outlives any launcher's patience).
    RCCL gets `device_id` (the communicator is bound to this rank's GPU and created eagerly: a bad device mapping fails HERE, with
    a message, not inside the first collective).  force: create the group for a world of one too (GradSync(force=True)).
    banner: one line per rank on stderr -- rank, device, visible devices, backend -- so that a first multi-GPU run explains
    itself."""
    import datetime
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("LVAE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if timeout_s is None:
            timeout_s = float(os.environ.get("LVAE_DIST_TIMEOUT", "180"))
        kw = {}
        if torch.cuda.is_available():
            # one process per GPU; LVAE_DIST_BACKEND=gloo lets several ranks share a device (single-GPU smoke tests)
            local = local % torch.cuda.device_count()
            torch.cuda.set_device(local)
            if backend == "nccl":
                kw["device_id"] = torch.device("cuda", local)
                # RCCL's own warnings (topology, transport fallbacks).  Its sink is STDOUT (version banner included, buffered until exit):
                # a caller that owns a one-line stdout contract points fd 1 elsewhere first (bench.protect_stdout); NCCL_DEBUG_FILE is
                # deliberately not set -- RCCL opens it with "w", and /dev/stderr of a redirected job is a file every rank would truncate
                os.environ.setdefault("NCCL_DEBUG", "WARN")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if banner:
            import sys
            print("vae_lagging_encoder_amd.dist: rank %d/%d local %d -> %s of %d visible device(s), backend %s, timeout %.0f s"
                  % (rank, world, local, ("cuda:%d" % local) if torch.cuda.is_available() else "cpu",
                     torch.cuda.device_count() if torch.cuda.is_available() else 0, backend, timeout_s), file=sys.stderr, flush=True)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s), **kw)
    return rank, local, world


class _HostStaged(object):
    """Handle of a collective on DEVICE tensors over gloo (ranks sharing a GPU: the functional-check layout of bench.py / the GPU
    tests, never the product's): the tensors travel as host copies made by us -- one D2H before, one H2D in wait() -- instead of
    through gloo's own CUDA path, whose chunked staging costs a device synchronisation per chunk; with four processes
    time-slicing one GPU that turned a 66 MB all-reduce into 13 s per step (profiles/r06b_four_ranks_on_one_gpu.txt)."""

    def __init__(self, work, back):
        self.work, self.back = work, back           # back: [(device tensor, host tensor)] copied at wait()

    def wait(self):
        self.work.wait()
        for dst, host in self.back:
            dst.copy_(host)
        return True


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

    def __init__(self, group=None, mode="strict", decoder="auto", payload="auto", force=False):
        assert mode in ("strict", "encoder_only")
        assert decoder in ("auto", "norm", "allreduce")
        assert payload in ("auto", "f32", "bf16")
        self.payload = payload
        self.profile = False
        self._prof = []           # per step: list of (name, start event, end event)
        self.group = group
        self.mode = mode
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # `active`: the exchange runs.  A world of ONE is normally a no-op; LVAE_DP_FORCE=1 (or force=True) runs every collective of
        # the schedule anyway -- a one-rank process group over RCCL on a one-GPU box executes the product's own calls (bf16
        # reduce-scatter, asynchronous all-reduces beside the engines' streams, the scalar exchanges) on the real backend, which
        # is all of RCCL a box with one device can reach (tests/test_rccl_single_rank.py, bench.py --force-dp)
        force = bool(force) or os.environ.get("LVAE_DP_FORCE", "") not in ("", "0")
        if force and not dist.is_initialized():
            raise RuntimeError("GradSync(force=True) needs an initialised process group (dist.init_from_env(force=True))")
        self.active = self.world > 1 or force
        # conservative schedule (LVAE_DP_CONSERVATIVE=1; a rung of bench.py's launch ladder): nothing is issued from inside the
        # backward, every collective is issued by sync() with the device idle before and after -- no kernel of ours is ever in
        # flight beside an RCCL kernel.  Slower (nothing overlaps), and the fallback if the overlapped schedule misbehaves on
        # a transport it has not met.
        self.conservative = os.environ.get("LVAE_DP_CONSERVATIVE", "") not in ("", "0")
        # gloo with device tensors (several ranks on one GPU): host-staged by us, see _HostStaged
        self._stage = dist.is_initialized() and dist.get_backend(group) == "gloo"
        self.decoder = decoder
        self._inv = None
        # per-slot state (slot = micro-batch slice of a step, trainer.micro_batches; slot 0 alone without gradient accumulation):
        # encoder buckets in flight, the decoder handles, the wire images and the reduce-scatter shard of that slice
        self._slots = {}
        self._slot = 0
        self._ss = None           # device scalar: sum of squares of the mean decoder gradient (decoder="norm")
        self._shard_sum = None
        # row-list exchange of the encoder's embedding gradient (prepare_rows): per-batch sorted token ids of EVERY rank, padded
        self.rows = None          # dict(cap, ni, V, n_emb, ids_all [n_batches][world][cap] int64, mine [n_batches][1][cap] int64)

    class _Slot(object):
        def __init__(self):
            self.buckets = []     # encoder buckets already in flight: (handle, wire tensor or None, lo, hi)
            self.h_dec = None
            self.h_rs = None      # reduce-scatter issued early by start_decoder()
            self.enc_rest = []
            self.b16 = {}         # bf16 wire images of the flat gradient buffers (payload "bf16")
            self.shard16 = None
            self.shard = None

    # ---- the three collectives of the schedule (RCCL: as they are; gloo + device tensors: host-staged) -------------------------
    def _c_all_reduce(self, t, async_op=False):
        if not (self._stage and t.is_cuda):
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        h = t.cpu()
        w = _HostStaged(dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group, async_op=True), [(t, h)])
        return w if async_op else w.wait() and None

    def _c_reduce_scatter(self, dst, src, async_op=False):
        if not (self._stage and src.is_cuda):
            return dist.reduce_scatter_tensor(dst, src, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        hs, hd = src.cpu(), torch.empty(dst.shape, dtype=dst.dtype)
        w = _HostStaged(dist.reduce_scatter_tensor(hd, hs, op=dist.ReduceOp.SUM, group=self.group, async_op=True), [(dst, hd)])
        return w if async_op else w.wait() and None

    def _c_all_gather(self, dst, src, async_op=False):
        if not (self._stage and src.is_cuda):
            return dist.all_gather_into_tensor(dst, src, group=self.group, async_op=async_op)
        hs, hd = src.cpu(), torch.empty(dst.shape, dtype=dst.dtype)
        w = _HostStaged(dist.all_gather_into_tensor(hd, hs, group=self.group, async_op=True), [(dst, hd)])
        return w if async_op else w.wait() and None

    @property
    def _cur(self):
        st = self._slots.get(self._slot)
        if st is None:
            st = self._slots[self._slot] = GradSync._Slot()
        return st

    def set_slot(self, i):
        """Select the micro-batch slice the next calls belong to (the trainer also switches the flat buffers' gradient slot)."""
        self._slot = int(i)

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
        stale = []
        for st in self._slots.values():
            stale += [b[0] for b in st.buckets + st.enc_rest] + [h[0] for h in (st.h_dec,) if h is not None] + [h for h in (st.h_rs,) if h is not None]
            st.buckets, st.enc_rest, st.h_dec, st.h_rs = [], [], None, None
        self._slot = 0
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
        b16 = self._cur.b16
        t = b16.get(id(flat))
        if t is None or t.numel() != n or t.device != src.device:
            t = torch.empty(n, dtype=torch.int16, device=src.device)
            b16[id(flat)] = t
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
            return (self._c_all_reduce(t16.view(torch.bfloat16), async_op=True), t16, lo, hi)
        return (self._c_all_reduce(flat.grad_padded[lo:hi], async_op=True), None, lo, hi)

    def _all_reduce_mean_finish(self, flat, started):
        h, t16, lo, hi = started
        h.wait()
        if isinstance(t16, tuple):            # the row-list bucket: rebuild the dense mean of the embedding gradient
            _, j, recv = t16
            R = self.rows
            lib = _eng.backend_for(flat.device)
            lib.lv_rows_merge_f32(P(R["ids_all"][j]), P(recv), 1 if self.payload == "bf16" else 0, self.world, R["cap"], R["ni"], R["V"],
                                  1.0 / self.world, P(flat.grad), _eng.stream_ptr(flat.device))
            return
        if t16 is not None:
            self._unwire(flat, t16, 1.0 / self.world, lo, hi)
        else:
            self._scale(flat, lo, hi)

    def start_encoder_bucket(self, enc_flat, lo, hi):
        """Issue the mean all-reduce of encoder-gradient elements [lo, hi) now (lo, hi multiples of 1024 elements, or hi = the
        padded end): called from inside the encoder's backward as soon as that part of the gradient is final, so that the
        collective runs under the backward work still queued behind it.  sync() completes it and exchanges what is left."""
        if not self.active or self.conservative:
            return
        pad = enc_flat.grad_padded.numel()
        hi = min(hi, pad)
        assert lo < hi
        if self.payload == "bf16":          # the wire conversion works on rows of 1024 elements
            assert lo % 1024 == 0 and (hi % 1024 == 0 or hi == pad), (lo, hi)
        assert all(hi <= b[2] or lo >= b[3] for b in self._cur.buckets), "encoder buckets must not overlap"
        self._cur.buckets.append(self._all_reduce_mean_start(enc_flat, lo, hi))

    # ---- row-list exchange of the embedding gradient ---------------------------------------------------------------------
    def prepare_rows(self, unique_ids, V, ni, n_emb, mode="auto"):
        """unique_ids: for every batch of THIS rank's pool, in pool order, the sorted token ids that occur in it (1-D int64 device
        tensors).  One-time exchange (all ranks call it with pools of the same length, and later step on the same pool index --
        the aggressive loop replicates its host draws, text.py:389): the longest list of any rank fixes the capacity, every
        rank's id lists are all-gathered and kept.  From then on start_encoder_rows(flat, j) replaces the dense all-reduce of
        the first n_emb elements of the encoder gradient (the embedding table leads the flat buffer) by an all-gather of
        `cap` gradient rows per rank (lv_rows_merge_f32 rebuilds the dense mean).  mode "auto": only when that is fewer bytes
        than the ring all-reduce (cap * world < 2 V: Zipf-distributed natural text, small worlds), else dense; "rows" forces it.
        Returns True when the row-list exchange is in use."""
        if not self.active or self.conservative:          # (conservative: the dense exchange inside sync() only)
            return False
        dev = unique_ids[0].device
        cap = torch.tensor([max(int(u.numel()) for u in unique_ids), len(unique_ids)], dtype=torch.int64)
        if dist.get_backend(self.group) == "nccl":
            cap = cap.to(dev)
        dist.all_reduce(cap, op=dist.ReduceOp.MAX, group=self.group)
        cap, nb = int(cap[0]), int(cap[1])
        if nb != len(unique_ids):
            raise _eng._lib.LvaeError("prepare_rows: the ranks' pools differ in length (%d here, %d elsewhere)" % (len(unique_ids), nb))
        cap = (cap + 7) // 8 * 8
        use = mode == "rows" or (mode == "auto" and cap * self.world < 2 * V)
        if not use:
            self.rows = None
            return False
        mine = torch.full((nb, 1, cap), -1, dtype=torch.int64, device=dev)
        for j, u in enumerate(unique_ids):
            mine[j, 0, :u.numel()] = u
        ids_all = torch.empty(self.world, nb, cap, dtype=torch.int64, device=dev)
        self._c_all_gather(ids_all.view(-1), mine.view(-1))
        self.rows = dict(cap=cap, ni=int(ni), V=int(V), n_emb=int(n_emb), mine=mine,
                         ids_all=ids_all.permute(1, 0, 2).contiguous())      # [batch][rank][cap]
        return True

    def start_encoder_rows(self, enc_flat, j):
        """Issue the row-list exchange of the embedding gradient of pool batch j (the dense gradient is complete in enc_flat.grad[0 :
        n_emb]): gather this rank's `cap` rows into the wire buffer (f32, or bf16 under the bf16 payload), all-gather them.
        sync_collect() rebuilds the dense mean in place."""
        R = self.rows
        if os.environ.get("LVAE_DP_DEBUG"):
            # the gathered rows are paired with id lists by POOL INDEX: every rank must be on the same j (the aggressive loop's
            # replicated host draws give that; a rank-specific seed or skip would silently mis-pair rows).  Debug mode: one small
            # host-synchronising all-gather per step.
            js = [None] * self.world
            dist.all_gather_object(js, int(j), group=self.group)
            if any(v != js[0] for v in js):
                raise _eng._lib.LvaeError("row-list exchange: the ranks are on different pool batches this step (%r); their host "
                                          "random streams have diverged" % (js,))
        lib, s = _eng.backend_for(enc_flat.device), _eng.stream_ptr(enc_flat.device)
        st = self._cur
        cap, ni, V = R["cap"], R["ni"], R["V"]
        bf = self.payload == "bf16"
        key = ("rows", bf)
        buf = st.b16.get(key)
        if buf is None:
            dt = torch.int16 if bf else torch.float32
            buf = (torch.empty(cap, ni, dtype=dt, device=enc_flat.device), torch.empty(self.world, cap, ni, dtype=dt, device=enc_flat.device))
            st.b16[key] = buf
        send, recv = buf
        ids = R["mine"][j]                                   # [1][cap]: batch-first ids of the gather kernels (B = 1, T = cap)
        if bf:
            lib.lv_embed_gather_b16(P(enc_flat.grad), P(ids), cap, None, 1.0, cap, 1, ni, V, P(send), ni, None, 0, s)
            h = self._c_all_gather(recv.view(torch.bfloat16).view(-1), send.view(torch.bfloat16).view(-1), async_op=True)
        else:
            lib.lv_embed_gather_f32(P(enc_flat.grad), P(ids), cap, None, 1.0, P(send), cap, 1, ni, V, s)
            h = self._c_all_gather(recv.view(-1), send.view(-1), async_op=True)
        align = 1024 if bf else 4
        st.buckets.append((h, ("rows", j, recv), 0, R["n_emb"] // align * align))

    def _norm_only(self, dec_flat, update):
        return (self.mode == "strict" and update == "encoder" and self.decoder in ("auto", "norm")
                and dec_flat.grad_padded.numel() % self.world == 0)

    def start_decoder(self, dec_flat, update="encoder"):
        """Issue the decoder-gradient exchange (strict mode) once the decoder's backward has been queued: RCCL runs it on its
        own stream, behind everything queued on the compute stream so far and beside what is queued afterwards (the
        encoder's backward, or its weight-gradient GEMMs when the BPTT is a persistent launch).  Completed by sync()."""
        if not self.active or self.mode != "strict" or self.conservative:
            return
        if self._norm_only(dec_flat, update):
            self._cur.h_rs = self._reduce_scatter(dec_flat, update, async_op=True)
            return
        self._cur.h_dec = self._all_reduce_mean_start(dec_flat)

    def _reduce_scatter(self, dec_flat, update, async_op):
        """Reduce-scatter of the (padded) decoder gradient into this rank's shard; returns a waitable handle or None."""
        src = dec_flat.grad_padded
        n = src.numel() // self.world
        self.ss_handle(dec_flat, update)
        st = self._cur
        if st.shard is None or st.shard.numel() != n or st.shard.device != src.device:
            st.shard = torch.empty(n, dtype=torch.float32, device=src.device)
        bf = self.payload == "bf16"
        if bf:
            src = self._wire(dec_flat).view(torch.bfloat16)
            if st.shard16 is None or st.shard16.numel() != n or st.shard16.device != src.device:
                st.shard16 = torch.empty(n, dtype=torch.int16, device=src.device)
        # the same call on RCCL and on gloo (torch >= 2.10's gloo has reduce_scatter_tensor, bf16 included): the CPU tests
        # exercise the product's collective, not a substitute
        dst = st.shard16.view(torch.bfloat16) if bf else st.shard
        return self._c_reduce_scatter(dst, src, async_op=async_op)

    def sync(self, enc_flat, dec_flat, update="encoder"):
        """Exchange the gradients of one step.  Returns None when both buffers now hold the global mean gradient, or a
        device scalar with the sum of squares of the mean decoder gradient when only that was exchanged (the trainer adds
        it to the encoder's sum of squares for the clip coefficient)."""
        if not self.active:
            return None
        quiesce = self.conservative and enc_flat.device.type == "cuda"
        if quiesce:
            torch.cuda.synchronize(enc_flat.device)
        self.sync_issue(enc_flat, dec_flat, update)
        self.sync_collect(enc_flat, dec_flat, update)
        ss = self.sync_norm(dec_flat, update, (self._slot,))
        if quiesce:
            torch.cuda.synchronize(enc_flat.device)
        return ss

    def sync_issue(self, enc_flat, dec_flat, update="encoder"):
        """First half of sync() for the current slot: every collective of this slice's gradients that is not in flight yet is
        issued (asynchronously); nothing is waited for.  With micro-batches the trainer calls this after each slice's backward
        and goes on to the next slice -- the exchange runs underneath it (weights are frozen within a step)."""
        if not self.active:
            return
        st = self._cur
        dev = enc_flat.device
        if self._norm_only(dec_flat, update):
            if st.h_rs is None:
                st.h_rs = self._reduce_scatter(dec_flat, update, async_op=True)      # not started early (hipGraph split, direct callers)
        elif self.mode == "strict" and st.h_dec is None:
            st.h_dec = self._all_reduce_mean_start(dec_flat)
        with self._Phase(self, "encoder_allreduce_issue", dev):
            st.enc_rest = self._start_encoder_rest(enc_flat)

    def sync_collect(self, enc_flat, dec_flat, update="encoder"):
        """Second half for the current slot (the flat buffers must have the same gradient slot selected): wait for this slice's
        collectives and unpack them -- the encoder (and, where it was all-reduced, the decoder) buffer then holds the mean over
        ranks of this slice's gradient, the reduce-scatter shard the SUM over ranks of this rank's share of the decoder's."""
        if not self.active:
            return
        st = self._cur
        lib, s = _eng.backend_for(enc_flat.device), _eng.stream_ptr(enc_flat.device)
        dev = enc_flat.device
        h_rs, st.h_rs = st.h_rs, None
        h_dec, st.h_dec = st.h_dec, None
        if h_rs is not None:
            n = dec_flat.grad_padded.numel() // self.world
            with self._Phase(self, "decoder_reduce_scatter_wait", dev):
                h_rs.wait()
                if self.payload == "bf16":
                    lib.lv_cvt_f32_bf16_scaled(P(st.shard16), n, 1.0, P(st.shard), s)
        if h_dec is not None:
            with self._Phase(self, "decoder_allreduce_wait", dev):
                self._all_reduce_mean_finish(dec_flat, h_dec)
        rest, st.enc_rest = st.enc_rest, []
        with self._Phase(self, "encoder_allreduce_wait", dev):
            for h in rest:
                self._all_reduce_mean_finish(enc_flat, h)

    def sync_norm(self, dec_flat, update, slots):
        """Norm-only decoder exchange: sum of squares of the MEAN decoder gradient from the reduce-scatter shards of the listed
        slots (their sum = this rank's share of the gradient summed over ranks and micro-batches) + one scalar all-reduce.
        Returns the device scalar, or None when the decoder gradient was all-reduced in full (or not exchanged)."""
        if not self.active or not self._norm_only(dec_flat, update):
            return None
        lib, s = _eng.backend_for(dec_flat.device), _eng.stream_ptr(dec_flat.device)
        dev = dec_flat.device
        n = dec_flat.grad_padded.numel() // self.world
        shards = [self._slots[i].shard for i in slots]
        with self._Phase(self, "decoder_reduce_scatter_wait", dev):
            total = shards[0]
            if len(shards) > 1:
                if self._shard_sum is None or self._shard_sum.numel() != n or self._shard_sum.device != dev:
                    self._shard_sum = torch.empty(n, dtype=torch.float32, device=dev)
                total = self._shard_sum
                lib.lv_add_f32(P(shards[0]), P(shards[1]), P(total), n, s)
                for sh in shards[2:]:
                    lib.lv_add_f32(P(total), P(sh), P(total), n, s)
            lib.lv_sumsq_f32(P(total), n, P(self._ws), P(self._ss), 0, s)
            lib.lv_scale_f32(P(self._ss), 1, P(self._inv2), s)          # shard of the SUM -> shard of the mean
        with self._Phase(self, "scalar_allreduce", dev):
            self._c_all_reduce(self._ss)
        return self._ss

    def _start_encoder_rest(self, enc_flat):
        """Issue the all-reduce of every encoder range no bucket has covered; returns all encoder handles in flight, lowest
        range first (finished by sync in that order)."""
        st = self._cur
        started, st.buckets = sorted(st.buckets, key=lambda b: b[2]), []
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
        if not self.active or not self._norm_only(dec_flat, update):
            return None
        src = dec_flat.grad_padded
        if self._ss is None or self._ss.device != src.device:
            lib = _eng.backend_for(src.device)
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

    def bytes_on_wire(self, enc_flat, dec_flat, update="encoder", n_emb=0, micro_batches=1):
        """Bytes this rank SENDS per step, per phase of the exchange (ring algorithms, as bytes_per_step), for the bench's
        dp_breakdown: the embedding bucket issued from inside the encoder backward (n_emb elements, 0 = no bucket), the rest of the
        encoder buffer, the decoder's reduce-scatter or all-reduce, the scalar all-reduce; times the micro-batch count."""
        f = (self.world - 1) / max(1, self.world)
        b = 2 if self.payload == "bf16" else 4
        pad_e, pad_d = enc_flat.grad_padded.numel(), dec_flat.grad_padded.numel()
        out = {"encoder_bucket_embedding": (int(2 * f * b * n_emb) if self.rows is None else int(f * self.world * b * self.rows["cap"] * self.rows["ni"])) * micro_batches,
               "embedding_exchange": "dense all-reduce" if self.rows is None else "row list (all-gather of %d rows per rank)" % self.rows["cap"],
               "encoder_rest": int(2 * f * b * (pad_e - n_emb)) * micro_batches,
               "decoder_reduce_scatter": 0, "decoder_allreduce": 0, "scalar_allreduce": 0}
        if self.mode == "strict":
            if self._norm_only(dec_flat, update):
                out["decoder_reduce_scatter"] = int(f * b * pad_d) * micro_batches
                out["scalar_allreduce"] = int(2 * f * 4)
            else:
                out["decoder_allreduce"] = int(2 * f * b * pad_d) * micro_batches
        out["total"] = sum(v for v in out.values() if not isinstance(v, str))
        return out

    def window_mean(self, loss_sum, num_words):
        """Global mean loss per word of one exit window (text.py:393-396): sum of the ranks' loss sums over the sum of
        their word counts, identical on every rank, so all ranks take the same data-dependent `break`."""
        if not self.active:
            return loss_sum / num_words
        t = torch.tensor([loss_sum, float(num_words)], dtype=torch.float64)
        if dist.get_backend(self.group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        v = t.tolist()
        return v[0] / v[1]
