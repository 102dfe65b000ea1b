"""Fused driver for the aggressive inference-network loop (reference text.py:366-424).

`AggressiveTextTrainer.step()` is one body of the inner loop -- zero_grad, VAE.loss, loss.mean().backward(),
clip_grad_norm_(all params, 5.0), encoder SGD step (text.py:373-387) -- as one stream-ordered sequence of C-ABI
kernel calls with NO host synchronisation and no autograd: scalars (kl weight, lr, norm, clip coefficient, the
running loss sum that text.py:381 pulls to the host every iteration) stay in device memory, so the sequence can be
captured once per (B, T) bucket into a hipGraph and replayed (`use_graph=True`).
`inner_loop()` reproduces the data-dependent exit of text.py:366-400 with one host read every 15 iterations
instead of one per iteration.  Data parallelism (one process per GPU) plugs in through `dist.GradSync`.
"""
import collections
import ctypes

import numpy as np
import contextlib
import gc
import torch

from . import engine as _eng
from .engine import P



@contextlib.contextmanager
def _capture_graph(graph, stream):
    """`torch.cuda.graph(graph, stream=stream)` with Python's cyclic garbage collector held off for the duration of the capture.
    The collector runs whenever allocation counts say so -- also in the middle of a capture -- and whatever it frees then runs its
    destructor there: a `torch.cuda.CUDAGraph` of an earlier trainer (they sit in reference cycles with the trainer that replays
    them) synchronises the device in `~CUDAGraph`, which HIP refuses while a stream is capturing ("operation not permitted when
    stream is capturing"), and an exception in a destructor is `std::terminate`: the process aborts.  Seen on the GPU suite once the
    test count moved the collector's schedule (profiles/r06x4_*); `torch.cuda.graph.__enter__` itself collects BEFORE the capture
    begins, so everything unreachable by then is gone in an orderly way."""
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, stream=stream):
            yield
    finally:
        if was_enabled:
            gc.enable()

class _Static(object):
    pass


# eager mode: how many (B, T) shapes keep their per-shape step buffers (inputs, masks, per-row scalars; <= 10 MB each at the Yahoo
# dims: 5 GB at most).  A corpus bucketed by sentence length (data/text_data.py:219-255) cycles through one shape per length plus
# the tails: with 64 slots bench.py's 71-shape mixed pool rebuilt a slot -- one allocator request -- on every step (round 5).
STATIC_SHAPES = 512


def _rank_seed(seed, grad_sync):
    """Philox key for this process: `seed` itself on a single GPU, a rank-specific key under data parallelism."""
    rank = grad_sync.rank if grad_sync is not None else 0
    return (int(seed) + 0x9E3779B1 * rank) & 0x7FFFFFFFFFFFFFFF


def _log_demotion(rung):
    import sys
    print("vae_lagging_encoder_amd: a persistent LSTM launch reported a hand-off timeout; the voided steps are replayed on ladder rung "
          "%d (%s)" % (rung, _eng.PERSIST_RUNGS[rung]), file=sys.stderr)


class AggressiveTextTrainer(object):
    # data parallel: the embedding gradient goes out as its own all-reduce bucket when it has at least this many elements
    # (below ~4 MB a second collective costs more in latency than its overlap buys)
    BUCKET_MIN_ELEMS = 1 << 20

    def __init__(self, vae, lr=1.0, clip=5.0, seed=783435, grad_sync=None, use_graph=False, device=None,
                 precision="f32", micro_batches=1, fold_norm=True, decoder_grads="full", encoder_forward=None, forward_operands=None):
        """encoder_forward = "f32" (with precision="bf16"): the encoder's FORWARD (input projection + recurrence) with f32-like
        weights inside the bf16 configuration (engine._exact_forward_split: split-bf16 operands + a two-pass recurrence; on a
        fallback rung the exact-f32 kernels).  mu / logvar -- hence z and the KL of encoder.py:55 -- depend on the forward's last
        state alone (enc_lstm.py:60-62), which the WEIGHTS' rounding moves (profiles/r05a_kl_ablation.txt): KL within 2e-5 of the
        reference instead of 1e-5..7e-5 (the default: binary16 forward operands, engine.LSTMEncoderEngine.fwd_operands) or
        2e-4..5e-4 (bf16 forward operands, rounds 1-4), at +0.6 ms per step; gradient products and the decoder stay on the bf16
        matrix pipe.  None / "bf16": the plain bf16 configuration.
        decoder_grads = "norm" (needs fold_norm): in ENCODER-ONLY steps the decoder's two vocabulary-sized gradient tensors
        (embedding table, dW_pred) are not written to memory at all -- text.py:383-387 needs them for the clip norm alone, the
        next backward overwrites them -- and their .grad is left unspecified by such a step; "full" (default) keeps every .grad
        as clip_grad_norm_ would leave it.
        fold_norm: the vocabulary-sized gradient tensors hand their sums of squares to the clip kernel from their producers'
        own passes instead of being read again (_plan_fold; the same norm up to summation order).
        micro_batches = m > 1: gradient accumulation -- every step's batch is cut into m row slices that run one after the
        other on the frozen weights, slice i's gradient exchange (data parallel) is issued when its backward has been queued and
        runs under slice i + 1's forward and backward, the m slice gradients are summed and ONE clip + update follows: the same
        mean gradient as the whole batch in one piece (text.py:382-387; SURVEY.md 8e's overlap window).  Eager mode only."""
        self.vae = vae
        self.micro_batches = int(micro_batches)
        assert self.micro_batches >= 1
        self.fold_norm = bool(fold_norm)
        assert decoder_grads in ("full", "norm")
        self.decoder_grads = decoder_grads
        self._fold, self._fold_plans = None, {}
        if self.micro_batches > 1 and use_graph:
            raise ValueError("micro_batches > 1 runs in eager mode (every slice writes its own gradient slot)")
        self._slice_views = {}
        self._row_index = {}
        self.enc = vae.encoder._hip
        self.dec = vae.decoder._hip
        self.device = torch.device(device) if device is not None else next(vae.parameters()).device
        self.enc.ensure(self.device)
        self.dec.ensure(self.device)
        assert precision in ("f32", "bf16")
        self.enc.precision = self.dec.precision = precision   # large GEMMs: exact f32 (parity) or bf16 pipe (throughput)
        assert encoder_forward in (None, "bf16", "f32")
        self.enc.exact_forward = ("gx", "rec") if (encoder_forward == "f32" and precision == "bf16") else ()
        # forward_operands ("f16" / "bf16"; None keeps the engine's setting, default "f16"): number format of the encoder forward's
        # matrix-pipe operands under precision "bf16" (VAE.set_forward_operands: binary16 puts the KL within 1e-4; range assumption
        # |weights|, |embeddings| << 65504)
        if forward_operands is not None:
            assert forward_operands in ("f16", "bf16")
            self.enc.fwd_operands = forward_operands
        if grad_sync is not None:
            grad_sync.resolve_payload(precision)              # "auto": bf16 wire for the bf16 configuration, exact fp32 otherwise
            if self.micro_batches > 1 and grad_sync.active and precision == "bf16" and (self.enc.persistent or self.dec.persistent):
                # slice i's collectives are in flight during slice i + 1's recurrences BY DESIGN here; a persistent launch needs all 256
                # CUs resident at once and would run into its bounded hand-off spin beside an RCCL kernel (a timeout per step, then the
                # ladder): micro-batch mode under data parallelism runs on the launch-per-timestep kernels from the start
                self.enc.persistent = self.dec.persistent = False
                import sys
                print("vae_lagging_encoder_amd: micro_batches > 1 under data parallelism overlaps collectives with the next slice's "
                      "recurrences; the persistent LSTM launches (which need the whole GPU) are off: %s" % _eng.PERSIST_RUNGS[2],
                      file=sys.stderr)
        self.enc.flat.attach_grads()
        self.dec.flat.attach_grads()
        self.lib = _eng.backend_for(self.device)
        self.clip = float(clip)
        self.grad_sync = grad_sync
        self.use_graph = bool(use_graph) and self.device.type == "cuda"
        self._capturing = False
        self._update = "encoder"
        d = self.device
        # device scalars: [0 kl_weight, 1 lr, 2 sumsq, 3 coef, 4 norm, 5 loss_sum, 6 rec_sum, 7 kl_sum,
        #                  8 void flag, 9 steps committed, 10..12 (loss, rec, kl) sums of the step in flight]  -- 8..12 = the
        # transaction block of lv_clip_*_txn_f32: a step whose persistent recurrences timed out is voided on the device
        self.scal = torch.zeros(16, dtype=torch.float32, device=d)
        self.scal[1] = lr
        self._klw_host = 0.0                  # host copy of scal[0]
        self._journal = []                    # steps queued since the last host check: (x, kl_weight, noise, update)
        self._committed_base = 0.0            # scal[9] at the last host check
        self.on_demote = _log_demotion        # callback(rung) after a move down the persistent-launch ladder; default: one line on stderr
        self.recoveries = 0                   # how many times a voided run of steps was replayed
        self.norm_ws = torch.empty(self.lib.lv_sumsq_workspace_floats(), dtype=torch.float32, device=d)
        # Philox (seed, offset), uint64 bits.  Data parallel: every rank draws from its own substream (the rank is folded
        # into the key), otherwise row i of every rank's batch would see the same eps / dropout masks.
        self.rng_state = torch.tensor([_rank_seed(seed, grad_sync), 0], dtype=torch.int64, device=d)
        self.static = collections.OrderedDict()
        # hipGraph mode: captured graphs hold raw pointers into the engines' workspaces, so nothing may be dropped (graph mode is
        # for a fixed set of (B, T) buckets); eager mode: least-recently-used shapes are dropped (engine._WS, STATIC_SHAPES)
        # Sticky: once a hipGraph trainer has pinned an engine's workspaces, a later eager trainer on the same VAE must not
        # re-enable eviction underneath the captured graphs.
        for e in (self.enc, self.dec):
            e.ws_evictable = e.ws_evictable and not self.use_graph
            if e.wsc is not None:
                e.wsc.evictable = e.ws_evictable

    # -- scalar views ------------------------------------------------------------------------------
    def _s(self, i):
        return P(self.scal, i)

    def set_lr(self, lr):
        self.scal[1] = lr

    # how many steps may be queued before the driver looks at the transaction block on its own (one host read)
    JOURNAL_MAX = 64

    def read_stats(self):
        """One host read: dict(loss_sum, rec_sum, kl_sum, norm, coef) accumulated since reset_stats() -- over COMMITTED steps:
        the same read settles the transaction block (a run of steps voided by a persistent-launch timeout is replayed first)."""
        v = self._settle(self.scal.cpu().tolist())
        return dict(loss_sum=v[5], rec_sum=v[6], kl_sum=v[7], norm=v[4], coef=v[3])

    def commit(self):
        """One host read that makes sure every step queued so far has been applied (text.py:385-387 semantics: an update is
        computed from complete recurrences or not at all).  Returns the ladder rung the engines run on afterwards."""
        self._settle(self.scal.cpu().tolist())
        return max(_eng.persist_rung(self.enc), _eng.persist_rung(self.dec))

    def _settle(self, v):
        """v: host copy of the device scalars.  While the void flag is up: the first (steps committed) entries of the journal were
        applied, the rest was voided ON THE DEVICE (no weight, no report sum moved) -- move both engines one rung down the
        fallback ladder (engine.demote_persistent: write-through hand-off, then the launch-per-timestep kernels), clear the
        gate and queue the voided steps again.  Data parallel: the guard element makes every rank void the same step, so all
        ranks take these decisions identically.  The random draws of a replayed step are new ones (the Philox offset is not
        rewound); injected noise is replayed as given."""
        while v[8] != 0.0:
            done = int(round(v[9] - self._committed_base))
            redo = self._journal[done:]
            if all(_eng.persist_rung(e) == 2 for e in (self.enc, self.dec)):
                raise _eng._lib.LvaeError("a step was voided although both engines run the launch-per-timestep kernels (status words "
                                          "%s / %s)" % (self.enc.status.tolist(), self.dec.status.tolist()))
            for e in (self.enc, self.dec):
                if _eng.persist_rung(e) < 2:
                    _eng.demote_persistent(e)
                else:
                    _eng.reset_persistent_status(e)
            self.recoveries += 1
            for st in self.static.values():
                st.graphs = {}                 # captured graphs hold the launches of the rung that failed
            self.scal[8:13] = 0
            self._committed_base = 0.0
            self._journal = []
            if self.on_demote is not None:
                self.on_demote(max(_eng.persist_rung(self.enc), _eng.persist_rung(self.dec)))
            for x, klw, noise, update in redo:
                self._queue_step(x, klw, noise, update)
                self._journal.append((x, klw, noise, update))
            v = self.scal.cpu().tolist()
        self._committed_base = v[9]
        self._journal = []
        if v[9] >= 4194304.0:
            # the committed-step counter is a float32 incremented by 1.0 per step: long before increments stop registering (2^24)
            # it goes back to zero -- nothing is in flight here (this is behind a host read) and the journal is empty
            self.scal[9] = 0
            self._committed_base = 0.0
        return v

    def reset_stats(self):
        self.scal[5:8] = 0

    def prepare_batches(self, batches):
        """Batch construction for the fused driver: the (token, row) sort the embedding backward needs depends on the token ids
        alone, so it is computed ONCE per batch tensor, when the batch list is built (as data/text_data.py builds the id tensors
        themselves once), and kept with the tensor (engine._TokenSortCache) instead of being redone in every step that meets
        the batch.  Optional: a batch that was not prepared is sorted on first use.  Eager mode only (a captured hipGraph owns
        fixed buffers and sorts inside the graph)."""
        if self.use_graph:
            return
        lib = self.lib
        for x in batches:
            if not (x.is_contiguous() and x.device == self.device and x.dtype == torch.int64 and x.dim() == 2):
                continue
            B, T = x.shape
            s = _eng.stream_ptr(self.device)
            for eng, Tu in ((self.enc, T), (self.dec, T - 1)):
                if Tu <= 0 or eng._sorts.get(x, (Tu, B)) is not None:
                    continue
                V = eng.dims()[0]
                srows, stok = eng.wsc.i32(Tu * B), eng.wsc.i32(Tu * B)
                tmp = eng.wsc.i32(2 * Tu * B)
                lib.lv_token_sort(P(x), T, Tu, B, V, P(srows), P(stok), P(tmp), s)
                eng._sorts.put(x, (Tu, B), srows, stok)

    def enable_row_exchange(self, batches, mode="auto"):
        """Data parallel: exchange the encoder's embedding gradient as a ROW LIST -- an all-gather of the gradient rows of the tokens
        that occur in each rank's batch -- instead of inside the dense all-reduce (41 of the 66 MB encoder payload at the Yahoo
        shape, of which at most B * T of V rows are non-zero per rank).  `batches`: this rank's whole pool, in the
        order every rank uses (later steps must be on these tensors, and all ranks on the same pool index at a time: the
        aggressive loop's replicated host draws).  mode "auto" switches it on only where it is fewer bytes on the wire than the
        ring all-reduce (capacity * world < 2 V: natural, Zipf-distributed text; small worlds); "rows" forces it.  Collective:
        every rank must call it.  Returns True when the row-list exchange is in use afterwards."""
        gs = self.grad_sync
        if gs is None or not gs.active:
            return False
        if self.micro_batches != 1 or self.use_graph:
            raise ValueError("the row-list exchange runs with micro_batches = 1 in eager mode")
        self.prepare_batches(batches)
        V, ni = self.enc.dims()[:2]
        uniq = []
        self._row_index = {}
        for j, x in enumerate(batches):
            B, T = x.shape
            _, stok = self.enc._sorts.get(x, (T, B))
            uniq.append(torch.unique_consecutive(stok).to(torch.int64))      # batch preparation, once: sorted ids of the tokens that occur
            self._row_index[id(x)] = j
        self._row_keep = list(batches)                       # the ids in _row_index must stay valid
        ef = self.enc.flat
        return gs.prepare_rows(uniq, V, ni, ef.offsets[ef.names[1]], mode=mode)

    def invalidate_batch(self, x=None):
        """A batch tensor was rewritten behind torch's version counter (x.data.copy_, a numpy view, a DLPack / custom-kernel
        write): drop its cached sorted token lists (None: all of them).  See engine._TokenSortCache."""
        self.enc._sorts.invalidate(x)
        self.dec._sorts.invalidate(x)

    # -- per-(B,T) static state ----------------------------------------------------------------------
    def _static_for(self, B, T, parts=1):
        """parts: the step's batch is B * parts rows of which this static state serves one slice (mean over ALL rows)."""
        key = (B, T) if parts == 1 else (B, T, parts)
        st = self.static.get(key)
        if st is not None:
            self.static.move_to_end(key)
        else:
            d = self.device
            V, ni, H, nz = self.dec.dims()
            st = _Static()
            st.x = torch.zeros(B, T, dtype=torch.int64, device=d)
            st.eps = torch.zeros(B, 1, nz, dtype=torch.float32, device=d)
            st.m_in = torch.ones(B, T - 1, ni, dtype=torch.uint8, device=d)
            st.m_out = torch.ones(B, T - 1, H, dtype=torch.uint8, device=d)
            st.z = torch.empty(B, 1, nz, dtype=torch.float32, device=d)
            st.kl = torch.empty(B, dtype=torch.float32, device=d)
            st.loss = torch.empty(B, dtype=torch.float32, device=d)
            st.rec = torch.empty(B, dtype=torch.float32, device=d)
            st.gl = torch.full((B,), 1.0 / (B * parts), dtype=torch.float32, device=d)   # d(mean_b loss_b)/d loss_b
            st.rowscale = torch.empty(B, dtype=torch.float32, device=d)
            st.dkl = torch.empty(B, dtype=torch.float32, device=d)
            st.dmulv = torch.empty(B, 2 * nz, dtype=torch.float32, device=d)
            st.graphs = {}
            st.x_key = None
            st.xin = st.x
            self.static[key] = st
            if not self.use_graph:
                while len(self.static) > STATIC_SHAPES:
                    self.static.popitem(last=False)
        return st

    # -- the step ----------------------------------------------------------------------------------------
    def _fwd_bwd(self, st, draw):
        self._arm_fold()
        try:
            self._fwd_bwd_armed(st, draw)
        finally:
            self.enc.fold = self.dec.fold = None

    def _fwd_bwd_armed(self, st, draw):
        lib, s = self.lib, _eng.stream_ptr(self.device)
        B, T = st.x.shape
        dec = self.vae.decoder
        train = self.vae.training
        use_in = train and dec.dropout_in.p > 0
        use_out = train and dec.dropout_out.p > 0
        if draw:
            # throughput mode: eps and both dropout keep-masks from the on-device Philox stream, one launch
            # (offset advanced by the loss assembly's launch below: inc = 0 here)
            lib.lv_rng_noise_step(P(st.eps), st.eps.numel(), P(st.m_in) if use_in else None, st.m_in.numel(),
                                  1.0 - dec.dropout_in.p, P(st.m_out) if use_out else None, st.m_out.numel(),
                                  1.0 - dec.dropout_out.p, P(self.rng_state), 0, s)
        m_in = st.m_in if use_in else None
        m_out = st.m_out if use_out else None
        # encoder: ... LSTM, then head + reparameterise + KL in one launch
        mulv = self.enc.forward(st.xin, head=(st.eps, st.z, st.kl), x_key=st.x_key)
        self.dec.forward(st.xin, st.z, m_in, m_out, dec.dropout_in.p, dec.dropout_out.p, want_rec=False, x_key=st.x_key)
        w = self.dec._ws(B, T - 1)
        # rec, loss, the running report sums and the seeds of mean_b(loss_b).backward(), one launch
        # (report sums into the pending slots: committed by the transaction gate; the Philox offset moves on here when drawn)
        if draw:
            lib.lv_loss_assemble_rng_f32(P(w.nll), P(st.kl), self._s(0), P(st.gl), P(st.loss), P(st.rec), P(st.rowscale), P(st.dkl),
                                         self._s(10), T - 1, B, P(self.rng_state), 1, s)
        else:
            lib.lv_loss_assemble_f32(P(w.nll), P(st.kl), self._s(0), P(st.gl), P(st.loss), P(st.rec), P(st.rowscale), P(st.dkl),
                                     self._s(10), T - 1, B, s)
        dzp, parts = self.dec.backward(st.rowscale, partial_dz=True)
        hook = bucket = None
        if self.grad_sync is not None and not self._capturing and self.grad_sync.active:
            ef = self.enc.flat
            align = 1024 if self.grad_sync.payload == "bf16" else 4      # bf16 wire rows are 1024 elements wide
            n_emb = ef.offsets[ef.names[1]] // align * align             # the embedding table leads the flat buffer
            if self.grad_sync.rows is not None:
                # ... as a ROW LIST (enable_row_exchange): this rank's touched rows are all-gathered instead of the dense table
                j = self._row_index.get(id(st.x_key))
                if j is None:
                    raise _eng._lib.LvaeError("row-list exchange is on, but this batch tensor was not among those given to "
                                              "enable_row_exchange()")
                def bucket():
                    self.grad_sync.start_encoder_rows(ef, j)
            elif n_emb >= self.BUCKET_MIN_ELEMS and n_emb > 0:
                # first bucket of the encoder exchange: the embedding gradient (41 of 66 MB at the Yahoo shape) goes out as
                # soon as the scatter is queued and runs under the LSTM weight-gradient GEMMs; sync() sends the rest
                def bucket():
                    self.grad_sync.start_encoder_bucket(ef, 0, n_emb)
        if self.grad_sync is not None and not self._capturing:
            def start():
                # data parallel: the decoder-gradient exchange (149 MB all-reduce, or 75 MB reduce-scatter when only its norm
                # is needed) is issued as soon as it can run beside ordinary kernels
                self.dec.join()
                self.grad_sync.start_decoder(self.dec.flat, self._update)
            if self._collective_after_bptt():
                # not beside a PERSISTENT encoder BPTT (that launch needs every CU resident at once): issued right after the
                # BPTT has been queued, so the collective starts behind it and runs under the encoder's weight-gradient GEMMs
                hook = start
            else:
                start()                   # step kernels: the collective runs under the whole encoder backward
        self.enc.backward(None, head=(st.eps, dzp, parts, st.dkl), after_bptt=hook, after_embed=bucket)
        self.dec.join()           # decoder weight-gradient GEMMs ran on the side stream underneath the BPTT chains
        if self.grad_sync is not None and self.grad_sync.active:
            # data parallel: this rank's timeout flag rides in the tail padding of the encoder gradient (exchanged by sync())
            ef = self.enc.flat
            lib.lv_txn_guard_f32(_eng.status_ptr(self.enc), _eng.status_ptr(self.dec), P(ef.grad_padded, ef.guard_index), s)

    def _collective_after_bptt(self):
        """True when the encoder's BPTT may be a persistent launch: a collective must then not be in flight beside it."""
        return bool(self.enc.persistent and self.enc.precision == "bf16")

    def _clip_and_step(self, update, dec_ss=None):
        lib, s = self.lib, _eng.stream_ptr(self.device)
        ef, df = self.enc.flat, self.dec.flat
        # the transaction gate rides in the clip coefficient's launch: status words of both engines, data parallel also the
        # exchanged guard element; it commits the step's report sums (scal[10..12] -> scal[5..7]) or raises the void flag scal[8]
        dp = self.grad_sync is not None and self.grad_sync.active
        gate = (_eng.status_ptr(self.enc), _eng.status_ptr(self.dec), P(ef.grad_padded, ef.guard_index) if dp else None,
                self._s(8), self._s(5))
        if dec_ss is None and self._fold is not None:
            # the vocabulary-sized tensors' squares came out of their producers (_plan_fold): stage 1 reads the rest only
            f = self._fold
            lib.lv_clip_norm2_fold_txn_f32(P(ef.grad, f.enc_off), ef.numel - f.enc_off, P(df.grad, f.dec_off), f.dec_end - f.dec_off,
                                           P(self.norm_ws), P(f.parts), f.n, self.clip, self._s(2), self._s(3), self._s(4), *gate, s)
        elif dec_ss is None:
            lib.lv_clip_norm2_txn_f32(P(ef.grad), ef.numel, P(df.grad), df.numel, P(self.norm_ws), self.clip, self._s(2), self._s(3),
                                      self._s(4), *gate, s)
        else:
            # data parallel, encoder-only step: the decoder gradient was exchanged as its sum of squares only (GradSync)
            lib.lv_sumsq_f32(P(ef.grad), ef.numel, P(self.norm_ws), self._s(2), 0, s)
            lib.lv_sum_accum_f32(P(dec_ss), 1, self._s(2), s)
            lib.lv_clip_coef_txn_f32(self._s(2), self.clip, self._s(3), self._s(4), *gate, s)
        # clip_grad_norm_ scales every grad in place; the update only touches the stepped side; both are no-ops under the void flag
        if update == "both":
            lib.lv_sgd_step_txn_f32(P(ef.data), P(ef.grad), ef.numel, self._s(1), self._s(3), 1, self._s(8), s)
            lib.lv_sgd_step_txn_f32(P(df.data), P(df.grad), df.numel, self._s(1), self._s(3), 1, self._s(8), s)
        else:
            a, b = (ef, df) if update == "encoder" else (df, ef)      # a is stepped, b's gradient is only scaled: one launch
            b_off, b_n = 0, b.numel
            if update == "encoder" and self._fold is not None and self.decoder_grads == "norm":
                b_off, b_n = self._fold.dec_off, self._fold.dec_end - self._fold.dec_off      # what of the decoder's gradient exists
            lib.lv_sgd_step_scale_txn_f32(P(a.data), P(a.grad), a.numel, self._s(1), self._s(3), 1, P(b.grad, b_off), b_n, self._s(8), s)

    def _plan_fold(self, st, update):
        """Norm folding (single GPU, one micro-batch): the three vocabulary-sized gradient tensors -- both embedding tables and
        dW_pred, 164 of the 215 MB at the Yahoo shape -- hand their sums of squares to the clip kernel from the kernels that
        complete them, and lv_clip_norm2's streaming pass covers only the rest of the two flat buffers (embedding first in both,
        pred_linear last in the decoder's: the rest is one contiguous range each).  Off where the norm is not that of this
        backward's own tensors: data parallel (the norm is the averaged gradient's) and micro-batches (of the slots' sum)."""
        self._fold = None
        if not self.fold_norm or self.grad_sync is not None or self.micro_batches != 1:
            return
        B, T = st.x.shape
        key = (B, T)
        f = self._fold_plans.get(key)
        if f is None:
            pe, pd = self.enc.fold_parts(B, T), self.dec.fold_parts(B, T - 1)
            f = _eng._NS()
            f.n = pe["embed"] + pd["embed"] + pd.get("pred", 0)
            f.parts = torch.zeros(f.n, dtype=torch.float32, device=self.device)
            f.enc_embed = f.parts[:pe["embed"]]
            f.dec_embed = f.parts[pe["embed"]:pe["embed"] + pd["embed"]]
            f.dec_pred = f.parts[pe["embed"] + pd["embed"]:] if "pred" in pd else None
            ef, df = self.enc.flat, self.dec.flat
            f.enc_off = ef.offsets["lstm.weight_ih_l0"]              # behind the encoder's embedding table
            f.dec_off = df.offsets["trans_linear.weight"]            # behind the decoder's
            f.dec_end = df.offsets["pred_linear.weight"] if f.dec_pred is not None else df.numel
            assert ef.offsets["embed.weight"] == 0 and df.offsets["embed.weight"] == 0 and \
                df.offsets["pred_linear.weight"] + self.vae.decoder.pred_linear.weight.numel() <= df.numel
            self._fold_plans[key] = f
        self._fold = f

    def _arm_fold(self):
        """Tell the engines where this backward's partial sums of squares go (and whether the decoder's big tensors are
        norm-only); _fwd_bwd disarms them again when everything is queued, so that a backward of the drop-in autograd path on
        the same modules never runs with a trainer's plan."""
        f = self._fold
        if f is None:
            return
        only = self.decoder_grads == "norm" and self._update == "encoder"
        self.enc.fold = {"embed": (f.enc_embed, False)}
        self.dec.fold = {"embed": (f.dec_embed, only)}
        if f.dec_pred is not None:
            self.dec.fold["pred"] = (f.dec_pred, only)

    def _run(self, st, update, draw):
        self._update = update
        self._plan_fold(st, update)
        if self.grad_sync is not None:
            self.grad_sync.begin_step()
        self._fwd_bwd(st, draw)
        dec_ss = None
        if self.grad_sync is not None:
            dec_ss = self.grad_sync.sync(self.enc.flat, self.dec.flat, update)
        self._clip_and_step(update, dec_ss)

    def step(self, x, kl_weight, noise=None, update="encoder"):
        """One body of the aggressive loop on batch x (int64 [B][T] on device).

        noise=(eps, mask_in, mask_out) injects the random draws (parity mode); None draws them on device.
        update: 'encoder' (inner loop, text.py:387), 'decoder' (joint step while aggressive, text.py:419-424)
        or 'both' (joint step after aggressive mode ends).

        The step is queued, not awaited.  It is also journalled until the next host read (read_stats / commit / every JOURNAL_MAX
        steps): should a persistent recurrence of it time out, the device voids it -- and everything queued behind it -- and the
        read replays the voided steps one rung down the fallback ladder (_settle).  x (and injected noise) must therefore stay
        unmodified until then, as they must for the sorted-token cache."""
        self._queue_step(x, kl_weight, noise, update)
        self._journal.append((x, float(kl_weight), noise, update))
        if len(self._journal) >= self.JOURNAL_MAX:
            self.commit()

    def _slices(self, x, m):
        """The m row slices of batch tensor x as stable view objects (kept per batch tensor, weakly: the sorted-token cache is
        keyed by tensor identity, and a fresh view per step would re-sort every slice every time)."""
        import weakref
        ent = self._slice_views.get(id(x))
        if ent is not None and ent[0]() is x and ent[1] == (x._version, m):
            return ent[2]
        Bs = x.shape[0] // m
        views = [x[i * Bs:(i + 1) * Bs] for i in range(m)]
        if len(self._slice_views) > 4096:
            self._slice_views.clear()
        self._slice_views[id(x)] = (weakref.ref(x, lambda _r, i=id(x), d=self._slice_views: d.pop(i, None)), (x._version, m), views)
        return views

    def _queue_micro(self, x, kl_weight, noise, update, m):
        """One step as m micro-batches (see __init__)."""
        lib, s = self.lib, _eng.stream_ptr(self.device)
        B, T = x.shape
        if B % m != 0:
            raise ValueError("micro_batches = %d does not divide the batch of %d sequences" % (m, B))
        if not (x.is_contiguous() and x.device == self.device and x.dtype == torch.int64):
            x = x.to(self.device, torch.int64).contiguous()
        Bs = B // m
        if self._klw_host != float(kl_weight):
            self.scal[0] = float(kl_weight)
            self._klw_host = float(kl_weight)
        draw = noise is None
        ef, df = self.enc.flat, self.dec.flat
        gs = self.grad_sync
        dp = gs is not None and gs.active
        self._update = update
        self._fold = None                 # (the norm is that of the slots' sum: no folding, _plan_fold)
        if gs is not None:
            gs.begin_step()
        try:
            for i, xs in enumerate(self._slices(x, m)):
                st = self._static_for(Bs, T, m)
                st.xin, st.x_key = xs, xs
                if not draw:
                    eps, m_in, m_out = noise
                    rows = slice(i * Bs, (i + 1) * Bs)
                    st.eps.copy_(eps.reshape(B, 1, -1)[rows])
                    if m_in is not None:
                        st.m_in.copy_(m_in[rows])
                    if m_out is not None:
                        st.m_out.copy_(m_out[rows])
                ef.use_slot(i)
                df.use_slot(i)
                if gs is not None:
                    gs.set_slot(i)
                self._fwd_bwd(st, draw)
                if dp:
                    gs.sync_issue(ef, df, update)       # slice i's exchange: in flight underneath slice i + 1
            dec_ss = None
            if dp:
                for i in range(m):
                    ef.use_slot(i)
                    df.use_slot(i)
                    gs.set_slot(i)
                    gs.sync_collect(ef, df, update)
                dec_ss = gs.sync_norm(df, update, tuple(range(m)))
        finally:
            ef.use_slot(0)
            df.use_slot(0)
            if gs is not None:
                gs.set_slot(0)
        # sum of the slice gradients (each carries the 1 / B of the whole batch), over the padded buffers: the guard elements add up too
        for f in (ef, df):
            for i in range(1, m):
                lib.lv_add_f32(P(f._slots[0][0]), P(f._slots[i][0]), P(f._slots[0][0]), f._slots[0][0].numel(), s)
        self._clip_and_step(update, dec_ss)
        if update in ("encoder", "both"):
            self.enc.wgen += 1
        if update in ("decoder", "both"):
            self.dec.wgen += 1

    def _queue_step(self, x, kl_weight, noise, update):
        if self.micro_batches > 1:
            return self._queue_micro(x, kl_weight, noise, update, self.micro_batches)
        B, T = x.shape
        for e in (self.enc, self.dec):
            if e.flat.detached:           # a drop-in autograd backward moved some .grad off the flat buffer (FlatBuffer.before_autograd_backward)
                e.flat.attach_grads()
        st = self._static_for(B, T)
        if self.use_graph or not (x.is_contiguous() and x.device == self.device and x.dtype == torch.int64):
            st.x.copy_(x)                 # captured graphs read the per-shape static buffer
            st.xin = st.x
        else:
            st.xin = x                    # eager: the kernels read the caller's batch in place (no copy launch)
        st.x_key = x                      # the caller's batch tensor: its identity keys the engines' sorted-token cache
        if self._klw_host != float(kl_weight):
            self.scal[0] = float(kl_weight)       # one H2D fill per change of the weight, not per step (text.py anneals it per outer iteration)
            self._klw_host = float(kl_weight)
        draw = noise is None
        if not draw:
            eps, m_in, m_out = noise
            st.eps.copy_(eps.reshape(st.eps.shape))
            if m_in is not None:
                st.m_in.copy_(m_in)
            if m_out is not None:
                st.m_out.copy_(m_out)
        try:
            if not self.use_graph:
                self._run(st, update, draw)
                return
            # cached weight images (the frozen decoder's) are brought up to date OUTSIDE the captured region: a graph
            # replays exactly what was queued at capture time, and at capture time they are fresh
            self.enc.refresh_weight_images(B, self.device)
            self.dec.refresh_weight_images(B, self.device)
            gens = (getattr(self.enc, "ladder_gen", 0), getattr(self.dec, "ladder_gen", 0))
            key = (update, draw, self.vae.training) + gens
            g = st.graphs.get(key)
            if g is None:
                # graphs captured on an earlier rung of the persistent-launch ladder (a demotion by this trainer's _settle or by
                # training.guarded_eval) replay launches of the rung that failed: dropped, never replayed
                for k in [k for k in st.graphs if k[3:] != gens]:
                    del st.graphs[k]
                # eager warm-up (allocates every workspace), then capture
                self._run(st, update, draw)
                torch.cuda.synchronize(self.device)
                g = self._capture(st, update, draw)
                st.graphs[key] = g
                return
            for part in g:
                part()
        finally:
            # raw-pointer weight updates are invisible to torch's version counters
            if update in ("encoder", "both"):
                self.enc.wgen += 1
            if update in ("decoder", "both"):
                self.dec.wgen += 1

    def _capture(self, st, update, draw):
        """Capture the step as hipGraph(s).  With a gradient all-reduce the step is split around it
        (fwd+bwd graph | RCCL all-reduce | clip+SGD graph): collectives are not captured."""
        parts = []
        self._update = update
        self._plan_fold(st, update)
        stream = torch.cuda.Stream(self.device)
        stream.wait_stream(torch.cuda.current_stream(self.device))

        def cap(fn):
            gr = torch.cuda.CUDAGraph()
            self._capturing = True
            try:
                with _capture_graph(gr, stream):
                    fn()
            finally:
                self._capturing = False
            return gr.replay
        if self.grad_sync is None:
            parts.append(cap(lambda: (self._fwd_bwd(st, draw), self._clip_and_step(update))))
        else:
            self._update = update
            parts.append(cap(lambda: self._fwd_bwd(st, draw)))
            # the exchange returns the same device scalar (or None) on every call of a given `update`: safe to capture
            ss = self.grad_sync.ss_handle(self.dec.flat, update)
            parts.append(lambda: self.grad_sync.sync(self.enc.flat, self.dec.flat, update))
            parts.append(cap(lambda: self._clip_and_step(update, ss)))
        return parts

    # -- the loop of text.py:366-400 ------------------------------------------------------------------------
    def inner_loop(self, batches, first, kl_weight, np_rng=None, max_iter=100, window=15, fixed_k=None, noise_fn=None):
        """Run the aggressive inner loop starting on batch `first`; later batches are drawn with
        np_rng.random_integers(0, len-1) semantics (text.py:389).  Returns the number of encoder steps taken.

        fixed_k: run exactly that many steps with no data-dependent exit (BASELINE.json stress config).

        Data parallel (grad_sync set): the windowed exit test of text.py:393-396 is taken on the GLOBAL mean loss per word
        (sum of the ranks' loss sums / sum of their word counts, one 2-element all-reduce per window), so every rank
        leaves the loop at the same iteration -- a rank deciding on its local window would pair its joint-step all-reduce
        with the other ranks' encoder-step all-reduces.  The device accumulators are cleared on exit, so a read_stats()
        after the joint step reports that step alone (text.py:426-427)."""
        rng = np_rng if np_rng is not None else np.random
        sub_iter = 1
        x = first
        burn_num_words = 0
        burn_pre_loss = 1e4
        self.reset_stats()
        steps = 0
        while sub_iter < max_iter:
            B, T = x.shape
            burn_num_words += (T - 1) * B
            self.step(x, kl_weight, noise=None if noise_fn is None else noise_fn(x), update="encoder")
            steps += 1
            idx = int(rng.randint(0, len(batches)))
            x = batches[idx]
            if fixed_k is not None:
                if steps >= fixed_k:
                    break
            elif sub_iter % window == 0:
                loss_sum = self.read_stats()["loss_sum"]                   # the only host sync of the window
                if self.grad_sync is not None:
                    cur = self.grad_sync.window_mean(loss_sum, burn_num_words)
                else:
                    cur = loss_sum / burn_num_words
                if burn_pre_loss - cur < 0:
                    break
                burn_pre_loss = cur
                burn_num_words = 0
                self.reset_stats()
            sub_iter += 1
        # every step of the loop has been applied when it returns (one host read; a voided tail is replayed here)
        self.commit()
        self.reset_stats()
        return steps


class AggressiveImageTrainer(object):
    """Fused driver for the Omniglot aggressive loop (reference image.py:295-348): one `step()` = zero_grad, VAE.loss,
    loss.mean().backward(), clip_grad_norm_(all params, 5.0), Adam step on the encoder (image.py:302-314), as a
    stream-ordered sequence of C-ABI kernel calls with device-resident scalars (kl weight, lr, Adam step, norm)."""

    def __init__(self, vae, lr=1e-3, clip=5.0, seed=783435, device=None, precision="f32", betas=(0.9, 0.999), eps=1e-8,
                 use_graph=False):
        self.vae = vae
        self.enc = vae.encoder._hip
        self.dec = vae.decoder._hip
        self.device = torch.device(device) if device is not None else next(vae.parameters()).device
        self.enc.ensure(self.device)
        self.dec.ensure(self.device)
        # "f32": exact; "bf16x3": the decoder's direct convolutions on split-bf16 operands (f32-like results: the f32 fixtures hold),
        # the rest exact; "bf16": plain bf16 operands there and in the im2col GEMMs (image_engine.CONV_TERMS)
        assert precision in ("f32", "bf16x3", "bf16")
        self.enc.precision = self.dec.precision = precision
        self.enc.flat.attach_grads()
        self.dec.flat.attach_grads()
        self.lib = _eng.backend_for(self.device)
        self.clip, self.betas, self.adam_eps = float(clip), betas, float(eps)
        # ~1500 small launches per step: replaying the captured step as a hipGraph removes the host launch overhead
        self.use_graph = bool(use_graph) and self.device.type == "cuda"
        self._static = {}
        d = self.device
        # device scalars: [kl_weight, lr, sumsq, coef, norm, loss_sum, rec_sum, kl_sum, adam_step_enc, adam_step_dec, zero]
        self.scal = torch.zeros(12, dtype=torch.float32, device=d)
        self.scal[1] = lr
        self.norm_ws = torch.empty(self.lib.lv_sumsq_workspace_floats(), dtype=torch.float32, device=d)
        self.rng_state = torch.tensor([seed, 0], dtype=torch.int64, device=d)
        self.m = {"enc": torch.zeros_like(self.enc.flat.data), "dec": torch.zeros_like(self.dec.flat.data)}
        self.v = {"enc": torch.zeros_like(self.enc.flat.data), "dec": torch.zeros_like(self.dec.flat.data)}

    def _s(self, i):
        return P(self.scal, i)

    def read_stats(self):
        v = self.scal.cpu().tolist()
        return dict(loss_sum=v[5], rec_sum=v[6], kl_sum=v[7], norm=v[4], coef=v[3])

    def reset_stats(self):
        self.scal[5:8] = 0

    def set_lr(self, lr):
        self.scal[1] = lr

    def reset_optimizer(self, lr):
        """image.py:419-420: a learning-rate decay builds NEW Adam optimizers -- first and second moments and the step counts
        start from zero again, for the encoder's and the decoder's optimizer alike."""
        self.scal[1] = lr
        self.scal[8:10] = 0
        for k in ("enc", "dec"):
            self.m[k].zero_()
            self.v[k].zero_()

    def binarize(self, probs):
        """torch.bernoulli(batch) (image.py:287,318) on device."""
        probs = probs.contiguous().float()
        out = torch.empty_like(probs)
        s = _eng.stream_ptr(self.device)
        self.lib.lv_rng_bernoulli_f32(P(probs), P(out), probs.numel(), P(self.rng_state), 7, s)
        self.lib.lv_rng_advance(P(self.rng_state), 1, s)
        return out

    def inner_loop(self, x_train, first, kl_weight, batch_size=50, np_rng=None, max_iter=100, window=10, fixed_k=None,
                   eps_fn=None, binarize_fn=None):
        """The aggressive inner loop of image.py:295-327 starting on the binarised batch `first`: encoder-only steps,
        each followed by the pick of the next batch -- np.random.choice(N, batch_size, replace=False) rows of `x_train`
        (image.py:316), dynamically binarised on device (torch.bernoulli, image.py:318) -- and, every `window` (10)
        iterations, the exit test on the mean loss per example (image.py:320-325).  One host read per window instead of
        the reference's one per iteration.  Returns the number of encoder steps taken.

        eps_fn(x) / binarize_fn(probs) inject the reparameterisation noise / the binarisation draw (parity tests)."""
        rng = np_rng if np_rng is not None else np.random
        N = int(x_train.shape[0])
        sub_iter = 1
        x = first
        burn_num_examples = 0
        burn_pre_loss = 1e4
        self.reset_stats()
        steps = 0
        while sub_iter < max_iter:
            burn_num_examples += int(x.shape[0])
            self.step(x, kl_weight, eps=None if eps_fn is None else eps_fn(x), update="encoder")
            steps += 1
            id_ = rng.choice(N, batch_size, replace=False)
            probs = x_train[torch.from_numpy(np.asarray(id_, dtype=np.int64)).to(x_train.device)].to(self.device)
            x = binarize_fn(probs) if binarize_fn is not None else self.binarize(probs)
            if fixed_k is not None:
                if steps >= fixed_k:
                    break
            elif sub_iter % window == 0:
                cur = self.read_stats()["loss_sum"] / burn_num_examples
                if burn_pre_loss - cur < 0:
                    break
                burn_pre_loss = cur
                burn_num_examples = 0
                self.reset_stats()
            sub_iter += 1
        self.reset_stats()
        return steps

    def step(self, x, kl_weight, eps=None, update="encoder"):
        """x (B,1,28,28) binarised; eps (B,1,nz) injects the reparameterisation noise (parity mode)."""
        d = self.device
        B = x.shape[0]
        nz = self.vae.nz
        self.scal[0] = float(kl_weight)
        try:
            self._step(x, eps, update, d, B, nz)
        finally:
            if update in ("decoder", "both"):
                self.dec.wgen += 1          # raw-pointer update: invalidates the cached packed conv weights

    def _step(self, x, eps, update, d, B, nz):
        if not self.use_graph:
            if eps is not None:
                eps = eps.to(d).float().contiguous()
            self._body(x, eps, update)
            return
        self.dec.refresh_packs(d)
        key = (B, update, eps is None, self.vae.training)
        st = self._static.get(key)
        if st is None:
            st = dict(x=torch.zeros(B, 1, 28, 28, dtype=torch.float32, device=d),
                      eps=torch.zeros(B, 1, nz, dtype=torch.float32, device=d), graph=None)
            self._static[key] = st
        st["x"].copy_(x.reshape(B, 1, 28, 28))
        if eps is not None:
            st["eps"].copy_(eps.reshape(B, 1, nz))
        if st["graph"] is None:
            self._body(st["x"], None if eps is None else st["eps"], update)      # eager warm-up performs this call's step
            torch.cuda.synchronize(d)
            g = torch.cuda.CUDAGraph()
            stream = torch.cuda.Stream(d)
            stream.wait_stream(torch.cuda.current_stream(d))
            # capture must not advance state twice: snapshot the mutable state the body touches, capture, restore
            snap = (self.enc.flat.data.clone(), self.dec.flat.data.clone(), self.scal.clone(), self.rng_state.clone(),
                    {k: v.clone() for k, v in self.m.items()}, {k: v.clone() for k, v in self.v.items()},
                    {k: b.clone() for k, b in self.vae.named_buffers()})
            with _capture_graph(g, stream):
                self._body(st["x"], None if eps is None else st["eps"], update)
            self.enc.flat.data.copy_(snap[0]); self.dec.flat.data.copy_(snap[1]); self.scal.copy_(snap[2])
            self.rng_state.copy_(snap[3])
            for k in self.m:
                self.m[k].copy_(snap[4][k]); self.v[k].copy_(snap[5][k])
            for k, b in self.vae.named_buffers():
                b.copy_(snap[6][k])
            st["graph"] = g
            return
        st["graph"].replay()

    def _body(self, x, eps, update):
        lib, s, d = self.lib, _eng.stream_ptr(self.device), self.device
        B = x.shape[0]
        nz = self.vae.nz
        if eps is None:
            eps = torch.empty(B, 1, nz, dtype=torch.float32, device=d)
            lib.lv_rng_normal_f32(P(eps), eps.numel(), P(self.rng_state), 0, s)
            lib.lv_rng_advance(P(self.rng_state), 1, s)
        mulv = self.enc.forward(x)
        z = torch.empty(B, 1, nz, dtype=torch.float32, device=d)
        kl = torch.empty(B, dtype=torch.float32, device=d)
        lib.lv_reparam_kl_fwd_f32(P(mulv), P(eps), P(z), P(kl), B, 1, nz, s)
        rec = self.dec.forward(x, z.view(B, nz))
        loss = torch.empty(B, dtype=torch.float32, device=d)
        rec2 = torch.empty(B, dtype=torch.float32, device=d)
        lib.lv_vae_loss_f32(P(rec), P(kl), self._s(0), P(loss), P(rec2), 1, B, s)
        lib.lv_sum_accum_f32(P(loss), B, self._s(5), s)
        lib.lv_sum_accum_f32(P(rec), B, self._s(6), s)
        lib.lv_sum_accum_f32(P(kl), B, self._s(7), s)
        gl = torch.full((B,), 1.0 / B, dtype=torch.float32, device=d)
        drec = torch.empty(B, dtype=torch.float32, device=d)
        dkl = torch.empty(B, dtype=torch.float32, device=d)
        lib.lv_loss_bwd_scales_f32(P(gl), None, None, self._s(0), P(drec), P(dkl), B, s)
        dz = self.dec.backward(drec)
        dmulv = torch.empty(B, 2 * nz, dtype=torch.float32, device=d)
        lib.lv_reparam_kl_bwd_f32(P(mulv), P(eps), P(dz), P(dkl), P(dmulv), B, 1, nz, s)
        self.enc.backward(dmulv)
        ef, df = self.enc.flat, self.dec.flat
        lib.lv_sumsq_f32(P(ef.grad), ef.numel, P(self.norm_ws), self._s(2), 0, s)
        lib.lv_sumsq_f32(P(df.grad), df.numel, P(self.norm_ws), self._s(2), 1, s)
        lib.lv_clip_coef_f32(self._s(2), self.clip, self._s(3), self._s(4), s)
        for name, flat, slot in (("enc", ef, 8), ("dec", df, 9)):
            stepped = (update in ("encoder", "both")) if name == "enc" else (update in ("decoder", "both"))
            if stepped:
                lib.lv_add_scalar_f32(self._s(slot), 1.0, s)
                lib.lv_adam_step_f32(P(flat.data), P(flat.grad), P(self.m[name]), P(self.v[name]), flat.numel, self._s(1),
                                     self._s(3), self._s(slot), self.betas[0], self.betas[1], self.adam_eps, 1, s)
            else:
                lib.lv_scale_f32(P(flat.grad), flat.numel, self._s(3), s)
