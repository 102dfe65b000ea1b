"""Host-side sequencing of the gfx950 kernels for the LSTM-VAE hot path.

This is the layer between the reference's Python class surface (modules/*.py mirrors) and the C ABI in
include/lvae.h.  PyTorch is used only for device memory (torch.empty), the current HIP stream and autograd
plumbing; every arithmetic op on the hot path is a call into liblvae_hip.so.

Layout contract: token ids stay batch-first int64 [B][T] exactly as the reference's
create_data_batch hands them over (data/text_data.py:219-255); every activation is time-major ([T][B][*]) so one
LSTM timestep is a contiguous slab; parameters of one module live in one flat fp32 buffer (FlatBuffer) whose
segments back the module's nn.Parameters, gradients in a parallel flat buffer.
"""
import ctypes
import os

import torch

from . import _lib

# ---- backend selection -------------------------------------------------------------------------------
def backend_for(device):
    """The C-ABI library for tensors on `device`: the gfx950 build for ROCm 'cuda' devices, nothing else.  (The GPU-less CI
    substitutes this function from outside the package -- tests/emu/install.py -- to drive the same host code against an
    emulator build of the kernel sources; the package itself has no such path.)"""
    device = torch.device(device)
    if device.type == "cuda":
        return _lib.load()
    raise _lib.LvaeError(
        "vae_lagging_encoder_amd runs on MI355X (ROCm 'cuda' tensors) through its HIP extension; got a %s tensor. "
        "There is no CPU fallback." % device.type)


def stream_ptr(device):
    device = torch.device(device)
    if device.type == "cuda":
        return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    return None


def P(t, offset_elems=0):
    """Raw device pointer of tensor t (+ element offset); None -> NULL."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr() + offset_elems * t.element_size())


def _round_up(x, m):
    return (x + m - 1) // m * m


class FlatBuffer(object):
    """One flat fp32 buffer for a module's parameters and one for its gradients; each nn.Parameter's .data is
    re-pointed to a 16-byte-aligned segment so the global grad-norm / clip+SGD are single streaming passes."""

    def __init__(self, named_params, device):
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        self.offsets = {}
        off = 0
        for n, p in named_params:
            self.offsets[n] = off
            off += _round_up(p.numel(), 4)
        self.numel = off
        self.device = torch.device(device)
        self.data = torch.zeros(off, dtype=torch.float32, device=self.device)
        # gradients: zero tail padding to a multiple of 1024 elements, so the buffer splits evenly across any power-of-two
        # world size (reduce-scatter of dist.GradSync); kernels only ever touch [0, numel).  There is always at least one
        # padding element: the LAST one is the data-parallel transaction guard (lv_txn_guard_f32 writes 1 there when this rank's
        # persistent launches reported a timeout; after the mean all-reduce it is non-zero on every rank iff any rank did)
        self.grad_padded = torch.zeros(_round_up(off + 1, 1024), dtype=torch.float32, device=self.device)
        self.guard_index = self.grad_padded.numel() - 1
        self.grad = self.grad_padded[:off]
        self.views, self.gviews = {}, {}
        for n, p in named_params:
            o = self.offsets[n]
            v = self.data[o:o + p.numel()].view(p.shape)
            v.copy_(p.data.to(self.device, torch.float32))
            p.data = v
            self.views[n] = v
            self.gviews[n] = self.grad[o:o + p.numel()].view(p.shape)

        self._slots = [(self.grad_padded, self.grad, self.gviews)]
        self.detached = False

    def use_slot(self, i):
        """Gradient accumulation over micro-batches (trainer.micro_batches): slice i of a step writes its gradients into slot i --
        a second, third, ... flat gradient buffer of the same layout, created on first use -- so that slice i's exchange can be in
        flight while slice i + 1 is computed; slot 0 is the buffer param.grad aliases and the one the update reads."""
        while len(self._slots) <= i:
            gp = torch.zeros_like(self._slots[0][0])
            g = gp[:self.numel]
            gv = {n: g[self.offsets[n]:self.offsets[n] + p.numel()].view(p.shape) for n, p in zip(self.names, self.params)}
            self._slots.append((gp, g, gv))
        self.grad_padded, self.grad, self.gviews = self._slots[i]

    def bound(self):
        return all(p.data_ptr() == self.views[n].data_ptr() for n, p in zip(self.names, self.params))

    def attach_grads(self):
        """Make param.grad alias the flat gradient segments (fused driver path)."""
        for n, p in zip(self.names, self.params):
            p.grad = self._slots[0][2][n]
        self.detached = False

    def grads_are_views(self):
        """True when every parameter's .grad IS its segment of the flat gradient buffer (slot 0)."""
        gv = self._slots[0][2]
        return all(p.grad is not None and p.grad.data_ptr() == gv[n].data_ptr() and p.grad.shape == gv[n].shape
                   for n, p in zip(self.names, self.params))

    def before_autograd_backward(self):
        """Called by the drop-in autograd Functions BEFORE the engine writes a backward's gradients into the flat buffer: a .grad
        that still aliases the buffer holds something the caller has not cleared (no zero_grad since the last backward, or a fused
        trainer attached it), which autograd's += must see intact -- it is moved to storage of its own first."""
        gv = self._slots[0][2]
        for n, p in zip(self.names, self.params):
            if p.grad is not None and p.grad.data_ptr() == gv[n].data_ptr():
                p.grad = p.grad.clone()
                self.detached = True          # (a fused trainer re-attaches before its next step)

    # the drop-in path hands gradients out as views of the flat buffer (see deliver_grads); False = always as clones
    flat_grads = True

    @staticmethod
    def _accumulates_into(p):
        """True when the running backward will ACCUMULATE into leaf p's .grad -- i.e. this is a plain `loss.backward()` (or
        `backward(inputs=[..., p, ...])`).  False when the engine will not visit p's accumulation node (`backward(inputs=` without
        p, `autograd.grad` for other tensors); None under `torch.autograd.grad(..., p)`, which must be handed a real tensor to
        capture (torch refuses the query for a leaf in that mode and says so)."""
        try:
            node = torch.autograd.graph.get_gradient_edge(p).node
            return bool(torch._C._will_engine_execute_node(node))
        except (RuntimeError, AttributeError):        # autograd.grad mode -- or a torch without these hooks: hand out real tensors
            return None

    def deliver_grads(self):
        """What a drop-in autograd backward returns for the parameters once the engine has written this backward's gradients into
        the flat buffer, PER PARAMETER:
          * frozen (requires_grad False): None, and its .grad is not touched;
          * a plain `loss.backward()` reaching a parameter whose .grad is None (the state `zero_grad()` leaves, text.py:373): the
            .grad becomes a VIEW of the flat buffer and autograd is handed nothing to accumulate -- no 215 MB of clones per step at
            the Yahoo shape, and lvae.clip_grad_norm_ / lvae.SGD can run as single streaming launches over the flat buffers;
          * a backward that will not accumulate into it (`backward(inputs=...)` without it): None;
          * everything else -- `torch.autograd.grad(loss, params)` (the caller wants tensors back, .grad must stay untouched), a
            .grad that already holds something (accumulation across backward calls, zero_grad(set_to_none=False)), gradients being
            written to a micro-batch slot, `flat_grads = False` -- a clone, with autograd's usual semantics.
        ALIASING (documented contract of the zero-copy route): such a .grad is overwritten in place by the NEXT backward of the
        module; code that keeps a gradient tensor across steps (`g = p.grad; zero_grad(); ...; backward()`) must clone it, or
        switch the route off (`VAE.use_flat_grads(False)`)."""
        slot0 = self.gviews is self._slots[0][2]
        out = []
        for n, p in zip(self.names, self.params):
            if not p.requires_grad:
                out.append(None)
                continue
            acc = self._accumulates_into(p)
            if acc is False:
                out.append(None)
            elif acc and slot0 and self.flat_grads and p.grad is None:
                p.grad = self.gviews[n]
                out.append(None)
            else:
                out.append(self.gviews[n].clone())
        return tuple(out)


def _tensor_bytes(obj, depth=0):
    """Device bytes reachable from a workspace object (tensors in attributes / dict values / lists, nested up to 3 levels)."""
    if torch.is_tensor(obj):
        return obj.numel() * obj.element_size()
    if depth > 3:
        return 0
    if isinstance(obj, dict):
        return sum(_tensor_bytes(v, depth + 1) for v in obj.values())
    if isinstance(obj, (list, tuple)):
        return sum(_tensor_bytes(v, depth + 1) for v in obj)
    d = getattr(obj, "__dict__", None)
    return sum(_tensor_bytes(v, depth + 1) for v in d.values()) if d else 0


# fraction of the device memory one engine's shape-keyed workspaces may hold before the least recently used shapes are dropped
WORKSPACE_BUDGET_FRACTION = 0.2


class _WS(object):
    """Shape-keyed cache of device workspaces, least-recently-used shapes dropped beyond a byte budget.

    A (B, T) batch shape owns ~1.5 GB of activations per engine at the Yahoo dims and real corpora cycle through hundreds of
    sentence lengths (data/text_data.py batches by length): without a bound the cache grows to the whole device (a soak over
    1500 random shapes ran a 288 GB MI355X out of memory).  The shapes of the step in flight are always the most recent, so
    the budget only needs to cover one step.  `evictable = False` (set by the trainers in hipGraph mode, where captured graphs
    hold raw pointers into these buffers) keeps everything: graph mode is meant for a fixed set of shape buckets."""

    def __init__(self, device, budget_bytes=None):
        import collections
        self.device = torch.device(device)
        self.cache = collections.OrderedDict()
        self.nbytes = {}
        self.total = 0
        self.evictable = True
        self.hits = self.misses = self.evictions = 0      # (measurement: bench.py's mixed-shape run reports the hit rate)
        self.before_evict = None      # called once before workspaces are dropped (the engine orders its side streams first)
        if budget_bytes is None:
            if self.device.type == "cuda":
                budget_bytes = int(WORKSPACE_BUDGET_FRACTION * torch.cuda.get_device_properties(self.device).total_memory)
            else:
                budget_bytes = 8 << 30
        self.budget = budget_bytes

    def get(self, key, builder):
        ws = self.cache.get(key)
        if ws is not None:
            self.cache.move_to_end(key)
            self.hits += 1
            return ws
        self.misses += 1
        ws = builder()
        nb = _tensor_bytes(ws)
        self.cache[key] = ws
        self.nbytes[key] = nb
        self.total += nb
        self._trim(key)
        return ws

    def grew(self, key, nbytes):
        """A workspace of `key` allocated another buffer after it was built (lazily created images): count it."""
        if key in self.nbytes:
            self.nbytes[key] += nbytes
            self.total += nbytes
            self._trim(key)

    def _trim(self, key):
        if not self.evictable:
            return
        # drop old shapes (never one touched since the current key's shape was first used in this step: those sit at the end)
        told = False
        while self.total > self.budget and len(self.cache) > 8:
            k = next(iter(self.cache))
            if k == key:
                break
            if not told and self.before_evict is not None:
                # side-stream work of the previous step (weight-gradient GEMMs, token sort) may still read what is dropped here;
                # the caching allocator would hand the memory to the current stream at once
                self.before_evict()
                told = True
            del self.cache[k]
            self.total -= self.nbytes.pop(k)
            self.evictions += 1

    def f32(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self.device)

    def i16(self, *shape):
        return torch.empty(*shape, dtype=torch.int16, device=self.device)

    def i32(self, *shape):
        return torch.empty(*shape, dtype=torch.int32, device=self.device)


class _NS(object):
    pass


class _DecWS(_NS):
    """Decoder workspace; `logits` (f32 [T*B][ldl], 510 MB at the bench shape) is created on first use."""

    @property
    def logits(self):
        if self._logits is None:
            self._logits = self._alloc()
            self._grew(self._logits.numel() * self._logits.element_size())
        return self._logits


# bench.py's roofline leg: when set to a dict, kernel-launch groups are bracketed by HIP events recorded on the launch
# stream and (start, end, work, launches) is appended under the group name (measurement only; None normally).
# work = flops for the GEMM groups, timesteps for the LSTM groups.
PROFILE = None
# ... and only the groups whose name starts with this prefix (None: all).  A HIP event record costs ~5 us of an otherwise gap-free
# queue (kernel trace: profiles/r04d_timeline.txt -- 28 records = 150 us of a 3.5 ms step), so the bench brackets only the
# dominant group inside its timed region and takes the other group's times from a separate, untimed pass.
PROFILE_PREFIX = None


class _prof(object):
    def __init__(self, name, work, launches=1):
        self.name, self.work, self.launches = name, work, launches
        self.on = PROFILE is not None and (PROFILE_PREFIX is None or name.startswith(PROFILE_PREFIX))

    def __enter__(self):
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *a):
        if self.on:
            self.e1.record()
            PROFILE.setdefault(self.name, []).append((self.e0, self.e1, self.work, self.launches))


_GEMM_WS = {}
_PERSISTENT_DEFAULT = True


def _gemm_ws(lib, s):
    """Per-device split-K scratch (64M floats = 256 MB of the 288 GB) shared by every GEMM launch on that device's stream order."""
    dev = torch.device("cuda", torch.cuda.current_device()) if s is not None else torch.device("cpu")
    ws = _GEMM_WS.get(dev)
    if ws is None:
        ws = torch.empty(1 << 26 if dev.type == "cuda" else 1 << 20, dtype=torch.float32, device=dev)
        _GEMM_WS[dev] = ws
    return ws


def _gemm(lib, s, tA, tB, M, N, K, A, lda, B, ldb, C, ldc, alpha=1.0, acc=0, add1=None, ld1=0, mod1=1,
          add2=None, ld2=0, mod2=1, prec="f32", ws=None):
    """prec: 'f32' = exact-f32 MFMA (parity path); 'bf16' = bf16 matrix pipe with f32 accumulate (throughput path,
    only requested for the large contractions)."""
    if ws is None:
        ws = _gemm_ws(lib, s)
    fn = lib.lv_gemm_bf16 if prec == "bf16" else lib.lv_gemm_f32
    with _prof("gemm_" + prec, 2.0 * M * N * K):
        fn(tA, tB, M, N, K, alpha, A, lda, B, ldb, C, ldc, acc, add1, ld1, mod1, add2, ld2, mod2, P(ws), ws.numel(), s)


def _gemm16(lib, s, tA, M, N, K, A16, lda, B16, ldb, C, ldc, add1=None, ld1=0, mod1=1, add2=None, ld2=0, mod2=1, ws=None, acc=0):
    """C = op(A) . B^T (+ add1 + add2) with operands already rounded to bf16 in HBM (lv_gemm_b16): B stored [N][K];
    A stored [M][K] (tA = 0) or [K][M] (tA = 1).  Bit-identical to _gemm(prec='bf16') on the f32 data (up to where
    split-K cuts), at half the operand bytes and with no conversion work in the GEMM."""
    if ws is None:
        ws = _gemm_ws(lib, s)
    with _prof("gemm_bf16", 2.0 * M * N * K):
        lib.lv_gemm_b16(tA, M, N, K, 1.0, A16, lda, B16, ldb, C, ldc, acc, add1, ld1, mod1, add2, ld2, mod2,
                        P(ws), ws.numel(), s)


class _AuxStream(object):
    """A per-engine auxiliary HIP stream for work that depends only on the token ids and must merely be finished by the
    time the embedding gradient is scattered: the stable token sort (one workgroup, ~60 us at the bench shape) and the
    zero fill of the embedding-gradient table.  Runs them beside the latency-bound LSTM chains, which leave almost every
    CU idle; inline on the test backend."""

    def __init__(self):
        self.stream = None
        self.done = None

    def run(self, device, fn, keep=()):
        """keep: tensors fn reads on the side stream (told to the caching allocator, so that a caller that drops them
        right after a forward-only call cannot have them recycled underneath the side stream)."""
        device = torch.device(device)
        # inline on the test backend, and under hipGraph capture: a forked branch in a replayed graph was measured to
        # cost ~0.5 ms per step, far more than the 0.15 ms the side stream saves in eager mode
        if device.type != "cuda" or torch.cuda.is_current_stream_capturing():
            fn(None)
            return
        if self.stream is None:
            self.stream = torch.cuda.Stream(device)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        self.stream.wait_event(ev)
        for t in keep:
            t.record_stream(self.stream)
        with torch.cuda.stream(self.stream):
            fn(self.stream.cuda_stream)
            self.done = torch.cuda.Event()
            self.done.record(self.stream)

    def join(self, device):
        if self.done is not None:
            torch.cuda.current_stream(torch.device(device)).wait_event(self.done)
            self.done = None


class _TokenSortCache(object):
    """The stable (token, row) sort of the embedding backward depends on the token ids alone, and the aggressive loop keeps
    meeting the same batch tensors (text.py:389 draws from the fixed list train_data_batch): the sorted (rows, tokens) pair is
    kept per batch TENSOR -- weakly (it dies with the tensor) and only while the tensor's version counter is unchanged -- so that
    in steady state the two single-workgroup sort launches of a step (75 us each, and nothing can run beside the persistent
    recurrences they used to hide behind) disappear.  Not used under hipGraph capture (a captured step owns fixed buffers).

    CONTRACT: a batch tensor handed to step() is immutable, as the reference's train_data_batch entries are.  In-place torch ops
    bump the version counter and are noticed; writes that bypass it are NOT -- `x.data.copy_(...)`, a numpy view of a CPU tensor,
    DLPack / custom-kernel writes into a reused device buffer: after those call `invalidate(x)` (or pass a fresh tensor), else the
    embedding gradients of both networks are scattered with the old batch's token lists.  At most LIMIT batches are kept, the least
    recently used ones go first."""
    LIMIT = 8192

    def __init__(self):
        import collections
        self.map = collections.OrderedDict()   # id(tensor) -> (weak reference to the tensor, {tag: (version, rows, tokens)})

    def _entries(self, key_tensor, create):
        i = id(key_tensor)
        ent = self.map.get(i)
        if ent is not None and ent[0]() is key_tensor:
            self.map.move_to_end(i)
            return ent[1]
        if not create:
            return None
        import weakref
        while len(self.map) >= self.LIMIT:
            self.map.popitem(last=False)       # least recently used batch (its buffers go back to the caching allocator)
        # (tensors compare element-wise, so they cannot key a WeakKeyDictionary: identity + a weak reference that retires the id)
        self.map[i] = (weakref.ref(key_tensor, lambda _r, i=i, m=self.map: m.pop(i, None)), {})
        return self.map[i][1]

    def get(self, key_tensor, tag):
        ent = self._entries(key_tensor, False)
        hit = ent.get(tag) if ent is not None else None
        return hit[1:] if hit is not None and hit[0] == key_tensor._version else None

    def put(self, key_tensor, tag, srows, stok):
        self._entries(key_tensor, True)[tag] = (key_tensor._version, srows, stok)

    def invalidate(self, key_tensor=None):
        """Forget the sorted lists of one batch tensor (None: of all) -- for callers that rewrote a batch behind the version counter."""
        if key_tensor is None:
            self.map.clear()
        else:
            self.map.pop(id(key_tensor), None)


def reset_persistent_status(eng):
    """Clear the status word (after the caller has handled a reported timeout, e.g. by moving down the fallback ladder)."""
    if eng.status is not None:
        eng.status.zero_()


# ---- the fallback ladder of the persistent recurrences -----------------------------------------------------------------------
# rung 0: persistent launches, hand-off granules kept in the XCD's L2 (persist_flags bit 0) -- the fast form, which ASSUMES that
#         the 32 workgroups of a group (blockIdx % 8) share an XCD: checked by the bounded spins, never proven;
# rung 1: persistent launches with agent-scope (write-through) granules: any placement works as long as all 256 workgroups are
#         resident together (they are not when something else holds compute units, e.g. a collective's kernels);
# rung 2: the launch-per-timestep kernels: no co-residency assumption at all.
# A timeout is reported through the engine's device status word; the fused trainers gate every update on it ON THE DEVICE
# (lv_clip_*_txn_f32), notice it at their next host read, call demote_persistent() and replay the voided steps.
PERSIST_RUNGS = ("persistent, XCD-local hand-off", "persistent, write-through hand-off", "launch-per-timestep kernels")


def persist_rung(eng):
    if not eng.persistent:
        return 2
    return 0 if (eng.persist_flags & 1) else 1


def demote_persistent(eng):
    """Move `eng` one rung down the ladder and clear its status word; returns the new rung.  Raises when there is nothing left
    (the launch-per-timestep kernels have no hand-off that could time out)."""
    r = persist_rung(eng)
    if r == 0:
        eng.persist_flags &= ~1
    elif r == 1:
        eng.persistent = False
    else:
        raise _lib.LvaeError("a persistent-launch timeout was reported although the engine runs the launch-per-timestep kernels")
    reset_persistent_status(eng)
    # whoever demotes (a trainer's _settle, training.guarded_eval): captured hipGraphs hold the launches of the rung that failed --
    # the trainers key their graphs by this counter and re-capture (guarded_eval demoted behind the trainer's back)
    eng.ladder_gen = getattr(eng, "ladder_gen", 0) + 1
    return r + 1


def status_ptr(eng):
    """Device pointer of the engine's persistent-launch status word (None before ensure())."""
    return P(eng.status) if eng.status is not None else None


_PERSIST_H = 1024        # lv_lstm_persist16.hip is built for this hidden size
# The persistent launches (lv_lstm_persist16.hip): <= 16 rows per XCD group, so B <= 128 (BASELINE.json configs[4] runs 16 rows on
# each of the 8 groups); eng.persist_rows asks for a row count per group (e.g. 8 rows on four groups = a B = 32 recurrence on
# half the chip); eng.persist_flags bit 0 keeps the hand-off granules in the XCD's L2 (no agent-scope write-through; measured
# 3.2 -> 2.2 us per forward timestep, 3.1 -> 2.7 us per BPTT timestep at B = 32) -- rung 0 of the ladder above.
_PERSIST_MAX_B = 128


def _persist_rows(eng, B):
    """Rows per XCD group for a persistent launch of batch B on this engine."""
    rows = getattr(eng, "persist_rows", None)
    if rows is not None and 8 * rows >= B and 1 <= rows <= 16:
        return int(rows)
    return (B + 7) // 8


def _saved_floats(eng, T, B, H):
    """Floats of one LSTM layer's saved-activation buffer: gates [T][B][4H] for the step kernels; the persistent kernels keep
    gates AND cell states in a workgroup-major record buffer of their own size."""
    n = T * B * 4 * H
    if H == _PERSIST_H and eng.precision == "bf16" and eng.persistent and B <= _PERSIST_MAX_B:
        n = max(n, eng.lib.lv_lstm_persist16_saved_floats(T, _persist_rows(eng, B)))
    return n


def _persistent_ok(eng, img, B, H, device, max_b):
    return (eng.precision == "bf16" and img is not None and eng.persistent and H == _PERSIST_H and B <= max_b
            and torch.device(device).type == "cuda" and torch.cuda.get_device_properties(device).multi_processor_count >= 256)


def weights_version(eng):
    """Changes whenever the module's weights may have changed: in-place torch updates bump the parameters' version
    counters (optimizer.step, load_state_dict, user edits), raw-pointer updates by the fused trainer bump eng.wgen."""
    return (sum(p._version for p in eng.flat.params), eng.wgen)


def _weight_images(eng, lib, s, device, want_persist, lstm=True, pred=False):
    """Engine-level bf16 images of the module's weights for the throughput path -- W_ih rows in unit-major gate order
    (4u + g: Gx comes out with each unit's (i,f,g,o) side by side), W_ih^T [ni][4H] (contraction index of dX), for the
    decoder also pred_linear.weight / its transpose, and the packed register images of W_hh for the persistent
    recurrences -- rebuilt when weights_version() changed, or on every call when eng.cache_weight_images is off.
    The decoder is frozen for the whole aggressive inner loop (text.py:371-400 steps the encoder only), so its images
    are built once per outer iteration."""
    wi = eng._wimg
    V, ni, H = eng.dims()[:3]
    v = eng.flat.views
    c = eng.wsc
    if wi is None:
        wi = eng._wimg = _NS()
        wi.ver = object()
        wi.W = wi.WT = wi.pred = wi.predT = wi.fwd16 = wi.bwd16 = wi.xch = None
        if lstm:
            wi.W = c.i16(4 * H, ni)
            wi.WT = c.i16(ni, 4 * H)
        wi.ldv = _round_up(V, 32)
        if pred:
            wi.pred = c.i16(V, H)
            wi.predT = c.i16(H, wi.ldv)
        wi.packed16 = False
    ver = weights_version(eng) if eng.cache_weight_images else object()
    h16 = bool(getattr(eng, "_h16_now", False))      # the forward's operand images in IEEE binary16 (LSTMEncoderEngine.fwd_operands)
    stale = wi.ver != ver or getattr(wi, "h16", False) != h16
    if stale:
        wih = v["lstm.weight_ih_l0"]
        if wi.W is not None and h16:
            # W (forward: Gx) as binary16, W^T (backward: dX) as bf16, one launch
            lib.lv_cvt_h16_f32(P(wih), wih.shape[1], 4 * H, ni, H, None, 0, 1, 0, P(wi.W), ni, P(wi.WT), 4 * H, s)
        elif wi.W is not None:
            lib.lv_cvt_bf16_gates_f32(P(wih), wih.shape[1], H, ni, P(wi.W), ni, P(wi.WT), 4 * H, s)
        wi.h16 = h16
        if wi.pred is not None:
            lib.lv_cvt_bf16_f32(P(v["pred_linear.weight"]), H, V, H, P(wi.pred), H, P(wi.predT), wi.ldv, s)
        wi.ver = ver
        wi.packed16 = False
    if want_persist and not wi.packed16:
        if wi.fwd16 is None:
            n = lib.lv_lstm_persist16_wpk_floats()
            wi.fwd16, wi.bwd16 = c.f32(n), c.f32(n)
            # hand-off exchange buffer, zeroed once: the launches alternate its halves and clear the other one themselves
            wi.xch = torch.zeros(lib.lv_lstm_persist16_xch_floats(), dtype=torch.float32, device=c.device)
            wi.xstate = {"f": 0, "g": 0, "gcls": [0, 0]}      # next half per kind of launch; instantiation that last used each BPTT half
        (lib.lv_lstm_persist16_pack2_h16 if h16 else lib.lv_lstm_persist16_pack2)(P(v["lstm.weight_hh_l0"]), P(wi.fwd16), P(wi.bwd16), H, s)
        wi.packed16 = True
    return wi


def _xch_flags(wi, kind, rows, device, T=1):
    """Flag bits 1.. of a persistent launch (lv_lstm_persist16_xch_floats): the exchange buffer's halves alternate per kind of launch
    ("f" forward, "g" BPTT) and every launch clears the other half of its kind itself -- no memset launch in front (4 per training
    step before: ~26 us).  Not under hipGraph capture: a replay repeats the captured half, so captured launches keep half 0 behind
    their memset (and leave the alternation state alone)."""
    st = wi.xstate
    if T <= 0:
        # the launch returns before it touches the exchange (and before it would clear the other half): the alternation must not
        # advance, or the next launch of this kind would poll a half that still holds the tags of the launch before last
        return 0
    if torch.device(device).type == "cuda" and torch.cuda.is_current_stream_capturing():
        st["captured"] = True
    if st.get("captured"):
        # a captured launch dirties half 0 at every replay, unseen from here: once an engine's launches live in a graph, ALL its
        # launches (eager ones too) stay in the memset form
        return 0
    half = st[kind]
    st[kind] = 1 - half
    fl = 2 | (half << 2)
    if kind == "g":
        cls = 1 if rows <= 4 else (2 if rows <= 8 else 3)
        fl |= st["gcls"][1 - half] << 3          # clear the other half over what its last user dirtied (0: never used -> own extent)
        st["gcls"][half] = cls
        st["gcls"][1 - half] = 0
    return fl


def _lstm_forward(eng, lib, s, img, w, Gx, whh, mask, scale, hdrop, T, B, H, device):
    """The forward recurrence of one LSTM layer: exact f32, bf16 launch-per-step, or (bf16 image path on a >= 256-CU
    device, H = 1024, B <= 128) the single persistent launch of lv_lstm_persist16.hip."""
    args = (Gx, whh, P(w.hs), P(w.cs), P(w.gates), mask, scale, hdrop)
    w.saved_layout = ("canonical", T, B, 0)
    if eng.precision != "bf16":
        lib.lv_lstm_fwd_f32(*args, P(w.lstm_ws), T, B, H, s)
    elif img is None:
        lib.lv_lstm_fwd_bf16(*args, P(w.lstm_ws), T, B, H, s)
    elif _persistent_ok(eng, img, B, H, device, _PERSIST_MAX_B):
        wi = eng._wimg          # packed by _weight_images(want_persist=True) at the top of the forward
        rows = _persist_rows(eng, B)
        if mask is not None or hdrop is not None:
            raise _lib.LvaeError("the persistent forward has no in-kernel dropout (the engine applies it on the images)")
        need = lib.lv_lstm_persist16_saved_floats(T, rows)
        if w.gates.numel() < need:              # eng.persist_rows / eng.persistent changed after the workspace was built
            w.gates = torch.empty(need, dtype=torch.float32, device=w.gates.device)
        lib.lv_lstm_fwd_bf16_persist16(Gx, P(wi.fwd16), P(w.hs), P(w.cs), P(w.gates), P(wi.xch), P(eng.status), T, B, rows,
                                       eng.persist_flags | _xch_flags(wi, "f", rows, device, T) | (32 if getattr(wi, "h16", False) else 0), H, s)
        w.saved_layout = ("persist16", T, B, rows)      # what w.gates holds now: the BPTT must be given the same T, B, R
    else:
        lib.lv_lstm_fwd_bf16_ug(*args, P(w.lstm_ws), T, B, H, s)


_PERSIST_BWD_MAX_B = 128


def check_persistent_status(eng):
    """Raise if a persistent LSTM launch of this engine ever reported a hand-off timeout (device status word; one host
    read -- call it where the host synchronises anyway)."""
    st = eng.status
    if st is not None and int(st.item()) != 0:
        raise _lib.LvaeError("persistent LSTM kernel reported hand-off timeout (status %d) on ladder rung %d (%s): not all 256 "
                             "workgroups were resident (e.g. another kernel held compute units for seconds), or -- with "
                             "eng.persist_flags bit 0 set -- a group's workgroups were not placed on one XCD.  The fused trainers "
                             "recover from this by themselves (engine.demote_persistent); on the drop-in autograd path call it and "
                             "redo the step" % (int(st.item()), persist_rung(eng), PERSIST_RUNGS[persist_rung(eng)]))


def _need_canonical_saved(w):
    lay = getattr(w, "saved_layout", None)
    if lay is not None and lay[0] != "canonical":
        raise _lib.LvaeError("the saved activations are in the 16-row persistent kernels' own order %r, but the backward was routed to a "
                             "kernel that reads gates [T][B][4H] / cs: eng.persistent / eng.persist_rows changed between forward and "
                             "backward" % (lay,))


def _lstm_backward(eng, lib, s, img, w, dh_ext, dh_last, mask, scale, whh, dh0, dc0, tanh_init, T, B, H, device):
    """BPTT of one LSTM layer: exact f32, bf16 two launches per step, or the single persistent launch where supported."""
    if eng.precision != "bf16":
        lib.lv_lstm_bwd_f32(dh_ext, dh_last, mask, scale, whh, P(w.gates), P(w.hs), P(w.cs), P(w.dG), P(w.dGsum), P(w.lstm_ws),
                            dh0, dc0, tanh_init, T, B, H, s)
        return
    dG = P(w.dG) if img is None else None
    dG16 = P(img.dG) if img is not None else None
    if _persistent_ok(eng, img, B, H, device, _PERSIST_BWD_MAX_B):
        wi = eng._wimg
        rows = _persist_rows(eng, B)
        if mask is not None:
            raise _lib.LvaeError("the persistent BPTT has no in-kernel dropout mask (the engine applies it on dO)")
        if getattr(w, "saved_layout", None) != ("persist16", T, B, rows):
            raise _lib.LvaeError("the saved activations were not written by a persistent forward with the same T, B and rows per "
                                 "group (%r): eng.persist_rows / eng.persistent changed between forward and backward" % (getattr(w, "saved_layout", None),))
        lib.lv_lstm_bwd_bf16_persist16(dh_ext, dh_last, P(wi.bwd16), P(w.gates), P(w.hs), P(w.cs), dG16, P(w.dGsum), P(wi.xch),
                                       P(eng.status), dh0, dc0, tanh_init, T, B, rows, eng.persist_flags | _xch_flags(wi, "g", rows, device, T), H, s)
    else:
        _need_canonical_saved(w)
        lib.lv_lstm_bwd_bf16_img(dh_ext, dh_last, mask, scale, whh, P(w.gates), P(w.hs), P(w.cs), dG, dG16, P(w.dGsum),
                                 P(w.lstm_ws), dh0, dc0, tanh_init, T, B, H, s)


# dW_ih and dW_hh of an LSTM layer as one product where the shape allows (LVAE_DUAL_WGRAD=0: two products, for A/B measurements)
DUAL_WGRAD = os.environ.get("LVAE_DUAL_WGRAD", "1") != "0"
# ... and dX with them in one grouped stream-K launch on the 256 x 256 tile where the shapes are big enough (LVAE_PAIR_WGRAD=0: the
# separate launches, for A/B measurements)
PAIR_WGRAD = os.environ.get("LVAE_PAIR_WGRAD", "1") != "0"


class _LstmImages(object):
    """bf16 operand images of one LSTM layer's input-side GEMMs (Gx = X.W_ih^T forward; dX = dG.W_ih,
    dW_ih = dG^T.X, dW_hh = dG^T.h_prev backward), built with lv_cvt_bf16_f32 next to the f32 originals."""

    def __init__(self, c, TB, ni, H, key=None):
        self.TB, self.ni, self.H = TB, ni, H
        self.key = key                      # this object's key in the workspace cache (byte accounting of the lazy addend)
        self.ldr = _round_up(TB, 8)
        self.X = c.i16(TB, ni)              # layer input rows            [T*B][ni]
        # X^T [ni][T*B] and h_{t-1}^T [H][T*B] in ONE buffer: the two weight gradients that share dG as their A operand,
        # dW_ih = dG^T X and dW_hh = dG^T h_prev, are then one product over the image [X^T ; h_prev^T] (lv_gemm_b16_dual)
        self.XhT = c.i16(ni + H, self.ldr)
        self.addend = None                  # unit-major copy of the Gx epilogue addend (biases / z-projection)
        self.dG = c.i16(TB, 4 * H)          # gate pre-activation grads   [T*B][4H]

    @property
    def XT(self):                           # layer input rows, transposed [ni][T*B]
        return self.XhT[:self.ni]

    @property
    def hT(self):                           # h_{t-1} rows, transposed     [H][T*B]
        return self.XhT[self.ni:]

    @staticmethod
    def usable(precision, native16, ni, H):
        return precision == "bf16" and native16 and ni % 8 == 0 and H % 8 == 0

    def forward(self, lib, s, X, W16, Gx, add_a, add_b, rows, wsc, addend_um=None, gather=None, h16=False):
        """Gx[r][4u + g] = X[r] . W_ih[g*H + u] + (add_a + add_b)[r % rows][g*H + u]; W16: the unit-major bf16 image of W_ih
        (engine-level, _weight_images); add_a/add_b: gate-major [rows][4H] (add_b may be None), or addend_um: the addend
        already in unit-major order.  gather = (emb, ids, ids_stride, keep, kscale, T, B, V): the layer input is an embedding
        lookup (+ dropout) -- its bf16 images are gathered straight from the table (lv_embed_gather_b16) and X (f32) is not read."""
        TB, ni, H = self.TB, self.ni, self.H
        if addend_um is not None:
            addend = addend_um
        else:
            if self.addend is None or self.addend.shape[0] != rows:
                self.addend = wsc.f32(rows, 4 * H)
                if self.key is not None:
                    wsc.grew(self.key, rows * 4 * H * 4)
            lib.lv_gate_interleave_f32(add_a, add_b, rows, H, P(self.addend), s)
            addend = P(self.addend)
        if h16:
            # binary16 forward operands (W16 is the binary16 image of W_ih): X as binary16 for this product, X^T as bf16 for dW_ih
            emb, ids, ids_stride, keep, kscale, T, B, V = gather
            assert keep is None
            lib.lv_cvt_h16_f32(emb, ni, TB, ni, 0, ids, ids_stride, B, V, P(self.X), ni, P(self.XT), self.ldr, s)
            ws = _gemm_ws(lib, s)
            with _prof("gemm_bf16", 2.0 * TB * 4 * H * ni):
                lib.lv_gemm_h16(TB, 4 * H, ni, 1.0, P(self.X), ni, W16, ni, Gx, 4 * H, 0, addend, 4 * H if rows > 1 else 0, rows, None, 0, 1,
                                P(ws), ws.numel(), s)
            return
        if gather is not None:
            emb, ids, ids_stride, keep, kscale, T, B, V = gather
            lib.lv_embed_gather_b16(emb, ids, ids_stride, keep, kscale, T, B, ni, V, P(self.X), ni, P(self.XT), self.ldr, s)
        else:
            lib.lv_cvt_bf16_f32(X, ni, TB, ni, P(self.X), ni, P(self.XT), self.ldr, s)
        _gemm16(lib, s, 0, TB, 4 * H, ni, P(self.X), ni, W16, ni, Gx, 4 * H,
                add1=addend, ld1=4 * H if rows > 1 else 0, mod1=rows)

    def backward(self, lib, s, dG, h_prev, WT16, dX, gW_ih, ld_gw, gW_hh, ws=None, between=None):
        """dG: the f32 gate gradients, or None when the BPTT kernel already wrote their bf16 image into self.dG; WT16: the
        bf16 image of W_ih^T [ni][4H] (engine-level).  between(): called once dX is queued and before the two weight-gradient
        products (the encoder scatters its embedding gradient there, so that a data-parallel exchange of it can start under
        the weight-gradient GEMMs)."""
        TB, ni, H = self.TB, self.ni, self.H
        if dG is not None:
            lib.lv_cvt_bf16_f32(dG, 4 * H, TB, 4 * H, P(self.dG), 4 * H, None, 0, s)
        wsd = ws if ws is not None else _gemm_ws(lib, s)
        if PAIR_WGRAD and DUAL_WGRAD and lib.lv_gemm_b16_pair_supported(1, 4 * H, ni + H, TB, 0, TB, ni, 4 * H, wsd.numel()):
            # all three products of the layer in ONE grouped launch on the 256 x 256 tile (lv_gemm_b16_pair): [dW_ih | dW_hh] on the
            # share of the CUs its flops ask for, dX on the rest, tiles shared between workgroups summed inside the launch
            lib.lv_cvt_bf16_f32(h_prev, H, TB, H, None, 0, P(self.hT), self.ldr, s)
            with _prof("gemm_bf16", 2.0 * 4 * H * (ni + H) * TB + 2.0 * TB * ni * 4 * H):
                lib.lv_gemm_b16_pair(1, 4 * H, ni + H, TB, P(self.dG), 4 * H, P(self.XhT), self.ldr, gW_ih, ld_gw, ni, gW_hh, H,
                                     0, TB, ni, 4 * H, P(self.dG), 4 * H, WT16, 4 * H, dX, ni, P(wsd), wsd.numel(), s)
            if between is not None:
                between()
            return
        _gemm16(lib, s, 0, TB, ni, 4 * H, P(self.dG), 4 * H, WT16, 4 * H, dX, ni, ws=ws)
        if between is not None:
            between()
        lib.lv_cvt_bf16_f32(h_prev, H, TB, H, None, 0, P(self.hT), self.ldr, s)
        if DUAL_WGRAD and ni % 4 == 0 and lib.lv_gemm_b16_dual_supported(4 * H, ni + H, TB, wsd.numel()):
            # both weight gradients as one product (N = ni + H columns, split between the two destinations in its reduction stage)
            with _prof("gemm_bf16", 2.0 * 4 * H * (ni + H) * TB):
                lib.lv_gemm_b16_dual(1, 4 * H, ni + H, TB, P(self.dG), 4 * H, P(self.XhT), self.ldr, gW_ih, ld_gw, ni, gW_hh, H,
                                     P(wsd), wsd.numel(), s)
            return
        _gemm16(lib, s, 1, 4 * H, ni, TB, P(self.dG), 4 * H, P(self.XT), self.ldr, gW_ih, ld_gw, ws=ws)
        _gemm16(lib, s, 1, 4 * H, H, TB, P(self.dG), 4 * H, P(self.hT), self.ldr, gW_hh, H, ws=ws)


def _wgrad(lib, s, M, N, K, A, lda, Bm, ldb, C, ldc, prec, ws=None):
    """Weight gradient C[M,N] = A^T . B with A stored [K][M] (lda) and B stored [K][N] (ldb).

    Both kernels read the operands as stored (TN form): each wave-level load is 256 contiguous bytes of one k-row.
    (Measured on MI355X: transposing the operands first to use the NT form is SLOWER for the bf16 kernel -- dW_pred
    1.63 ms + 0.27 ms of transposes vs ~0.9 ms as TN; profiles/r01f_*.)"""
    _gemm(lib, s, 1, 0, M, N, K, A, lda, Bm, ldb, C, ldc, prec=prec, ws=ws)


def _sorted_tokens(eng, lib, s, x, x_key, ids_stride, T_used, B, V, w):
    """(rows, tokens) of the embedding backward's stable token sort for batch x: from the engine's per-batch cache when this
    batch tensor has been seen (and not modified) before, else sorted now -- into cache-owned buffers in eager mode, into the
    workspace under hipGraph capture or when the batch has no stable identity."""
    key = x_key if x_key is not None else x
    capturing = x.device.type == "cuda" and torch.cuda.is_current_stream_capturing()
    use_cache = eng.ws_evictable and not capturing           # hipGraph trainers pin the workspaces (ws_evictable False): fixed buffers
    if use_cache:
        hit = eng._sorts.get(key, (T_used, B))
        if hit is not None:
            return hit
        srows, stok = eng.wsc.i32(T_used * B), eng.wsc.i32(T_used * B)
    else:
        srows, stok = w.srows, w.stok
    eng._aux.run(x.device, lambda sa: lib.lv_token_sort(P(x), ids_stride, T_used, B, V, P(srows), P(stok), P(w.stmp), sa if sa is not None else s),
                 keep=(x, srows, stok))
    if use_cache:
        eng._sorts.put(key, (T_used, B), srows, stok)
    return srows, stok


class LSTMEncoderEngine(object):
    """Forward/backward of LSTMEncoder.forward (reference modules/encoders/enc_lstm.py:47-64)."""

    def __init__(self, module):
        self.m = module
        self.flat = None
        self.wsc = None
        self.ws_evictable = True  # False under hipGraph capture (graphs hold pointers into the workspaces)
        self.gen = 0
        self.precision = "f32"    # precision of the large GEMMs: "f32" (parity) or "bf16" (throughput)
        self.native16 = True      # bf16 path: pre-rounded bf16 operand images (lv_gemm_b16) where the shapes allow
        self.persistent = _PERSISTENT_DEFAULT   # bf16 image path: forward recurrence as one persistent launch where supported
        self.persist_rows = None                # rows per XCD group of the persistent launches (None: B / 8; 8 at B = 32 = half the chip)
        self.persist_flags = 1                  # bit 0: hand-off granules stay in the XCD's L2 (ladder rung 0; see demote_persistent)
        self.status = None                      # device int32: a persistent launch's hand-off timeout is reported here
        self.fold = None                        # norm folding (trainer._plan_fold): {"embed": (partials tensor, norm-only flag)}
        self.cache_weight_images = False        # the encoder is stepped every inner iteration: its images are rebuilt per call
        self.wgen = 0                           # bumped by the fused trainer after a raw-pointer weight update
        # bf16 configuration only: which parts of the FORWARD run f32-accurately all the same -- "gx" (the input projection
        # X W_ih^T) and / or "rec" (the recurrence).  mu / logvar, hence z and the KL (encoder.py:55), are functions of the forward's
        # last state alone (enc_lstm.py:60-62); the weights' rounding to bf16 moves it by 2e-4..5e-4 relative (to binary16, the
        # default below: 1e-5..7e-5), the exact forward holds 2e-5 while every gradient product and the decoder stay on the bf16 pipe.
        self.exact_forward = ()
        # how: "auto" = on the bf16 pipe itself wherever the forward recurrence is a persistent launch -- split-bf16 operands for
        # the input projection and a two-pass recurrence whose second pass carries W_lo . h as part of gx (_exact_forward_split) --
        # else, and with "f32", the exact-f32 GEMM and launch-per-timestep recurrence
        self.exact_impl = "auto"
        # bf16 configuration: number format of the FORWARD's matrix-pipe operands (embedded rows, W_ih, W_hh, the h hand-off).
        # "f16" (default): IEEE binary16 -- the same instructions at the same rates with 11 instead of 8 bits of significand.  What
        # moves the forward's last state (hence mu / logvar, z and the KL of encoder.py:55) in the bf16 configuration is the
        # rounding of the WEIGHTS and embeddings, the same perturbation at every timestep (profiles/r05a_kl_ablation.txt: W_hh
        # 1.4e-4, embedding 0.4-1.9e-4, W_ih 0.2-1.2e-4 relative on the KL; the h hand-off <= 2.5e-5); these operands are bounded
        # (weights ~U(-0.01, 0.01)-ish, h in (-1, 1)), so they do not need bf16's exponent range.  The gradient products and the
        # BPTT keep bf16.  "bf16": the round-1..4 arithmetic.
        self.fwd_operands = "f16"
        self._h16_now = False
        self._wimg = None
        self._aux = _AuxStream()
        self._sorts = _TokenSortCache()

    def _quiesce_side_streams(self):
        dev = self.flat.device if self.flat is not None else None
        if dev is not None and dev.type == "cuda" and self._aux.stream is not None:
            torch.cuda.current_stream(dev).wait_stream(self._aux.stream)

    def _b16(self, B, T):
        V, ni, H, nz2 = self.dims()
        if not _LstmImages.usable(self.precision, self.native16, ni, H):
            return None
        return self.wsc.get(("b16", B, T), lambda: _LstmImages(self.wsc, T * B, ni, H, key=("b16", B, T)))

    def _exact(self, img):
        """The parts of the forward that run in exact f32 although the engine is in the bf16 configuration."""
        ex = tuple(self.exact_forward or ())
        assert all(e in ("gx", "rec") for e in ex), ex
        return ex if (img is not None and self.precision == "bf16") else ()

    def fold_parts(self, B, T):
        """Partial sums of squares this engine's backward can emit from the kernels that complete its big gradient tensors
        (name -> number of partials): the embedding table's gradient, from the scatter."""
        return {"embed": self.lib.lv_embed_scatter_sumsq_parts(T, B)}

    def refresh_weight_images(self, B, device):
        """Bring the bf16 weight images (and, where the persistent launches apply, the packed recurrent weights) up to
        date on the current stream; a no-op when caching is on and the weights have not changed."""
        V, ni, H, nz2 = self.dims()
        if not _LstmImages.usable(self.precision, self.native16, ni, H):
            return None
        self.ensure(device)
        persist = self.persistent and H == _PERSIST_H and B <= _PERSIST_MAX_B and torch.device(device).type == "cuda" and \
            torch.cuda.get_device_properties(device).multi_processor_count >= 256
        return _weight_images(self, self.lib, stream_ptr(device), device, persist)

    def ensure(self, device):
        device = torch.device(device)
        if self.flat is None or self.flat.device != device or not self.flat.bound():
            named = [("embed.weight", self.m.embed.weight), ("lstm.weight_ih_l0", self.m.lstm.weight_ih_l0),
                     ("lstm.weight_hh_l0", self.m.lstm.weight_hh_l0), ("lstm.bias_ih_l0", self.m.lstm.bias_ih_l0),
                     ("lstm.bias_hh_l0", self.m.lstm.bias_hh_l0), ("linear.weight", self.m.linear.weight)]
            self.flat = FlatBuffer(named, device)
            self.flat.flat_grads = getattr(self, "_flat_grads_pref", True)      # VAE.use_flat_grads before the buffers existed
            self.wsc = _WS(device)
            self.wsc.evictable = self.ws_evictable
            self.wsc.before_evict = self._quiesce_side_streams
            # the cached bf16 / packed weight images describe the OLD flat buffer (p.data = X, module._apply and device moves keep
            # the parameters' version counters, so weights_version() alone would not notice)
            self._wimg = None
            for p in self.flat.params:
                p._lvae_engine = self         # lets optim.clip_grad_norm_ / optim.SGD find the flat buffers from the parameters
        self.lib = backend_for(device)
        if self.status is None or self.status.device != device:
            self.status = torch.zeros(1, dtype=torch.int32, device=device)
        return self.flat

    def dims(self):
        V, ni = self.m.embed.weight.shape
        H = self.m.lstm.weight_hh_l0.shape[1]
        nz2 = self.m.linear.weight.shape[0]
        return V, ni, H, nz2

    def _ws(self, B, T):
        V, ni, H, nz2 = self.dims()
        c = self.wsc

        def build():
            w = _NS()
            w.X = c.f32(T * B, ni)
            w.Gx = c.f32(T * B, 4 * H)
            # index 0 = the initial state: zero for the encoder (enc_lstm.py:60), and no kernel ever writes slot 0
            w.hs = torch.zeros(T + 1, B, H, dtype=torch.float32, device=c.device)
            w.cs = torch.zeros(T + 1, B, H, dtype=torch.float32, device=c.device)
            w.gates = c.f32(_saved_floats(self, T, B, H))
            w.mulv = c.f32(B, nz2)
            w.dmulv = c.f32(B, nz2)
            w.dG = c.f32(T * B, 4 * H)
            w.dGsum = c.f32(B, 4 * H)
            w.lstm_ws = c.f32(self.lib.lv_lstm_ws_floats(B, H))
            w.dhT = c.f32(B, H)
            w.dX = c.f32(T * B, ni)
            w.srows = c.i32(T * B)
            w.stok = c.i32(T * B)
            w.stmp = c.i32(2 * T * B)
            return w
        return c.get((B, T), build)

    def forward(self, x, head=None, x_key=None):
        """x int64 [B][T] on device -> mulv [B][2nz] (mu | logvar).  Keeps activations for backward().

        head = (eps [B][ns][nz], z, kl): also reparameterise and compute the KL in the head's launch (fused driver).
        x_key: the batch tensor x was copied from (fused driver: x is its per-shape static buffer), whose identity keys the
        cache of sorted token lists; default x itself."""
        assert x.dtype == torch.int64 and x.dim() == 2
        x = x.contiguous()
        B, T = x.shape
        f = self.ensure(x.device)
        lib, s = self.lib, stream_ptr(x.device)
        V, ni, H, nz2 = self.dims()
        w = self._ws(B, T)
        v = f.views
        img = self._b16(B, T)
        exact = self._exact(img)
        self._h16_now = img is not None and not exact and self.fwd_operands == "f16"
        split = (self.exact_impl == "auto" and set(exact) == {"gx", "rec"} and _persistent_ok(self, img, B, H, x.device, _PERSIST_MAX_B))
        if img is None or ("gx" in exact and not split):
            lib.lv_embed_gather_f32(P(v["embed.weight"]), P(x), T, None, 1.0, P(w.X), T, B, ni, V, s)      # f32 rows: the exact-f32 / fallback GEMM's operand
        # the backward's token sort depends on x only: taken from the per-batch cache, or queued now (auxiliary stream)
        self._sort = _sorted_tokens(self, lib, s, x, x_key, T, T, B, V, w)
        biases = dict(add1=P(v["lstm.bias_ih_l0"]), ld1=0, mod1=1, add2=P(v["lstm.bias_hh_l0"]), ld2=0, mod2=1)
        gx_unit_major = True
        if split:
            self._exact_forward_split(lib, s, img, w, x, T, B, V, ni, H)
        elif img is not None and "gx" in exact:
            # exact-f32 input projection (gate-major columns, as the f32 recurrence reads them); the bf16 images of the embedded
            # rows are still gathered: the backward's dW_ih product reads them
            self.refresh_weight_images(B, x.device)
            lib.lv_embed_gather_b16(P(v["embed.weight"]), P(x), T, None, 1.0, T, B, ni, V, P(img.X), ni, P(img.XT), img.ldr, s)
            _gemm(lib, s, 0, 1, T * B, 4 * H, ni, P(w.X), ni, P(v["lstm.weight_ih_l0"]), ni, P(w.Gx), 4 * H, prec="f32", **biases)
            gx_unit_major = False
            if "rec" not in exact:          # (measurement only: the bf16 recurrences read a unit-major gx)
                if getattr(w, "Gx_gm", None) is None:
                    w.Gx_gm = self.wsc.f32(T * B, 4 * H)
                w.Gx, w.Gx_gm = w.Gx_gm, w.Gx
                lib.lv_gate_interleave_f32(P(w.Gx_gm), None, T * B, H, P(w.Gx), s)
                gx_unit_major = True
        elif img is not None:
            wi = self.refresh_weight_images(B, x.device)
            img.forward(lib, s, None, P(wi.W), P(w.Gx), P(v["lstm.bias_ih_l0"]), P(v["lstm.bias_hh_l0"]), 1, self.wsc,
                        gather=(P(v["embed.weight"]), P(x), T, None, 1.0, T, B, V), h16=self._h16_now)
        else:
            _gemm(lib, s, 0, 1, T * B, 4 * H, ni, P(w.X), ni, P(v["lstm.weight_ih_l0"]), ni, P(w.Gx), 4 * H,
                  prec=self.precision, **biases)
        gx_round = getattr(self, "gx_round", None)
        if gx_round is not None and not split:
            # MEASUREMENT ONLY (profiles/microbench/kl_ablation.py): what a 16-bit Gx image would cost the KL -- the f32 Gx rounded to
            # binary16 / bf16 in place before the recurrence reads it (never set by the product)
            w.Gx.copy_(w.Gx.to(torch.float16 if gx_round == "f16" else torch.bfloat16).to(torch.float32))
        if split:
            pass
        elif "rec" in exact:
            self._exact_recurrence(lib, s, img, w, gx_unit_major, T, B, H, x.device)
        else:
            with _prof("lstm_fwd_enc", float(T), 1 if _persistent_ok(self, img, B, H, x.device, _PERSIST_MAX_B) else T):
                _lstm_forward(self, lib, s, img, w, P(w.Gx), P(v["lstm.weight_hh_l0"]), None, 1.0, None, T, B, H, x.device)
        if head is not None and not fused_ends_ok(B, nz2 // 2, head[0].shape[1]):
            eps, z, kl = head
            _gemm(lib, s, 0, 1, B, nz2, H, P(w.hs, T * B * H), H, P(v["linear.weight"]), H, P(w.mulv), nz2)
            lib.lv_reparam_kl_fwd_f32(P(w.mulv), P(eps), P(z), P(kl), B, eps.shape[1], nz2 // 2, s)
        elif head is not None:
            eps, z, kl = head
            lib.lv_enc_head_fwd_f32(P(w.hs, T * B * H), P(v["linear.weight"]), P(eps), P(w.mulv), P(z), P(kl), B, H,
                                    eps.shape[1], nz2 // 2, s)
        else:
            _gemm(lib, s, 0, 1, B, nz2, H, P(w.hs, T * B * H), H, P(v["linear.weight"]), H, P(w.mulv), nz2)
        self.gen += 1
        self.last = (x, B, T, self.gen)
        return w.mulv

    def _exact_forward_split(self, lib, s, img, w, x, T, B, V, ni, H):
        """The encoder's forward with f32-like WEIGHTS on the bf16 pipe (exact_forward = gx + rec where the recurrence is a
        persistent launch).  What moves the forward's last state -- hence mu / logvar, z and the KL of encoder.py:55 -- in the
        bf16 configuration is the rounding of the weights, a perturbation that acts the same way at every timestep (measured,
        profiles/r05a_kl_ablation.txt: W_hh alone 1.4e-4 relative on the KL, the embedding rows 0.4e-4..1.9e-4, W_ih 0.2e-4..1.2e-4),
        not the rounding of h_{t-1} in the hand-off (<= 2.5e-5: fresh noise at every step, which averages out).  So:
          * input projection: every operand split into hi + lo (both bf16; x = hi + lo up to 2^-17 |x|) and
            Gx = X_hi W_hi^T + X_lo W_hi^T + X_hi W_lo^T as ONE product over K = 3 ni ([X_hi | X_lo | X_hi] . [W_hi | W_hi | W_lo]^T);
          * recurrence: pass 1 = the persistent bf16 recurrence as it is -> h(1); then Gx += bf16(h(1)_{t-1}) . W_hh,lo^T for all t
            at once (one GEMM); pass 2 = the same persistent launch on the corrected Gx.  Pass 2 computes W_hi . bf16(h(2)) +
            W_lo . bf16(h(1)); against the exact W . h(2) that leaves W_lo . (h(1) - h(2)) ~ 2^-9 x 1e-3 and the hand-off rounding.
        Everything the BPTT reads (gate records, cell states, hs) is what pass 2 wrote, in the persistent kernels' own layout."""
        v = self.flat.views
        TB = T * B
        wi = self.refresh_weight_images(B, x.device)
        c = self.wsc
        if getattr(wi, "Ww", None) is None:
            wi.Ww = c.i16(4 * H, 3 * ni)           # [W_hi | W_hi | W_lo], rows unit-major
            wi.Whh_lo = c.i16(4 * H, H)            # low half of W_hh, rows unit-major (the column order of Gx)
        wih, whh = v["lstm.weight_ih_l0"], v["lstm.weight_hh_l0"]
        lib.lv_cvt_bf16_gates_f32(P(wih), wih.shape[1], H, ni, P(wi.Ww), 3 * ni, None, 0, s)
        lib.lv_cvt_bf16_gates_f32(P(wih), wih.shape[1], H, ni, P(wi.Ww, ni), 3 * ni, None, 0, s)
        lib.lv_cvt_bf16_lo_f32(P(wih), wih.shape[1], 4 * H, ni, H, None, 0, 1, 0, P(wi.Ww, 2 * ni), 3 * ni, None, 0, s)
        lib.lv_cvt_bf16_lo_f32(P(whh), H, 4 * H, H, H, None, 0, 1, 0, P(wi.Whh_lo), H, None, 0, s)
        if getattr(img, "Xw", None) is None:
            img.Xw = c.i16(TB, 3 * ni)             # [X_hi | X_lo | X_hi]
            img.h16 = c.i16(TB, H)                 # bf16(h(1)_{t-1}), rows t*B + b
            if img.key is not None:
                c.grew(img.key, TB * (3 * ni + H) * 2)
        emb = P(v["embed.weight"])
        lib.lv_embed_gather_b16(emb, P(x), T, None, 1.0, T, B, ni, V, P(img.Xw), 3 * ni, P(img.XT), img.ldr, s)      # (+ X^T for dW_ih)
        lib.lv_cvt_bf16_lo_f32(emb, ni, TB, ni, 0, P(x), T, B, V, P(img.Xw, ni), 3 * ni, None, 0, s)
        lib.lv_embed_gather_b16(emb, P(x), T, None, 1.0, T, B, ni, V, P(img.Xw, 2 * ni), 3 * ni, None, 0, s)
        if img.addend is None or img.addend.shape[0] != 1:
            img.addend = c.f32(1, 4 * H)
        lib.lv_gate_interleave_f32(P(v["lstm.bias_ih_l0"]), P(v["lstm.bias_hh_l0"]), 1, H, P(img.addend), s)
        _gemm16(lib, s, 0, TB, 4 * H, 3 * ni, P(img.Xw), 3 * ni, P(wi.Ww), 3 * ni, P(w.Gx), 4 * H, add1=P(img.addend), ld1=0, mod1=1)
        with _prof("lstm_fwd_enc", float(T), 1):
            _lstm_forward(self, lib, s, img, w, P(w.Gx), P(whh), None, 1.0, None, T, B, H, x.device)
        lib.lv_cvt_bf16_f32(P(w.hs), H, TB, H, P(img.h16), H, None, 0, s)
        _gemm16(lib, s, 0, TB, 4 * H, H, P(img.h16), H, P(wi.Whh_lo), H, P(w.Gx), 4 * H, acc=1)
        with _prof("lstm_fwd_enc", float(T), 1):
            _lstm_forward(self, lib, s, img, w, P(w.Gx), P(whh), None, 1.0, None, T, B, H, x.device)

    def _exact_recurrence(self, lib, s, img, w, gx_unit_major, T, B, H, device):
        """The forward recurrence on the exact-f32 launch-per-timestep kernels inside the bf16 configuration (exact_forward has
        "rec").  They save gates [T][B][H][4] / cs [T+1][B][H]; where the BPTT will be the persistent launch, those are copied
        into its workgroup-major record buffer (lv_lstm_persist16_import_saved) -- the same T, B, R contract as a persistent
        forward."""
        v = self.flat.views
        n = T * B * 4 * H
        if getattr(w, "gates_canon", None) is None or w.gates_canon.numel() < n:
            w.gates_canon = self.wsc.f32(n)
        fwd = lib.lv_lstm_fwd_f32_ug if gx_unit_major else lib.lv_lstm_fwd_f32
        persist_bwd = _persistent_ok(self, img, B, H, device, _PERSIST_BWD_MAX_B)
        gates = w.gates_canon if persist_bwd else w.gates
        with _prof("lstm_fwd_enc", float(T), T):
            fwd(P(w.Gx), P(v["lstm.weight_hh_l0"]), P(w.hs), P(w.cs), P(gates), None, 1.0, None, P(w.lstm_ws), T, B, H, s)
        w.saved_layout = ("canonical", T, B, 0)
        if persist_bwd:
            rows = _persist_rows(self, B)
            need = lib.lv_lstm_persist16_saved_floats(T, rows)
            if w.gates.numel() < need:
                w.gates = torch.empty(need, dtype=torch.float32, device=w.gates.device)
            lib.lv_lstm_persist16_import_saved(P(w.gates_canon), P(w.cs), P(w.gates), T, B, rows, H, s)
            w.saved_layout = ("persist16", T, B, rows)

    def backward(self, dmulv, gen=None, head=None, after_bptt=None, after_embed=None):
        """dmulv [B][2nz] -> fills self.flat.grad (all encoder parameter grads, '=' semantics).

        after_bptt: called once the BPTT recurrence has been queued (data parallel: the point where a collective can be issued
        so that it starts behind the persistent launch and runs under the weight-gradient GEMMs that follow).
        after_embed: called once the embedding gradient -- the first 41 MB of the flat gradient buffer at the Yahoo shape -- has
        been queued; the LSTM weight-gradient GEMMs follow it (data parallel: that bucket's all-reduce runs under them).

        head = (eps, dz [parts][B][ns][nz], parts, dkl [B]) instead of dmulv: the backward of reparameterise + KL runs in
        the head's launch (fused driver; dmulv is then produced into the workspace)."""
        x, B, T, g = self.last
        if gen is not None and gen != g:
            raise _lib.LvaeError("encoder activations were overwritten by a later forward(); the HIP engine keeps "
                                 "one in-flight step per module")
        f = self.flat
        lib, s = self.lib, stream_ptr(x.device)
        V, ni, H, nz2 = self.dims()
        w = self._ws(B, T)
        v, gv = f.views, f.gviews
        if head is not None and not fused_ends_ok(B, nz2 // 2, head[0].shape[1]):
            eps, dz, parts, dkl = head
            assert parts == 1
            lib.lv_reparam_kl_bwd_f32(P(w.mulv), P(eps), P(dz), P(dkl), P(w.dmulv), B, eps.shape[1], nz2 // 2, s)
            _gemm(lib, s, 0, 0, B, H, nz2, P(w.dmulv), nz2, P(v["linear.weight"]), H, P(w.dhT), H)
            _gemm(lib, s, 1, 0, nz2, H, B, P(w.dmulv), nz2, P(w.hs, T * B * H), H, P(gv["linear.weight"]), H)
        elif head is not None:
            eps, dz, parts, dkl = head
            lib.lv_enc_head_bwd_f32(P(w.mulv), P(eps), P(dz), parts, P(dkl), P(w.hs, T * B * H), P(v["linear.weight"]), P(w.dmulv),
                                    P(w.dhT), P(gv["linear.weight"]), B, H, eps.shape[1], nz2 // 2, s)
        else:
            dmulv = dmulv.contiguous()
            # head: dh_T = dmulv . W_lin ; dW_lin = dmulv^T . h_T
            _gemm(lib, s, 0, 0, B, H, nz2, P(dmulv), nz2, P(v["linear.weight"]), H, P(w.dhT), H)
            _gemm(lib, s, 1, 0, nz2, H, B, P(dmulv), nz2, P(w.hs, T * B * H), H, P(gv["linear.weight"]), H)
        img = self._b16(B, T)
        with _prof("lstm_bwd_enc", float(T), 1 if _persistent_ok(self, img, B, H, x.device, _PERSIST_BWD_MAX_B) else 2 * T):
            _lstm_backward(self, lib, s, img, w, None, P(w.dhT), None, 1.0, P(v["lstm.weight_hh_l0"]), None, None, 0, T, B, H, x.device)
        if after_bptt is not None:
            after_bptt()
        def embed_grad():
            # dX -> embedding rows (the embedding table leads the flat buffer: its gradient is the first, and largest, bucket)
            self._aux.join(x.device)                   # token sort queued by forward()
            # every row of the table's gradient is written by this launch (zeros where a token does not occur): no fill in front
            if self.fold and "embed" in self.fold:
                sq, only = self.fold["embed"]         # the table gradient's squares go to the clip kernel from here (no second read)
                lib.lv_embed_scatter_full_sumsq_f32(P(w.dX), None, 1.0, P(self._sort[0]), P(self._sort[1]), T, B, P(gv["embed.weight"]), ni,
                                                    V, -1, P(sq), int(only), s)
            else:
                lib.lv_embed_scatter_full_f32(P(w.dX), None, 1.0, P(self._sort[0]), P(self._sort[1]), T, B, P(gv["embed.weight"]), ni, V, -1, s)
            if after_embed is not None:
                after_embed()
        # input-side grads: dX first, then the embedding scatter, then the two weight-gradient products
        if img is not None:
            img.backward(lib, s, None, P(w.hs), P(self._wimg.WT), P(w.dX), P(gv["lstm.weight_ih_l0"]), ni, P(gv["lstm.weight_hh_l0"]),
                         between=embed_grad)
        else:
            _gemm(lib, s, 0, 0, T * B, ni, 4 * H, P(w.dG), 4 * H, P(v["lstm.weight_ih_l0"]), ni, P(w.dX), ni, prec=self.precision)
            embed_grad()
            _wgrad(lib, s, 4 * H, ni, T * B, P(w.dG), 4 * H, P(w.X), ni, P(gv["lstm.weight_ih_l0"]), ni, self.precision)
            _wgrad(lib, s, 4 * H, H, T * B, P(w.dG), 4 * H, P(w.hs), H, P(gv["lstm.weight_hh_l0"]), H, self.precision)
        lib.lv_colsum_f32(P(w.dGsum), 4 * H, B, 4 * H, P(gv["lstm.bias_ih_l0"]), P(gv["lstm.bias_hh_l0"]), s)


class LSTMDecoderEngine(object):
    """Forward/backward of LSTMDecoder.reconstruct_error (reference modules/decoders/dec_lstm.py:66-148)."""

    def __init__(self, module):
        self.m = module
        self.flat = None
        self.wsc = None
        self.ws_evictable = True  # False under hipGraph capture (graphs hold pointers into the workspaces)
        self.gen = 0
        self.precision = "f32"    # precision of the large GEMMs: "f32" (parity) or "bf16" (throughput)
        # The BPTT chains are latency-bound (one small launch per timestep) and leave most CUs idle: the decoder's
        # weight-gradient GEMMs run on a side HIP stream underneath them (dW_pred under the decoder BPTT; dX / dW_ih /
        # dW_hh / embedding scatter under the encoder's backward).  join() orders them before anything reads the grads.
        self.overlap = None       # None = auto policy (_overlap_on); True / False force it
        self.native16 = True      # bf16 path: feed the vocabulary-sized GEMMs pre-rounded bf16 operand images (lv_gemm_b16)
        self.persistent = _PERSISTENT_DEFAULT   # bf16 image path: forward recurrence as one persistent launch where supported
        self.persist_rows = None                # rows per XCD group of the persistent launches (see LSTMEncoderEngine)
        self.persist_flags = 1
        self.status = None
        self.fold = None                        # norm folding: {"embed": (partials, only), "pred": (partials, only)}
        # The decoder is frozen for the whole aggressive inner loop (text.py:371-400 steps the encoder only): its bf16
        # weight images and packed recurrent weights are rebuilt only when weights_version() changes.
        self.cache_weight_images = True
        # bf16 image path: vocabulary projection fused with the NLL statistics, binary16 logits (lv_gemm_b16_nll)
        self.fused_nll = True
        self.wgen = 0
        self._wimg = None
        self._side = None
        self._side_ws = None
        self._pending = None
        self._aux = _AuxStream()
        self._sorts = _TokenSortCache()

    def _quiesce_side_streams(self):
        dev = self.flat.device if self.flat is not None else None
        if dev is not None and dev.type == "cuda":
            for st in (self._aux.stream, self._side):
                if st is not None:
                    torch.cuda.current_stream(dev).wait_stream(st)

    def _fork(self, device):
        """Returns a context manager that runs its body on the side stream, ordered after everything queued so far
        on the current stream (inline when overlap is off or on the test backend)."""
        import contextlib
        if not (torch.device(device).type == "cuda" and self._overlap_on()):
            return contextlib.nullcontext(), None
        if self._side is None:
            self._side = torch.cuda.Stream(device)
            self._side_ws = torch.empty(1 << 26, dtype=torch.float32, device=device)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        self._side.wait_event(ev)
        return torch.cuda.stream(self._side), self._side_ws

    def _overlap_on(self):
        if self.overlap is not None:
            return bool(self.overlap)
        # auto (measured on MI355X): on for the f32 path (-2.2 ms/step); on the bf16 path the side
        # stream is worth <= 0.1 ms in eager mode, costs 0.5 ms inside a captured hipGraph and blurs the per-kernel
        # timings the bench reports, so it stays off there
        return self.precision == "f32"

    def _mark_pending(self, device):
        if self._side is not None and torch.device(device).type == "cuda" and self._overlap_on():
            ev = torch.cuda.Event()
            ev.record(self._side)
            self._pending = ev

    def join(self):
        """Order the current stream after the side-stream gradient work of the last backward()."""
        if self._pending is not None:
            torch.cuda.current_stream().wait_event(self._pending)
            self._pending = None

    def ensure(self, device):
        device = torch.device(device)
        if self.flat is None or self.flat.device != device or not self.flat.bound():
            m = self.m
            named = [("embed.weight", m.embed.weight), ("trans_linear.weight", m.trans_linear.weight),
                     ("lstm.weight_ih_l0", m.lstm.weight_ih_l0), ("lstm.weight_hh_l0", m.lstm.weight_hh_l0),
                     ("lstm.bias_ih_l0", m.lstm.bias_ih_l0), ("lstm.bias_hh_l0", m.lstm.bias_hh_l0),
                     ("pred_linear.weight", m.pred_linear.weight)]
            self.flat = FlatBuffer(named, device)
            self.flat.flat_grads = getattr(self, "_flat_grads_pref", True)      # VAE.use_flat_grads before the buffers existed
            self.wsc = _WS(device)
            self.wsc.evictable = self.ws_evictable
            self.wsc.before_evict = self._quiesce_side_streams
            # the cached bf16 / packed weight images describe the OLD flat buffer (p.data = X, module._apply and device moves keep
            # the parameters' version counters, so weights_version() alone would not notice)
            self._wimg = None
            for p in self.flat.params:
                p._lvae_engine = self         # lets optim.clip_grad_norm_ / optim.SGD find the flat buffers from the parameters
        self.lib = backend_for(device)
        if self.status is None or self.status.device != device:
            self.status = torch.zeros(1, dtype=torch.int32, device=device)
        return self.flat

    def dims(self):
        V, ni = self.m.embed.weight.shape
        H = self.m.lstm.weight_hh_l0.shape[1]
        nz = self.m.trans_linear.weight.shape[1]
        return V, ni, H, nz

    def _ws(self, Bd, Td):
        V, ni, H, nz = self.dims()
        c = self.wsc
        ldl = _round_up(V, 32)

        def build():
            w = _DecWS()
            w.ldl = ldl
            w._alloc = lambda: c.f32(Td * Bd, ldl)
            w._grew = lambda nb: c.grew((Bd, Td), nb)
            w.X = c.f32(Td * Bd, ni)
            w.Zp = c.f32(Bd, 4 * H)
            w.Gx = c.f32(Td * Bd, 4 * H)
            w.hs = c.f32(Td + 1, Bd, H)
            w.cs = c.f32(Td + 1, Bd, H)
            w.gates = c.f32(_saved_floats(self, Td, Bd, H))
            w.O = c.f32(Td * Bd, H)
            w._logits = None                 # f32 logits image: allocated on first use (the fused bf16 route never needs it)
            w.lse = c.f32(Td * Bd)
            w.nll = c.f32(Td * Bd)
            w.rec = c.f32(Bd)
            w.dO = c.f32(Td * Bd, H)
            w.dG = c.f32(Td * Bd, 4 * H)
            w.dGsum = c.f32(Bd, 4 * H)
            w.lstm_ws = c.f32(self.lib.lv_lstm_ws_floats(Bd, H))
            w.dc0 = c.f32(Bd, H)
            w.dX = c.f32(Td * Bd, ni)
            w.dz = c.f32(Bd, nz)
            w.dz_parts = self.lib.lv_dec_tail_parts(H)
            w.dzp = c.f32(w.dz_parts, Bd, nz)
            w.srows = c.i32(Td * Bd)
            w.stok = c.i32(Td * Bd)
            w.stmp = c.i32(2 * Td * Bd)
            w.zero1 = torch.zeros(1, dtype=torch.float32, device=c.device)
            w.klz = torch.zeros(Bd, dtype=torch.float32, device=c.device)
            w.loss = c.f32(Bd)
            return w
        return c.get((Bd, Td), build)

    def fold_parts(self, B, Td):
        """As TextEncoderEngine.fold_parts: the embedding table's gradient from the scatter and, where the bf16 product takes the
        256 x 256 tile, dW_pred from its own epilogue (the ws size enters the tile plan, so it must be the launch's)."""
        out = {"embed": self.lib.lv_embed_scatter_sumsq_parts(Td, B)}
        V, ni, H, nz = self.dims()
        if self._b16(B, Td) is not None:
            dev = self.flat.device
            on_side = torch.device(dev).type == "cuda" and self._overlap_on()
            ws_n = (1 << 26) if on_side else _gemm_ws(self.lib, stream_ptr(dev)).numel()
            n = self.lib.lv_gemm_b16_sumsq_parts(V, H, Td * B, ws_n)
            if n > 0:
                out["pred"] = n
        return out

    def _b16(self, Bd, Td):
        """bf16 images of the vocabulary-sized GEMMs' operands (throughput path), or None when the shapes do not meet
        lv_gemm_b16's 16-byte row alignment (then lv_gemm_bf16 rounds the f32 operands on the fly)."""
        V, ni, H, nz = self.dims()
        if self.precision != "bf16" or not self.native16 or H % 8 != 0:
            return None
        c = self.wsc

        def build():
            b = _NS()
            b.ldr = _round_up(Td * Bd, 8)
            b.ldv = _round_up(V, 32)
            b.O = c.i16(Td * Bd, H)           # dropout(h_t) rows      [T*B][H]
            b.OT = c.i16(H, b.ldr)            # ... transposed         [H][T*B]
            # (pred_linear.weight [V][H] and its transpose [H][V]: engine-level images, _weight_images)
            b.dl = c.i16(Td * Bd, b.ldv)      # dlogits                [T*B][V]
            # fused projection + NLL statistics (lv_gemm_b16_nll): binary16 logits image, per-piece (max, sum exp), target logit
            b.l16 = c.i16(Td * Bd, b.ldv)
            b.nparts = self.lib.lv_gemm_b16_nll_parts(V)
            b.part = c.f32(Td * Bd, 2 * b.nparts)
            b.tgt = c.f32(Td * Bd)
            return b
        return c.get(("b16", Bd, Td), build)

    def refresh_weight_images(self, B, device):
        """See LSTMEncoderEngine.refresh_weight_images; covers pred_linear.weight as well."""
        V, ni, H, nz = self.dims()
        use_pred = self.precision == "bf16" and self.native16 and H % 8 == 0
        use_lstm = _LstmImages.usable(self.precision, self.native16, ni, H)
        if not (use_pred or use_lstm):
            return None
        self.ensure(device)
        persist = use_lstm and self.persistent and H == _PERSIST_H and B <= _PERSIST_MAX_B and \
            torch.device(device).type == "cuda" and torch.cuda.get_device_properties(device).multi_processor_count >= 256
        return _weight_images(self, self.lib, stream_ptr(device), device, persist, lstm=use_lstm, pred=use_pred)

    def _lstm_images(self, Bd, Td):
        V, ni, H, nz = self.dims()
        if not _LstmImages.usable(self.precision, self.native16, ni, H):
            return None
        return self.wsc.get(("b16lstm", Bd, Td), lambda: _LstmImages(self.wsc, Td * Bd, ni, H, key=("b16lstm", Bd, Td)))

    def forward(self, x, z, mask_in, mask_out, p_in, p_out, want_rec=True, x_key=None):
        """x int64 [B][T]; z [B][1][nz] (ns = 1 on the HIP path); masks uint8 keep-masks in the reference's
        batch-first layout ([B][T-1][ni], [B][T-1][H]) or None (eval mode).  Returns rec [B].  x_key: see LSTMEncoderEngine.forward."""
        assert x.dtype == torch.int64 and x.dim() == 2
        x = x.contiguous()
        B, T = x.shape
        Td = T - 1
        if z.dim() != 3 or z.shape[1] != 1:
            raise _lib.LvaeError("the HIP decoder path takes nsamples == 1 (z of shape [B,1,nz]); got %s" % (tuple(z.shape),))
        z2 = z.reshape(B, -1).contiguous()
        f = self.ensure(x.device)
        lib, s = self.lib, stream_ptr(x.device)
        V, ni, H, nz = self.dims()
        w = self._ws(B, Td)
        v = f.views
        sc_in = 1.0 / (1.0 - p_in) if mask_in is not None else 1.0
        sc_out = 1.0 / (1.0 - p_out) if mask_out is not None else 1.0
        if mask_in is not None:
            assert mask_in.dtype == torch.uint8 and tuple(mask_in.shape) == (B, Td, ni) and mask_in.is_contiguous()
        if mask_out is not None:
            assert mask_out.dtype == torch.uint8 and tuple(mask_out.shape) == (B, Td, H) and mask_out.is_contiguous()
        gather = (P(v["embed.weight"]), P(x), T, P(mask_in), sc_in, Td, B, V)
        if self._lstm_images(B, Td) is None:
            lib.lv_embed_gather_f32(P(v["embed.weight"]), P(x), T, P(mask_in), sc_in, P(w.X), Td, B, ni, V, s)
        self._sort = _sorted_tokens(self, lib, s, x, x_key, T, Td, B, V, w)
        # c0 = z W_trans^T ; h0 = tanh(c0) (dec_lstm.py:99-101) ; Zp = z W_ih[:, ni:]^T + b_ih + b_hh, so that
        # Gx = X W_ih[:, :ni]^T + Zp[b]   (cat((word_embed, z_)) never materialised) -- one launch
        wih = v["lstm.weight_ih_l0"]
        img = self._lstm_images(B, Td)
        fused = fused_ends_ok(B, nz)
        if fused:
            lib.lv_dec_init_f32(P(z2), P(v["trans_linear.weight"]), P(wih), ni + nz, ni, P(v["lstm.bias_ih_l0"]),
                                P(v["lstm.bias_hh_l0"]), P(w.cs), P(w.hs), P(w.Zp), 1 if img is not None else 0, B, H, nz, s)
        else:
            _gemm(lib, s, 0, 1, B, H, nz, P(z2), nz, P(v["trans_linear.weight"]), nz, P(w.cs), H)
            lib.lv_tanh_f32(P(w.cs), P(w.hs), B * H, s)
            _gemm(lib, s, 0, 1, B, 4 * H, nz, P(z2), nz, P(wih, ni), ni + nz, P(w.Zp), 4 * H,
                  add1=P(v["lstm.bias_ih_l0"]), ld1=0, mod1=1, add2=P(v["lstm.bias_hh_l0"]), ld2=0, mod2=1)
        if img is not None:
            wi = self.refresh_weight_images(B, x.device)
            if fused:
                img.forward(lib, s, None, P(wi.W), P(w.Gx), None, None, B, self.wsc, addend_um=P(w.Zp), gather=gather)
            else:
                img.forward(lib, s, None, P(wi.W), P(w.Gx), P(w.Zp), None, B, self.wsc, gather=gather)
        else:
            _gemm(lib, s, 0, 1, Td * B, 4 * H, ni, P(w.X), ni, P(wih), ni + nz, P(w.Gx), 4 * H,
                  add1=P(w.Zp), ld1=4 * H, mod1=B, prec=self.precision)
        b16 = self._b16(B, Td)
        # persistent recurrence: dropout_out is applied while the output is converted to its bf16 images (mask loads along H)
        # instead of inside the recurrence (8 bytes per row and step: +0.3 us per timestep); same arithmetic, same bits
        late_mask = b16 is not None and _persistent_ok(self, img, B, H, x.device, _PERSIST_MAX_B)
        with _prof("lstm_fwd_dec", float(Td), 1 if _persistent_ok(self, img, B, H, x.device, _PERSIST_MAX_B) else Td):
            if late_mask:
                _lstm_forward(self, lib, s, img, w, P(w.Gx), P(v["lstm.weight_hh_l0"]), None, 1.0, None, Td, B, H, x.device)
            else:
                _lstm_forward(self, lib, s, img, w, P(w.Gx), P(v["lstm.weight_hh_l0"]), P(mask_out), sc_out, P(w.O), Td, B, H, x.device)
        if b16 is not None:
            if late_mask and mask_out is not None:
                lib.lv_cvt_bf16_keep_f32(P(w.hs, B * H), H, Td, B, H, P(mask_out), sc_out, P(b16.O), H, P(b16.OT), b16.ldr, s)
            elif late_mask:
                lib.lv_cvt_bf16_f32(P(w.hs, B * H), H, Td * B, H, P(b16.O), H, P(b16.OT), b16.ldr, s)
            else:
                lib.lv_cvt_bf16_f32(P(w.O), H, Td * B, H, P(b16.O), H, P(b16.OT), b16.ldr, s)
            wi = self.refresh_weight_images(B, x.device)
            if self.fused_nll:
                # logits leave the GEMM once, as binary16, with the online-softmax statistics taken in its epilogue
                with _prof("gemm_bf16", 2.0 * Td * B * V * H):
                    lib.lv_gemm_b16_nll(Td * B, V, H, P(b16.O), H, P(wi.pred), H, P(b16.l16), b16.ldv, P(x), T, 1, B,
                                        P(b16.part), P(b16.tgt), s)
                lib.lv_softmax_nll_merge_f32(P(b16.part), b16.nparts, P(b16.tgt), P(w.lse), P(w.nll), Td * B, s)
            else:
                _gemm16(lib, s, 0, Td * B, V, H, P(b16.O), H, P(wi.pred), H, P(w.logits), w.ldl)
        else:
            _gemm(lib, s, 0, 1, Td * B, V, H, P(w.O), H, P(v["pred_linear.weight"]), H, P(w.logits), w.ldl, prec=self.precision)
        if not (b16 is not None and self.fused_nll):
            lib.lv_softmax_nll_fwd_f32(P(w.logits), w.ldl, P(x), T, 1, P(w.lse), P(w.nll), Td, B, V, s)
        if want_rec:
            # rec[b] = sum_t nll[t][b]  (loss assembly kernel with kl weight 0); the fused driver sums nll itself
            lib.lv_vae_loss_f32(P(w.nll), P(w.klz), P(w.zero1), P(w.loss), P(w.rec), Td, B, s)
        self.gen += 1
        self.last = (x, z2, mask_in, mask_out, sc_in, sc_out, B, T, self.gen)
        return w.rec

    def backward(self, drec, gen=None, partial_dz=False):
        """drec [B] = dL/d rec_b -> fills self.flat.grad; returns dz [B][nz] (partial_dz: the tail kernel's partial sums
        [parts][B][nz] and their count instead)."""
        x, z2, mask_in, mask_out, sc_in, sc_out, B, T, g = self.last
        if gen is not None and gen != g:
            raise _lib.LvaeError("decoder activations were overwritten by a later forward(); the HIP engine keeps "
                                 "one in-flight step per module")
        Td = T - 1
        f = self.flat
        lib, s = self.lib, stream_ptr(x.device)
        V, ni, H, nz = self.dims()
        w = self._ws(B, Td)
        v, gv = f.views, f.gviews
        drec = drec.contiguous()
        wih = v["lstm.weight_ih_l0"]
        gwih = gv["lstm.weight_ih_l0"]
        dev = x.device
        b16 = self._b16(B, Td)
        if b16 is not None and self.fused_nll:
            lib.lv_softmax_nll_bwd_h16(P(b16.l16), b16.ldv, P(w.lse), P(x), T, 1, P(drec), P(b16.dl), b16.ldv, Td, B, V, s)
        elif b16 is not None:
            lib.lv_softmax_nll_bwd_b16(P(w.logits), w.ldl, P(w.lse), P(x), T, 1, P(drec), P(b16.dl), b16.ldv, Td, B, V, s)
        else:
            lib.lv_softmax_nll_bwd_f32(P(w.logits), w.ldl, P(w.lse), P(x), T, 1, P(drec), Td, B, V, s)
        ctx, sws = self._fork(dev)                    # side: dW_pred = dlogits^T . O (only needs dlogits and O)
        with ctx:
            if b16 is not None and self.fold and "pred" in self.fold:
                sq, only = self.fold["pred"]          # |dW_pred|^2 leaves the product's own epilogue
                gws = sws if sws is not None else _gemm_ws(lib, s)
                if lib.lv_gemm_b16_sumsq_parts(V, H, Td * B, gws.numel()) != sq.numel():
                    raise RuntimeError("norm folding: the plan's partial count does not match this launch's tile plan")
                with _prof("gemm_bf16", 2.0 * V * H * Td * B):
                    lib.lv_gemm_b16_sumsq(1, V, H, Td * B, P(b16.dl), b16.ldv, P(b16.OT), b16.ldr, P(gv["pred_linear.weight"]), H,
                                          P(gws), gws.numel(), P(sq), int(only), stream_ptr(dev))
            elif b16 is not None:
                _gemm16(lib, stream_ptr(dev), 1, V, H, Td * B, P(b16.dl), b16.ldv, P(b16.OT), b16.ldr,
                        P(gv["pred_linear.weight"]), H, ws=sws)
            else:
                _wgrad(lib, stream_ptr(dev), V, H, Td * B, P(w.logits), w.ldl, P(w.O), H, P(gv["pred_linear.weight"]), H,
                       self.precision, ws=sws)
        img = self._lstm_images(B, Td)
        late_mask = _persistent_ok(self, img, B, H, dev, _PERSIST_BWD_MAX_B)
        if b16 is not None and late_mask and mask_out is not None:
            # dO = dlogits . W_pred with the dropout_out backward applied where the product's K pieces are summed (no pass of its own)
            gws = _gemm_ws(lib, s)
            with _prof("gemm_bf16", 2.0 * Td * B * H * V):
                lib.lv_gemm_b16_keep(Td * B, H, V, P(b16.dl), b16.ldv, P(self._wimg.predT), b16.ldv, P(w.dO), P(mask_out), sc_out, B,
                                     P(gws), gws.numel(), s)
        else:
            if b16 is not None:
                _gemm16(lib, s, 0, Td * B, H, V, P(b16.dl), b16.ldv, P(self._wimg.predT), b16.ldv, P(w.dO), H)
            else:
                _gemm(lib, s, 0, 0, Td * B, H, V, P(w.logits), w.ldl, P(v["pred_linear.weight"]), H, P(w.dO), H, prec=self.precision)
            if late_mask and mask_out is not None:
                lib.lv_keep_scale_f32(P(w.dO), P(mask_out), sc_out, Td, B, H, s)      # dropout_out backward, once, loads along H
        with _prof("lstm_bwd_dec", float(Td), 1 if late_mask else 2 * Td):
            _lstm_backward(self, lib, s, img, w, P(w.dO), None, None if late_mask else P(mask_out), 1.0 if late_mask else sc_out,
                           P(v["lstm.weight_hh_l0"]), None, P(w.dc0), 1, Td, B, H, dev)
        ctx, sws = self._fork(dev)                    # side: everything that only needs dG (runs under the encoder's backward)
        with ctx:
            s2 = stream_ptr(dev)
            if img is not None:
                img.backward(lib, s2, None, P(w.hs), P(self._wimg.WT), P(w.dX), P(gwih), ni + nz, P(gv["lstm.weight_hh_l0"]), ws=sws)
            else:
                _gemm(lib, s2, 0, 0, Td * B, ni, 4 * H, P(w.dG), 4 * H, P(wih), ni + nz, P(w.dX), ni, prec=self.precision, ws=sws)
                _wgrad(lib, s2, 4 * H, ni, Td * B, P(w.dG), 4 * H, P(w.X), ni, P(gwih), ni + nz, self.precision, ws=sws)
                _wgrad(lib, s2, 4 * H, H, Td * B, P(w.dG), 4 * H, P(w.hs), H, P(gv["lstm.weight_hh_l0"]), H,
                       self.precision, ws=sws)
            self._aux.join(dev)                        # token sort queued by forward()
            if self.fold and "embed" in self.fold:
                sq, only = self.fold["embed"]
                lib.lv_embed_scatter_full_sumsq_f32(P(w.dX), P(mask_in), sc_in, P(self._sort[0]), P(self._sort[1]), Td, B,
                                                    P(gv["embed.weight"]), ni, V, V - 1, P(sq), int(only), s2)
            else:
                lib.lv_embed_scatter_full_f32(P(w.dX), P(mask_in), sc_in, P(self._sort[0]), P(self._sort[1]), Td, B, P(gv["embed.weight"]), ni,
                                              V, V - 1, s2)
        self._mark_pending(dev)
        if not fused_ends_ok(B, nz):
            _gemm(lib, s, 1, 0, 4 * H, nz, B, P(w.dGsum), 4 * H, P(z2), nz, P(gwih, ni), ni + nz)
            lib.lv_colsum_f32(P(w.dGsum), 4 * H, B, 4 * H, P(gv["lstm.bias_ih_l0"]), P(gv["lstm.bias_hh_l0"]), s)
            _gemm(lib, s, 0, 0, B, nz, 4 * H, P(w.dGsum), 4 * H, P(wih, ni), ni + nz, P(w.dz), nz)
            _gemm(lib, s, 0, 0, B, nz, H, P(w.dc0), H, P(v["trans_linear.weight"]), nz, P(w.dz), nz, acc=1)
            _gemm(lib, s, 1, 0, H, nz, B, P(w.dc0), H, P(z2), nz, P(gv["trans_linear.weight"]), nz)
            return (w.dz, 1) if partial_dz else w.dz
        # batch-sized tail in one launch (critical path: dz feeds the encoder's backward): the z-columns of dW_ih, both
        # bias gradients, dW_trans, and dz = dGsum . W_ih[:, ni:] + dc0 . W_trans
        lib.lv_dec_tail_bwd_f32(P(w.dGsum), P(w.dc0), P(z2), P(wih), ni + nz, ni, P(v["trans_linear.weight"]), P(gwih), ni + nz,
                                P(gv["trans_linear.weight"]), P(gv["lstm.bias_ih_l0"]), P(gv["lstm.bias_hh_l0"]), P(w.dzp),
                                B, H, nz, s)
        if partial_dz:
            return w.dzp, w.dz_parts         # the fused driver's encoder head sums the parts itself
        lib.lv_colsum_f32(P(w.dzp), B * nz, w.dz_parts, B * nz, P(w.dz), None, s)
        return w.dz


def fused_ends_ok(B, nz, ns=1):
    """LDS budgets of lv_head.hip (64 KB of dynamic LDS): encoder head (B*2nz + 16*(2nz + B) floats at the narrowest column
    block), decoder init / tail (B*nz + B*65 + 64*nz floats).  Beyond them the engines use the GEMM-based sequences."""
    return (B * 2 * nz + B * ns * nz + 16 * (2 * nz + B)) * 4 <= 64000 and (B * nz + B * 65 + 64 * (nz + 1)) * 4 <= 64000


def reparam_kl_forward(mulv, eps):
    """mulv [B][2nz], eps [B][ns][nz] -> z [B][ns][nz], kl [B]   (encoder.py:53-55, 71-79)."""
    lib, s = backend_for(mulv.device), stream_ptr(mulv.device)
    B, nz2 = mulv.shape
    nz = nz2 // 2
    ns = eps.shape[1]
    mulv = mulv.contiguous()
    eps = eps.contiguous()
    z = torch.empty(B, ns, nz, dtype=torch.float32, device=mulv.device)
    kl = torch.empty(B, dtype=torch.float32, device=mulv.device)
    lib.lv_reparam_kl_fwd_f32(P(mulv), P(eps), P(z), P(kl), B, ns, nz, s)
    return z, kl


def reparam_kl_backward(mulv, eps, dz, dkl):
    lib, s = backend_for(mulv.device), stream_ptr(mulv.device)
    B, nz2 = mulv.shape
    nz = nz2 // 2
    ns = eps.shape[1]
    dmulv = torch.empty_like(mulv)
    # (contiguous copies are held in locals until the launch is queued: a temporary's block may be handed out again at once)
    mulv_c, eps_c, dz_c, dkl_c = mulv.contiguous(), eps.contiguous(), dz.contiguous(), dkl.contiguous()
    lib.lv_reparam_kl_bwd_f32(P(mulv_c), P(eps_c), P(dz_c), P(dkl_c), P(dmulv), B, ns, nz, s)
    return dmulv


# ---- evaluation statistics (lv_eval.hip) ------------------------------------------------------------------
def gauss_logpdf(z, mu, logvar):
    """z [B][ns][nz]; mu, logvar [B][nz] or None (standard normal) -> log density [B][ns]."""
    lib, s = backend_for(z.device), stream_ptr(z.device)
    B, ns, nz = z.shape
    z = z.contiguous().float()
    out = torch.empty(B, ns, dtype=torch.float32, device=z.device)
    if mu is not None:
        mu, logvar = mu.contiguous().float(), logvar.contiguous().float()
    lib.lv_gauss_logpdf_f32(P(z), P(mu), P(logvar), P(out), B, ns, nz, s)
    return out


def logsumexp_rows(x, add=0.0):
    """x [R][C] -> log(sum_c exp(x[r][c])) + add, [R]."""
    lib, s = backend_for(x.device), stream_ptr(x.device)
    x = x.contiguous().float()
    R, C = x.shape
    out = torch.empty(R, dtype=torch.float32, device=x.device)
    lib.lv_logsumexp_rows_f32(P(x), C, R, C, float(add), P(out), s)
    return out


def calc_mi(mu, logvar, z):
    """mu, logvar [Bx][nz], z [Bz][nz] -> device tensor (MI, E log q(z|x), E log q(z))."""
    lib, s = backend_for(mu.device), stream_ptr(mu.device)
    mu, logvar, z = mu.contiguous().float(), logvar.contiguous().float(), z.contiguous().float()
    Bx, nz = mu.shape
    Bz = z.shape[0]
    ws = torch.empty(Bz, dtype=torch.float32, device=mu.device)
    out = torch.empty(3, dtype=torch.float32, device=mu.device)
    lib.lv_calc_mi_f32(P(mu), P(logvar), P(z), P(ws), P(out), Bx, Bz, nz, s)
    return out


def au_accumulate(mu, mean, acc):
    """acc[k] += sum_b mu[b][k] (mean None) or sum_b (mu[b][k] - mean[k])^2."""
    lib, s = backend_for(mu.device), stream_ptr(mu.device)
    mu = mu.contiguous().float()
    lib.lv_au_accum_f32(P(mu), P(mean), P(acc), mu.shape[0], mu.shape[1], s)


# ---- generation -------------------------------------------------------------------------------------------
class LSTMDecodeStepper(object):
    """One decoder timestep at a time for greedy / sample / beam decoding (reference modules/decoders/dec_lstm.py:163-367):
    embed(token) ++ z -> LSTM cell -> pred_linear, through the same C-ABI kernels as the training path in exact f32
    (lv_embed_gather_f32, lv_gemm_f32 with the z-projection folded into its epilogue, lv_lstm_fwd_f32 with T = 1)."""

    def __init__(self, eng, device):
        self.eng = eng
        self.device = torch.device(device)
        eng.ensure(self.device)
        self.lib = eng.lib
        self.ws = {}

    def _refresh(self):
        """The decoder's parameters may have been rebound since this stepper was built (p.data = X, a dtype round trip): bring
        the flat copy up to date before reading it."""
        self.eng.ensure(self.device)
        self.lib = self.eng.lib

    def _w(self, n):
        w = self.ws.get(n)
        if w is None:
            V, ni, H, nz = self.eng.dims()
            d = self.device
            w = _NS()
            w.X = torch.empty(n, ni, dtype=torch.float32, device=d)
            w.Zp = torch.empty(n, 4 * H, dtype=torch.float32, device=d)
            w.Gx = torch.empty(n, 4 * H, dtype=torch.float32, device=d)
            w.hs = torch.empty(2, n, H, dtype=torch.float32, device=d)
            w.cs = torch.empty(2, n, H, dtype=torch.float32, device=d)
            w.gates = torch.empty(n, 4 * H, dtype=torch.float32, device=d)
            w.lstm_ws = torch.empty(self.lib.lv_lstm_ws_floats(n, H), dtype=torch.float32, device=d)
            w.ldl = _round_up(V, 32)
            w.logits = torch.empty(n, w.ldl, dtype=torch.float32, device=d)
            self.ws[n] = w
        return w

    def init_state(self, z2):
        """z2 [n][nz] -> (h0, c0) [n][H]: c0 = trans_linear(z), h0 = tanh(c0)   (dec_lstm.py:181-182, 284-285)."""
        self._refresh()
        V, ni, H, nz = self.eng.dims()
        v = self.eng.flat.views
        lib, s = self.lib, stream_ptr(self.device)
        n = z2.shape[0]
        z2 = z2.contiguous().float()
        c0 = torch.empty(n, H, dtype=torch.float32, device=self.device)
        _gemm(lib, s, 0, 1, n, H, nz, P(z2), nz, P(v["trans_linear.weight"]), nz, P(c0), H)
        h0 = torch.empty_like(c0)
        lib.lv_tanh_f32(P(c0), P(h0), n * H, s)
        return h0, c0

    def step(self, tokens, z2, h, c):
        """tokens int64 [n], z2 [n][nz], (h, c) [n][H] -> logits [n][V] (a view, valid until the next call), (h', c')."""
        self._refresh()
        V, ni, H, nz = self.eng.dims()
        v = self.eng.flat.views
        lib, s = self.lib, stream_ptr(self.device)
        n = tokens.shape[0]
        w = self._w(n)
        tok = tokens.reshape(n, 1).contiguous()
        z2 = z2.contiguous().float()
        lib.lv_embed_gather_f32(P(v["embed.weight"]), P(tok), 1, None, 1.0, P(w.X), 1, n, ni, V, s)
        wih = v["lstm.weight_ih_l0"]
        _gemm(lib, s, 0, 1, n, 4 * H, nz, P(z2), nz, P(wih, ni), ni + nz, P(w.Zp), 4 * H,
              add1=P(v["lstm.bias_ih_l0"]), ld1=0, mod1=1, add2=P(v["lstm.bias_hh_l0"]), ld2=0, mod2=1)
        _gemm(lib, s, 0, 1, n, 4 * H, ni, P(w.X), ni, P(wih), ni + nz, P(w.Gx), 4 * H, add1=P(w.Zp), ld1=4 * H, mod1=n)
        w.hs[0].copy_(h)
        w.cs[0].copy_(c)
        lib.lv_lstm_fwd_f32(P(w.Gx), P(v["lstm.weight_hh_l0"]), P(w.hs), P(w.cs), P(w.gates), None, 1.0, None, P(w.lstm_ws), 1, n, H, s)
        _gemm(lib, s, 0, 1, n, V, H, P(w.hs, n * H), H, P(v["pred_linear.weight"]), H, P(w.logits), w.ldl)
        return w.logits[:, :V], w.hs[1].clone(), w.cs[1].clone()

    def argmax(self, logits):
        n, V = logits.shape
        idx = torch.empty(n, dtype=torch.int64, device=self.device)
        self.lib.lv_argmax_rows_f32(P(logits), logits.stride(0), n, V, P(idx), stream_ptr(self.device))
        return idx

    def sample(self, logits, u):
        n, V = logits.shape
        idx = torch.empty(n, dtype=torch.int64, device=self.device)
        u = u.contiguous().float()
        self.lib.lv_sample_rows_f32(P(logits), logits.stride(0), n, V, P(u), P(idx), stream_ptr(self.device))
        return idx

    def log_softmax(self, logits, addrow=None):
        n, V = logits.shape
        out = torch.empty(n, V, dtype=torch.float32, device=self.device)
        if addrow is not None:
            addrow = addrow.contiguous().float()
        self.lib.lv_log_softmax_rows_f32(P(logits), logits.stride(0), n, V, P(addrow), P(out), V, stream_ptr(self.device))
        return out
