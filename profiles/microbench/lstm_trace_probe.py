"""Microbenchmark (measurement tooling): where a timestep of the persistent recurrences goes.  Uses liblvae_trace.so (build_trace.sh:
the kernel library with -DLV_TRACE), whose persistent kernels store the shader clock of lane 0 of workgroup 8 at phase boundaries:
forward  0 step start | 1 gather complete (polls + LDS stores issued) | 2 LDS visible | 3 MFMAs + quarter-product stores issued |
         4 after the barrier | 5 cell update done, h_t published
BPTT     0 step start | 1 receive complete (polls + sums + shuffles) | 2 gate gradients + dG image stored | 3 after the barrier |
         4 product + sends issued"""
import ctypes, os, sys, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vae_lagging_encoder_amd import _lib
from vae_lagging_encoder_amd.engine import P, stream_ptr
here = os.path.dirname(os.path.abspath(__file__))
cdll = ctypes.CDLL(os.path.join(here, "liblvae_trace.so"))
lib = _lib.bind(cdll, "liblvae_trace.so")
dev = torch.device("cuda:0"); s = stream_ptr(dev)
T, H = 200, 1024
whh = (torch.rand(4 * H, H, device=dev) * 2 - 1) * 0.03
n = lib.lv_lstm_persist16_wpk_floats()
wf, wb = torch.empty(n, device=dev), torch.empty(n, device=dev)
lib.lv_lstm_persist16_pack(P(whh), P(wf), 0, H, s)
lib.lv_lstm_persist16_pack(P(whh), P(wb), 1, H, s)
xch = torch.empty(lib.lv_lstm_persist16_xch_floats(), device=dev)
st = torch.zeros(1, dtype=torch.int32, device=dev)
trace = torch.zeros(T, 8, dtype=torch.int64, device=dev)
cdll.lv_trace_set.argtypes = [ctypes.c_void_p]
for B, R in ((32, 4), (64, 8), (128, 16)):
    g = torch.Generator().manual_seed(B)
    gx = (torch.randn(T, B, 4 * H, generator=g) * 0.5).to(dev)
    hs = torch.zeros(T + 1, B, H, device=dev); cs = torch.zeros(T + 1, B, H, device=dev)
    gates = torch.empty(max(T * B * 4 * H, lib.lv_lstm_persist16_saved_floats(T, 16)), device=dev)      # also the 16-row kernels' saved-activation buffer
    dO = (torch.randn(T, B, H, generator=g) * 0.1).to(dev)
    dG16 = torch.empty(T, B, 4 * H, dtype=torch.int16, device=dev)
    dGsum = torch.empty(B, 4 * H, device=dev); dc0 = torch.empty(B, H, device=dev)
    for name, nm, fn in (("forward", 6, lambda: lib.lv_lstm_fwd_bf16_persist16(P(gx), P(wf), P(hs), P(cs), P(gates), P(xch), P(st), T, B, R, 1, H, s)),
                         ("BPTT", 5, lambda: lib.lv_lstm_bwd_bf16_persist16(P(dO), None, P(wb), P(gates), P(hs), P(cs), P(dG16), P(dGsum), P(xch), P(st), None, P(dc0), 1, T, B, R, 1, H, s))):
        cdll.lv_trace_set(None)
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        us_plain = e0.elapsed_time(e1) * 1e3 / T
        cdll.lv_trace_set(ctypes.c_void_p(trace.data_ptr()))
        trace.zero_()
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        us_traced = e0.elapsed_time(e1) * 1e3 / T
        full = trace.cpu().numpy().astype(np.float64)
        spins = full[:, 6:8]
        tr = full[:, :nm]
        order = np.argsort(tr[:, 0])                       # BPTT walks t downwards
        tr = tr[order]
        period = np.diff(tr[:, 0])
        inner = np.array([i for i in range(1, T - 1) if (i % 8) not in (0, 7)])      # steps away from the I/O block boundaries
        ticks_per_us = np.median(period[inner - 1]) / us_traced
        ph = np.diff(tr, axis=1)[inner]
        tail = (tr[inner + 1, 0] - tr[inner, nm - 1])
        print("B=%d R=%d %s: %.2f us/step untraced, %.2f traced; median phase times (us): %s | to next step start %.2f  (clock %.0f ticks/us); "
              "failed poll rounds per step of the last polling round(s): mean %s" % (
            B, R, name, us_plain, us_traced, " ".join("%.2f" % (v / ticks_per_us) for v in np.median(ph, axis=0)), np.median(tail) / ticks_per_us, ticks_per_us,
            " ".join("%.2f" % v for v in spins[5:-5].mean(axis=0))))
cdll.lv_trace_set(None)
