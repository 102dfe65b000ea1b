"""Microbenchmark (measurement tooling): forward LSTM recurrence, one persistent launch vs one launch per timestep."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vae_lagging_encoder_amd import _lib
from vae_lagging_encoder_amd.engine import P, stream_ptr
dev = torch.device("cuda:0"); lib = _lib.load(); s = stream_ptr(dev)
T, B, H = 200, 32, 1024
gx = torch.randn(T, B, 4 * H, device=dev) * 0.5
whh = torch.randn(4 * H, H, device=dev) / H ** 0.5
hs = torch.zeros(T + 1, B, H, device=dev); cs = torch.zeros(T + 1, B, H, device=dev)
gates = torch.empty(T, B, 4 * H, device=dev)
hdrop = torch.empty(T, B, H, device=dev)
mask = (torch.rand(B, T, H, device=dev) < 0.5).to(torch.uint8)
wsp = torch.empty(lib.lv_lstm_persist_xch_floats(), device=dev)
wpk_f = torch.empty(lib.lv_lstm_persist_wpk_floats(), device=dev)
wpk_b = torch.empty(lib.lv_lstm_persist_wpk_floats(), device=dev)
lib.lv_lstm_persist_pack(P(whh), P(wpk_f), 0, H, s)
lib.lv_lstm_persist_pack(P(whh), P(wpk_b), 1, H, s)
ws = torch.empty(lib.lv_lstm_ws_floats(B, H), device=dev)
st = torch.zeros(1, dtype=torch.int32, device=dev)


def t(f, n=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


a = t(lambda: lib.lv_lstm_fwd_bf16_ug(P(gx), P(whh), P(hs), P(cs), P(gates), P(mask), 2.0, P(hdrop), P(ws), T, B, H, s))
b = t(lambda: lib.lv_lstm_fwd_bf16_persist(P(gx), P(wpk_f), P(hs), P(cs), P(gates), P(mask), 2.0, P(hdrop), P(wsp), P(st), T, B, H, s))
print("launch per step : %8.1f us  (%.2f us/step)" % (a, a / T))
print("persistent (column split, 16x16x32): %8.1f us  (%.2f us/step)   status %d" % (b, b / T, int(st.item())))
wpk_k = torch.empty(lib.lv_lstm_persist_wpk_floats(), device=dev)
lib.lv_lstm_persist_pack(P(whh), P(wpk_k), 3, H, s)
b = t(lambda: lib.lv_lstm_fwd_bf16_persist_ks(P(gx), P(wpk_k), P(hs), P(cs), P(gates), P(mask), 2.0, P(hdrop), P(wsp), P(st), T, B, H, s))
print("persistent (K split, 4x4x4)         : %8.1f us  (%.2f us/step)   status %d" % (b, b / T, int(st.item())))
b = t(lambda: lib.lv_lstm_fwd_bf16_persist_ks(P(gx), P(wpk_k), P(hs), P(cs), P(gates), None, 1.0, None, P(wsp), P(st), T, B, H, s))
print("persistent (K split), no mask / dropped output: %.2f us/step" % (b / T))

# ---- BPTT: one persistent launch vs elementwise + split-K matmul launches per step -------------------------------------
gates_std = torch.empty(T, B, 4 * H, device=dev)
lib.lv_lstm_fwd_bf16(P(gx), P(whh), P(hs), P(cs), P(gates_std), P(mask), 2.0, P(hdrop), P(ws), T, B, H, s)
dO = torch.randn(T, B, H, device=dev)
dG16 = torch.empty(T, B, 4 * H, dtype=torch.int16, device=dev)
dGsum = torch.empty(B, 4 * H, device=dev); dc0 = torch.empty(B, H, device=dev)
a = t(lambda: lib.lv_lstm_bwd_bf16_img(P(dO), None, P(mask), 2.0, P(whh), P(gates_std), P(hs), P(cs), None, P(dG16), P(dGsum), P(ws), None, P(dc0), 1, T, B, H, s))
b = t(lambda: lib.lv_lstm_bwd_bf16_persist(P(dO), None, P(mask), 2.0, P(wpk_b), P(gates_std), P(hs), P(cs), None, P(dG16), P(dGsum), P(wsp), P(st), None, P(dc0), 1, T, B, H, s))
print("BPTT two launches per step : %8.1f us  (%.2f us/step)" % (a, a / T))
print("BPTT persistent (all-gather): %8.1f us  (%.2f us/step)   status %d" % (b, b / T, int(st.item())))
wpk_r = torch.empty(lib.lv_lstm_persist_wpk_floats(), device=dev)
lib.lv_lstm_persist_pack(P(whh), P(wpk_r), 2, H, s)
st.zero_()
b = t(lambda: lib.lv_lstm_bwd_bf16_persist_rs(P(dO), None, P(mask), 2.0, P(wpk_r), P(gates_std), P(hs), P(cs), None, P(dG16), P(dGsum), P(wsp), P(st), None, P(dc0), 1, T, B, H, s))
print("BPTT persistent (reduce-scatter): %8.1f us  (%.2f us/step)   status %d" % (b, b / T, int(st.item())))

# ---- what the block I/O costs: the same launches with fewer per-step inputs / outputs -----------------------------------
b = t(lambda: lib.lv_lstm_fwd_bf16_persist(P(gx), P(wpk_f), P(hs), P(cs), P(gates), None, 1.0, None, P(wsp), P(st), T, B, H, s))
print("forward persistent, no dropout mask / dropped output: %.2f us/step" % (b / T))
b = t(lambda: lib.lv_lstm_fwd_bf16_persist(P(gx), P(wpk_f), P(hs), P(cs), P(gates), None, 1.0, P(hdrop), P(wsp), P(st), T, B, H, s))
print("forward persistent, dropped output without mask      : %.2f us/step" % (b / T))
dlast = torch.randn(B, H, device=dev)
b = t(lambda: lib.lv_lstm_bwd_bf16_persist_rs(P(dO), None, None, 1.0, P(wpk_r), P(gates_std), P(hs), P(cs), None, P(dG16), P(dGsum), P(wsp), P(st), None, P(dc0), 1, T, B, H, s))
print("BPTT rs, dh_ext without mask: %.2f us/step" % (b / T))
b = t(lambda: lib.lv_lstm_bwd_bf16_persist_rs(None, P(dlast), None, 1.0, P(wpk_r), P(gates_std), P(hs), P(cs), None, P(dG16), P(dGsum), P(wsp), P(st), None, P(dc0), 1, T, B, H, s))
print("BPTT rs, dh_last only       : %.2f us/step" % (b / T))
