"""Microbenchmark (measurement tooling): the per-launch fixed cost of the persistent recurrences -- time of one launch at
T = 1, 2, 4, 8, 16, 50, 100, 200 (B = 32, 4 rows per group, hand-off in the XCD's L2), and the least-squares split a + b T."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vae_lagging_encoder_amd import _lib
from vae_lagging_encoder_amd.engine import P, stream_ptr
dev = torch.device("cuda:0"); s = stream_ptr(dev)
lib = _lib.load()
H, B, R = 1024, 32, 4
whh = (torch.rand(4 * H, H, device=dev) * 2 - 1) * 0.03
n = lib.lv_lstm_persist16_wpk_floats()
wf, wb = torch.empty(n, device=dev), torch.empty(n, device=dev)
lib.lv_lstm_persist16_pack2(P(whh), P(wf), P(wb), H, s)
xch = torch.zeros(lib.lv_lstm_persist16_xch_floats(), device=dev)
st = torch.zeros(1, dtype=torch.int32, device=dev)
from types import SimpleNamespace
from vae_lagging_encoder_amd.engine import _xch_flags
wi = SimpleNamespace(xstate={"f": 0, "g": 0, "gcls": [0, 0]})


def t(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


Ts = [1, 2, 4, 8, 16, 50, 100, 200]
rows = []
for mode in ("memset", "alternating halves"):
  rows = []
  lib.lv_lstm_persist16_xch_clear(P(xch), s)
  wi.xstate = {"f": 0, "g": 0, "gcls": [0, 0]}
  FF = (lambda: 1) if mode == "memset" else (lambda: 1 | _xch_flags(wi, "f", R, "cpu"))
  FB = (lambda: 1) if mode == "memset" else (lambda: 1 | _xch_flags(wi, "g", R, "cpu"))
  print("== exchange buffer: %s" % ("zeroed by a memset launch in front of every launch (round 4)" if mode == "memset" else
                                    "halves alternate, every launch clears the other half in its prologue (round 5)"))
  for T in Ts:
      gx = (torch.randn(T, B, 4 * H, device=dev) * 0.5)
      hs = torch.zeros(T + 1, B, H, device=dev); cs = torch.zeros(T + 1, B, H, device=dev)
      saved = torch.empty(lib.lv_lstm_persist16_saved_floats(T, R), device=dev)
      dO = torch.randn(T, B, H, device=dev) * 0.1
      dG16 = torch.empty(T, B, 4 * H, dtype=torch.int16, device=dev)
      dGsum = torch.empty(B, 4 * H, device=dev); dc0 = torch.empty(B, H, device=dev)
      f = t(lambda: lib.lv_lstm_fwd_bf16_persist16(P(gx), P(wf), P(hs), P(cs), P(saved), P(xch), P(st), T, B, R, FF(), H, s))
      b = t(lambda: lib.lv_lstm_bwd_bf16_persist16(P(dO), None, P(wb), P(saved), P(hs), P(cs), P(dG16), P(dGsum), P(xch), P(st), None, P(dc0), 1, T, B, R, FB(), H, s))
      rows.append((T, f, b))
      print("T=%3d: forward %7.1f us, BPTT %7.1f us (back-to-back launches)" % (T, f, b))
  import numpy as np
  A = np.array([[1.0, r[0]] for r in rows])
  for name, col in (("forward", 1), ("BPTT", 2)):
      a, b = np.linalg.lstsq(A, np.array([r[col] for r in rows]), rcond=None)[0]
      print("%s: fixed %.1f us per launch + %.3f us per timestep" % (name, a, b))
print("status", int(st.item()))
