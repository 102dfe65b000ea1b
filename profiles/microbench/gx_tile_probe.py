"""Microbenchmark (measurement tooling): the decoder's input projection Gx = X . W_ih^T (M = T B, N = 4H, K = ni, + bias addend) on the
128 x 128 kernel (the shipped choice below 1e11 flop) against the 256 x 256 tile on its three schedules (lv_gemm_b16_tile: 256 / 257 / 258)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vae_lagging_encoder_amd import _lib
from vae_lagging_encoder_amd.engine import P, stream_ptr
lib = _lib.load(); dev = torch.device("cuda:0"); s = stream_ptr(dev)
ws = torch.empty(1 << 26, device=dev)


def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


H = 1024
for TB, ni, label in ((6400, 512, "Yahoo encoder / decoder (K = 512)"), (6368, 544, "decoder incl. z columns (K = 544)"), (3200, 512, "Yelp (T = 100)"), (25600, 512, "stress (B = 128)")):
    X = torch.randn(TB, ni, device=dev).to(torch.bfloat16).view(torch.int16)
    W = torch.randn(4 * H, ni, device=dev).to(torch.bfloat16).view(torch.int16)
    add = torch.randn(4 * H, device=dev)
    C = torch.empty(TB, 4 * H, device=dev)
    ref = None
    out = []
    for tile in (128, 256, 257, 258):
        f = lambda: lib.lv_gemm_b16_tile(tile, 0, TB, 4 * H, ni, 1.0, P(X), ni, P(W), ni, P(C), 4 * H, 0, P(add), 0, 1, None, 0, 1, P(ws), ws.numel(), s)
        f(); torch.cuda.synchronize()
        if ref is None: ref = C.clone()
        err = float((C - ref).abs().max())
        out.append("tile %d: %.1f us (max diff %.1e)" % (tile, timeit(f), err))
    print(label + ": " + " | ".join(out), flush=True)
