"""Microbenchmark (measurement tooling): the operand-image conversions of a Yahoo step in their 16-byte form (cvt_b16_v4_kernel: aligned
strides) against the 4-byte / 2-byte form (cvt_b16_kernel: the same call with the row strides padded by one element, which makes the
dispatcher fall back).  us per call, 50 calls each."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vae_lagging_encoder_amd import _lib
from vae_lagging_encoder_amd.engine import P, stream_ptr
lib = _lib.load(); dev = torch.device("cuda:0"); s = stream_ptr(dev)


def timeit(f, n=50):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


T, B, H, ni, V = 200, 32, 1024, 512, 20001
TB = T * B
for pad, name in ((0, "16-byte form"), (1, "4 / 2-byte form")):
    res = []
    # (a) LSTM output -> O, O^T (with the dropout mask folded in)
    hs = torch.randn(TB, H, device=dev); keep = (torch.rand(B, T, H, device=dev) < 0.5).to(torch.uint8)
    O = torch.empty(TB, H + pad, dtype=torch.int16, device=dev); OT = torch.empty(H, TB + 8 + pad, dtype=torch.int16, device=dev)
    res.append(("h -> O, O^T (dropout folded)", timeit(lambda: lib.lv_cvt_bf16_keep_f32(P(hs), H, T, B, H, P(keep), 2.0, P(O), H + pad, P(OT), TB + 8 + pad, s))))
    # (b) h_prev -> h^T only
    res.append(("h_prev -> h^T", timeit(lambda: lib.lv_cvt_bf16_f32(P(hs), H, TB, H, None, 0, P(OT), TB + 8 + pad, s))))
    # (c) embedding gather -> X (binary16), X^T
    emb = torch.randn(V, ni, device=dev); ids = torch.randint(0, V, (B, T), device=dev)
    X = torch.empty(TB, ni + pad, dtype=torch.int16, device=dev); XT = torch.empty(ni, TB + 8 + pad, dtype=torch.int16, device=dev)
    res.append(("embedding rows -> X (binary16), X^T", timeit(lambda: lib.lv_cvt_h16_f32(P(emb), ni, TB, ni, 0, P(ids), T, B, V, P(X), ni + pad, P(XT), TB + 8 + pad, s))))
    keep2 = (torch.rand(B, T, ni, device=dev) < 0.5).to(torch.uint8)
    res.append(("embedding rows + dropout -> X, X^T", timeit(lambda: lib.lv_embed_gather_b16(P(emb), P(ids), T, P(keep2), 2.0, T, B, ni, V, P(X), ni + pad, P(XT), TB + 8 + pad, s))))
    # (d) W_ih -> unit-major binary16 image + transposed bf16 image
    W = torch.randn(4 * H, ni, device=dev); Wi = torch.empty(4 * H, ni + pad, dtype=torch.int16, device=dev); WT = torch.empty(ni, 4 * H + pad, dtype=torch.int16, device=dev)
    res.append(("W_ih -> image, W_ih^T", timeit(lambda: lib.lv_cvt_h16_f32(P(W), ni, 4 * H, ni, H, None, 0, 1, 0, P(Wi), ni + pad, P(WT), 4 * H + pad, s))))
    print(name + ": " + " | ".join("%s %.1f us" % r for r in res) + " | sum %.1f us" % sum(r[1] for r in res), flush=True)
