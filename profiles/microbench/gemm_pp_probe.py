"""Microbenchmark (measurement tooling): the 256 x 256 bf16 GEMM kernel, lockstep K loop (tile code 256) against the ping-pong
schedule (257: the two waves of a SIMD half a k-step apart), on the step's three vocabulary-sized products, the fused logits + NLL
form and two squares.  Every library named in LVAE_PROBE_LIBS (comma-separated paths of alternative builds, e.g. made with
build_alt.sh -DLV_B16_PP_DMA=1 -o ...) is timed beside the product build, columns interleaved and repeated (the first timing of a
kernel in a process runs slow)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vae_lagging_encoder_amd import _lib
from vae_lagging_encoder_amd.engine import P, stream_ptr

dev = torch.device("cuda:0")
libs = [("prod", _lib.load())]
for path in [p for p in os.environ.get("LVAE_PROBE_LIBS", "").split(",") if p]:
    try:
        libs.append((os.path.basename(path).replace("liblvae_", "").replace(".so", ""), _lib.bind(ctypes.CDLL(path), path)))
    except Exception as e:      # noqa
        print("skipping %s: %s" % (path, str(e)[:100]))
B, T, V, H = 32, 200, 20001, 1024
R = (T - 1) * B
ldl = (V + 31) // 32 * 32
s = stream_ptr(dev)
ws = torch.empty(1 << 26, device=dev)


def timeit(fn, n=8):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def med(fn, reps=3):
    return sorted(timeit(fn) for _ in range(reps))[reps // 2]


def b16(*shape):
    return torch.randn(*shape, device=dev).to(torch.bfloat16).view(torch.int16)


O16, O16T = b16(R, H), b16(H, R)
W16, W16T = b16(V, H), b16(H, ldl)
dl16 = b16(R, ldl)
logits = torch.empty(R, ldl, device=dev)
dO = torch.empty(R, H, device=dev)
dW = torch.empty(V, H, device=dev)
S8 = 8192
sqA, sqB = b16(S8, S8), b16(S8, S8)
sqC = torch.empty(S8, S8, device=dev)
shapes = [("logits", 0, R, V, H, O16, H, W16, H, logits, ldl), ("dO", 0, R, H, V, dl16, ldl, W16T, ldl, dO, H),
          ("dW_pred", 1, V, H, R, dl16, ldl, O16T, R, dW, H), ("sq8k", 0, S8, S8, S8, sqA, S8, sqB, S8, sqC, S8),
          ("sq8kTN", 1, S8, S8, S8, sqA, S8, sqB, S8, sqC, S8), ("sq4k", 0, 4096, 4096, 4096, sqA, S8, sqB, S8, sqC, S8)]
only = [n for n in os.environ.get("LVAE_PROBE_SHAPES", "").split(",") if n]
tiles = tuple(int(t) for t in os.environ.get("LVAE_PROBE_TILES", "256,257").split(","))
for name, tA, M, N, K, A, lda, Bm, ldb, C, ldc in shapes:
    if only and name not in only:
        continue
    line = "%-8s M=%5d N=%5d K=%5d" % (name, M, N, K)
    for rep in range(2):
        for ln, L in libs:
            for tile in tiles:
                us = med(lambda: L.lv_gemm_b16_tile(tile, tA, M, N, K, 1.0, P(A), lda, P(Bm), ldb, P(C), ldc, 0, None, 0, 1, None, 0, 1,
                                                    P(ws), ws.numel(), s))
                line += " | %s/%d %6.1f us %6.1f TF" % (ln, tile, us, 2.0 * M * N * K / us / 1e6)
    print(line, flush=True)
x = torch.randint(0, V, (B, T), device=dev)
l16 = torch.empty(R, ldl, dtype=torch.int16, device=dev)
part = torch.empty(R, 2 * libs[0][1].lv_gemm_b16_nll_parts(V), device=dev)
tg = torch.empty(R, device=dev)
line = "logits+NLL fused M=%5d N=%5d K=%5d" % (R, V, H)
for rep in range(0 if (only and "nll" not in only) else 2):
    for ln, L in libs:
        for tile in tiles:
            us = med(lambda: L.lv_gemm_b16_nll_tile(tile, R, V, H, P(O16), H, P(W16), H, P(l16), ldl, P(x), T, 1, B, P(part), P(tg), s))
            line += " | %s/%d %6.1f us %6.1f TF" % (ln, tile, us, 2.0 * R * V * H / us / 1e6)
print(line, flush=True)
