"""Microbenchmark (measurement tooling): the eight LSTM-sized GEMMs of the step (Gx, dX, dW_ih, dW_hh of both networks) on the
128 x 128 kernel, with their split-K reductions, for every library in LVAE_PROBE_LIBS (comma-separated alternative builds) beside
the product build; columns interleaved and repeated."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vae_lagging_encoder_amd import _lib
from vae_lagging_encoder_amd.engine import P, stream_ptr
dev = torch.device("cuda:0")
libs = [("prod", _lib.load())]
for path in [p for p in os.environ.get("LVAE_PROBE_LIBS", "").split(",") if p]:
    libs.append((os.path.basename(path).replace("liblvae_", "").replace(".so", ""), _lib.bind(ctypes.CDLL(path), path)))
s = stream_ptr(dev)
ws = torch.empty(1 << 26, device=dev)
H, ni, TB = 1024, 512, 6400


def b16(*shape):
    return torch.randn(*shape, device=dev).to(torch.bfloat16).view(torch.int16)


def med(fn, n=10, reps=3):
    out = []
    for _ in range(reps):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / n)
    return sorted(out)[reps // 2]


X16, XT16 = b16(TB, ni), b16(ni, TB)
Wi16, WiT16 = b16(4 * H, ni), b16(ni, 4 * H)
dG16, hT16 = b16(TB, 4 * H), b16(H, TB)
Gx = torch.empty(TB, 4 * H, device=dev); dX = torch.empty(TB, ni, device=dev)
dWi = torch.empty(4 * H, ni, device=dev); dWh = torch.empty(4 * H, H, device=dev)
bias = torch.randn(4 * H, device=dev)
shapes = [("Gx", 0, TB, 4 * H, ni, X16, ni, Wi16, ni, Gx, 4 * H, bias), ("dX", 0, TB, ni, 4 * H, dG16, 4 * H, WiT16, 4 * H, dX, ni, None),
          ("dW_ih", 1, 4 * H, ni, TB, dG16, 4 * H, XT16, TB, dWi, ni, None), ("dW_hh", 1, 4 * H, H, TB, dG16, 4 * H, hT16, TB, dWh, H, None)]
tot = {ln: 0.0 for ln, _ in libs}
for name, tA, M, N, K, A, lda, Bm, ldb, C, ldc, add in shapes:
    line = "%-6s M=%5d N=%5d K=%5d" % (name, M, N, K)
    for rep in range(2):
        for ln, L in libs:
            us = med(lambda: L.lv_gemm_b16(tA, M, N, K, 1.0, P(A), lda, P(Bm), ldb, P(C), ldc, 0, P(add) if add is not None else None, 0, 1, None, 0, 1,
                                           P(ws), ws.numel(), s))
            line += " | %s %6.1f us %6.1f TF" % (ln, us, 2.0 * M * N * K / us / 1e6)
            if rep == 1:
                tot[ln] += us
    for tile in (256, 257, 258):
        L = libs[0][1]
        us = med(lambda: L.lv_gemm_b16_tile(tile, tA, M, N, K, 1.0, P(A), lda, P(Bm), ldb, P(C), ldc, 0, P(add) if add is not None else None, 0, 1, None, 0, 1,
                                            P(ws), ws.numel(), s))
        line += " | tile%d %6.1f us %6.1f TF" % (tile, us, 2.0 * M * N * K / us / 1e6)
    print(line, flush=True)
print("sum of the four (GEMM + split-K reduce), x2 networks per step: " + ", ".join("%s %.1f us" % (k, 2 * v) for k, v in tot.items()))
