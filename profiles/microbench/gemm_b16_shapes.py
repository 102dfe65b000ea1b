"""Microbenchmark (measurement tooling): the decoder's three vocabulary-sized GEMMs, lv_gemm_bf16 (f32 operands rounded
on the fly) vs lv_gemm_b16 (operands pre-rounded to bf16), plus the producers (lv_cvt_bf16_f32, softmax bwd)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vae_lagging_encoder_amd import _lib
from vae_lagging_encoder_amd.engine import P, stream_ptr
dev = torch.device("cuda:0")
if os.environ.get("LVAE_PROBE_LIB"):            # an alternative build of the kernel library (a measurement knob compiled in)
    import ctypes
    lib = _lib.bind(ctypes.CDLL(os.environ["LVAE_PROBE_LIB"]), os.environ["LVAE_PROBE_LIB"])
else:
    lib = _lib.load()
B, T, V, H = 32, 200, 20001, 1024
Td = T - 1
R = Td * B
ldl = (V + 31) // 32 * 32
s = stream_ptr(dev)
ws = torch.empty(1 << 26, device=dev)


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


O = torch.randn(R, H, device=dev)
W = torch.randn(V, H, device=dev) * 0.05
dl = torch.randn(R, ldl, device=dev)
O16 = torch.empty(R, H, dtype=torch.int16, device=dev)
O16T = torch.empty(H, R, dtype=torch.int16, device=dev)
W16 = torch.empty(V, H, dtype=torch.int16, device=dev)
W16T = torch.empty(H, ldl, dtype=torch.int16, device=dev)
dl16 = dl.to(torch.bfloat16).view(torch.int16).contiguous()
logits = torch.empty(R, ldl, device=dev)
dO = torch.empty(R, H, device=dev)
dW = torch.empty(V, H, device=dev)
GF = 2.0 * R * V * H
print("cvt O  (+T): %7.1f us" % timeit(lambda: lib.lv_cvt_bf16_f32(P(O), H, R, H, P(O16), H, P(O16T), R, s)))
print("cvt Wp (+T): %7.1f us" % timeit(lambda: lib.lv_cvt_bf16_f32(P(W), H, V, H, P(W16), H, P(W16T), ldl, s)))
rows = [
    ("logits", lambda: lib.lv_gemm_bf16(0, 1, R, V, H, 1.0, P(O), H, P(W), H, P(logits), ldl, 0, None, 0, 1, None, 0, 1, P(ws), ws.numel(), s),
     lambda: lib.lv_gemm_b16(0, R, V, H, 1.0, P(O16), H, P(W16), H, P(logits), ldl, 0, None, 0, 1, None, 0, 1, P(ws), ws.numel(), s)),
    ("dO", lambda: lib.lv_gemm_bf16(0, 0, R, H, V, 1.0, P(dl), ldl, P(W), H, P(dO), H, 0, None, 0, 1, None, 0, 1, P(ws), ws.numel(), s),
     lambda: lib.lv_gemm_b16(0, R, H, V, 1.0, P(dl16), ldl, P(W16T), ldl, P(dO), H, 0, None, 0, 1, None, 0, 1, P(ws), ws.numel(), s)),
    ("dW_pred", lambda: lib.lv_gemm_bf16(1, 0, V, H, R, 1.0, P(dl), ldl, P(O), H, P(dW), H, 0, None, 0, 1, None, 0, 1, P(ws), ws.numel(), s),
     lambda: lib.lv_gemm_b16(1, V, H, R, 1.0, P(dl16), ldl, P(O16T), R, P(dW), H, 0, None, 0, 1, None, 0, 1, P(ws), ws.numel(), s)),
]
# LSTM input-side GEMMs (both networks): Gx = X.W_ih^T, dX = dG.W_ih, dW_ih = dG^T.X, dW_hh = dG^T.h
ni, TB = 512, 6400
X16 = torch.randn(TB, ni, device=dev).to(torch.bfloat16).view(torch.int16)
XT16 = torch.randn(ni, TB, device=dev).to(torch.bfloat16).view(torch.int16)
Wi16 = torch.randn(4 * H, ni, device=dev).to(torch.bfloat16).view(torch.int16)
WiT16 = torch.randn(ni, 4 * H, device=dev).to(torch.bfloat16).view(torch.int16)
dG16 = torch.randn(TB, 4 * H, device=dev).to(torch.bfloat16).view(torch.int16)
hT16 = torch.randn(H, TB, device=dev).to(torch.bfloat16).view(torch.int16)
Gx = torch.empty(TB, 4 * H, device=dev); dX = torch.empty(TB, ni, device=dev)
dWi = torch.empty(4 * H, ni, device=dev); dWh = torch.empty(4 * H, H, device=dev)
import ctypes
libs = {"default": lib}
alt = os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblvae_alt.so")
if os.path.exists(alt):
    try:
        libs["alt"] = _lib.bind(ctypes.CDLL(alt), alt)      # an A/B build made with build_alt.sh (must be of the same ABI revision)
    except Exception as e:
        print("skipping stale A/B library:", str(e)[:80])
small = [("Gx", 0, TB, 4 * H, ni, X16, ni, Wi16, ni, Gx, 4 * H), ("dX", 0, TB, ni, 4 * H, dG16, 4 * H, WiT16, 4 * H, dX, ni),
         ("dW_ih", 1, 4 * H, ni, TB, dG16, 4 * H, XT16, TB, dWi, ni), ("dW_hh", 1, 4 * H, H, TB, dG16, 4 * H, hT16, TB, dWh, H),
         ("dO", 0, R, H, V, dl16, ldl, W16T, ldl, dO, H), ("dW_pred", 1, V, H, R, dl16, ldl, O16T, R, dW, H), ("logits", 0, R, V, H, O16, H, W16, H, logits, ldl)]
def timeit_med(fn, reps=3):
    return sorted(timeit(fn, 8) for _ in range(reps))[reps // 2]


S8 = 8192
sqA = torch.randn(S8, S8, device=dev).to(torch.bfloat16).view(torch.int16)
sqB = torch.randn(S8, S8, device=dev).to(torch.bfloat16).view(torch.int16)
sqC = torch.empty(S8, S8, device=dev)
small += [("sq8k", 0, S8, S8, S8, sqA, S8, sqB, S8, sqC, S8), ("sq8kTN", 1, S8, S8, S8, sqA, S8, sqB, S8, sqC, S8),
          ("sq4k", 0, 4096, 4096, 4096, sqA, S8, sqB, S8, sqC, S8)]
for name, tA, M, N, K, A, lda, Bm, ldb, C, ldc in small:
    line = "%-7s M=%5d N=%5d K=%5d" % (name, M, N, K)
    for ln, L in list(libs.items()) * (2 if len(libs) > 1 else 1):      # A/B libraries: two interleaved passes (the first timing of a kernel runs slow)
        for tile in (128, 256):
            us = timeit_med(lambda: L.lv_gemm_b16_tile(tile, tA, M, N, K, 1.0, P(A), lda, P(Bm), ldb, P(C), ldc, 0, None, 0, 1, None, 0, 1, P(ws), ws.numel(), s))
            line += " | %s/%d %7.1f us %6.1f TF" % (ln, tile, us, 2.0 * M * N * K / us / 1e6)
    print(line)
# the input projection as the step runs it: with the bias addend, and with output buffers that are not cache-resident
bias = torch.randn(4 * H, device=dev)
zp = torch.randn(B, 4 * H, device=dev)
outs = [torch.empty(TB, 4 * H, device=dev) for _ in range(4)]
X544 = torch.randn(TB, 544, device=dev).to(torch.bfloat16).view(torch.int16)
W544 = torch.randn(4 * H, 544, device=dev).to(torch.bfloat16).view(torch.int16)
it = [0]
def gx(variant, tile=0):
    it[0] += 1
    C = outs[it[0] % 4] if "cold" in variant else Gx
    if "544" in variant:
        lib.lv_gemm_b16_tile(tile, 0, TB, 4 * H, 544, 1.0, P(X544), 544, P(W544), 544, P(C), 4 * H, 0, P(zp), 4 * H, B, None, 0, 1, P(ws), ws.numel(), s)
    elif "bias" in variant:
        lib.lv_gemm_b16_tile(tile, 0, TB, 4 * H, ni, 1.0, P(X16), ni, P(Wi16), ni, P(C), 4 * H, 0, P(bias), 0, 1, None, 0, 1, P(ws), ws.numel(), s)
    else:
        lib.lv_gemm_b16_tile(tile, 0, TB, 4 * H, ni, 1.0, P(X16), ni, P(Wi16), ni, P(C), 4 * H, 0, None, 0, 1, None, 0, 1, P(ws), ws.numel(), s)
for variant in ("plain", "bias", "cold", "bias cold", "544 zp", "544 zp cold"):
    line = "Gx %-12s" % variant
    for tile in (128, 256):
        line += " | %d %7.1f us" % (tile, timeit_med(lambda: gx(variant, tile)))
    print(line)
# the fused vocabulary projection + NLL statistics
x = torch.randint(0, V, (B, T), device=dev)
l16 = torch.empty(R, ldl, dtype=torch.int16, device=dev)
part = torch.empty(R, 2 * lib.lv_gemm_b16_nll_parts(V), device=dev)
tg = torch.empty(R, device=dev)
line = "logits+NLL fused   M=%5d N=%5d K=%5d" % (R, V, H)
for ln, L in libs.items():
    for tile in (128, 256):
        us = timeit_med(lambda: L.lv_gemm_b16_nll_tile(tile, R, V, H, P(O16), H, P(W16), H, P(l16), ldl, P(x), T, 1, B, P(part), P(tg), s))
        line += " | %s/%d %7.1f us %6.1f TF" % (ln, tile, us, GF / us / 1e6)
print(line)
for name, f_old, f_new in rows:
    a, b = timeit(f_old), timeit(f_new)
    print("%-8s on-the-fly %7.1f us %6.1f TF | pre-rounded %7.1f us %6.1f TF" % (name, a, GF / a / 1e6, b, GF / b / 1e6))
