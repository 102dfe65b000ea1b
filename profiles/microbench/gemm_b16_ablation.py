"""Microbenchmark (measurement tooling): where does lv_gemm_b16's time go on the logits shape?  Compares the product
library with ablation builds (LV_B16_ABL: 1 = no MFMA, 2 = no global loads inside the K loop, 4 = no LDS fragment reads)
built next to this script as liblvae_abl<N>.so by profiles/microbench/build_ablation.sh."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vae_lagging_encoder_amd import _lib
from vae_lagging_encoder_amd.engine import P, stream_ptr
dev = torch.device("cuda:0")
here = os.path.dirname(os.path.abspath(__file__))
libs = {"product": _lib.load()}
for n in (1, 2, 3, 4, 6, 7):
    f = os.path.join(here, "liblvae_abl%d.so" % n)
    if os.path.exists(f):
        libs["abl%d" % n] = _lib.bind(ctypes.CDLL(f), f)
R, V, H = 6368, 20001, 1024
ldl = (V + 31) // 32 * 32
s = stream_ptr(dev)
ws = torch.empty(1 << 26, device=dev)
O16 = torch.randn(R, H, device=dev).to(torch.bfloat16).view(torch.int16)
W16 = torch.randn(V, H, device=dev).to(torch.bfloat16).view(torch.int16)
dl16 = torch.randn(R, ldl, device=dev).to(torch.bfloat16).view(torch.int16)
W16T = torch.randn(H, ldl, device=dev).to(torch.bfloat16).view(torch.int16)
logits = torch.empty(R, ldl, device=dev)
dO = torch.empty(R, H, device=dev)


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


for name, L in libs.items():
    a = timeit(lambda: L.lv_gemm_b16(0, R, V, H, 1.0, P(O16), H, P(W16), H, P(logits), ldl, 0, None, 0, 1, None, 0, 1, P(ws), ws.numel(), s))
    b = timeit(lambda: L.lv_gemm_b16(0, R, H, V, 1.0, P(dl16), ldl, P(W16T), ldl, P(dO), H, 0, None, 0, 1, None, 0, 1, P(ws), ws.numel(), s))
    print("%-8s logits %7.1f us   dO %7.1f us" % (name, a, b))
