"""Microbenchmark (measurement tooling): per-shape timing of the step's GEMMs, f32 vs bf16 kernels, optional alt .so."""
import ctypes, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vae_lagging_encoder_amd import _lib
from vae_lagging_encoder_amd.engine import P, stream_ptr
dev = torch.device("cuda:0")
libs = {"default": _lib.load()}
alt = os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblvae_alt.so")
if os.path.exists(alt):
    libs["alt(soft-cvt)"] = _lib.bind(ctypes.CDLL(alt), alt)
B, T, V, ni, H = 32, 200, 20001, 512, 1024
Td = T - 1
shapes = [("Gx_enc NT", 0, 1, T * B, 4 * H, ni), ("logits NT", 0, 1, Td * B, V, H), ("dO NN", 0, 0, Td * B, H, V),
          ("dW_pred TN", 1, 0, V, H, Td * B), ("dX NN", 0, 0, Td * B, ni, 4 * H), ("dW_ih TN", 1, 0, 4 * H, ni, Td * B),
          ("dW_hh TN", 1, 0, 4 * H, H, Td * B)]
ws = torch.empty(1 << 26, device=dev)
s = stream_ptr(dev)
for name, tA, tB, M, N, K in shapes:
    lda = (M if tA else K); ldb = (K if tB else N)
    lda_p = (lda + 31) // 32 * 32; ldb_p = ldb
    A = torch.randn((K if tA else M), lda_p, device=dev)
    Bm = torch.randn((N if tB else K), ldb_p, device=dev)
    C = torch.empty(M, N, device=dev)
    line = "%-12s M=%6d N=%6d K=%6d " % (name, M, N, K)
    for lname, lib in libs.items():
        for kind in ("f32", "bf16"):
            fn = lib.lv_gemm_f32 if kind == "f32" else lib.lv_gemm_bf16
            for _ in range(2):
                fn(tA, tB, M, N, K, 1.0, P(A), lda_p, P(Bm), ldb_p, P(C), N, 0, None, 0, 1, None, 0, 1, P(ws), ws.numel(), s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn(tA, tB, M, N, K, 1.0, P(A), lda_p, P(Bm), ldb_p, P(C), N, 0, None, 0, 1, None, 0, 1, P(ws), ws.numel(), s)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 5
            line += "| %s %s %8.1f us %6.1f TF " % (lname[:7], kind, us, 2.0 * M * N * K / us / 1e6)
    print(line)
