"""Microbenchmark (measurement tooling): where a K tile of the 256 x 256 x 64 bf16 GEMM kernel goes.  liblvae_trace.so (build_trace.sh)
stores the shader clock of lane 0 of workgroup 8 per K tile: 0 tile start | 1 fragments + 32 MFMAs + 8 LDS-DMA issued | 2 DMA landed
(vmcnt(0)) | 3 after the workgroup barrier."""
import ctypes, os, sys, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vae_lagging_encoder_amd import _lib
from vae_lagging_encoder_amd.engine import P, stream_ptr
here = os.path.dirname(os.path.abspath(__file__))
cdll = ctypes.CDLL(os.path.join(here, "liblvae_trace.so"))
lib = _lib.bind(cdll, "liblvae_trace.so")
cdll.lv_trace_set_gemm.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0"); s = stream_ptr(dev)
for name, M, N, K in (("square", 8192, 8192, 8192), ("dO", 6368, 1024, 20001), ("logits", 6368, 20001, 1024)):
    ldk = (K + 31) // 32 * 32
    A = torch.randn(M, ldk, device=dev).to(torch.bfloat16).view(torch.int16)
    Bm = torch.randn(N, ldk, device=dev).to(torch.bfloat16).view(torch.int16)
    C = torch.empty(M, N, device=dev)
    ws = torch.empty(1 << 26, device=dev)
    nk = (K + 63) // 64
    trace = torch.zeros(nk + 8, 8, dtype=torch.int64, device=dev)
    def run(): lib.lv_gemm_b16_tile(256, 0, M, N, K, 1.0, P(A), ldk, P(Bm), ldk, P(C), N, 0, None, 0, 1, None, 0, 1, P(ws), ws.numel(), s)
    cdll.lv_trace_set_gemm(None)
    for _ in range(2): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    cdll.lv_trace_set_gemm(ctypes.c_void_p(trace.data_ptr()))
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    us_t = e0.elapsed_time(e1) * 1e3
    tr = trace.cpu().numpy().astype(np.float64)[:, :4]
    ok = tr[:, 3] > 0
    tr = tr[ok][2:-1]
    ph = np.diff(tr, axis=1)
    nxt = tr[1:, 0] - tr[:-1, 3]
    per = np.diff(tr[:, 0])
    print("%-7s M=%d N=%d K=%d: %.1f us (%.0f TF), traced %.1f us; K tiles seen %d; cycles per K tile median %.0f (MFMA pipe alone: 2048 for the two waves of a SIMD); "
          "phases (clock ticks, median): MFMAs+DMA issue %.0f | wait DMA %.0f | barrier %.0f | to next tile %.0f" % (
          name, M, N, K, us, 2.0 * M * N * K / us / 1e6, us_t, len(tr), np.median(per), *np.median(ph, axis=0), np.median(nxt)))
cdll.lv_trace_set_gemm(None)
