"""Times the direct 32 -> 32 convolution entries (lv_conv32_f32 forward / data gradient, lv_conv32_wgrad_f32, pointwise and
BatchNorm entries) at the Omniglot decoder's shapes and at batch sizes that change the workgroups-per-CU balance.
usage (GPU box): python profiles/microbench/conv32_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vae_lagging_encoder_amd.engine import backend_for, stream_ptr, P

dev = torch.device("cuda:0")
lib = backend_for(dev)
s = stream_ptr(dev)


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for N in (36, 50, 73, 100):
    for k in (3, 5, 7):
        nt = (k // 2) * k + k // 2 + 1
        x = torch.randn(N * 784, 32, device=dev)
        dy = torch.randn(N * 784, 32, device=dev)
        w = torch.randn(32, 32, k, k, device=dev) * 0.05
        wp = torch.empty(lib.lv_conv32_wpack_floats(nt), device=dev)
        lib.lv_conv32_pack_f32(P(w), P(wp), k, nt, 0, s)
        y = torch.empty(N * 784, 32, device=dev)
        part = torch.empty(lib.lv_conv32_blocks(N) * 64, device=dev)
        dw = torch.empty(32, 32, k, k, device=dev)
        ws = torch.empty(lib.lv_conv32_wgrad_ws_floats(N, k), device=dev)
        t_f = timeit(lambda: lib.lv_conv32_f32(P(x), P(wp), P(y), N, k, nt, 0, 0, s))
        t_s = timeit(lambda: lib.lv_conv32_bnstat_f32(P(x), P(wp), P(y), P(part), N, k, nt, s))
        t_w = timeit(lambda: lib.lv_conv32_wgrad_f32(P(x), P(dy), P(dw), P(ws), N, k, 0, s))
        fl = 2.0 * N * 784 * 32 * 32
        wp16 = torch.empty(lib.lv_conv32_wpack_floats(nt), device=dev)
        lib.lv_conv32_pack_b16(P(w), P(wp16), k, nt, 0, s)
        t_b3 = timeit(lambda: lib.lv_conv32_b16(P(x), P(wp16), P(y), P(part), N, k, nt, 0, 0, 3, s))
        t_b1 = timeit(lambda: lib.lv_conv32_b16(P(x), P(wp16), P(y), P(part), N, k, nt, 0, 0, 1, s))
        t_w1 = timeit(lambda: lib.lv_conv32_wgrad_f32(P(x), P(dy), None, P(ws), N, k, 0, s))
        t_w3 = timeit(lambda: lib.lv_conv32_wgrad_b16(P(x), P(dy), None, P(ws), N, k, 0, 3, s))
        t_wb = timeit(lambda: lib.lv_conv32_wgrad_b16(P(x), P(dy), None, P(ws), N, k, 0, 1, s))
        print("N=%3d k=%d nt=%2d  fwd %6.1f us (%5.1f TF)  fwd+bnstat %6.1f us  split-bf16 x3 %6.1f us  bf16 %6.1f us | wgrad(+reduce) %6.1f us (%5.1f TF)"
              "  stage 1 alone: f32 %6.1f  x3 %6.1f  bf16 %6.1f us" %
              (N, k, nt, t_f, fl * nt / t_f / 1e6, t_s, t_b3, t_b1, t_w, fl * k * k / t_w / 1e6, t_w1, t_w3, t_wb))
for N in (50,):
    Pn = N * 784
    for Cin, Cout in ((64, 32), (32, 64)):
        x = torch.randn(Pn, Cin, device=dev)
        dy = torch.randn(Pn, Cout, device=dev)
        w = torch.randn(Cout, Cin, device=dev)
        y = torch.empty(Pn, Cout, device=dev)
        dx = torch.empty(Pn, Cin, device=dev)
        dw = torch.empty(Cout, Cin, device=dev)
        ws = torch.empty(lib.lv_conv1x1_wgrad_ws_floats(Cin, Cout), device=dev)
        t_f = timeit(lambda: lib.lv_conv1x1_f32(P(x), P(w), P(y), Pn, Cin, Cout, 0, 0, s))
        t_d = timeit(lambda: lib.lv_conv1x1_f32(P(dy), P(w), P(dx), Pn, Cout, Cin, 1, 0, s))
        t_w = timeit(lambda: lib.lv_conv1x1_wgrad_f32(P(x), P(dy), P(dw), P(ws), Pn, Cin, Cout, 0, s))
        mb = Pn * (Cin + Cout) * 4 / 1e6
        print("1x1 %d->%d  fwd %5.1f us (%4.2f TB/s)  dgrad %5.1f us  wgrad(+reduce) %5.1f us" % (Cin, Cout, t_f, mb / t_f, t_d, t_w))
    for C in (32, 64):
        x = torch.randn(Pn, C, device=dev)
        res = torch.randn(Pn, C, device=dev)
        dy = torch.randn(Pn, C, device=dev)
        y, dv, dx = torch.empty(Pn, C, device=dev), torch.empty(Pn, C, device=dev), torch.empty(Pn, C, device=dev)
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        mean, invstd, rm, rv = (torch.zeros(C, device=dev) for _ in range(4))
        dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
        ws = torch.empty(lib.lv_bn_workspace_floats(C) + 2 * C, device=dev)
        t_f = timeit(lambda: lib.lv_bn_fwd_f32(P(x), P(g), P(b), P(res), 1, P(y), P(mean), P(invstd), P(rm), P(rv), 1e-5, 0.1, P(ws), Pn, C, s))
        t_p = timeit(lambda: lib.lv_bn_fwd_partials_f32(P(x), P(g), P(b), P(res), 1, P(y), P(mean), P(invstd), P(rm), P(rv), 1e-5, 0.1, P(ws), 245, Pn, C, s))
        t_p7 = timeit(lambda: lib.lv_bn_fwd_partials_f32(P(x), P(g), P(b), P(res), 1, P(y), P(mean), P(invstd), P(rm), P(rv), 1e-5, 0.1, P(ws), 700, Pn, C, s))
        t_b = timeit(lambda: lib.lv_bn_bwd_f32(P(x), P(dy), P(y), P(mean), P(invstd), P(g), 1, P(dv), P(dx), P(dg), P(db), 0, P(ws), Pn, C, s))
        mb = Pn * C * 4 / 1e6
        print("BN C=%d  fwd(2 launches) %5.1f us  fwd from 245 / 700 partial blocks %5.1f / %5.1f us (%4.2f TB/s)  bwd(2 launches) %5.1f us (%4.2f TB/s)" %
              (C, t_f, t_p, t_p7, 3 * mb / t_p, t_b, 7 * mb / t_b))
