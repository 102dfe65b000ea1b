"""Microbenchmark (measurement tooling): the persistent recurrences of two builds of the library side by side -- every output compared
BIT FOR BIT on the same inputs (where the bits differ: the largest difference beside the largest value) (hs, cs, the saved records, dG16, dGsum, dc0), then microseconds per timestep (slope between T = 40 and
T = 200, 20 launches each, the two libraries alternating).   usage: lstm_swap_ab.py <other .so> [label]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from types import SimpleNamespace
from vae_lagging_encoder_amd import _lib
from vae_lagging_encoder_amd.engine import P, stream_ptr, _xch_flags
dev = torch.device("cuda:0"); s = stream_ptr(dev)
H = 1024
other_path = sys.argv[1]
label = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(other_path)
libs = [("shipped", _lib.load()), (label, _lib.bind(ctypes.CDLL(other_path), other_path))]


def timeit(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def run(lib, B, R, f16, T, seed, time_it):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    whh = (torch.rand(4 * H, H, device=dev, generator=g) * 2 - 1) * 0.03
    n = lib.lv_lstm_persist16_wpk_floats()
    wf, wb = torch.empty(n, device=dev), torch.empty(n, device=dev)
    lib.lv_lstm_persist16_pack(P(whh), P(wf), 2 if f16 else 0, H, s)
    lib.lv_lstm_persist16_pack(P(whh), P(wb), 1, H, s)
    xch = torch.zeros(lib.lv_lstm_persist16_xch_floats(), device=dev)
    st = torch.zeros(1, dtype=torch.int32, device=dev)
    wi = SimpleNamespace(xstate={"f": 0, "g": 0, "gcls": [0, 0]})
    gx = torch.randn(T, B, 4 * H, device=dev, generator=g) * 0.5
    hs = torch.zeros(T + 1, B, H, device=dev); cs = torch.zeros(T + 1, B, H, device=dev)
    hs[0] = torch.randn(B, H, device=dev, generator=g) * 0.1; cs[0] = torch.randn(B, H, device=dev, generator=g) * 0.1
    saved = torch.zeros(lib.lv_lstm_persist16_saved_floats(T, R), device=dev)
    dO = torch.randn(T, B, H, device=dev, generator=g) * 0.1
    dG16 = torch.zeros(T, B, 4 * H, dtype=torch.int16, device=dev)
    dGsum = torch.zeros(B, 4 * H, device=dev); dc0 = torch.zeros(B, H, device=dev)
    fwd = lambda: lib.lv_lstm_fwd_bf16_persist16(P(gx), P(wf), P(hs), P(cs), P(saved), P(xch), P(st), T, B, R,
                                                 1 | _xch_flags(wi, "f", R, "cpu") | (32 if f16 else 0), H, s)
    bwd = lambda: lib.lv_lstm_bwd_bf16_persist16(P(dO), None, P(wb), P(saved), P(hs), P(cs), P(dG16), P(dGsum), P(xch), P(st),
                                                 None, P(dc0), 1, T, B, R, 1 | _xch_flags(wi, "g", R, "cpu"), H, s)
    fwd(); bwd(); torch.cuda.synchronize()
    outs = [t.clone() for t in (hs, cs, saved, dG16, dGsum, dc0)]
    assert int(st.item()) == 0
    tf = tb = None
    if time_it:
        tf, tb = timeit(fwd), timeit(bwd)
    return outs, tf, tb


names = ["hs", "cs", "saved", "dG16", "dGsum", "dc0"]
for B, R, f16 in [(32, 4, True), (32, 4, False), (20, 4, True), (64, 8, True), (128, 16, True), (100, 16, False)]:
    res = {}
    for T in (40, 200):
        per = []
        for name, lib in libs:
            per.append(run(lib, B, R, f16, T, 1234 + T + B, True))
        def cmp(a, b):
            if torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a, b.view(torch.int32) if b.dtype == torch.float32 else b):
                return True
            if a.dtype == torch.int16:      # bf16 images: compare as values
                a, b = (a.view(torch.bfloat16).float(), b.view(torch.bfloat16).float())
            return "%.1e of %.1e" % (float((a - b).abs().max()), float(b.abs().max()))      # max |difference| of max |value|
        same = [cmp(a, b) for a, b in zip(per[0][0], per[1][0])]
        res[T] = (per, same)
    sl = [((res[200][0][i][1] - res[40][0][i][1]) / 160.0, (res[200][0][i][2] - res[40][0][i][2]) / 160.0) for i in range(2)]
    print("B=%3d R=%2d %s | bit-identical outputs (T=40 / T=200): %s / %s | us per timestep fwd / BPTT: %s %.3f / %.3f | %s %.3f / %.3f"
          % (B, R, "f16 forward" if f16 else "bf16 forward", dict(zip(names, res[40][1])) if not all(x is True for x in res[40][1]) else "all six",
             dict(zip(names, res[200][1])) if not all(x is True for x in res[200][1]) else "all six", libs[0][0], sl[0][0], sl[0][1], libs[1][0], sl[1][0], sl[1][1]), flush=True)
