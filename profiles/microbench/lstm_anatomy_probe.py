"""Microbenchmark (measurement tooling): anatomy of one timestep of the persistent recurrences by WHAT-IF builds of the final
kernels -- liblvae_p16abl<N>.so = the product sources with -DLV_P16_ABL=N (lv_lstm_persist16.hip; bit 0: no MFMAs, bit 1: hand-off
tags not tested, i.e. no step waits for its producers, bit 2: no transcendental cell / gate-gradient math, bit 3: no workgroup
barrier).  A variant's results are garbage; its TIME is the product kernel's minus the removed phase's share of the critical path.
Unlike the LV_TRACE build (clock stores at every phase boundary: +35 % per timestep even with the trace switched off) the what-if
builds leave every other instruction of the kernel as it is.  Slope per timestep = (t(T = 200) - t(T = 40)) / 160, 20 launches each."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from types import SimpleNamespace
from vae_lagging_encoder_amd import _lib
from vae_lagging_encoder_amd.engine import P, stream_ptr, _xch_flags
here = os.path.dirname(os.path.abspath(__file__))
dev = torch.device("cuda:0"); s = stream_ptr(dev)
H = 1024
NAMES = {0: "product kernel", 1: "no MFMAs", 2: "no waiting for producers", 3: "no MFMAs, no waiting", 4: "no transcendental math",
         6: "no waiting, no transcendental math", 7: "no MFMAs, no waiting, no transcendental math", 8: "no workgroup barrier",
         10: "no waiting, no barrier", 15: "none of the four (loads, LDS traffic, stores, loop)",
         16: "gathered granules not staged through LDS (fwd)", 18: "no waiting, no LDS staging of the gather (fwd)",
         32: "h granules not loaded (fwd)", 34: "no waiting, h granules not loaded (fwd)",
         50: "no waiting, no granule loads, no LDS staging (fwd)",
         64: "BPTT exchange as 16-byte accesses (what-if: half the load / store instructions; tags not tested)",
         66: "no waiting + BPTT exchange as 16-byte accesses", 63: "none of the six (fwd: Gx loads, quarter-sum LDS, stores, loop)"}


def timeit(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def slopes(lib, B, R, f16):
    whh = (torch.rand(4 * H, H, device=dev) * 2 - 1) * 0.03
    n = lib.lv_lstm_persist16_wpk_floats()
    wf, wb = torch.empty(n, device=dev), torch.empty(n, device=dev)
    lib.lv_lstm_persist16_pack(P(whh), P(wf), 2 if f16 else 0, H, s)
    lib.lv_lstm_persist16_pack(P(whh), P(wb), 1, H, s)
    xch = torch.zeros(lib.lv_lstm_persist16_xch_floats(), device=dev)
    st = torch.zeros(1, dtype=torch.int32, device=dev)
    wi = SimpleNamespace(xstate={"f": 0, "g": 0, "gcls": [0, 0]})
    res = {}
    for T in (40, 200):
        gx = torch.randn(T, B, 4 * H, device=dev) * 0.5
        hs = torch.zeros(T + 1, B, H, device=dev); cs = torch.zeros(T + 1, B, H, device=dev)
        saved = torch.empty(lib.lv_lstm_persist16_saved_floats(T, R), device=dev)
        dO = torch.randn(T, B, H, device=dev) * 0.1
        dG16 = torch.empty(T, B, 4 * H, dtype=torch.int16, device=dev)
        dGsum = torch.empty(B, 4 * H, device=dev); dc0 = torch.empty(B, H, device=dev)
        lib.lv_lstm_persist16_xch_clear(P(xch), s)
        wi.xstate = {"f": 0, "g": 0, "gcls": [0, 0]}
        f = timeit(lambda: lib.lv_lstm_fwd_bf16_persist16(P(gx), P(wf), P(hs), P(cs), P(saved), P(xch), P(st), T, B, R,
                                                          1 | _xch_flags(wi, "f", R, "cpu") | (32 if f16 else 0), H, s))
        b = timeit(lambda: lib.lv_lstm_bwd_bf16_persist16(P(dO), None, P(wb), P(saved), P(hs), P(cs), P(dG16), P(dGsum), P(xch), P(st),
                                                          None, P(dc0), 1, T, B, R, 1 | _xch_flags(wi, "g", R, "cpu"), H, s))
        res[T] = (f, b)
    st.zero_()
    return (res[200][0] - res[40][0]) / 160.0, (res[200][1] - res[40][1]) / 160.0


KNOBS = {"sbb4": "LV_SBB16=4 (BPTT input block: 4 timesteps; shipped 8)", "sbb2": "LV_SBB16=2", "sbb16": "LV_SBB16=16 (spills: 1040 B/lane)",
         "hb4sbb4": "LV_HB16=4 LV_SBB16=4 (one receive round of 32 granules per lane)", "hb1": "LV_HB16=1 (four receive rounds)",
         "sbk2": "LV_SBK16=2 (forward I/O block: 2 timesteps; shipped 4)", "sbk8": "LV_SBK16=8 (spills: 60 B/lane)",
         "gj8": "LV_GJ16=8 (forward: 8 granules per lane per poll round; shipped 16)", "gj32": "LV_GJ16=32 (one poll round)"}
variants = [(int(a) if a.isdigit() else a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4, 6, 7, 8, 10, 15, 16, 18, 32, 34, 50, 63]
print("what-if builds of the final persistent kernels (-DLV_P16_ABL=N): microseconds per timestep, slope between T = 40 and T = 200")
print("%-58s | %-23s | %-23s" % ("variant", "B = 32, 4 rows / group", "B = 128, 16 rows / group"))
print("%-58s | %-11s %-11s | %-11s %-11s" % ("", "forward", "BPTT", "forward", "BPTT"))
prod = _lib.load()
for tag, lib in [("shipped library (liblvae_hip.so)", prod)] + [(None, v) for v in variants]:
    if tag is None:
        path = os.path.join(here, ("liblvae_p16abl%d.so" % lib) if isinstance(lib, int) else ("liblvae_p16%s.so" % lib))
        if not os.path.exists(path):
            continue
        tag = ("ABL=%-2d %s" % (lib, NAMES.get(lib, ""))) if isinstance(lib, int) else KNOBS.get(lib, lib)
        lib = _lib.bind(ctypes.CDLL(path), path)
    a = slopes(lib, 32, 4, True)
    b = slopes(lib, 128, 16, True)
    print("%-72s | %-11.3f %-11.3f | %-11.3f %-11.3f" % (tag, a[0], a[1], b[0], b[1]), flush=True)
