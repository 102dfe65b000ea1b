"""Microbenchmark (measurement tooling): the persistent recurrences of lv_lstm_persist16.hip by rows per
XCD group -- us per timestep of the forward and the BPTT at T = 200 -- and experiment A of the round-2 review: a B = 32
recurrence on FOUR XCD groups (8 rows each) with a large GEMM queued beside it on a second stream (does the GEMM get the four
idle XCDs, and what does the recurrence pay?)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vae_lagging_encoder_amd import _lib
from vae_lagging_encoder_amd.engine import P, stream_ptr
dev = torch.device("cuda:0"); s = stream_ptr(dev)
if os.environ.get("LVAE_PROBE_LIB"):            # an alternative build of the kernel library (a measurement knob compiled in)
    import ctypes
    lib = _lib.bind(ctypes.CDLL(os.environ["LVAE_PROBE_LIB"]), os.environ["LVAE_PROBE_LIB"])
else:
    lib = _lib.load()
T, H = 200, 1024
whh = (torch.rand(4 * H, H, device=dev) * 2 - 1) * 0.03
n = lib.lv_lstm_persist16_wpk_floats()
wf16, wb16 = (torch.empty(n, device=dev) for _ in range(2))
lib.lv_lstm_persist16_pack(P(whh), P(wf16), 0, H, s)
lib.lv_lstm_persist16_pack(P(whh), P(wb16), 1, H, s)
xch = torch.empty(lib.lv_lstm_persist16_xch_floats(), device=dev)
st = torch.zeros(1, dtype=torch.int32, device=dev)


def t(f, n=6):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def bufs(B):
    g = torch.Generator(device="cpu").manual_seed(B)
    gx = (torch.randn(T, B, 4 * H, generator=g) * 0.5).to(dev)
    hs = torch.zeros(T + 1, B, H, device=dev); cs = torch.zeros(T + 1, B, H, device=dev)
    gates = torch.empty(max(T * B * 4 * H, lib.lv_lstm_persist16_saved_floats(T, 16)), device=dev)      # also the 16-row kernels' saved-activation buffer
    dO = (torch.randn(T, B, H, generator=g) * 0.1).to(dev)
    dG16 = torch.empty(T, B, 4 * H, dtype=torch.int16, device=dev)
    dGsum = torch.empty(B, 4 * H, device=dev); dc0 = torch.empty(B, H, device=dev)
    return gx, hs, cs, gates, dO, dG16, dGsum, dc0


print("forward / BPTT, us per timestep (T = %d)" % T)
for B, R in ((32, 4), (32, 8), (32, 16), (64, 8), (128, 16)):
    gx, hs, cs, gates, dO, dG16, dGsum, dc0 = bufs(B)
    line = "B=%3d R=%2d (%d groups): " % (B, R, (B + R - 1) // R)
    for fl in (0, 1):
        a = t(lambda: lib.lv_lstm_fwd_bf16_persist16(P(gx), P(wf16), P(hs), P(cs), P(gates), P(xch), P(st), T, B, R, fl, H, s))
        line += "fwd k16%s %.2f | " % ("/L2" if fl else "", a / T)
    for fl in (0, 1):
        a = t(lambda: lib.lv_lstm_bwd_bf16_persist16(P(dO), None, P(wb16), P(gates), P(hs), P(cs), P(dG16), P(dGsum), P(xch), P(st), None, P(dc0), 1, T, B, R, fl, H, s))
        line += "bwd rs16%s %.2f | " % ("/L2" if fl else "", a / T)
    line += "status %d" % int(st.item())
    print(line)

# ---- experiment A: half-chip recurrence with a GEMM beside it -------------------------------------------------------------------------
B = 32
gx, hs, cs, gates, dO, dG16, dGsum, dc0 = bufs(B)
V, R_ = 20001, 6368
ldv = (V + 31) // 32 * 32
dl = torch.randn(R_, ldv, device=dev).to(torch.bfloat16).view(torch.int16)
OT = torch.randn(H, R_, device=dev).to(torch.bfloat16).view(torch.int16)
dW = torch.empty(V, H, device=dev)
ws = torch.empty(1 << 26, device=dev)
side = torch.cuda.Stream(dev)


def gemm(stream):
    lib.lv_gemm_b16(1, V, H, R_, 1.0, P(dl), ldv, P(OT), R_, P(dW), H, 0, None, 0, 1, None, 0, 1, P(ws), ws.numel(), stream)


def wall(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


g_alone = wall(lambda: gemm(s))
print("dW_pred GEMM alone (full chip): %.1f us" % g_alone)
for R in (4, 8):
    def bptt():
        lib.lv_lstm_bwd_bf16_persist16(P(dO), None, P(wb16), P(gates), P(hs), P(cs), P(dG16), P(dGsum), P(xch), P(st), None, P(dc0), 1, T, B, R, 0, H, s)
    alone = wall(bptt)

    def both():
        ev = torch.cuda.Event(); ev.record()
        bptt()                                              # the recurrence first: its groups take their XCDs at once
        side.wait_event(ev)
        with torch.cuda.stream(side):
            gemm(side.cuda_stream)
        torch.cuda.current_stream().wait_stream(side)
    tot = wall(both)

    def both_gemm_first():
        ev = torch.cuda.Event(); ev.record()
        side.wait_event(ev)
        with torch.cuda.stream(side):
            gemm(side.cuda_stream)                          # the GEMM first: the recurrence's workgroups take CUs as its tiles retire
        bptt()
        torch.cuda.current_stream().wait_stream(side)
    tot2 = wall(both_gemm_first)
    print("  (GEMM queued first: %.1f us)" % tot2)
    print("BPTT R=%d alone %.1f us (%.2f us/step); BPTT + dW_pred on a second stream: %.1f us  (serial would be %.1f)  status %d" % (
        R, alone, alone / T, tot, alone + g_alone, int(st.item())))
