"""Microbenchmark (measurement tooling): the input-side backward of one LSTM layer -- dX = dG . W_ih and [dW_ih | dW_hh] = dG^T [X ; h_prev] --
as the separate launches (lv_gemm_b16 with its split-K reduce + lv_gemm_b16_dual with its reduce) against ONE grouped stream-K launch
(lv_gemm_b16_pair), interleaved, at the bench shapes; plus a bit-identity soak of the in-launch hand-off while a second stream keeps
some CUs busy (uneven load)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vae_lagging_encoder_amd import _lib
from vae_lagging_encoder_amd.engine import P, stream_ptr
dev = torch.device("cuda:0")
L = _lib.load()
alts = [(os.path.basename(q).replace("liblvae_", "").replace(".so", ""), _lib.bind(ctypes.CDLL(q), q))
        for q in os.environ.get("LVAE_PROBE_LIBS", "").split(",") if q]           # alternative builds (profiles/microbench/build_gemm_variant.sh)
s = stream_ptr(dev)
ws = torch.empty(1 << 26, device=dev)
H, ni = 1024, 512


def b16(*shape):
    return (torch.randn(*shape, device=dev) * 0.1).to(torch.bfloat16).view(torch.int16)


def med(fn, n=20, reps=5):
    out = []
    for _ in range(reps):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / n)
    return sorted(out)[reps // 2]


for name, TB in (("yahoo B=32 T=200", 6400), ("yelp B=32 T~100", 3200), ("stress B=128 T=200", 25600), ("short B=32 T=50", 1600)):
    ldr = TB
    dG = b16(TB, 4 * H); XhT = b16(ni + H, ldr); WT = b16(ni, 4 * H)
    dX = torch.empty(TB, ni, device=dev); gWi = torch.empty(4 * H, ni, device=dev); gWh = torch.empty(4 * H, H, device=dev)
    dX2 = torch.empty_like(dX); gWi2 = torch.empty_like(gWi); gWh2 = torch.empty_like(gWh)

    dual_ok = L.lv_gemm_b16_dual_supported(4 * H, ni + H, TB, ws.numel())

    def old():
        L.lv_gemm_b16(0, TB, ni, 4 * H, 1.0, P(dG), 4 * H, P(WT), 4 * H, P(dX), ni, 0, None, 0, 1, None, 0, 1, P(ws), ws.numel(), s)
        if dual_ok:
            L.lv_gemm_b16_dual(1, 4 * H, ni + H, TB, P(dG), 4 * H, P(XhT), ldr, P(gWi), ni, ni, P(gWh), H, P(ws), ws.numel(), s)
        else:       # (the 256-tile route: two products, as the engine falls back)
            L.lv_gemm_b16(1, 4 * H, ni, TB, 1.0, P(dG), 4 * H, P(XhT), ldr, P(gWi), ni, 0, None, 0, 1, None, 0, 1, P(ws), ws.numel(), s)
            L.lv_gemm_b16(1, 4 * H, H, TB, 1.0, P(dG), 4 * H, P(XhT, ni * ldr), ldr, P(gWh), H, 0, None, 0, 1, None, 0, 1, P(ws), ws.numel(), s)

    def new(L=L):
        L.lv_gemm_b16_pair(1, 4 * H, ni + H, TB, P(dG), 4 * H, P(XhT), ldr, P(gWi2), ni, ni, P(gWh2), H,
                           0, TB, ni, 4 * H, P(dG), 4 * H, P(WT), 4 * H, P(dX2), ni, P(ws), ws.numel(), s)

    def new_dual_only():
        L.lv_gemm_b16_pair(1, 4 * H, ni + H, TB, P(dG), 4 * H, P(XhT), ldr, P(gWi2), ni, ni, P(gWh2), H,
                           0, 0, 0, 0, None, 0, None, 0, None, 0, P(ws), ws.numel(), s)

    def new_dx_only():
        L.lv_gemm_b16_pair(0, TB, ni, 4 * H, P(dG), 4 * H, P(WT), 4 * H, P(dX2), ni, 0, None, 0,
                           0, 0, 0, 0, None, 0, None, 0, None, 0, P(ws), ws.numel(), s)

    gf = (2.0 * 4 * H * (ni + H) * TB + 2.0 * TB * ni * 4 * H) / 1e9
    print("%-20s supported=%d  %.1f GFLOP" % (name, L.lv_gemm_b16_pair_supported(1, 4 * H, ni + H, TB, 0, TB, ni, 4 * H, ws.numel()), gf), flush=True)
    old(); new(); torch.cuda.synchronize()
    for a, b, nm in ((dX, dX2, "dX"), (gWi, gWi2, "dW_ih"), (gWh, gWh2, "dW_hh")):
        sc = float(a.abs().max())
        print("   %-6s max|pair - separate| / max = %.2e" % (nm, float((a - b).abs().max()) / sc))
    ref = (dG.view(torch.bfloat16).double() @ WT.view(torch.bfloat16).double().t())
    print("   dX vs f64: %.2e" % (float((dX2.double() - ref).abs().max()) / float(ref.abs().max())))
    ref = (dG.view(torch.bfloat16).double().t() @ XhT.view(torch.bfloat16).double()[:, :TB].t())
    got = torch.cat([gWi2, gWh2], 1).double()
    print("   dW vs f64: %.2e" % (float((got - ref).abs().max()) / float(ref.abs().max())))
    del ref, got
    for rep in range(3):
        a, b = med(old), med(new)
        c, d = med(new_dual_only), med(new_dx_only)
        line = "   separate %7.1f us %6.1f TF | pair %7.1f us %6.1f TF | pair(dual only) %7.1f us | pair(dX only) %7.1f us" % (a, gf / a * 1e3, b, gf / b * 1e3, c, d)
        for nm, La in alts:
            line += " | %s %7.1f us" % (nm, med(lambda: new(La)))
        print(line, flush=True)

# the two ways a tile can be summed give the same bits: the product build (the closing contributor usually finds the others arrived and
# sums into its registers) against a build whose closing contributor never looks (LV_SK_SPIN=0: every piece travels, the last ticket sums)
for nm, La in alts:
    if "spin0" not in nm:
        continue
    TB = 6400
    dG = b16(TB, 4 * H); XhT = b16(ni + H, TB); WT = b16(ni, 4 * H)
    res = []
    for lib in (L, La):
        o = [torch.empty(TB, ni, device=dev), torch.empty(4 * H, ni, device=dev), torch.empty(4 * H, H, device=dev)]
        lib.lv_gemm_b16_pair(1, 4 * H, ni + H, TB, P(dG), 4 * H, P(XhT), TB, P(o[1]), ni, ni, P(o[2]), H,
                             0, TB, ni, 4 * H, P(dG), 4 * H, P(WT), 4 * H, P(o[0]), ni, P(ws), ws.numel(), s)
        torch.cuda.synchronize()
        res.append(o)
    print("product build vs %s: bit-identical = %s" % (nm, all(torch.equal(a, b) for a, b in zip(*res))))

# soak: the hand-off under uneven load -- a second stream runs a long, narrow kernel chain (a few CUs busy, the rest free), the
# grouped launch runs 300 times; every result must be bit-identical to the first
TB = 6400
dG = b16(TB, 4 * H); XhT = b16(ni + H, TB); WT = b16(ni, 4 * H)
outs = [torch.empty(TB, ni, device=dev), torch.empty(4 * H, ni, device=dev), torch.empty(4 * H, H, device=dev)]


def run():
    L.lv_gemm_b16_pair(1, 4 * H, ni + H, TB, P(dG), 4 * H, P(XhT), TB, P(outs[1]), ni, ni, P(outs[2]), H,
                       0, TB, ni, 4 * H, P(dG), 4 * H, P(WT), 4 * H, P(outs[0]), ni, P(ws), ws.numel(), s)


run(); torch.cuda.synchronize()
first = [o.clone() for o in outs]
side = torch.cuda.Stream()
junk = torch.randn(64, 4096, device=dev)
bad = 0
for it in range(300):
    with torch.cuda.stream(side):
        for _ in range(4):
            junk = torch.tanh(junk @ junk.t()[:, :64].contiguous() @ junk)
    for o in outs:
        o.fill_(float("nan"))
    run()
    torch.cuda.synchronize()
    bad += sum(int(not torch.equal(o, f)) for o, f in zip(outs, first))
print("soak under uneven load: %d launches, %d mismatching outputs" % (300, bad))
