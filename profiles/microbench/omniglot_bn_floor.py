"""Microbenchmark (measurement tooling): every BatchNorm call of ONE Omniglot inner step (image.py:300-314, B = 50) against its own
floor -- VERDICT r5 item 7.  The engine already brackets each BatchNorm call with HIP events and knows its algorithmic bytes
(engine._prof("batchnorm", bytes)); here the bracket also records WHERE it was made (forward with the statistics from the producing
convolution's epilogue: one launch; full forward: reduce + apply; backward apply-only; full backward: reduce + apply), the
tensor shape, and the floor  bytes / 6.3 TB/s (measured copy rate) + 1.7 us per launch (dependent kernel boundary between real
streaming kernels; /opt/skills/guides/MI355X_MICROARCH.md price list).  Steps are queued behind a device-side sleep so that the
kernels run back to back (the eager step is host-bound); times are medians over 6 profiled steps."""
import os, sys, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vae_lagging_encoder_amd import engine, image_engine
from vae_lagging_encoder_amd.factory import build_image_vae
from vae_lagging_encoder_amd.trainer import AggressiveImageTrainer

dev = torch.device("cuda:0")
precision = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
META = []
_base = engine._prof


class Tagged(_base):
    def __init__(self, name, work, launches=1):
        super().__init__(name, work, launches)
        if self.on and name == "batchnorm":
            f = sys._getframe(1)
            loc = f.f_locals
            Pn, C = loc.get("Pn"), loc.get("C")
            mult = work / (4.0 * Pn * C)
            if f.f_code.co_name == "bwd":
                kind, nl = ("backward, apply only (stage 1 in the data-gradient conv's epilogue)", 1) if mult == 3 else \
                           ("backward, reduce + apply, %d gradient summand(s)" % int(mult - 6), 2)
            else:
                x = loc.get("x")
                part = bool(getattr(x, "bn_nblk", 0))
                kind, nl = ("forward, apply only (statistics from the conv's epilogue)%s" % (" + residual" if loc.get("res") is not None else ""), 1) if part else \
                           ("forward, reduce + apply%s" % (" + residual" if loc.get("res") is not None else ""), 2)
            META.append((kind, nl, int(Pn), int(C), work))


engine._prof = Tagged
image_engine._eng._prof = Tagged
B = 50
vae = build_image_vae(dev, 783435)
tr = AggressiveImageTrainer(vae, lr=1e-3, clip=5.0, seed=783435, precision=precision, use_graph=False)
probs = torch.rand(8, B, 1, 28, 28).to(dev)
for i in range(3):
    tr.step(tr.binarize(probs[i]), 1.0)
torch.cuda.synchronize()
prof = {}
engine.PROFILE, engine.PROFILE_PREFIX = prof, "batchnorm"
NSTEP = 6
for i in range(NSTEP):
    torch.cuda._sleep(40_000_000)
    tr.step(tr.binarize(probs[i % 8]), 1.0)
torch.cuda.synchronize()
engine.PROFILE = engine.PROFILE_PREFIX = None
ev = prof["batchnorm"]
n = len(ev) // NSTEP
assert len(ev) == n * NSTEP == len(META), (len(ev), len(META))
us = np.array([[ev[s * n + i][0].elapsed_time(ev[s * n + i][1]) * 1e3 for i in range(n)] for s in range(NSTEP)])
med = np.median(us, axis=0)
HBM, BOUNDARY = 6.3e12, 1.7
rows = collections.OrderedDict()
tot_t = tot_f = 0.0
worst = []
for i in range(n):
    kind, nl, Pn, C, work = META[i]
    floor = work / HBM * 1e6 + BOUNDARY * nl
    key = (kind, Pn, C)
    r = rows.setdefault(key, [0, 0.0, 0.0, work, nl])
    r[0] += 1; r[1] += med[i]; r[2] += floor
    tot_t += med[i]; tot_f += floor
    worst.append((med[i] / floor, kind, Pn, C, med[i], floor))
print("# Omniglot inner step (B = 50, precision %s), eager, every BatchNorm call bracketed by HIP events: %d calls per step, "
      "%.3f ms per step in the group" % (precision, n, tot_t / 1e3))
print("# floor of a call = algorithmic bytes / 6.3 TB/s + 1.7 us per launch;  ratio = measured / floor")
print("%-78s %7s %4s %6s %9s %9s %9s %6s" % ("call kind", "P", "C", "calls", "MB/call", "us/call", "floor us", "ratio"))
for (kind, Pn, C), (cnt, t, f, work, nl) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print("%-78s %7d %4d %6d %9.2f %9.2f %9.2f %6.2f" % (kind, Pn, C, cnt, work / 1e6, t / cnt, f / cnt, t / f))
print("# group total: measured %.1f us, floor %.1f us, ratio %.2f; calls within 1.3x of their floor: %d of %d (%.0f %% of the group's time)" % (
    tot_t, tot_f, tot_t / tot_f, sum(1 for w in worst if w[0] <= 1.3), n, 100.0 * sum(w[4] for w in worst if w[0] <= 1.3) / tot_t))
nl_step = sum(m[1] for m in META[:n])
print("# of the floor, launch boundaries are %.1f us (%d launches x 1.7) and bytes / 6.3 TB/s %.1f us" % (
    BOUNDARY * nl_step, nl_step, tot_f - BOUNDARY * nl_step))
print("# NOTE: a call's time is taken between two HIP event records on the stream, which adds a constant of a few microseconds to every "
      "bracket (kernel-trace durations of the same launches: profiles/r06h_omniglot_kernel_stats.txt)")
