"""Soak test (measurement tooling): many aggressive inner steps on batches of varying shape (B <= 32, T <= 200) through the
persistent LSTM launches, checking the finiteness of the loss and, at the end, that no step had to be replayed (ladder rung 0, no recoveries); prints steps/s.
usage (GPU box): python profiles/microbench/soak_persistent.py [steps] [max batch] [exact | dp]
dp (round 6): the same soak with the data-parallel exchange FORCED ON over a one-rank RCCL process group (GradSync(force=True)): every
step issues its asynchronous bf16 all-reduces / reduce-scatter from inside the encoder backward, RCCL's stream beside the persistent launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from vae_lagging_encoder_amd.factory import build_text_vae
from vae_lagging_encoder_amd.trainer import AggressiveTextTrainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
BMAX = int(sys.argv[2]) if len(sys.argv) > 2 else 32          # 128: also the 8- and 16-row instantiations of the persistent kernels
dev = torch.device("cuda:0")
V = 20001
vae = build_text_vae(V, 512, 1024, 32, dev, seed=3)
EXACT = len(sys.argv) > 3 and sys.argv[3] == "exact"           # the encoder's two-pass exact forward (two forward launches in a row)
DP = len(sys.argv) > 3 and sys.argv[3] == "dp"
gs = None
if DP:
    from vae_lagging_encoder_amd import dist as lvdist
    os.environ.setdefault("MASTER_PORT", "29577")
    lvdist.init_from_env(force=True, banner=True)
    gs = lvdist.GradSync(mode="strict", force=True)
tr = AggressiveTextTrainer(vae, lr=0.05, clip=5.0, precision="bf16", seed=11, encoder_forward="f32" if EXACT else None, grad_sync=gs)
rs = np.random.RandomState(5)
t0 = time.time()
shapes = set()
for i in range(steps):
    B = int(rs.randint(1, BMAX + 1))
    T = int(rs.randint(3, 201))
    shapes.add((B, T))
    x = torch.from_numpy(rs.randint(4, V - 1, size=(B, T))).to(dev)
    tr.step(x, 0.7)
    if i % 100 == 99:
        st = tr.read_stats()            # settles the transaction block: a timed-out hand-off would be replayed one ladder rung down
        assert np.isfinite(st["loss_sum"]), st
        tr.reset_stats()
torch.cuda.synchronize()
from vae_lagging_encoder_amd import engine
print("ladder rung at the end: enc %d / dec %d (0 = XCD-local hand-off), recoveries %d, status words %s / %s" % (
    engine.persist_rung(tr.enc), engine.persist_rung(tr.dec), tr.recoveries, tr.enc.status.tolist(), tr.dec.status.tolist()))
if DP:
    import torch.distributed as dist
    print("data-parallel exchange forced on: backend %s, payload %s" % (dist.get_backend(), gs.payload))
print("soak ok: %d steps, %d distinct (B, T) shapes, %.1f steps/s, peak device memory %.1f GB (workspace caches: enc %.1f GB / %d shapes"
      ", dec %.1f GB / %d shapes)" % (steps, len(shapes), steps / (time.time() - t0), torch.cuda.max_memory_allocated() / 1e9,
                                      tr.enc.wsc.total / 1e9, len(tr.enc.wsc.cache), tr.dec.wsc.total / 1e9, len(tr.dec.wsc.cache)))
