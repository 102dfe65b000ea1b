"""Measurement tooling: which call sites still ask the caching allocator for device memory in a STEADY-STATE fused step
(bench.py side_runs.mixed_shapes reports the count; this names the sites).  Yahoo dims, bf16, three shapes, counts per step."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vae_lagging_encoder_amd.factory import build_text_vae, synthetic_batch
from vae_lagging_encoder_amd.trainer import AggressiveTextTrainer
dev = torch.device("cuda:0")
V = 20001
vae = build_text_vae(V, 512, 1024, 32, dev, seed=1)
tr = AggressiveTextTrainer(vae, lr=1.0, clip=5.0, precision="bf16")
pool = [synthetic_batch(b, t, V, seed=i).to(dev) for i, (b, t) in enumerate([(32, 200), (32, 77), (11, 40)])]
tr.prepare_batches(pool)
for _ in range(2):
    for x in pool:
        tr.step(x, 0.1)
tr.commit()
torch.cuda.synchronize()
m0 = torch.cuda.memory_stats(dev)["allocation.all.allocated"]
torch.cuda.memory._record_memory_history(max_entries=100000)
N = 12
for i in range(N):
    tr.step(pool[i % 3], 0.1)
tr.read_stats()
torch.cuda.synchronize()
m1 = torch.cuda.memory_stats(dev)["allocation.all.allocated"]
snap = torch.cuda.memory._snapshot()
torch.cuda.memory._record_memory_history(enabled=None)
sites = collections.Counter()
for ev in snap.get("device_traces", [[]])[0]:
    if ev.get("action") == "alloc":
        fr = [f for f in ev.get("frames", []) if "/vae_lagging_encoder_amd/" in f.get("filename", "") or f.get("filename", "").endswith("bench.py")]
        key = "%s:%d %s" % (fr[0]["filename"].split("/")[-1], fr[0]["line"], fr[0]["name"]) if fr else "(outside the package)"
        sites[(key, ev.get("size"))] += 1
print("allocator requests over %d steady-state steps + 1 read_stats: %d" % (N, m1 - m0))
for (k, sz), n in sites.most_common(20):
    print("  %3d x %10d bytes  %s" % (n, sz, k))
