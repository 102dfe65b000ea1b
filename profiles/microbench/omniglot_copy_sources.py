"""Measurement tooling: which host-side torch calls of one eager Omniglot step turn into device-to-device copy kernels
(__amd_rocclr_copyBuffer in the rocprof summary).  torch.profiler with stacks, one step."""
import os, sys, collections, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vae_lagging_encoder_amd.factory import build_image_vae
from vae_lagging_encoder_amd.trainer import AggressiveImageTrainer
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
vae = build_image_vae(dev, 783435)
tr = AggressiveImageTrainer(vae, lr=1e-3, clip=5.0, seed=783435, precision="f32", use_graph=False)
probs = torch.rand(4, 50, 1, 28, 28).to(dev)
for i in range(3):
    tr.step(tr.binarize(probs[i]), 1.0)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(tr.binarize(probs[3]), 1.0)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::fill_", "aten::zero_", "aten::zeros", "aten::full", "aten::_foreach_add_"):
        st = [f for f in (ev.stack or []) if "vae_lagging_encoder_amd" in f or "bench" in f]
        cnt[(ev.name, st[0] if st else "?")] += 1
for (name, where), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print("%4d  %-22s %s" % (n, name, where))
kern = collections.Counter(ev.name for ev in prof.events() if ev.device_type is not None and str(ev.device_type).endswith("CUDA"))
for k, n in kern.most_common(12):
    print("%5d  %s" % (n, k[:110]))
