"""Microbenchmark (measurement tooling): the Omniglot inner step (hipGraph replay, B = 50, bf16x3) on VARIANT builds of the kernel
library -- one fresh process per library (profiles/microbench/liblvae_<name>.so, or "product"), alternating, several rounds, so
that box drift shows as spread inside a column rather than as a difference between columns."""
import ctypes, os, subprocess, sys, time
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
if len(sys.argv) > 2 and sys.argv[1] == "--child":
    sys.path.insert(0, root)
    import numpy as np, torch
    from vae_lagging_encoder_amd import _lib
    name, prec = sys.argv[2], sys.argv[3]
    if name != "product":
        path = os.path.join(here, "liblvae_%s.so" % name)
        _lib._lib = _lib.bind(ctypes.CDLL(path), path)
    from vae_lagging_encoder_amd.factory import build_image_vae
    from vae_lagging_encoder_amd.trainer import AggressiveImageTrainer
    dev = torch.device("cuda:0")
    vae = build_image_vae(dev, 783435)
    tr = AggressiveImageTrainer(vae, lr=1e-3, clip=5.0, seed=783435, precision=prec, use_graph=True)
    probs = torch.rand(8, 50, 1, 28, 28).to(dev)
    for i in range(6):
        tr.step(tr.binarize(probs[i % 8]), 1.0)
    torch.cuda.synchronize()
    best = []
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(40):
            tr.step(tr.binarize(probs[i % 8]), 1.0)
        torch.cuda.synchronize()
        best.append(50 * 40 / (time.perf_counter() - t0))
    print("RES %s %.1f %.1f %.1f loss %.4f" % (name, best[0], best[1], best[2], tr.read_stats()["loss_sum"]))
    sys.exit(0)
names = sys.argv[1:] or ["product"]
prec = os.environ.get("OMNI_PREC", "bf16x3")
print("# Omniglot inner step, hipGraph replay, B = 50, precision %s: img/s of three 40-step passes per process, processes alternating" % prec)
for rnd in range(3):
    for n in names:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", n, prec], capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RES ")]
        print(line[0][4:] if line else "%s FAILED %s" % (n, r.stderr[-300:]), flush=True)
