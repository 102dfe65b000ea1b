"""RCCL on the one-GPU box: a process group of ONE rank over backend "nccl" (= RCCL on ROCm), with the exchange forced on.

RCCL needs one device per communicator rank, so a box with one GPU cannot run two ranks; what it CAN run is every call the
data-parallel schedule makes -- communicator creation bound to the device (`device_id`), asynchronous bf16 / f32 all-reduces
issued from inside the encoder's backward beside the engines' streams, the bf16 reduce-scatter of the decoder buffer, the scalar
and float64 window exchanges, barrier, all_gather -- on the real backend and the real kernels.  With one rank the mean over
ranks is the rank's own gradient, so the step must land on the single-process reference fixture exactly as the plain step does.
(The two-rank halves of the same schedule run over gloo in tests/test_dist_gloo.py; the multi-GPU RCCL run itself is the
driver's, guarded by tests/test_bench_launch.py::test_rccl_run_keeps_rung_0_and_identical_replicas_on_a_multi_gpu_box.)"""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r'''
import json, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
from vae_lagging_encoder_amd import dist as lvdist, engine
from vae_lagging_encoder_amd.dist import GradSync
from vae_lagging_encoder_amd.trainer import AggressiveTextTrainer
from helpers import ENC_KEYS, build_vae, fixture_params, load
rank, local, world = lvdist.init_from_env(force=True, banner=True)
assert (rank, world) == (0, 1) and dist.get_backend() == "nccl"
dev = torch.device("cuda", local)
res = {}
for name, decoder, payload in (("text_small_wide", "norm", "f32"), ("text_small_wide", "allreduce", "f32"), ("text_mid", "norm", "bf16"),
                               ("text_mid", "allreduce", "bf16")):
    fx = load(name)
    V, ni, H, nz = (int(fx[k]) for k in ("V", "ni", "H", "nz"))
    vae = build_vae(V, ni, H, nz, dev, params=fixture_params(fx))
    gs = GradSync(mode="strict", decoder=decoder, payload=payload, force=True)
    assert gs.active and gs.world == 1
    tr = AggressiveTextTrainer(vae, lr=1.0, clip=5.0, grad_sync=gs)
    tr.BUCKET_MIN_ELEMS = 1                      # the embedding bucket goes out from inside the encoder backward
    tr._collective_after_bptt = lambda: True     # ... and the decoder exchange from the after-BPTT hook (the schedule beside persistent launches)
    x = torch.from_numpy(fx["x"]).to(dev)
    noise = tuple(torch.from_numpy(fx[k]).to(dev) for k in ("eps", "mask_in", "mask_out"))
    tr.step(x, float(fx["kl_weight"]), noise=noise)
    st = tr.read_stats()
    sd = vae.state_dict()
    res["%%s/%%s/%%s" %% (name, decoder, payload)] = {
        "norm_rel": abs(st["norm"] - float(fx["total_norm"])) / float(fx["total_norm"]),
        "loss_rel": abs(st["loss_sum"] - float(fx["loss"].sum())) / abs(float(fx["loss"].sum())),
        "w_rel": max(float((sd[k].cpu() - torch.from_numpy(fx["new/" + k])).abs().max() / np.abs(fx["new/" + k]).max()) for k in ENC_KEYS),
        "window_mean": gs.window_mean(3.0, 2)}
# the throughput configuration at a mid shape: persistent bf16 launches with the exchange forced on, 30 steps, drawn noise
from vae_lagging_encoder_amd.factory import build_text_vae, synthetic_batch
V, ni, H, nz, B, T = 2003, 512, 1024, 32, 32, 40
vae = build_text_vae(V, ni, H, nz, dev, seed=5)
gs = GradSync(mode="strict", payload="auto", force=True)
tr = AggressiveTextTrainer(vae, lr=1.0, clip=5.0, seed=11, grad_sync=gs, precision="bf16")
pool = [synthetic_batch(B, T, V, seed=i).to(dev) for i in range(4)]
tr.prepare_batches(pool)
gs.profile = True
for i in range(30):
    tr.step(pool[i %% 4], 0.1)
rung = tr.commit()
torch.cuda.synchronize()
dist.barrier()
bd = gs.breakdown()
st = tr.read_stats()
res["bf16_persistent"] = {"rung": rung, "payload": gs.payload, "loss_finite": bool(np.isfinite(st["loss_sum"])), "phases": sorted(k for k in bd if k != "steps"),
                          "recoveries": tr.recoveries}
print("RESULT " + json.dumps(res), flush=True)
dist.destroy_process_group()
'''


@pytest.mark.gpu
def test_every_collective_of_the_schedule_runs_on_rccl_with_one_rank(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("LVAE_DIST_BACKEND", None)
    child = tmp_path / "rccl_one_rank.py"
    child.write_text(_CHILD % {"root": ROOT})
    r = subprocess.run([sys.executable, str(child)], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, r.stdout[-2000:]
    res = json.loads(line[0][7:])
    for key, v in res.items():
        if key == "bf16_persistent":
            continue
        tol = 5e-3 if key.endswith("bf16") else 1e-4           # the bf16 wire rounds every gradient element to 8 bits
        assert v["norm_rel"] < tol and v["loss_rel"] < 1e-4 and v["w_rel"] < tol, (key, v)
        assert v["window_mean"] == 1.5
    p = res["bf16_persistent"]
    # rung 0 after 30 steps: no collective (RCCL's stream) took the device from a persistent launch; the bf16 configuration picked
    # the bf16 wire; the exchange phases were all seen
    assert p["rung"] == 0 and p["recoveries"] == 0 and p["payload"] == "bf16" and p["loss_finite"], p
    assert {"encoder_allreduce_issue", "encoder_allreduce_wait", "decoder_reduce_scatter_wait", "scalar_allreduce"} <= set(p["phases"]), p
