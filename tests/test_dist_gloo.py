"""Data-parallel inner step over gloo, world_size 2, on CPU (emulator backend for the kernels): two ranks each take
half of the fixture batch (rows + matching slices of eps / dropout masks), mean-all-reduce the flat gradient buffers
(strict mode: encoder AND decoder, because the clip norm spans both -- SURVEY.md G1) and must land on the
single-process reference result (strong-scaling parity)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, name, mode, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
        import ctypes
        import torch.distributed as dist
        from build_emu import build_emu
        from vae_lagging_encoder_amd import _lib, engine
        from vae_lagging_encoder_amd.dist import GradSync
        from vae_lagging_encoder_amd.trainer import AggressiveTextTrainer
        from helpers import ENC_KEYS, build_vae, fixture_params, load
        torch.set_num_threads(1)
        engine._install_test_backend(_lib.bind(ctypes.CDLL(build_emu())))
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
        fx = load(name)
        V, ni, H, nz, B = int(fx["V"]), int(fx["ni"]), int(fx["H"]), int(fx["nz"]), int(fx["B"])
        per = B // world
        sl = slice(rank * per, (rank + 1) * per)
        vae = build_vae(V, ni, H, nz, "cpu", params=fixture_params(fx))
        tr = AggressiveTextTrainer(vae, lr=1.0, clip=5.0, grad_sync=GradSync(mode=mode))
        x = torch.from_numpy(fx["x"])[sl].contiguous()
        noise = (torch.from_numpy(fx["eps"])[sl].contiguous(), torch.from_numpy(fx["mask_in"])[sl].contiguous(),
                 torch.from_numpy(fx["mask_out"])[sl].contiguous())
        tr.step(x, float(fx["kl_weight"]), noise=noise)
        st = tr.read_stats()
        sd = vae.state_dict()
        errs = {k: float((sd[k] - torch.from_numpy(fx["new/" + k])).abs().max() / np.abs(fx["new/" + k]).max()) for k in ENC_KEYS}
        q.put((rank, st["norm"], st["loss_sum"], errs, None))
        dist.destroy_process_group()
    except Exception as e:  # noqa
        import traceback
        q.put((rank, None, None, None, traceback.format_exc()))


@pytest.mark.parametrize("name", ["text_small_wide"])
def test_two_rank_strict_dp_equals_single_process_reference(name):
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    from build_emu import build_emu
    build_emu()   # build once in the parent
    from helpers import load
    fx = load(name)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, "strict", q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, norm, loss_sum, errs, tb in res:
        assert tb is None, tb
        # every rank sees the GLOBAL clipped norm (fixture has norm > 5: the clip is active)
        assert abs(norm - float(fx["total_norm"])) / float(fx["total_norm"]) < 1e-4
        for k, e in errs.items():
            assert e < 1e-4, (rank, k, e)
    # the ranks' local loss sums add up to the reference's batch loss sum
    assert abs(sum(r[2] for r in res) - float(fx["loss"].sum())) / abs(float(fx["loss"].sum())) < 1e-4
