"""Host-side pieces that need no device: the dataset hyper-parameter dicts, the model factory's construction order and
key names, the synthetic batch generator (CPU only)."""
import importlib

import torch

from vae_lagging_encoder_amd import factory

TEXT_KEYS = {"enc_type", "dec_type", "nz", "ni", "enc_nh", "dec_nh", "dec_dropout_in", "dec_dropout_out", "batch_size",
             "epochs", "test_nepoch", "train_data", "val_data", "test_data"}
IMAGE_KEYS = {"img_size", "nz", "enc_layers", "dec_kernel_size", "dec_layers", "latent_feature_map", "batch_size", "epochs",
              "test_nepoch", "data_file"}


def _params(name):
    return importlib.import_module("vae_lagging_encoder_amd.config.config_" + name).params


def test_text_configs_expose_the_reference_keys():
    for name, extra in (("yahoo", set()), ("yelp", {"label"}), ("synthetic", set())):
        p = _params(name)
        assert set(p) == TEXT_KEYS | extra, name
        assert p["enc_type"] == p["dec_type"] == "lstm"
    y = _params("yahoo")
    assert (y["nz"], y["ni"], y["enc_nh"], y["dec_nh"], y["batch_size"]) == (32, 512, 1024, 1024, 32)   # BASELINE.json shape
    s = _params("synthetic")
    assert s["val_data"] == s["test_data"] and (s["nz"], s["ni"], s["enc_nh"]) == (2, 50, 50)


def test_omniglot_config():
    p = _params("omniglot")
    assert set(p) == IMAGE_KEYS
    assert p["img_size"] == [1, 28, 28] and p["batch_size"] == 50 and p["latent_feature_map"] == 4
    assert p["dec_kernel_size"] == [9] * 3 + [7] * 3 + [5] * 3 + [3] * 3 and len(p["dec_layers"]) == 12


def test_synthetic_batch_matches_the_survey_distribution():
    x = factory.synthetic_batch(7, 13, 101, seed=3)
    assert x.dtype == torch.int64 and tuple(x.shape) == (7, 13)
    assert bool((x[:, 0] == 1).all()) and bool((x[:, -1] == 2).all())
    assert int(x[:, 1:-1].min()) >= 4 and int(x.max()) < 101
    assert torch.equal(x, factory.synthetic_batch(7, 13, 101, seed=3))


def test_text_vae_factory_key_names_and_seeding():
    a = factory.build_text_vae(53, 8, 12, 3, "cpu", seed=5)
    b = factory.build_text_vae(53, 8, 12, 3, "cpu", seed=5)
    ka = [k for k, _ in a.named_parameters()]
    assert ka == ["encoder.embed.weight", "encoder.lstm.weight_ih_l0", "encoder.lstm.weight_hh_l0", "encoder.lstm.bias_ih_l0",
                  "encoder.lstm.bias_hh_l0", "encoder.linear.weight", "decoder.embed.weight", "decoder.trans_linear.weight",
                  "decoder.lstm.weight_ih_l0", "decoder.lstm.weight_hh_l0", "decoder.lstm.bias_ih_l0",
                  "decoder.lstm.bias_hh_l0", "decoder.pred_linear.weight"]
    for (_, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(pa, pb)
    assert a.decoder.lstm.weight_ih_l0.shape == (48, 8 + 3)          # decoder input = embedding ++ z
    v = factory.SizedVocab(53)
    assert len(v) == 53 and v["<pad>"] == 0 and v["<s>"] == 1 and v["</s>"] == 2


def test_image_inner_loop_control_flow(emu_backend):
    """image.py:295-327 bookkeeping without any arithmetic: the batch pick is np.random.choice(N, B, replace=False) drawn
    AFTER each step, the window is 10 iterations, the first comparison is against 1e4, the loop stops on the first window
    whose mean loss per example went up, and at most 99 steps are taken."""
    import numpy as np
    import torch
    from vae_lagging_encoder_amd.factory import build_image_vae
    from vae_lagging_encoder_amd.trainer import AggressiveImageTrainer
    vae = build_image_vae("cpu", 3)
    tr = AggressiveImageTrainer(vae)
    N, B = 64, 8
    x_train = torch.rand(N, 1, 28, 28)
    for losses in ([5.0] * 10 + [4.0] * 10 + [4.5] * 10 + [1.0] * 100, [1.0] * 200):
        state = {"i": 0, "acc": 0.0, "picked": []}

        def fake_step(x, klw, eps=None, update="encoder"):
            state["acc"] += losses[state["i"]] * x.shape[0]
            state["i"] += 1
        tr.step = fake_step
        tr.read_stats = lambda: dict(loss_sum=state["acc"])
        tr.reset_stats = lambda: state.__setitem__("acc", 0.0)

        def fake_bin(p):
            state["picked"].append(p.clone())
            return p
        steps = tr.inner_loop(x_train, x_train[:B], 1.0, batch_size=B, np_rng=np.random.RandomState(4), binarize_fn=fake_bin)
        # replay of the host stream
        rs = np.random.RandomState(4)
        ids = [rs.choice(N, B, replace=False) for _ in range(steps)]
        assert len(state["picked"]) == steps
        for got, id_ in zip(state["picked"], ids):
            assert torch.equal(got, x_train[torch.from_numpy(id_)])
        assert steps == (30 if losses[0] == 5.0 else 99), steps


def test_outer_loop_policy(emu_backend):
    """text.py:447-489 as host logic: aggressive training stops the first time the validation MI drops; the learning rate
    halves after 2 epochs without improvement once epoch >= 15 (the best weights are reloaded), training stops after 5
    decays; the KL weight anneals linearly to 1 over warm_up epochs."""
    import argparse
    import torch
    from vae_lagging_encoder_amd import evaluation as E
    from vae_lagging_encoder_amd.factory import build_text_vae, synthetic_batch
    from vae_lagging_encoder_amd.training import TextTrainingLoop
    vae = build_text_vae(53, 8, 16, 4, "cpu", seed=1)
    batches = [synthetic_batch(4, 6, 53, seed=i) for i in range(5)]
    args = argparse.Namespace(kl_start=0.1, warm_up=2, batch_size=4, epochs=3, aggressive=1, nsamples=1, test_nepoch=5,
                              iw_nsamples=20, momentum=0)
    logs = []
    loop = TextTrainingLoop(vae, batches, batches[:2], batches[:1], args, log=logs.append)
    assert abs(loop.anneal_rate - 0.9 / (2 * 5)) < 1e-12
    # MI sequence 0.3, 0.5, 0.4: the drop at the third check ends aggressive training, and pre_mi always tracks the last value
    mis = iter([0.3, 0.5, 0.4])
    saved = E.calc_mi
    E.calc_mi = lambda model, data: next(mis)
    try:
        for want in (True, True, False):
            loop.check_aggressive()
            assert loop.aggressive is want
        assert loop.pre_mi == 0.4 and "STOP BURNING" in logs
    finally:
        E.calc_mi = saved
    # learning-rate policy: nothing before epoch 15; then a decay every second non-improving epoch; stop at the fifth
    w0 = vae.state_dict()["encoder.linear.weight"].clone()
    assert loop.end_of_epoch(0, 10.0, 10.0, 1.0, 50.0) is False and loop.best["loss"] == 10.0
    with torch.no_grad():
        vae.encoder.linear.weight.add_(1.0)                     # weights drift away from the best checkpoint
    for ep in range(1, 15):
        assert loop.end_of_epoch(ep, 11.0, 11.0, 1.0, 60.0) is False
    assert loop.opt["lr"] == 1.0 and loop.decay_cnt == 0
    stop_epoch = None
    for ep in range(15, 40):            # decays at epochs 15 (14 stale epochs already counted), 17, 19, 21, 23 -> stop
        if loop.end_of_epoch(ep, 11.0 + 0.1 * ep, 11.0, 1.0, 60.0):
            stop_epoch = ep
            break
    assert stop_epoch == 23 and loop.decay_cnt == 5 and abs(loop.opt["lr"] - 0.5 ** 5) < 1e-12
    assert float(loop.trainer.scal[1]) == loop.opt["lr"]          # the fused step's device-resident learning rate follows
    assert torch.equal(vae.state_dict()["encoder.linear.weight"], w0)   # a decay reloads the best weights


def test_outer_loop_runs_end_to_end(emu_backend):
    """Two epochs of the whole driver on the emulator: inner loops, joint steps, MI-based stop check, validation, history."""
    import argparse
    import numpy as np
    from vae_lagging_encoder_amd.factory import build_text_vae, synthetic_batch
    from vae_lagging_encoder_amd.training import TextTrainingLoop
    vae = build_text_vae(53, 8, 16, 4, "cpu", seed=2, model_scale=0.1)
    train = [synthetic_batch(4, T, 53, seed=10 + i) for i, T in enumerate((5, 6, 4))]
    args = argparse.Namespace(kl_start=0.1, warm_up=1, batch_size=4, epochs=2, aggressive=1, nsamples=1, test_nepoch=1,
                              iw_nsamples=20, momentum=0)
    loop = TextTrainingLoop(vae, train, train[:2], train[:1], args, log=lambda *_: None, np_rng=np.random.RandomState(3))
    loop.trainer.inner_loop.__func__            # (bound method exists)
    orig = loop.trainer.inner_loop
    calls = []
    loop.trainer.inner_loop = lambda *a, **k: calls.append(1) or orig(*a, max_iter=4, window=2, **{kk: v for kk, v in k.items()})
    out = loop.run()
    assert out["epochs"] == 2 and len(calls) >= 3 and np.isfinite(out["best_loss"])
    assert loop.history[-1]["kl_weight"] == 1.0                         # warm_up = 1 epoch ... reached by the end of epoch 2


def test_workspace_cache_drops_least_recently_used_shapes():
    """engine._WS: shape-keyed workspaces are bounded by a byte budget (a corpus cycles through hundreds of (B, T) shapes);
    the most recent shapes always survive, and nothing is dropped when hipGraphs hold pointers (evictable = False)."""
    import torch
    from vae_lagging_encoder_amd import engine as eng

    def build(n):
        ns = eng._NS()
        ns.a = torch.zeros(n, dtype=torch.float32)
        ns.sub = {"b": torch.zeros(n, dtype=torch.int16)}
        return ns

    c = eng._WS("cpu", budget_bytes=10 * 6000)
    first = c.get(("ws", 1, 1), lambda: build(1000))
    assert eng._tensor_bytes(first) == 6000
    for i in range(2, 40):
        c.get(("ws", 1, i), lambda: build(1000))
        assert c.get(("ws", 1, i), None) is not None          # a hit never rebuilds
    assert c.total <= 10 * 6000 and len(c.cache) == 10
    assert ("ws", 1, 39) in c.cache and ("ws", 1, 1) not in c.cache
    assert first.a.numel() == 1000                            # a dropped workspace stays valid for whoever still holds it
    c.get(("ws", 1, 30), None)                                # touch -> most recent
    for i in range(40, 49):
        c.get(("ws", 1, i), lambda: build(1000))
    assert ("ws", 1, 30) in c.cache and ("ws", 1, 31) not in c.cache
    # the last 8 entries survive even a budget smaller than one step's workspaces
    tiny = eng._WS("cpu", budget_bytes=1)
    for i in range(20):
        tiny.get(i, lambda: build(10))
    assert len(tiny.cache) == 8
    # hipGraph mode: nothing is dropped
    keep = eng._WS("cpu", budget_bytes=1)
    keep.evictable = False
    for i in range(20):
        keep.get(i, lambda: build(10))
    assert len(keep.cache) == 20


def test_image_outer_loop_policy():
    """image.py:381-425 as host logic (no kernels: a stub trainer): aggressive training ends at the FIFTH end-of-epoch check whose
    validation MI is below the best MI so far (not at the first drop, and the strikes need not be consecutive); the learning
    rate halves after `decay_epoch` epochs that did not set a new best validation loss (no epoch gate), the best weights are
    reloaded and both Adam optimizers start from scratch; five decays end training.  The transitions below were stepped through
    by hand against image.py; the full loop is replayed against a recorded reference run in tests/test_policy_replay.py."""
    import argparse
    import torch
    from vae_lagging_encoder_amd import evaluation as E
    from vae_lagging_encoder_amd.training import ImageTrainingLoop

    class StubTrainer(object):
        def __init__(self):
            self.resets = []
            self.dec = argparse.Namespace(wgen=0)

        def reset_optimizer(self, lr):
            self.resets.append(lr)
    vae = torch.nn.Linear(2, 2)
    args = argparse.Namespace(kl_start=0.1, warm_up=2, batch_size=10, epochs=1, aggressive=1, nsamples=1, test_nepoch=5)
    x = torch.rand(50, 1, 28, 28)
    logs = []
    loop = ImageTrainingLoop(vae, x, x[:20], None, args, trainer=StubTrainer(), log=logs.append, decay_epoch=2)
    assert abs(loop.anneal_rate - 0.9 / (2 * 5)) < 1e-12 and loop.opt["lr"] == 0.001
    # MI checks: best_mi only moves up; strikes at 0.4, 0.3, 0.45 (< 0.5), 0.2, 0.1 -> the fifth strike stops
    mis = iter([0.2, 0.5, 0.4, 0.3, 0.45, 0.6, 0.2, 0.1])
    saved = E.image_calc_mi
    E.image_calc_mi = lambda model, loader: next(mis)
    try:
        flags = []
        for _ in range(8):
            loop.check_aggressive()
            flags.append(loop.aggressive)
    finally:
        E.image_calc_mi = saved
    assert flags == [True] * 7 + [False] and loop.best_mi == 0.6 and loop.mi_not_improved == 5 and "STOP BURNING" in logs
    # learning-rate policy (decay_epoch = 2 here): new best at epochs 0 and 1; epochs 2, 3 are worse -> decay at 3; 4 improves
    w0 = {k: v.clone() for k, v in vae.state_dict().items()}
    assert loop.end_of_epoch(0, 10.0, 10.0, 1.0) is False
    assert loop.end_of_epoch(1, 9.0, 9.0, 1.0) is False and loop.best["loss"] == 9.0
    best_w = {k: v.clone() for k, v in vae.state_dict().items()}
    with torch.no_grad():
        vae.weight.add_(1.0)
    assert loop.end_of_epoch(2, 9.5, 9.5, 1.0) is False and loop.opt["not_improved"] == 1 and loop.trainer.resets == []
    assert loop.end_of_epoch(3, 9.0, 9.0, 1.0) is False          # equal to the best is neither `<` nor `>`: counters reset (image.py:423-425)
    assert loop.opt["not_improved"] == 0
    assert loop.end_of_epoch(4, 9.2, 9.2, 1.0) is False and loop.end_of_epoch(5, 9.3, 9.3, 1.0) is False
    assert loop.trainer.resets == [0.0005] and loop.decay_cnt == 1 and loop.opt["lr"] == 0.0005
    assert all(torch.equal(vae.state_dict()[k], best_w[k]) for k in best_w)           # the decay reloaded the best weights
    stop = [loop.end_of_epoch(6 + i, 9.4 + 0.01 * i, 9.4, 1.0) for i in range(8)]
    assert stop == [False, False, False, False, False, False, False, True] and loop.decay_cnt == 5
    assert loop.trainer.resets == [0.001 * 0.5 ** k for k in range(1, 6)]
    del w0


def test_sampler_order_equals_a_shuffled_dataloader_pass():
    """data.sampler_order must consume torch's global generator exactly as `for datum in DataLoader(..., shuffle=True)` does
    (image.py:219-221,281): the iterator's _base_seed draw first, then RandomSampler's seed -- same orders pass after pass, and
    the same generator state afterwards (so that whatever the training loop draws next matches the reference's stream)."""
    import torch
    from torch.utils.data import DataLoader, TensorDataset
    from vae_lagging_encoder_amd.data import ShuffledLoader, sampler_order
    n = 37
    torch.manual_seed(5)
    dl = DataLoader(TensorDataset(torch.arange(n)), batch_size=5, shuffle=True)
    ref = [[int(v) for b in dl for v in b[0]] for _ in range(3)]
    state = torch.get_rng_state().clone()
    torch.manual_seed(5)
    assert [sampler_order(n) for _ in range(3)] == ref
    assert torch.equal(torch.get_rng_state(), state)
    torch.manual_seed(5)
    sl = ShuffledLoader(torch.arange(n).float().view(n, 1), 5)
    assert [[int(v) for b, _ in sl for v in b.view(-1)] for _ in range(3)] == ref
