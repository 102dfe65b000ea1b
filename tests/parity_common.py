"""Parity checks shared by the emulator (CPU, `not gpu`) and MI355X (`gpu`) test files: the drop-in modules are
driven exactly as text.py:373-387 drives the reference's, and compared with the golden fixtures."""
import numpy as np
import torch

from helpers import ALL_KEYS, DEC_KEYS, ENC_KEYS, build_vae, fixture_params, load, rel_err

# north_star tolerance: ELBO / KL / rec within 1e-4 relative in fp32.  KL at the reference init is ~1e-5 and
# cancellation-dominated, hence the absolute floor 1e-6*(1+|rec|) there.
RTOL = 1e-4
GRAD_RTOL = 2e-4


def run_reference_style_step(name, device, clip=5.0, lr=1.0, optim="torch"):
    """text.py:373-387 literally, on the drop-in modules.  optim = "lvae": the same four lines with vae_lagging_encoder_amd.optim's
    clip_grad_norm_ / SGD (streaming launches over the flat buffers) in place of torch's."""
    from vae_lagging_encoder_amd import optim as lvo
    fx = load(name)
    V, ni, H, nz = int(fx["V"]), int(fx["ni"]), int(fx["H"]), int(fx["nz"])
    vae = build_vae(V, ni, H, nz, device, params=fixture_params(fx))
    x = torch.from_numpy(fx["x"]).to(device)
    noise = (torch.from_numpy(fx["eps"]).to(device), torch.from_numpy(fx["mask_in"]).to(device),
             torch.from_numpy(fx["mask_out"]).to(device))
    SGD = lvo.SGD if optim == "lvae" else torch.optim.SGD
    clipfn = lvo.clip_grad_norm_ if optim == "lvae" else torch.nn.utils.clip_grad_norm_
    enc_opt = SGD(vae.encoder.parameters(), lr=lr, momentum=0)
    dec_opt = SGD(vae.decoder.parameters(), lr=lr, momentum=0)
    enc_opt.zero_grad()
    dec_opt.zero_grad()
    loss, rec, kl = vae.loss(x, float(fx["kl_weight"]), nsamples=1, noise=noise)
    loss.mean(dim=-1).backward()
    # after zero_grad() the backward leaves every .grad as a VIEW of its module's flat gradient buffer (no clones)
    for m in (vae.encoder, vae.decoder):
        assert m._hip.flat.grads_are_views()
    grads = {k: p.grad.detach().clone() for k, p in vae.named_parameters()}
    total = float(clipfn(vae.parameters(), clip))
    enc_opt.step()
    return fx, vae, loss.detach(), rec.detach(), kl.detach(), grads, total


def check_grad_accumulation_semantics(device, V=97, ni=12, H=20, nz=4, B=6, T=7):
    """Two backward() calls without a zero_grad() in between ADD (autograd's semantics), although the first one left the .grad
    tensors aliasing the flat buffer the second one's engine writes into; zero_grad(set_to_none=False) keeps working too."""
    from oracle import text_vae_oracle as O
    P = O.random_params(V, ni, H, nz, seed=3, scale=0.3, emb_scale=0.5, head_scale=0.5)
    vae = build_vae(V, ni, H, nz, device, params=P)
    xs = [O.synthetic_batch(B, T, V, seed=10 + i).to(device) for i in range(2)]
    ns = []
    for i in range(2):
        eps, mi, mo = O.draw_noise(B, T, ni, H, nz, seed=20 + i)
        ns.append((eps.to(device), mi.to(torch.uint8).to(device), mo.to(torch.uint8).to(device)))
    single = []
    for i in range(2):
        vae.zero_grad()
        vae.loss(xs[i], 0.5, noise=ns[i])[0].mean().backward()
        single.append({k: p.grad.detach().clone() for k, p in vae.named_parameters()})
    vae.zero_grad()
    vae.loss(xs[0], 0.5, noise=ns[0])[0].mean().backward()
    assert vae.encoder._hip.flat.grads_are_views()
    vae.loss(xs[1], 0.5, noise=ns[1])[0].mean().backward()          # no zero_grad: must accumulate
    for k, p in vae.named_parameters():
        want = single[0][k] + single[1][k]
        assert float((p.grad - want).abs().max()) <= 1e-6 * float(want.abs().max()) + 1e-12, k
    vae.zero_grad(set_to_none=False)                                # zeros in place, .grad stays allocated
    vae.loss(xs[1], 0.5, noise=ns[1])[0].mean().backward()
    for k, p in vae.named_parameters():
        assert float((p.grad - single[1][k]).abs().max()) <= 1e-6 * float(single[1][k].abs().max()) + 1e-12, k


def check_autograd_grad_and_frozen_modules(device, V=97, ni=12, H=20, nz=4, B=6, T=7):
    """the zero-copy gradient route is taken only inside a plain `loss.backward()`, per parameter.
    (a) torch.autograd.grad(loss, params) returns real tensors (equal to what backward() leaves) and touches no .grad;
    (b) a frozen module (requires_grad False) gets no .grad, the other module still gets its views;
    (c) backward(inputs=[encoder params]) leaves the decoder's .grad alone;
    (d) VAE.use_flat_grads(False): .grad tensors own their storage (a kept gradient survives the next backward)."""
    from oracle import text_vae_oracle as O
    P = O.random_params(V, ni, H, nz, seed=3, scale=0.3, emb_scale=0.5, head_scale=0.5)
    vae = build_vae(V, ni, H, nz, device, params=P)
    x = O.synthetic_batch(B, T, V, seed=10).to(device)
    eps, mi, mo = O.draw_noise(B, T, ni, H, nz, seed=20)
    noise = (eps.to(device), mi.to(torch.uint8).to(device), mo.to(torch.uint8).to(device))
    params = list(vae.parameters())
    vae.zero_grad()
    vae.loss(x, 0.5, noise=noise)[0].mean().backward()
    assert vae.encoder._hip.flat.grads_are_views() and vae.decoder._hip.flat.grads_are_views()
    want = [p.grad.detach().clone() for p in params]
    # (a)
    vae.zero_grad()
    got = torch.autograd.grad(vae.loss(x, 0.5, noise=noise)[0].mean(), params, allow_unused=True)
    assert all(p.grad is None for p in params), "autograd.grad must not populate .grad"
    for g, w, (k, _) in zip(got, want, vae.named_parameters()):
        assert g is not None and float((g - w).abs().max()) <= 1e-6 * float(w.abs().max()) + 1e-12, k
    flat_ptrs = {id(e): (e.flat.grad.data_ptr(), e.flat.grad.data_ptr() + 4 * e.flat.numel) for e in (vae.encoder._hip, vae.decoder._hip)}
    for g in got:       # fresh storage, not views of the flat buffers
        assert not any(lo <= g.data_ptr() < hi for lo, hi in flat_ptrs.values())
    # (b)
    for p in vae.decoder.parameters():
        p.requires_grad_(False)
    vae.zero_grad()
    vae.loss(x, 0.5, noise=noise)[0].mean().backward()
    assert all(p.grad is None for p in vae.decoder.parameters()), "a frozen module must not receive gradients"
    assert vae.encoder._hip.flat.grads_are_views()
    for p, w in zip(params, want):
        if p.grad is not None:
            assert float((p.grad - w).abs().max()) <= 1e-6 * float(w.abs().max()) + 1e-12
    for p in vae.decoder.parameters():
        p.requires_grad_(True)
    # (c)
    vae.zero_grad()
    vae.loss(x, 0.5, noise=noise)[0].mean().backward(inputs=list(vae.encoder.parameters()))
    assert all(p.grad is None for p in vae.decoder.parameters()) and all(p.grad is not None for p in vae.encoder.parameters())
    # (d)
    assert vae.use_flat_grads(False) is vae
    vae.zero_grad()
    vae.loss(x, 0.5, noise=noise)[0].mean().backward()
    assert not vae.encoder._hip.flat.grads_are_views()
    kept = params[0].grad
    snapshot = kept.clone()
    vae.zero_grad()
    vae.loss(O.synthetic_batch(B, T, V, seed=11).to(device), 0.5, noise=noise)[0].mean().backward()
    assert torch.equal(kept, snapshot), "with flat grads off a kept gradient tensor must survive the next backward"
    vae.use_flat_grads(True)
    vae.zero_grad()
    vae.loss(x, 0.5, noise=noise)[0].mean().backward()
    assert vae.encoder._hip.flat.grads_are_views()


def check_step_against_fixture(name, device, optim="torch"):
    fx, vae, loss, rec, kl, grads, total = run_reference_style_step(name, device, optim=optim)
    rec_scale = float(np.abs(fx["rec"]).max())
    assert rel_err(loss, fx["loss"]) < RTOL, ("loss", rel_err(loss, fx["loss"]))
    assert rel_err(rec, fx["rec"]) < RTOL
    kl_err = float(np.abs(kl.cpu().numpy() - fx["kl"]).max())
    assert kl_err < RTOL * float(np.abs(fx["kl"]).max()) + 1e-6 * (1 + rec_scale), ("kl", kl_err)
    for k in ALL_KEYS:
        g = fx["grad/" + k]
        if np.abs(g).max() > 0:
            e = rel_err(grads[k], g)
            assert e < GRAD_RTOL, (k, e)
        else:
            assert float(grads[k].abs().max()) == 0.0, k
    assert abs(total - float(fx["total_norm"])) / float(fx["total_norm"]) < RTOL
    sd = vae.state_dict()
    for k in ENC_KEYS:
        assert rel_err(sd[k], fx["new/" + k]) < RTOL, k
    for k in DEC_KEYS:   # encoder-only step leaves the decoder untouched
        assert torch.equal(sd[k].cpu(), torch.from_numpy(fx["param/" + k])), k
    return fx


def check_trajectory_against_fixture(device, use_graph=False):
    """K fused inner steps on a pool of batches + the joint decoder step (text.py:366-424 without the
    data-dependent exit) through AggressiveTextTrainer, against the reference trajectory."""
    from vae_lagging_encoder_amd.trainer import AggressiveTextTrainer
    fx = load("traj_small")
    V, ni, H, nz = int(fx["V"]), int(fx["ni"]), int(fx["H"]), int(fx["nz"])
    vae = build_vae(V, ni, H, nz, device, params=fixture_params(fx))
    tr = AggressiveTextTrainer(vae, lr=1.0, clip=5.0, use_graph=use_graph)
    K = int(fx["K"])
    klw = float(fx["kl_weight"])
    pool = [torch.from_numpy(p).to(device) for p in fx["pool"]]
    order = list(fx["order"])
    for it in range(K + 1):
        joint = it == K
        bi = 0 if joint else int(order[it])
        noise = tuple(torch.from_numpy(fx[k][it]).to(device) for k in ("eps", "mask_in", "mask_out"))
        tr.reset_stats()
        tr.step(pool[bi], klw, noise=noise, update="decoder" if joint else "encoder")
        st = tr.read_stats()
        assert abs(st["loss_sum"] - float(fx["loss"][it].sum())) / abs(float(fx["loss"][it].sum())) < RTOL
        assert abs(st["norm"] - float(fx["total_norm"][it])) / float(fx["total_norm"][it]) < RTOL
    sd = vae.state_dict()
    for k in ALL_KEYS:
        assert rel_err(sd[k], fx["final/" + k]) < RTOL, k


# ---------------------------------------------------------------------------------------------------------------------
# Omniglot path (ResNetEncoderV2 + PixelCNNDecoderV2), image.py:300-314
def build_image_vae(device, seed):
    from vae_lagging_encoder_amd.factory import build_image_vae as build
    return build(device, seed)


def _check_image_outputs(fx, vae, loss, rec, kl, grads, total):
    assert rel_err(loss, fx["loss"]) < RTOL, ("loss", rel_err(loss, fx["loss"]))
    assert rel_err(rec, fx["rec"]) < RTOL
    assert rel_err(kl, fx["kl"]) < RTOL
    assert abs(total - float(fx["total_norm64"])) / float(fx["total_norm64"]) < 5e-4, (total, float(fx["total_norm64"]))
    worst = 0.0
    for k, g in grads.items():
        ref_n = float(fx["gradnorm/" + k])
        got_n = float(g.double().norm())
        assert abs(got_n - ref_n) <= 2e-3 * ref_n + 1e-7, (k, got_n, ref_n)
        idx = torch.from_numpy(fx["sample_idx/" + k]).to(g.device)
        e = float((g.reshape(-1)[idx].cpu() - torch.from_numpy(fx["sample_grad/" + k])).abs().max())
        worst = max(worst, e / (ref_n / max(1.0, g.numel() ** 0.5) + 1e-12))
    assert worst < 0.05, worst        # sampled entries within 5% of the tensor's RMS gradient
    sd = vae.state_dict()
    for k in sd:
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert rel_err(sd[k][:8], fx["stat/" + k]) < 1e-4, k


def check_image_step_dropin(name, device):
    """The reference's own call sequence (image.py:300-314) on the drop-in modules, against the fixture."""
    fx = load(name)
    vae = build_image_vae(device, int(fx["model_seed"]))
    sd0 = vae.state_dict()
    for k, p in vae.named_parameters():   # regenerated weights ARE the reference's
        idx = torch.from_numpy(fx["sample_idx/" + k])
        assert torch.equal(sd0[k].reshape(-1)[idx].cpu(), torch.from_numpy(fx["sample_p0/" + k])), k
    x = torch.from_numpy(fx["x"]).float().to(device)
    eps = torch.from_numpy(fx["eps"]).to(device)
    enc_opt = torch.optim.Adam(vae.encoder.parameters(), lr=0.001)
    dec_opt = torch.optim.Adam(vae.decoder.parameters(), lr=0.001)
    enc_opt.zero_grad()
    dec_opt.zero_grad()
    loss, rec, kl = vae.loss(x, float(fx["kl_weight"]), nsamples=1, noise=(eps, None, None))
    loss.mean(dim=-1).backward()
    grads = {k: p.grad.detach().clone() for k, p in vae.named_parameters()}
    total = float(torch.nn.utils.clip_grad_norm_(vae.parameters(), 5.0))
    enc_opt.step()
    _check_image_outputs(fx, vae, loss.detach(), rec.detach(), kl.detach(), grads, total)
    sd = vae.state_dict()
    for k, p in vae.named_parameters():
        if k.startswith("encoder."):
            idx = torch.from_numpy(fx["sample_idx/" + k])
            got = sd[k].reshape(-1)[idx].cpu()
            # Adam's first step moves every weight by ~lr*sign(g): compare the UPDATE, not the weight
            upd_ref = torch.from_numpy(fx["sample_new/" + k]) - torch.from_numpy(fx["sample_p0/" + k])
            upd_got = got - torch.from_numpy(fx["sample_p0/" + k])
            assert float((upd_got - upd_ref).abs().max()) < 2e-5, k
    return fx


def check_image_step_fused(name, device, use_graph=False, precision="f32", rtol=RTOL, norm_tol=5e-4, upd_tol=2e-5):
    """precision = "bf16x3" (the direct convolutions on split-bf16 operands) is held to the SAME bounds as the exact-f32 path;
    "bf16" (plain bf16 operands) to the ones its caller states.  Returns the measured errors."""
    from vae_lagging_encoder_amd.trainer import AggressiveImageTrainer
    fx = load(name)
    vae = build_image_vae(device, int(fx["model_seed"]))
    tr = AggressiveImageTrainer(vae, lr=1e-3, clip=5.0, use_graph=use_graph, precision=precision)
    x = torch.from_numpy(fx["x"]).float().to(device)
    tr.step(x, float(fx["kl_weight"]), eps=torch.from_numpy(fx["eps"]).to(device))
    st = tr.read_stats()
    errs = dict(loss=abs(st["loss_sum"] - float(fx["loss"].sum())) / abs(float(fx["loss"].sum())),
                kl=abs(st["kl_sum"] - float(fx["kl"].sum())) / abs(float(fx["kl"].sum())),
                norm=abs(st["norm"] - float(fx["total_norm64"])) / float(fx["total_norm64"]), upd=0.0)
    assert errs["loss"] < rtol and errs["kl"] < rtol and errs["norm"] < norm_tol, errs
    sd = vae.state_dict()
    for k, p in vae.named_parameters():
        if k.startswith("encoder."):
            idx = torch.from_numpy(fx["sample_idx/" + k])
            upd_ref = torch.from_numpy(fx["sample_new/" + k]) - torch.from_numpy(fx["sample_p0/" + k])
            upd_got = sd[k].reshape(-1)[idx].cpu() - torch.from_numpy(fx["sample_p0/" + k])
            errs["upd"] = max(errs["upd"], float((upd_got - upd_ref).abs().max()))
            assert float((upd_got - upd_ref).abs().max()) < upd_tol, (k, float((upd_got - upd_ref).abs().max()))
        else:   # decoder untouched except MaskedConv2d's in-place weight masking
            idx = torch.from_numpy(fx["sample_idx/" + k])
            got = sd[k].reshape(-1)[idx].cpu()
            ref = torch.from_numpy(fx["sample_p0/" + k])
            assert bool(((got == ref) | (got == 0)).all()), k
    return errs


# ---------------------------------------------------------------------------------------------------------------------
# the whole aggressive inner loop with its data-dependent exit (text.py:366-400), oracle-driven replica vs trainer
def check_inner_loop_exit_logic(device, window=3, max_iter=11):
    """Replays text.py:366-400 literally (sub_iter counter, burn_* bookkeeping, np.random.random_integers batch pick,
    windowed mean-loss-per-word comparison, break) with the CPU oracle doing the arithmetic, and checks that
    AggressiveTextTrainer.inner_loop takes the same number of steps on the same batches and lands on the same encoder."""
    import numpy as np
    from oracle import text_vae_oracle as O
    from vae_lagging_encoder_amd.trainer import AggressiveTextTrainer
    V, ni, H, nz, B = 97, 12, 20, 4, 6
    P = O.random_params(V, ni, H, nz, seed=21, scale=0.3, emb_scale=0.5, head_scale=0.5)
    Ts = [5, 7, 6, 9]
    batches = [O.synthetic_batch(B, T, V, seed=40 + i) for i, T in enumerate(Ts)]
    klw = 0.8

    def noise_for(step, x):
        return O.draw_noise(x.shape[0], x.shape[1], ni, H, nz, seed=900 + step)

    # ---- reference control flow, oracle arithmetic -------------------------------------------------------------------
    rs = np.random.RandomState(5)
    Pr = {k: v.clone() for k, v in P.items()}
    sub_iter, x = 1, batches[0]
    burn_num_words, burn_pre_loss, burn_cur_loss = 0, 1e4, 0.0
    ref_steps = 0
    while sub_iter < max_iter:
        bsz, slen = x.shape
        burn_num_words += (slen - 1) * bsz
        eps, mi, mo = noise_for(ref_steps, x)
        r = O.inner_step(Pr, x, klw, eps, mi, mo)
        burn_cur_loss += float(r["loss"].sum())
        Pr.update(r["new_params"])
        ref_steps += 1
        x = batches[int(rs.randint(0, len(batches)))]
        if sub_iter % window == 0:
            burn_cur_loss = burn_cur_loss / burn_num_words
            if burn_pre_loss - burn_cur_loss < 0:
                break
            burn_pre_loss = burn_cur_loss
            burn_cur_loss = burn_num_words = 0
        sub_iter += 1

    # ---- the fused driver -------------------------------------------------------------------------------------------------
    vae = build_vae(V, ni, H, nz, device, params=P)
    tr = AggressiveTextTrainer(vae, lr=1.0, clip=5.0)
    counter = {"n": 0}

    def noise_fn(xb):
        eps, mi, mo = noise_for(counter["n"], xb)
        counter["n"] += 1
        return eps.to(device), mi.to(torch.uint8).to(device), mo.to(torch.uint8).to(device)
    steps = tr.inner_loop([b.to(device) for b in batches], batches[0].to(device), klw, np_rng=np.random.RandomState(5),
                          max_iter=max_iter, window=window, noise_fn=noise_fn)
    assert steps == ref_steps, (steps, ref_steps)
    sd = vae.state_dict()
    for k in ENC_KEYS:
        assert rel_err(sd[k], Pr[k]) < 5e-4, (k, rel_err(sd[k], Pr[k]))
    for k in DEC_KEYS:
        assert torch.equal(sd[k].cpu(), P[k]), k
    return steps


# ---------------------------------------------------------------------------------------------------------------------
# bf16 throughput path: pre-rounded bf16 operand images (lv_gemm_b16) vs rounding on the fly (lv_gemm_bf16)
def check_bf16_native_operands_equal_on_the_fly(device, V=333, ni=24, H=64, nz=8, B=7, T=11):
    """Both routes round the same f32 values to bf16 (RNE) and run the same MFMA chain, so one fused step must agree
    to f32-summation-order precision (split-K partitions differ) on every statistic and every gradient."""
    from oracle import text_vae_oracle as O
    from vae_lagging_encoder_amd.trainer import AggressiveTextTrainer
    P = O.random_params(V, ni, H, nz, seed=31, scale=0.2, emb_scale=0.5, head_scale=0.5)
    x = O.synthetic_batch(B, T, V, seed=32)
    eps, mi, mo = O.draw_noise(B, T, ni, H, nz, seed=33)
    noise = (eps.to(device), mi.to(torch.uint8).to(device), mo.to(torch.uint8).to(device))
    res = []
    for native in (True, False):
        vae = build_vae(V, ni, H, nz, device, params=P)
        tr = AggressiveTextTrainer(vae, lr=1.0, clip=5.0, precision="bf16")
        tr.dec.native16 = tr.enc.native16 = native
        tr.enc.fwd_operands = "bf16"            # (the image path's default forward operands are binary16; this compares the two bf16 routes)
        tr.step(x.to(device), 0.6, noise=noise)
        st = tr.read_stats()
        grads = {k: p.grad.detach().cpu().clone() for k, p in vae.named_parameters()}
        res.append((st, grads, (tr.dec._b16(B, T - 1) is not None and tr.dec._lstm_images(B, T - 1) is not None
                                and tr.enc._b16(B, T) is not None)))
    (s1, g1, used1), (s0, g0, used0) = res
    assert used1 and not used0
    # forward: the same rounded operands through the same MFMA chains, up to where split-K cuts the input projections
    for k in ("loss_sum", "rec_sum", "kl_sum"):
        assert abs(s1[k] - s0[k]) <= 2e-5 * abs(s0[k]), (k, s1[k], s0[k])
    # backward: the dO GEMM splits K at different boundaries in the two kernels (f32 summation order), and the bf16
    # BPTT re-rounds what it is fed, so last-bit differences can flip a few bf16 roundings downstream
    assert abs(s1["norm"] - s0["norm"]) <= 1e-4 * abs(s0["norm"]), (s1["norm"], s0["norm"])
    for k in g0:       # max-norm relative; 5e-3 ~ one flipped bf16 rounding (2^-8) in a recurrent operand, amplified through BPTT
        assert rel_err(g1[k], g0[k]) < 5e-3, (k, rel_err(g1[k], g0[k]))


def check_token_sort_cache_follows_the_batch(device, V=97, ni=12, H=20, nz=4, B=6, T=7):
    """The sorted token lists of the embedding backward are cached per batch TENSOR (engine._TokenSortCache).  The cache must hit
    for a batch that comes back unchanged, and must not for a tensor whose contents were overwritten in place (version counter)
    or for a different tensor that happens to reuse its address."""
    from oracle import text_vae_oracle as O
    from vae_lagging_encoder_amd.trainer import AggressiveTextTrainer
    P = O.random_params(V, ni, H, nz, seed=71, scale=0.3, emb_scale=0.5, head_scale=0.5)
    xa, xb = O.synthetic_batch(B, T, V, seed=72).to(device), O.synthetic_batch(B, T, V, seed=73).to(device)
    eps, mi, mo = O.draw_noise(B, T, ni, H, nz, seed=74)
    noise = (eps.to(device), mi.to(torch.uint8).to(device), mo.to(torch.uint8).to(device))

    def grads_after(seq):
        vae = build_vae(V, ni, H, nz, device, params=P)
        tr = AggressiveTextTrainer(vae, lr=0.0, clip=5.0)            # lr 0: the weights stay put, every step sees the same model
        out = []
        for x in seq:
            tr.step(x() if callable(x) else x, 0.5, noise=noise)
            out.append({k: p.grad.detach().cpu().clone() for k, p in vae.named_parameters() if "embed" in k})
        return tr, out
    _, ref = grads_after([xa, xb])
    x = xa.clone()

    def overwrite():
        x.copy_(xb)                 # same tensor object, new contents
        return x
    y = xa.clone()

    def behind_the_counter():
        y.data.copy_(xb)            # a write torch's version counter does not see: the documented contract asks for invalidate_batch()
        holder["tr"].invalidate_batch(y)
        return y
    holder = {}

    def first():
        return y
    vae = build_vae(V, ni, H, nz, device, params=P)
    tr = AggressiveTextTrainer(vae, lr=0.0, clip=5.0)
    holder["tr"] = tr
    got = []
    for xx in [x, x, overwrite, lambda: xb.clone(), first, behind_the_counter]:
        tr.step(xx() if callable(xx) else xx, 0.5, noise=noise)
        got.append({k: p.grad.detach().cpu().clone() for k, p in vae.named_parameters() if "embed" in k})
    hits = tr.enc._sorts.get(x, (T, B))
    assert hits is not None                                     # the unchanged tensor is served from the cache ...
    for k in ref[0]:
        assert torch.equal(got[0][k], ref[0][k]) and torch.equal(got[1][k], ref[0][k]), k
        assert torch.equal(got[2][k], ref[1][k]), k            # ... the overwritten one is sorted again
        assert torch.equal(got[3][k], ref[1][k]), k            # and so is a new tensor
        assert torch.equal(got[4][k], ref[0][k]) and torch.equal(got[5][k], ref[1][k]), k      # invalidate_batch() after a hidden write
    # an int32 id batch is converted (the in-place fast path takes int64 only)
    tr.step(xa.to(torch.int32), 0.5, noise=noise)
    g32 = {k: p.grad.detach().cpu().clone() for k, p in vae.named_parameters() if "embed" in k}
    for k in ref[0]:
        assert torch.equal(g32[k], ref[0][k]), k
    # least-recently-used eviction instead of dropping everything
    c = tr.enc._sorts
    old_limit, type(c).LIMIT = type(c).LIMIT, 3
    try:
        keep = [O.synthetic_batch(B, T, V, seed=80 + i).to(device) for i in range(5)]
        for t_ in keep:
            tr.step(t_, 0.5, noise=noise)
        assert len(c.map) == 3 and c.get(keep[4], (T, B)) is not None and c.get(keep[2], (T, B)) is not None and c.get(keep[0], (T, B)) is None
    finally:
        type(c).LIMIT = old_limit


def check_weight_images_follow_rebound_parameters(device, V=333, ni=24, H=64, nz=8, B=7, T=11):
    """The decoder caches its bf16 weight images across calls (it is frozen for the whole inner loop).  `p.data = X` (what
    `module._apply`, a `.half().float()` round trip or a hand-written load do) keeps the parameters' version counters, so the
    cache must also be dropped when the engine re-binds its flat buffer -- otherwise the bf16 path keeps multiplying with the
    OLD weights.  Two decoders, same final weights, one of them rebound after a first step: same loss."""
    from oracle import text_vae_oracle as O
    P0 = O.random_params(V, ni, H, nz, seed=41, scale=0.2, emb_scale=0.5, head_scale=0.5)
    P1 = O.random_params(V, ni, H, nz, seed=42, scale=0.2, emb_scale=0.5, head_scale=0.5)
    x = O.synthetic_batch(B, T, V, seed=43).to(device)
    eps, mi, mo = O.draw_noise(B, T, ni, H, nz, seed=44)
    noise = (eps.to(device), mi.to(torch.uint8).to(device), mo.to(torch.uint8).to(device))

    def bf16(vae):
        vae.encoder._hip.precision = vae.decoder._hip.precision = "bf16"
        return vae
    ref = bf16(build_vae(V, ni, H, nz, device, params=P1))
    want = ref.loss(x, 0.6, noise=noise)[0].detach().cpu()
    vae = bf16(build_vae(V, ni, H, nz, device, params=P0))
    first = vae.loss(x, 0.6, noise=noise)[0].detach().cpu()
    assert vae.decoder._hip._wimg is not None                 # the cached images exist (bf16 image route in use)
    assert not torch.allclose(first, want, rtol=1e-3)
    sd = vae.state_dict()
    for k, p in vae.named_parameters():
        p.data = P1[k].to(device).clone()                      # rebinding: _version is preserved
        assert k in sd
    got = vae.loss(x, 0.6, noise=noise)[0].detach().cpu()
    assert torch.equal(got, want), float((got - want).abs().max())


def check_saved_activation_layout_is_checked(device, V=333, ni=24, H=1024, nz=8, B=16, T=9):
    """The persistent recurrences keep what the forward saves for the BPTT in a buffer of their own layout, whose record size
    depends on the rows per XCD group.  A backward that would read it differently (rows per group or the persistent route changed
    between forward and backward) must fail loudly instead of computing garbage."""
    from oracle import text_vae_oracle as O
    from vae_lagging_encoder_amd import _lib, engine
    P0 = O.random_params(V, ni, H, nz, seed=41, scale=0.05, emb_scale=0.5, head_scale=0.5)
    x = O.synthetic_batch(B, T, V, seed=43).to(device)
    vae = build_vae(V, ni, H, nz, device, params=P0)
    vae.encoder._hip.precision = vae.decoder._hip.precision = "bf16"
    enc = vae.encoder._hip
    if not engine._persistent_ok(enc, object(), B, H, device, engine._PERSIST_MAX_B):
        return False                                            # no persistent route on this device: nothing to check
    for change in ("rows", "route"):
        mu, logvar = vae.encoder(x)
        if change == "rows":
            enc.persist_rows = 8                                # forward ran with ceil(16 / 8) = 2 rows per group
        else:
            enc.persistent = False                              # backward routed to the step kernels
        try:
            raised = False
            try:
                (mu.sum() + logvar.sum()).backward()
            except _lib.LvaeError as e:
                raised = "saved activations" in str(e)
            assert raised, "backward accepted saved activations of another layout (%s changed)" % change
        finally:
            enc.persist_rows = None
            enc.persistent = True
    mu, logvar = vae.encoder(x)                                 # and the unchanged pair still works
    (mu.sum() + logvar.sum()).backward()
    return True


# ---------------------------------------------------------------------------------------------------------------------
# joint step after aggressive mode ends (text.py:418-421: encoder AND decoder stepped) and the fixed-K loop of the stress config
def check_update_both_and_fixed_k(device, V=97, ni=12, H=20, nz=4, B=6, K=4, precision="f32", tol=5e-4):
    """inner_loop(fixed_k=K) takes exactly K encoder steps on the batches the seeded host stream picks (text.py:389) with no
    data-dependent exit, then step(update='both') moves encoder and decoder (text.py:418-424 with aggressive_flag off);
    every step against oracle.inner_step on the same batches and noise."""
    import numpy as np
    from oracle import text_vae_oracle as O
    from vae_lagging_encoder_amd.trainer import AggressiveTextTrainer
    P = O.random_params(V, ni, H, nz, seed=51, scale=0.3, emb_scale=0.5, head_scale=0.5)
    Ts = [5, 8, 6]
    batches = [O.synthetic_batch(B, T, V, seed=60 + i) for i, T in enumerate(Ts)]
    klw = 0.7

    def noise_for(step, x):
        return O.draw_noise(x.shape[0], x.shape[1], ni, H, nz, seed=700 + step)
    rs = np.random.RandomState(9)
    Pr = {k: v.clone() for k, v in P.items()}
    x = batches[0]
    for step in range(K):
        eps, mi, mo = noise_for(step, x)
        Pr.update(O.inner_step(Pr, x, klw, eps, mi, mo)["new_params"])
        x = batches[int(rs.randint(0, len(batches)))]
    eps, mi, mo = noise_for(K, batches[1])
    rj = O.inner_step(Pr, batches[1], klw, eps, mi, mo, update="both")
    Pr.update(rj["new_params"])

    vae = build_vae(V, ni, H, nz, device, params=P)
    tr = AggressiveTextTrainer(vae, lr=1.0, clip=5.0, precision=precision)
    counter = {"n": 0}

    def dev_noise(n):
        e, a, b = n
        return e.to(device), a.to(torch.uint8).to(device), b.to(torch.uint8).to(device)

    def noise_fn(xb):
        n = dev_noise(noise_for(counter["n"], xb))
        counter["n"] += 1
        return n
    # window=1 would break at the first window; fixed_k must ignore the exit test altogether
    steps = tr.inner_loop([b.to(device) for b in batches], batches[0].to(device), klw, np_rng=np.random.RandomState(9),
                          max_iter=100, window=1, fixed_k=K, noise_fn=noise_fn)
    assert steps == K, steps
    tr.step(batches[1].to(device), klw, noise=dev_noise(noise_for(K, batches[1])), update="both")
    st = tr.read_stats()
    assert abs(st["loss_sum"] - float(rj["loss"].sum())) < tol * abs(float(rj["loss"].sum()))     # the joint step alone
    assert abs(st["norm"] - rj["total_norm"]) < tol * rj["total_norm"]
    sd = vae.state_dict()
    for k in ALL_KEYS:
        assert rel_err(sd[k], Pr[k]) < tol, (k, rel_err(sd[k], Pr[k]))


def check_micro_batches_against_fixture(name, device, m, precision="f32", tol=RTOL):
    """trainer.micro_batches = m: the step's batch cut into m row slices, their gradients summed, ONE clip + update -- against
    the reference fixture of the whole batch (text.py:379-387): report sums, clip norm, updated encoder weights."""
    from vae_lagging_encoder_amd.trainer import AggressiveTextTrainer
    fx = load(name)
    V, ni, H, nz = int(fx["V"]), int(fx["ni"]), int(fx["H"]), int(fx["nz"])
    vae = build_vae(V, ni, H, nz, device, params=fixture_params(fx))
    tr = AggressiveTextTrainer(vae, lr=1.0, clip=5.0, precision=precision, micro_batches=m)
    x = torch.from_numpy(fx["x"]).to(device)
    noise = (torch.from_numpy(fx["eps"]).to(device), torch.from_numpy(fx["mask_in"]).to(device), torch.from_numpy(fx["mask_out"]).to(device))
    tr.step(x, float(fx["kl_weight"]), noise=noise)
    st = tr.read_stats()
    assert abs(st["loss_sum"] - float(fx["loss"].sum())) < tol * abs(float(fx["loss"].sum()))
    assert abs(st["norm"] - float(fx["total_norm"])) < tol * float(fx["total_norm"])
    sd = vae.state_dict()
    for k in ENC_KEYS:
        assert rel_err(sd[k], fx["new/" + k]) < tol, (k, rel_err(sd[k], fx["new/" + k]))
    for k in DEC_KEYS:
        assert torch.equal(sd[k].cpu(), torch.from_numpy(fx["param/" + k])), k
    # param.grad (slot 0) holds the clipped mean gradient of the WHOLE batch
    g = dict(vae.named_parameters())["decoder.pred_linear.weight"].grad
    assert rel_err(g, fx["grad/decoder.pred_linear.weight"] * float(fx["coef"])) < 2 * tol


def check_fold_norm(device, V, ni, H, nz, B, T, precision="f32", use_graph=False, steps=3, decoder_grads="full"):
    """Norm folding (trainer._plan_fold): the clip norm assembled from the producers' partial sums of squares + a streaming pass over
    the rest must be the norm of the same gradients -- against a trainer with folding off, on the same weights, batches and noise:
    per-step norm and clip coefficient to f32 summation-order accuracy, gradients bit for bit, the encoder after `steps` updates to
    what those coefficient differences allow.  decoder_grads = "norm": the decoder's two big tensors are never written in these
    (encoder-only) steps; everything else must still agree.  (The fixture parity tests run with folding on: they compare the folded norm with
    the reference's total_norm.)"""
    from oracle import text_vae_oracle as O
    from vae_lagging_encoder_amd.trainer import AggressiveTextTrainer
    P = O.random_params(V, ni, H, nz, seed=91, scale=0.05 if H >= 512 else 0.3, emb_scale=0.5, head_scale=0.3)
    xs = [O.synthetic_batch(B, T, V, seed=20 + i).to(device) for i in range(steps)]
    out = {}
    for fold in (False, True):
        vae = build_vae(V, ni, H, nz, device, params=P)
        tr = AggressiveTextTrainer(vae, lr=1.0, clip=5.0, precision=precision, use_graph=use_graph, fold_norm=fold,
                                   decoder_grads=decoder_grads if fold else "full")
        rec = []
        for i, x in enumerate(xs):
            e, a, b = O.draw_noise(B, T, ni, H, nz, seed=40 + i)
            tr.reset_stats()
            tr.step(x, 0.5, noise=(e.to(device), a.to(torch.uint8).to(device), b.to(torch.uint8).to(device)))
            st = tr.read_stats()
            rec.append((st["norm"], st["coef"]))
            if i == 0:
                grads = {k: p.grad.detach().clone() for k, p in vae.named_parameters() if p.grad is not None}
        assert (tr._fold is not None) == fold
        assert tr.enc.fold is None and tr.dec.fold is None      # disarmed between steps: the autograd path never sees a trainer's plan
        if fold and precision == "bf16" and 2.0 * V * H * (T - 1) * B >= 1e11 and torch.device(device).type == "cuda":
            assert tr._fold.dec_pred is not None, "dW_pred's squares were expected to come from the product's epilogue at this shape"
        out[fold] = (rec, grads, {k: v.detach().clone() for k, v in vae.state_dict().items()})
    for (n0, c0), (n1, c1) in zip(out[False][0], out[True][0]):
        assert abs(n0 - n1) <= 2e-6 * n0 and abs(c0 - c1) <= 2e-6, ((n0, c0), (n1, c1))
    c0, c1 = out[False][0][0][1], out[True][0][0][1]
    for k, g in out[False][1].items():
        g1 = out[True][1][k]
        if decoder_grads == "norm" and k in ("decoder.embed.weight", "decoder.pred_linear.weight"):
            continue                                          # not materialised by an encoder-only step: unspecified
        if c0 == c1:
            assert torch.equal(g, g1), k                      # step 0: the same gradients; clipped in place by the same coefficient
        else:
            assert rel_err(g1, g) < 1e-5, k
    for k in ENC_KEYS:
        assert rel_err(out[True][2][k], out[False][2][k]) < 1e-4, k
    return out[True][0]


def check_transactional_recovery(device, V=97, ni=12, H=20, nz=4, B=6, K=5, precision="f32", fault_at=(2,), rungs_down=1,
                                 use_graph=False, decoder_grads="full"):
    """A persistent-launch hand-off timeout must never reach the weights (text.py:385-387: an update is computed from complete
    recurrences or not at all).  K inner steps + the joint decoder step with injected noise; before the steps listed in
    `fault_at` an engine's status word is set -- what a timed-out recurrence leaves behind.  The device-side gate then voids that
    step and every step queued behind it; the next host read moves both engines `rungs_down` rungs down the fallback ladder
    (the fault is re-raised through on_demote until then) and replays the voided steps.  Weights and committed report sums must
    equal, BIT FOR BIT, a run that never saw a fault and was moved to the final rung by hand before the first faulty step (the
    same kernels step for step; with fault_at = (0,) and rungs_down = 2 that is a run on the launch-per-timestep kernels)."""
    import numpy as np
    from oracle import text_vae_oracle as O
    from vae_lagging_encoder_amd import engine
    from vae_lagging_encoder_amd.trainer import AggressiveTextTrainer
    P = O.random_params(V, ni, H, nz, seed=51, scale=0.3, emb_scale=0.5, head_scale=0.5)
    Ts = [5, 8, 6] if H < 1024 else [12, 9, 14]
    batches = [O.synthetic_batch(B, T, V, seed=60 + i).to(device) for i, T in enumerate(Ts)]
    klw = 0.7
    picks = [0] + [int(i) for i in np.random.RandomState(9).randint(0, len(batches), size=K)]

    def noise_for(step, x):
        e, a, b = O.draw_noise(x.shape[0], x.shape[1], ni, H, nz, seed=700 + step)
        return e.to(device), a.to(torch.uint8).to(device), b.to(torch.uint8).to(device)

    def run(faulty):
        vae = build_vae(V, ni, H, nz, device, params=P)
        tr = AggressiveTextTrainer(vae, lr=1.0, clip=5.0, precision=precision, use_graph=use_graph, decoder_grads=decoder_grads)
        log = {"demotions": []}
        if faulty:
            def on_demote(rung):
                log["demotions"].append(rung)
                if len(log["demotions"]) < rungs_down:
                    tr.dec.status.fill_(300)            # the next rung "times out" as well
            tr.on_demote = on_demote
        sums = []
        tr.reset_stats()
        for step in range(K):
            x = batches[picks[step]]
            if faulty and step in fault_at:
                (tr.enc if step % 2 == 0 else tr.dec).status.fill_(100 + step)
            if not faulty and step == min(fault_at):
                # the clean comparison run changes rung by hand where the faulty one is forced to: same kernels step for step
                for _ in range(rungs_down):
                    for e in (tr.enc, tr.dec):
                        engine.demote_persistent(e)
            tr.step(x, klw, noise=noise_for(step, x))
            if step == K - 2:
                sums.append(tr.read_stats())            # a host read in the middle: settles what is queued so far
        tr.step(batches[1], klw, noise=noise_for(K, batches[1]), update="decoder")
        sums.append(tr.read_stats())
        return vae.state_dict(), sums, tr, log

    sd_f, sums_f, tr_f, log = run(True)
    sd_c, sums_c, tr_c, _ = run(False)
    assert log["demotions"] == list(range(1, rungs_down + 1)), log
    assert tr_f.recoveries == rungs_down and tr_c.recoveries == 0
    assert engine.persist_rung(tr_f.enc) == engine.persist_rung(tr_c.enc) == rungs_down
    assert int(tr_f.enc.status.item()) == 0 and int(tr_f.dec.status.item()) == 0
    for k in ALL_KEYS:
        assert torch.equal(sd_f[k], sd_c[k]), k
    for a, b in zip(sums_f, sums_c):
        for key in ("loss_sum", "rec_sum", "kl_sum"):
            assert a[key] == b[key], (key, a[key], b[key])
    return tr_f


# ---------------------------------------------------------------------------------------------------------------------
# Omniglot: the aggressive loop of image.py:295-327 and an eval-mode forward after a decoder update
def _image_params(vae):
    return {k: v.detach().cpu().clone() for k, v in vae.state_dict().items()}


def check_image_inner_loop(device, B=6, N=40, window=2, max_iter=7, seed=5):
    """image.py:295-327 replayed literally (np.random.choice(N, B, replace=False) batch pick, torch.bernoulli dynamic
    binarisation, window test on the mean loss per example, break) with the CPU oracle doing the arithmetic (Adam state
    and BatchNorm running statistics carried along), against AggressiveImageTrainer.inner_loop on the same picks, the
    same binarisation draws and the same eps."""
    import numpy as np
    from oracle import image_vae_oracle as IO
    from vae_lagging_encoder_amd.trainer import AggressiveImageTrainer
    vae = build_image_vae(device, 31)
    P = _image_params(vae)
    g = torch.Generator().manual_seed(seed)
    x_train = torch.rand(N, 1, 28, 28, generator=g)
    klw = 0.7

    def binarize(step, probs):
        gg = torch.Generator().manual_seed(4000 + step)
        return (torch.rand(probs.shape, generator=gg) < probs.cpu()).float()

    def eps_for(step):
        gg = torch.Generator().manual_seed(5000 + step)
        return torch.randn(B, 1, 32, generator=gg)
    first = binarize(999, x_train[:B])
    # ---- reference control flow, oracle arithmetic
    rs = np.random.RandomState(11)
    Pr = {k: v.clone() for k, v in P.items()}
    adam = None
    sub_iter, x = 1, first
    burn_n, burn_pre, burn_cur = 0, 1e4, 0.0
    ref_steps = 0
    while sub_iter < max_iter:
        burn_n += x.shape[0]
        r = IO.inner_step_adam(Pr, x, klw, eps_for(ref_steps), adam=adam)
        burn_cur += float(r["loss"].sum())
        adam = r["adam"]
        Pr.update(r["new_params"]); Pr.update(r["new_stats"]); Pr.update(r["masked_weights"])
        ref_steps += 1
        id_ = rs.choice(N, B, replace=False)
        x = binarize(ref_steps, x_train[torch.from_numpy(id_)])
        if sub_iter % window == 0:
            burn_cur = burn_cur / burn_n
            if burn_pre - burn_cur < 0:
                break
            burn_pre = burn_cur
            burn_cur = burn_n = 0
        sub_iter += 1
    # ---- the fused driver
    tr = AggressiveImageTrainer(vae, lr=1e-3, clip=5.0)
    cnt = {"eps": 0, "bin": 0}

    def eps_fn(xb):
        e = eps_for(cnt["eps"]).to(device)
        cnt["eps"] += 1
        return e

    def binarize_fn(probs):
        cnt["bin"] += 1
        return binarize(cnt["bin"], probs).to(device)
    steps = tr.inner_loop(x_train.to(device), first.to(device), klw, batch_size=B, np_rng=np.random.RandomState(11),
                          max_iter=max_iter, window=window, eps_fn=eps_fn, binarize_fn=binarize_fn)
    assert steps == ref_steps, (steps, ref_steps)
    sd = vae.state_dict()
    for k in sd:
        if k.startswith("encoder.") and not k.endswith("num_batches_tracked"):
            # Adam moves every weight by ~lr per step whatever the gradient's size: compare the accumulated UPDATE
            ref_u, got_u = Pr[k] - P[k], sd[k].cpu() - P[k]
            if k.endswith("running_mean") or k.endswith("running_var"):
                # the statistics see activations computed from weights that Adam moved: a 5 % difference in a few of the
                # ~lr-sized updates (see below) shows up here at the 1e-3 level
                assert rel_err(sd[k], Pr[k]) < 5e-3, (k, rel_err(sd[k], Pr[k]))
            else:
                assert float((got_u - ref_u).abs().max()) < 0.05 * 1e-3 * steps + 1e-7, (k, float((got_u - ref_u).abs().max()))
    return steps


def check_image_eval_after_decoder_update(device, B=6):
    """MaskedConv2d multiplies its weight by the mask on EVERY forward (dec_pixelcnn_v2.py:29), eval mode included.  After a
    step that updates the decoder (update='both') the masked taps carry non-zero values (their gradients are kept); an
    eval-mode loss must still not see the target pixel: compare with the oracle in eval mode on the updated weights."""
    from oracle import image_vae_oracle as IO
    from vae_lagging_encoder_amd.trainer import AggressiveImageTrainer
    vae = build_image_vae(device, 33)
    g = torch.Generator().manual_seed(8)
    x = (torch.rand(B, 1, 28, 28, generator=g) < 0.4).float()
    eps = torch.randn(B, 1, 32, generator=g)
    tr = AggressiveImageTrainer(vae, lr=1e-3, clip=5.0)
    tr.step(x.to(device), 1.0, eps=eps.to(device), update="both")
    sd = _image_params(vae)
    dirty = 0.0
    for k in sd:
        if k.endswith(".mask"):
            dirty = max(dirty, float((sd[k.replace(".mask", ".weight")] * (1 - sd[k])).abs().max()))
    assert dirty > 0.0            # the decoder Adam step did move the masked taps
    vae.eval()
    with torch.no_grad():
        loss, rec, kl = vae.loss(x.to(device), 1.0, nsamples=1, noise=(eps.to(device), None, None))
    vae.train()
    c = IO.Ctx(sd, train=False)
    with torch.no_grad():
        l_r, rec_r, kl_r = IO.vae_loss(c, x, 1.0, eps)
    assert rel_err(rec, rec_r) < 1e-4, rel_err(rec, rec_r)
    assert rel_err(kl, kl_r) < 1e-4


# ---------------------------------------------------------------------------------------------------------------------
# evaluation statistics: text.py:120-227 on the reference vs vae_lagging_encoder_amd.evaluation
def check_eval_against_fixture(device):
    """test / calc_mi / calc_au / calc_iwnll / nll_iw / eval_inference_dist on the drop-in modules against the numbers the
    reference's text.py reported on the same seeded model, batches and Gaussian draws (tests/golden/eval_small.npz)."""
    import argparse
    from vae_lagging_encoder_amd import evaluation as E
    fx = load("eval_small")
    V, ni, H, nz, nb = (int(fx[k]) for k in ("V", "ni", "H", "nz", "nb"))
    vae = build_vae(V, ni, H, nz, device, params=fixture_params(fx))
    vae.eval()
    batches = [torch.from_numpy(fx["x/%d" % i]).to(device) for i in range(nb)]
    orig = vae.encoder._draw_eps

    def feed(prefix):
        """Replace the encoder's Gaussian draw by the reference's recorded draws, in program order."""
        state = {"i": 0}

        def draw(batch, nsamples, nzz, dev, eps=None):
            e = torch.from_numpy(fx["%s/%d" % (prefix, state["i"])])
            state["i"] += 1
            assert tuple(e.shape) == (batch, nsamples, nzz), (prefix, state["i"], tuple(e.shape), (batch, nsamples, nzz))
            return e.to(dev)
        vae.encoder._draw_eps = draw
        return state
    try:
        with torch.no_grad():
            feed("mi_eps")
            mi = E.calc_mi(vae, batches)
            assert abs(mi - float(fx["mi"])) < 2e-4 * max(1.0, abs(float(fx["mi"]))), (mi, float(fx["mi"]))
            vae.encoder._draw_eps = orig
            n_au, au_var = E.calc_au(vae, batches, delta=0.01)
            assert n_au == int(fx["au"])
            assert rel_err(au_var, fx["au_var"]) < 1e-4
            args = argparse.Namespace(nsamples=1, iw_nsamples=20)
            feed("test_eps")
            res = E.test(vae, batches, "VAL", args, verbose=False, np_rng=np.random.RandomState(7))
            ref = fx["test"]
            for got, want in zip(res[:4], ref[:4]):
                assert abs(got - want) < 1e-4 * abs(want), (res, ref)
            assert abs(res[4] - ref[4]) < 2e-4 * max(1.0, abs(ref[4]))
            feed("iw_eps")
            nll, ppl = E.calc_iwnll(vae, batches, args, ns=10, np_rng=np.random.RandomState(8))
            assert abs(nll - fx["iwnll"][0]) < 1e-4 * abs(fx["iwnll"][0]) and abs(ppl - fx["iwnll"][1]) < 2e-4 * abs(fx["iwnll"][1])
            feed("nll_iw_b0_eps")
            v = vae.nll_iw(batches[0], nsamples=20, ns=10)
            assert rel_err(v, fx["nll_iw_b0"]) < 1e-4
            vae.encoder._draw_eps = orig
            mu, lv = vae.encoder(batches[1])
            z = torch.from_numpy(fx["infer_z"]).to(device)
            assert rel_err(vae.eval_inference_dist(batches[1], z, (mu, lv)), fx["infer_logq"]) < 1e-4
            assert rel_err(vae.eval_prior_dist(z), fx["prior_logp"]) < 1e-5
    finally:
        vae.encoder._draw_eps = orig
        vae.train()


# ---------------------------------------------------------------------------------------------------------------------
# generation: greedy / beam decoding vs the reference, sampling by its properties
def check_generation_against_fixture(device):
    fx = load("generate_small")
    V, ni, H, nz = (int(fx[k]) for k in ("V", "ni", "H", "nz"))
    vae = build_vae(V, ni, H, nz, device, params=fixture_params(fx))
    vae.eval()
    z = torch.from_numpy(fx["z"]).to(device)

    def ids(sents):
        return [[int(w[1:]) for w in s] for s in sents]
    greedy = ids(vae.decode(z, "greedy"))
    beam = ids(vae.decode(z, "beam", K=4))
    for name, got in (("greedy", greedy), ("beam", beam)):
        for i, s in enumerate(got):
            n = int(fx[name + "_len"][i])
            assert s == list(fx[name + "_ids"][i][:n]), (name, i, s[:12], list(fx[name + "_ids"][i][:12]))
    # sampling: valid words, stops right after </s>, at most 99 words, reproducible from a seeded generator, and the
    # inverse-CDF pick follows softmax(logits) (chi-square-free check: empirical frequencies of a 6-word distribution)
    gen = torch.Generator(device=device).manual_seed(3)
    a = ids(vae.decode(z, "sample") if False else vae.decoder.sample_decode(z, generator=gen))
    gen = torch.Generator(device=device).manual_seed(3)
    b = ids(vae.decoder.sample_decode(z, generator=gen))
    assert a == b
    for s in a:
        assert 1 <= len(s) <= 99 and all(0 <= w < V for w in s)
        assert 2 not in s[:-1]
    st = vae.decoder._stepper(device)
    logits = torch.tensor([[0.0, 1.0, -1.0, 2.0, 0.5, -3.0]], device=device).repeat(4000, 1)
    u = torch.rand(4000, generator=torch.Generator().manual_seed(1)).to(device)
    pick = st.sample(logits, u).cpu()
    p = torch.softmax(logits[0].cpu(), 0)
    freq = torch.bincount(pick, minlength=6).float() / 4000
    assert float((freq - p).abs().max()) < 0.03, (freq, p)
    cdf = torch.cumsum(p.double(), 0)
    want = torch.searchsorted(cdf, u.cpu().double().clamp(max=float(cdf[-1]) - 1e-9))
    assert float((pick != want).float().mean()) < 0.002           # boundary draws may fall either side in f32
    vae.train()


def check_pixelcnn_ancestral_sampling(device, B=2):
    """PixelCNNDecoderV2.decode (dec_pixelcnn_v2.py:201-232) in eval mode: (i) the final probabilities equal the CPU oracle's
    eval forward on the produced image; (ii) autoregressive consistency: pixel (i, j) was thresholded from a pass that saw
    exactly the pixels before it, so it must equal [final prob at (i, j) >= 0.5] -- the masks make that probability a
    function of earlier pixels only (any leak of the pixel itself or of later ones breaks the equality)."""
    from oracle import image_vae_oracle as IO
    vae = build_image_vae(device, 35)
    vae.eval()
    g = torch.Generator().manual_seed(4)
    z = torch.randn(B, 32, generator=g).to(device)
    img, probs = vae.decoder.decode(z, deterministic=True)
    assert tuple(img.shape) == (B, 1, 28, 28) and float(((img != 0) & (img != 1)).float().sum()) == 0
    sd = {k: v.detach().cpu() for k, v in vae.state_dict().items()}
    c = IO.Ctx(sd, train=False)
    with torch.no_grad():
        bce = IO.decoder_reconstruct_error(c, img.cpu(), z.cpu().view(B, 1, 32))
    p = probs.cpu().double().clamp(1e-12, 1 - 1e-12)
    x = img.cpu().double()
    bce_ours = -((p + 1e-12).log() * x + (1 - p + 1e-12).log() * (1 - x)).view(B, -1).sum(1)
    assert rel_err(bce_ours, bce.view(-1)) < 1e-4
    margin = (probs.cpu() - 0.5).abs()
    decided = margin > 1e-4                                     # pixels whose probability sits on the threshold may go either way
    assert bool(((probs.cpu() >= 0.5).float() == img.cpu())[decided].all())
    vae.train()


def check_pixelcnn_incremental_sampling(device, B=2, compare_full_path=True, seed=36):
    """Pixel-at-a-time sampling (image_engine.PixelCNNSampler / lv_pixelcnn_sample.hip) against the full forward, BIT FOR BIT:
    (i) every pixel's logit, computed at its own step from the cached maps, equals the logit a full eval-mode forward over the
    finished image gives at that position (causality + identical fma chains); (ii) with `compare_full_path`, decode() draws the
    same image and returns the same probabilities as the reference's literal procedure of one full decoder pass per pixel
    (decode(incremental=False)), for the deterministic rule and for Bernoulli draws from the same generator state."""
    from vae_lagging_encoder_amd import image_engine as IE
    vae = build_image_vae(device, seed)
    with torch.no_grad():                      # BatchNorm statistics and affine parameters away from their initial values
        for m in vae.decoder.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                g = torch.Generator().manual_seed(m.num_features + int(m.weight.sum().item() * 0))
                m.running_mean.copy_(torch.randn(m.num_features, generator=g).to(device) * 0.1)
                m.running_var.copy_((torch.rand(m.num_features, generator=g) + 0.5).to(device))
                m.weight.copy_((torch.rand(m.num_features, generator=g) + 0.5).to(device))
                m.bias.copy_((torch.randn(m.num_features, generator=g) * 0.1).to(device))
    vae.eval()
    g = torch.Generator().manual_seed(4)
    z = torch.randn(B, 32, generator=g).to(device)
    dec = vae.decoder
    with torch.no_grad():
        smp = IE.PixelCNNSampler(dec).start(z)
        for i in range(28):
            for j in range(28):
                p = torch.sigmoid(smp.step(i, j))
                smp.set_pixel(i, j, (p >= 0.5).float())
        img = smp.img.clone()
        inc_logits = smp.logit.clone().cpu()
        dec._hip.forward(img, z.contiguous().float())
        full_logits = dec._hip.logit.t.view(B, 784).cpu()
    nbad = int((inc_logits != full_logits).sum())
    assert nbad == 0, (nbad, float((inc_logits - full_logits).abs().max()))
    assert 0 < float(img.mean()) < 1                     # a non-trivial image
    if compare_full_path:
        for deterministic in (True, False):
            outs = []
            for incremental in (True, False):
                gen = torch.Generator(device=device).manual_seed(9) if torch.device(device).type == "cuda" else torch.Generator().manual_seed(9)
                outs.append(dec.decode(z, deterministic=deterministic, generator=gen, incremental=incremental))
            assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), deterministic
    vae.train()


# ---------------------------------------------------------------------------------------------------------------------
# a K-step trajectory of the throughput configuration at H = 1024 (persistent recurrences), step by step against the oracle
def check_bf16_trajectory_h1024(device, K=4, B=32, T=60, V=2003, ni=64, nz=32):
    """K consecutive inner steps (each on the encoder the previous step left) with the bf16 configuration at the hidden size
    and batch the persistent LSTM launches are built for, against the f32 oracle fed the same batches and noise: per-step
    loss / KL / clip norm, and the encoder after K updates.  What a single-step test cannot see is how the bf16 operand
    rounding of one step feeds the next through the weights."""
    from oracle import text_vae_oracle as O
    from vae_lagging_encoder_amd.trainer import AggressiveTextTrainer
    H = 1024
    P = O.random_params(V, ni, H, nz, seed=77, scale=0.03, emb_scale=0.3, head_scale=0.2)
    batches = [O.synthetic_batch(B, T, V, seed=300 + i) for i in range(K)]
    klw = 0.5
    Pr = {k: v.clone() for k, v in P.items()}
    ref = []
    for i, x in enumerate(batches):
        eps, mi, mo = O.draw_noise(B, T, ni, H, nz, seed=400 + i)
        r = O.inner_step(Pr, x, klw, eps, mi, mo)
        ref.append((float(r["loss"].sum()), float(r["kl"].sum()), float(r["total_norm"])))
        Pr.update(r["new_params"])
    vae = build_vae(V, ni, H, nz, device, params=P)
    tr = AggressiveTextTrainer(vae, lr=1.0, clip=5.0, precision="bf16")
    out = []
    for i, x in enumerate(batches):
        eps, mi, mo = O.draw_noise(B, T, ni, H, nz, seed=400 + i)
        tr.reset_stats()
        tr.step(x.to(device), klw, noise=(eps.to(device), mi.to(torch.uint8).to(device), mo.to(torch.uint8).to(device)))
        st = tr.read_stats()
        out.append((st["loss_sum"], st["kl_sum"], st["norm"]))
    errs = {"loss": max(abs(a[0] - b[0]) / abs(b[0]) for a, b in zip(out, ref)),
            "kl": max(abs(a[1] - b[1]) / abs(b[1]) for a, b in zip(out, ref)),
            "norm": max(abs(a[2] - b[2]) / abs(b[2]) for a, b in zip(out, ref))}
    sd = vae.state_dict()
    errs["enc_w"] = max(rel_err(sd[k], Pr[k]) for k in ENC_KEYS)
    # the update of the whole trajectory, relative to its own size (the weights barely move in K steps)
    errs["enc_update"] = max(float((sd[k].cpu().double() - Pr[k].double()).abs().max()) /
                             (float((Pr[k].double() - P[k].double()).abs().max()) + 1e-30) for k in ENC_KEYS)
    return errs


# ---------------------------------------------------------------------------------------------------------------------
# outer-loop policy (SURVEY.md 8f rows 2 + 3): replay of a recorded run of the reference's text.main() / image.main()
class _RecordingRng(object):
    """numpy RandomState with the host draws of the training loop logged (batch picks, permutations)."""

    def __init__(self, seed):
        import numpy as np
        self.rs = np.random.RandomState(seed)
        self.picks, self.perms = [], []

    def randint(self, lo, hi=None, *a, **k):
        v = self.rs.randint(lo, hi, *a, **k)
        self.picks.append(int(v))
        return v

    def permutation(self, n):
        v = self.rs.permutation(n)
        self.perms.append([int(i) for i in v])
        return v

    def __getattr__(self, name):
        return getattr(self.rs, name)


class _EpsQueue(object):
    """The Gaussian draws of the recorded run, handed out in program order; a shape mismatch means the replay took a
    different path through the program than the reference did."""

    def __init__(self, fx, nz, device):
        import numpy as np
        sizes = fx["eps_sizes"]
        flat = torch.from_numpy(fx["eps_flat"])
        self.items, off = [], 0
        for b in sizes:
            n = int(b) * nz
            self.items.append(flat[off:off + n].reshape(int(b), 1, nz))
            off += n
        self.pos, self.device = 0, device

    def pop(self, batch, nsamples, nz):
        e = self.items[self.pos]
        assert tuple(e.shape) == (batch, nsamples, nz), (self.pos, tuple(e.shape), (batch, nsamples, nz))
        self.pos += 1
        return e.to(self.device)


class _MaskQueue(object):
    """The decoder dropout keep-masks of the recorded free-running run (policy_text_free.npz), one (mask_in, mask_out) pair
    per training-mode loss call in program order."""

    def __init__(self, fx, ni, H, device):
        import numpy as np
        self.shapes = [(int(b), int(t)) for b, t in fx["mask_shapes"]]
        self.bits_in, self.bits_out = np.unpackbits(fx["mask_in_bits"]), np.unpackbits(fx["mask_out_bits"])
        self.off_in = np.concatenate([[0], np.cumsum([b * t * ni for b, t in self.shapes])])
        self.off_out = np.concatenate([[0], np.cumsum([b * t * H for b, t in self.shapes])])
        self.ni, self.H, self.pos, self.device = ni, H, 0, device

    def pop(self, x):
        b, t = self.shapes[self.pos]
        assert (b, t) == (x.shape[0], x.shape[1] - 1), (self.pos, (b, t), tuple(x.shape))
        i = self.pos
        self.pos += 1
        mi = torch.from_numpy(self.bits_in[self.off_in[i]:self.off_in[i + 1]].copy()).reshape(b, t, self.ni)
        mo = torch.from_numpy(self.bits_out[self.off_out[i]:self.off_out[i + 1]].copy()).reshape(b, t, self.H)
        return mi.to(self.device), mo.to(self.device)


def check_policy_replay_text(device, tmp_dir, max_epochs=None, rtol=2e-3, free_running=False):
    """The reference's text.main() (text.py:229-522) was run on a tiny corpus with every decision recorded
    (tests/golden/make_golden_policy.py -> policy_text.npz).  Here the SAME corpus goes through our input pipeline
    (MonoTextData -> create_data_batch on `device`), the same initial weights, the same Gaussian draws and the same host seed
    through TextTrainingLoop on the HIP path, and the run must take the same decisions: batch order, KL weight, number of
    inner encoder steps of every iteration (the windowed exit test, text.py:393-396), the batch picks inside the inner loop,
    the iteration at which aggressive training stops (MI check, text.py:447-455), which epochs update the best checkpoint, the
    learning-rate decay (text.py:469-479); and report the same statistics within `rtol`.

    free_running: the second recorded run (policy_text_free.npz: `--free`, 3 epochs with the reference's decoder dropout 0.5,
    every keep-mask recorded).  The replay is fed the masks and the Gaussian draws and is NEVER re-synchronised: 39 outer
    iterations and 1080 inner encoder steps, each on the weights the previous one left, through the aggressive phase, the MI
    check that ends it and the joint steps after it."""
    import argparse
    import os
    import numpy as np
    from vae_lagging_encoder_amd.data import MonoTextData
    from vae_lagging_encoder_amd.factory import build_text_vae
    from vae_lagging_encoder_amd.modules.encoders.encoder import GaussianEncoderBase
    from vae_lagging_encoder_amd.training import TextTrainingLoop
    fx = load("policy_text_free" if free_running else "policy_text")
    paths = {}
    for k in ("train", "val", "test"):
        paths[k] = os.path.join(str(tmp_dir), k + ".txt")
        with open(paths[k], "w") as fh:
            fh.write(str(fx[k + "_txt"]))
    train = MonoTextData(paths["train"])
    val = MonoTextData(paths["val"], vocab=train.vocab)
    test = MonoTextData(paths["test"], vocab=train.vocab)
    bs, nz, ni, H = int(fx["batch_size"]), int(fx["nz"]), int(fx["ni"]), int(fx["H"])
    tb = train.create_data_batch(bs, torch.device(device), batch_first=True)
    vb = val.create_data_batch(bs, torch.device(device), batch_first=True)
    sb = test.create_data_batch(bs, torch.device(device), batch_first=True)
    assert [len(tb), len(vb), len(sb)] == [int(v) for v in fx["n_lists"][:3]]
    V = len(train.vocab)
    init = {k[5:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("init/")}
    assert init["encoder.embed.weight"].shape[0] == V
    p_drop = float(fx["dropout"]) if free_running else 0.0
    vae = build_text_vae(V, ni, H, nz, device, seed=int(fx["seed"]), params=init, dropout_in=p_drop, dropout_out=p_drop, vocab=train.vocab)
    epochs = int(fx["epochs"]) if max_epochs is None else max_epochs
    args = argparse.Namespace(kl_start=float(fx["kl_start"]), warm_up=int(fx["warm_up"]), batch_size=bs, epochs=epochs, aggressive=1,
                              nsamples=1, test_nepoch=int(fx["test_nepoch"]), iw_nsamples=100, momentum=0)
    queue = _EpsQueue(fx, nz, device)
    rng = _RecordingRng(int(fx["seed"]))
    saved = GaussianEncoderBase._draw_eps
    GaussianEncoderBase._draw_eps = lambda self, batch, nsamples, nz_, dev, eps=None: queue.pop(batch, nsamples, nz_) if eps is None else saved(self, batch, nsamples, nz_, dev, eps)
    logs = []

    def resync(loop, epoch):
        # SGD at lr = 1.0 on 44 sentences amplifies f32 rounding differences ~3x per epoch (1e-7 after epoch 0, 1e-3 after epoch
        # 10, measured): every epoch starts from the weights the reference's epoch started from and is checked on its own
        st = {k[12:]: torch.from_numpy(fx[k][epoch]) for k in fx.files if k.startswith("epoch_start/")}
        if epoch > 0:
            sd = vae.state_dict()
            drift.append(max(rel_err(sd[k], st[k]) for k in ALL_KEYS))
        vae.load_state_dict(st, strict=False)
    drift = []
    masks = _MaskQueue(fx, ni, H, device) if free_running else None

    def noise_fn(x):
        eps = queue.pop(x.shape[0], 1, nz)
        return (eps,) + (masks.pop(x) if free_running else (None, None))
    try:
        loop = TextTrainingLoop(vae, tb, vb, sb, args, n_train_sentences=len(train), log=logs.append, np_rng=rng,
                                seed=int(fx["seed"]), noise_fn=noise_fn, epoch_hook=None if free_running else resync)
        out = loop.run()
    finally:
        GaussianEncoderBase._draw_eps = saved
    n_it = len(loop.iterations)
    per_epoch = len(tb)
    assert n_it == epochs * per_epoch
    it = loop.iterations
    # ---- per-iteration decisions: exact ------------------------------------------------------------------------------------
    assert [r["batch"] for r in it] == [int(v) for v in fx["it_batch"][:n_it]]
    assert np.allclose([r["kl_weight"] for r in it], fx["it_klw"][:n_it], rtol=0, atol=1e-12)
    assert [int(r["aggressive"]) for r in it] == [int(v) for v in fx["it_aggr"][:n_it]]
    assert [r["inner_steps"] for r in it] == [int(v) for v in fx["it_inner"][:n_it]], "inner-loop exit decisions differ"
    n_picks = int(sum(fx["it_inner"][:n_it]))
    assert rng.picks[:n_picks] == [int(v) for v in fx["picks"][:n_picks]]
    # ---- per-iteration statistics of the joint step -------------------------------------------------------------------------
    rec = np.array([r["rec_sum"] for r in it])
    kl = np.array([r["kl_sum"] for r in it])
    assert np.abs(rec - fx["it_rec"][:n_it]).max() <= rtol * np.abs(fx["it_rec"][:n_it]).max(), np.abs(rec - fx["it_rec"][:n_it]).max()
    assert np.abs(kl - fx["it_kl"][:n_it]).max() <= rtol * max(1.0, np.abs(fx["it_kl"][:n_it]).max()) * 5
    # ---- aggressive stop -------------------------------------------------------------------------------------------------------
    stop_ref = [int(v) for v in fx["stop_burning"]]
    flips = [i for i in range(1, n_it) if it[i - 1]["aggressive"] and not it[i]["aggressive"]]
    assert flips == [s for s in stop_ref if s < n_it], (flips, stop_ref)
    k = len(loop.mi_checks)
    assert k == len([1 for m in fx["mi_checks"]][:k]) and np.allclose(np.array(loop.mi_checks), fx["mi_checks"][:k], atol=2e-3)
    # ---- per-epoch table -------------------------------------------------------------------------------------------------------
    h = loop.history
    assert len(h) == epochs
    ref_val = fx["val"][:epochs]             # avg_loss, kl, mi, recon, nll, ppl as printed (4 decimals)
    for e in range(epochs):
        assert abs(h[e]["loss"] - ref_val[e][0]) <= rtol * abs(ref_val[e][0]) + 1e-4, (e, h[e]["loss"], ref_val[e][0])
        assert abs(h[e]["kl"] - ref_val[e][1]) <= 5 * rtol * max(1.0, abs(ref_val[e][1])) + 1e-4
        assert abs(h[e]["mi"] - ref_val[e][2]) <= 5 * rtol * max(1.0, abs(ref_val[e][2])) + 1e-4
        assert abs(h[e]["ppl"] - ref_val[e][5]) <= 5 * rtol * abs(ref_val[e][5])
        assert h[e]["au"] == int(fx["au"][e])
    assert [e for e in range(epochs) if h[e]["best_updated"]] == [int(v) for v in fx["best_epochs"] if v < epochs]
    assert np.allclose([h[e]["lr_after"] for e in range(epochs)], fx["lr_by_epoch"][:epochs])
    if max_epochs is None:
        best = {k[5:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("best/")}
        sd = vae.state_dict()
        worst = max(rel_err(sd[k], best[k]) for k in ALL_KEYS)
        assert worst < 50 * rtol, worst                 # run() ends on the best checkpoint, as text.py:506 does
    # what one epoch of the replay drifts from the reference's weights before it is re-synchronised
    assert max(drift + [0.0]) < 5e-3, drift
    if free_running:
        assert masks.pos == len(masks.shapes) or max_epochs is not None, (masks.pos, len(masks.shapes))
    return dict(iterations=n_it, inner_steps=int(sum(r["inner_steps"] for r in it)), eps_used=queue.pos, out=out, drift=drift)


def check_policy_replay_image(device, rtol=5e-3):
    """The reference's image.main() (image.py:189-440) was run for 7 epochs on a 50-image set with every decision recorded
    (tests/golden/make_golden_policy_image.py -> policy_image.npz: shuffled-loader orders, binarisation draws, Gaussian
    draws, inner-loop picks).  The same data, orders and draws go through ImageTrainingLoop on the HIP path; the run must
    take the same decisions -- inner encoder steps per iteration (windowed exit, image.py:320-325), the five-strike MI
    patience (image.py:381-395), best-checkpoint updates, the learning-rate decay with re-created Adam optimizers
    (image.py:411-420) -- and report the same statistics within `rtol`.  Adam at lr 1e-3 does not amplify f32 rounding
    differences the way the text loop's SGD at lr 1.0 does (two reference runs whose initial weights differ by 1e-6 relative
    stay within 1e-3 of each other over these 7 epochs: the fixture script's --perturb mode), so no re-synchronisation."""
    import argparse
    import numpy as np
    from vae_lagging_encoder_amd.modules.encoders.encoder import GaussianEncoderBase
    from vae_lagging_encoder_amd.training import ImageTrainingLoop
    fx = load("policy_image")
    dev = torch.device(device)
    xs = [torch.from_numpy(fx[k]).float().div(255.0).to(dev) for k in ("x_train", "x_val", "x_test")]
    vae = build_image_vae(device, int(fx["seed"]))
    sd = vae.state_dict()
    for k in [k for k in fx.files if k.startswith("init_idx/")]:
        name = k[9:]
        assert torch.equal(sd[name].reshape(-1)[torch.from_numpy(fx[k]).to(dev)].cpu(), torch.from_numpy(fx["init_val/" + name])), name
    nz = 32
    queue = _EpsQueue(fx, nz, dev)
    orders = np.split(fx["orders"].astype(np.int64), np.cumsum(fx["order_sizes"])[:-1])
    opos = [0]

    def order_fn(n):
        o = orders[opos[0]]
        assert len(o) == n, (opos[0], len(o), n)
        opos[0] += 1
        return o.tolist()
    bits = np.unpackbits(fx["bern_bits"])
    bsizes = [int(b) for b in fx["bern_sizes"]]
    boff = np.concatenate([[0], np.cumsum([b * 784 for b in bsizes])])
    bpos = [0]

    def binarize_fn(probs):
        i = bpos[0]
        assert probs.shape[0] == bsizes[i], (i, probs.shape, bsizes[i])
        bpos[0] += 1
        return torch.from_numpy(bits[boff[i]:boff[i + 1]].astype(np.float32)).reshape(bsizes[i], 1, 28, 28).to(dev)

    class Rng(object):
        def __init__(self, seed):
            self.rs, self.picks = np.random.RandomState(seed), []

        def choice(self, n, size, replace=True):
            v = self.rs.choice(n, size, replace=replace)
            self.picks.append([int(i) for i in v])
            return v
    rng = Rng(int(fx["seed"]))
    args = argparse.Namespace(kl_start=float(fx["kl_start"]), warm_up=int(fx["warm_up"]), batch_size=int(fx["batch_size"]),
                              epochs=int(fx["epochs"]), aggressive=1, nsamples=1, test_nepoch=int(fx["test_nepoch"]))
    saved = GaussianEncoderBase._draw_eps
    GaussianEncoderBase._draw_eps = lambda self, batch, nsamples, nz_, d, eps=None: queue.pop(batch, nsamples, nz_) if eps is None else saved(self, batch, nsamples, nz_, d, eps)
    logs = []
    try:
        loop = ImageTrainingLoop(vae, xs[0], xs[1], xs[2], args, log=logs.append, np_rng=rng, seed=int(fx["seed"]), order_fn=order_fn,
                                 binarize_fn=binarize_fn, eps_fn=lambda x: queue.pop(x.shape[0], 1, nz),
                                 decay_epoch=int(fx["decay_epoch"]))
        loop.run()
    finally:
        GaussianEncoderBase._draw_eps = saved
    it, h = loop.iterations, loop.history
    n_it = len(it)
    assert n_it == len(fx["it_inner"]) and len(h) == len(fx["val"])
    assert [r["inner_steps"] for r in it] == [int(v) for v in fx["it_inner"]], "inner-loop exit decisions differ"
    assert np.allclose([r["kl_weight"] for r in it], fx["it_klw"], rtol=0, atol=1e-12)
    n_picks = int(sum(fx["it_inner"]))
    assert rng.picks[:n_picks] == fx["picks"][:n_picks].astype(np.int64).tolist()
    rec = np.array([r["rec_sum"] for r in it])
    kl = np.array([r["kl_sum"] for r in it])
    assert np.abs(rec - fx["it_rec"]).max() <= rtol * np.abs(fx["it_rec"]).max(), float(np.abs(rec - fx["it_rec"]).max())
    assert np.abs(kl - fx["it_kl"]).max() <= 5 * rtol * max(1.0, float(np.abs(fx["it_kl"]).max()))
    flips = [i for i in range(1, n_it) if it[i - 1]["aggressive"] and not it[i]["aggressive"]]
    assert flips == [int(v) for v in fx["stop_burning"]], (flips, fx["stop_burning"])
    for e in range(len(h)):
        ref = fx["val"][e]                       # avg_loss, kl, mi, recon, nll as printed
        assert abs(h[e]["loss"] - ref[0]) <= rtol * abs(ref[0]) + 1e-4, (e, h[e]["loss"], ref[0])
        assert abs(h[e]["kl"] - ref[1]) <= 5 * rtol * max(1.0, abs(ref[1])) + 1e-4
        assert h[e]["au"] == int(fx["au"][e])
    assert [e for e in range(len(h)) if h[e]["best_updated"]] == [int(v) for v in fx["best_epochs"]]
    assert np.allclose([h[e]["lr_after"] for e in range(len(h))], fx["lr_after"])
    return dict(iterations=n_it, inner_steps=int(sum(r["inner_steps"] for r in it)), eps_used=queue.pos, orders_used=opos[0])
