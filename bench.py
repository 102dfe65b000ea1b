#!/usr/bin/env python
"""bench.py -- aggressive-loop sequences/sec on the BASELINE.json metric configuration.

    python bench.py --gpus N --steps K --warmup W          (N > 1: this process starts the N ranks itself, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W          (the same ranks under torchrun)

One "step" = one body of the aggressive inner loop (reference text.py:373-387: zero_grad, VAE.loss, backward,
clip_grad_norm_ over encoder+decoder grads, encoder SGD step) on one synthetic batch of the Yahoo LSTM-VAE
configuration (B=32 sequences per GPU, T=200, V=20001, ni=512, H=1024, nz=32).  Inputs (a pool of 64
batches) are resident in HBM before the timed region; noise (eps, dropout masks) is drawn on device.  Weak scaling:
every rank runs its own B=32 batch and the flat gradient buffers are mean-all-reduced over RCCL each step.

Default arithmetic = BASELINE.json's GPU configuration ("Yahoo LSTM-VAE bf16"): the large GEMMs run on the bf16 matrix
pipe with f32 accumulate, f32 master weights / activations / gradients, f32 LSTM recurrence; `--dtype f32` times the
exact-f32 parity path (the one the 1e-4 ELBO parity tests run), and the default line carries that number too
(`f32_parity_path`).

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` (the kernel group with the largest
share of the step, bracketed live by HIP events on its launch stream inside the timed region; the other group is
`roofline_secondary`) and `cpu_baseline` (the CPU oracle's ATen path = the reference's CPU op sequence, timed on this
box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json metric: "aggressive-loop seqs/sec (Yahoo LSTM-VAE, bsz=32, len=200)"
    "yahoo": dict(V=20001, ni=512, H=1024, nz=32, B=32, T=200),
    "yelp": dict(V=19997, ni=512, H=1024, nz=32, B=32, T=100),
    "toy": dict(V=1004, ni=50, H=50, nz=1, B=16, T=12),
    # BASELINE.json configs[4]: Yahoo dims, B=128 per GPU, fixed K=50 inner steps (no data-dependent exit)
    "stress": dict(V=20001, ni=512, H=1024, nz=32, B=128, T=200),
    # BASELINE.json configs[3]: Omniglot ResNet-enc + PixelCNN-dec, 28x28 binary (parity-test case; bench line on request)
    "omniglot": dict(B=50),
}
EVENT_EVERY = 4                   # timed region: steps between two event-bracketed ones (text workloads, eager mode)
PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md chip table
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 MFMA (same table)
PEAK_HBM_GBS = 8000.0             # HBM3E spec (6.29 TB/s measured copy)
GFLOP_PER_SEQ = {"yahoo": 39.67, "yelp": 19.75}   # SURVEY.md 8(d) / BASELINE.md section 4


def fwd_flops(V, ni, H, nz, B, T):
    return 2 * B * (T * ni * 4 * H + T * H * 4 * H + H * 2 * nz) + 2 * B * nz * H + \
        2 * B * (T - 1) * ((ni + nz) * 4 * H + H * 4 * H + H * V)


def bench_omniglot(args, dev, rank, world):
    out = measure_omniglot(args, dev, rank, world, cpu_baseline=(rank == 0 and not args.no_cpu_baseline))
    if rank == 0:
        emit(out)


_RESULT_FD = None       # the process's original stdout once protect_stdout() has pointed fd 1 at stderr


def protect_stdout():
    """ONE JSON line on stdout is the contract, and libraries write there too (RCCL prints its version banner on stdout -- buffered,
    so it lands AFTER the result line, exactly where a last-line parser looks).  From here on fd 1 is stderr for everybody --
    Python's sys.stdout, C stdio, child processes -- and emit() writes the result line to the saved original."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def emit(out):
    """The one JSON line; a supervised rank 0 then tells its supervisor that the line is out (whatever teardown does)."""
    line = (json.dumps(out) + "\n").encode()
    if _RESULT_FD is not None:
        view = memoryview(line)
        while len(view):                        # (a pipe may take the line in pieces)
            view = view[os.write(_RESULT_FD, view):]
    else:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    if os.environ.get("LVAE_BENCH_DONE"):
        open(os.environ["LVAE_BENCH_DONE"], "w").write("done\n")


def measure_omniglot(args, dev, rank, world, cpu_baseline=False, profile_eager=False):
    """images/sec through the aggressive inner step of the Omniglot VAE (image.py:300-314), B=50 per GPU, replicas only
    (BatchNorm batch statistics make naive data parallelism non-equivalent: SURVEY.md 8e).  profile_eager: take the per-group
    kernel times from an eager pass even when the timed steps were hipGraph replays (a second trainer on the same model)."""
    from vae_lagging_encoder_amd import engine
    from vae_lagging_encoder_amd.factory import build_image_vae
    from vae_lagging_encoder_amd.trainer import AggressiveImageTrainer
    B = WORKLOADS["omniglot"]["B"]
    vae = build_image_vae(dev, 783435)
    tr = AggressiveImageTrainer(vae, lr=1e-3, clip=5.0, seed=783435 + rank, precision=args.dtype, use_graph=bool(args.graph))
    g = torch.Generator().manual_seed(1 + rank)
    probs = torch.rand(args.pool, B, 1, 28, 28, generator=g).to(dev)
    rs = np.random.RandomState(783435)

    def one_step():
        tr.step(tr.binarize(probs[int(rs.randint(0, args.pool))]), 1.0)
    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize(dev)
    tr.reset_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    # per-group kernel times: a separate, untimed pass.  The eager step is HOST-bound (~660 launches of 5-35 us), so HIP events
    # around a launch would measure the host's gaps; each profiled step is therefore queued behind a ~20 ms device-side sleep,
    # the host runs ahead, and the kernels (and the event records between them) execute back to back.
    prof = {}
    prof_steps = 0
    stats = tr.read_stats()
    if not args.graph or profile_eager:
        if args.graph:
            tr = AggressiveImageTrainer(vae, lr=1e-3, clip=5.0, seed=783435 + rank, precision=args.dtype, use_graph=False)
            one_step()                                      # allocate the eager path's buffers outside the profiled steps
        engine.PROFILE = prof
        prof_steps = 4
        for _ in range(prof_steps):
            torch.cuda._sleep(40_000_000)
            one_step()
        torch.cuda.synchronize(dev)
        engine.PROFILE = None
    value = world * B * args.steps / dt
    out = {"metric": "aggressive-loop images/sec", "value": round(value, 2), "unit": "img/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
           "dtype_detail": {"f32": "exact-f32 MFMA products, f32 everywhere",
                            "bf16x3": "decoder's direct convolutions on split-bf16 operands (three bf16 MFMAs per product, f32 accumulate: f32-like results), the rest exact f32",
                            "bf16": "bf16 MFMA operands, f32 accumulate"}.get(args.dtype),
           "data": "synthetic",
           "config": {"workload": "omniglot ResNetEncoderV2 + PixelCNNDecoderV2 aggressive inner step (fwd+bwd+clip+encoder Adam), "
                                  "B=%d/GPU, 28x28 binary, nz=32, fm=4" % B, "global_batch": world * B, "parallelism": "replicas%d" % world,
                      "hipgraph": bool(args.graph)},
           "mean_loss_per_image": round(stats["loss_sum"] / (B * args.steps), 4)}
    # kernel groups of the step, each bracketed live by HIP events on the launch stream (eager mode): the direct masked
    # convolutions and the im2col GEMMs against the f32 MFMA peak (flops over the taps the mask keeps), BatchNorm (+ residual +
    # ELU) and the pointwise convolutions against HBM (bytes of the activations they stream)
    peak = PEAK_BF16_MFMA_TFLOPS if args.dtype == "bf16" else PEAK_F32_MFMA_TFLOPS
    gemm_group = "gemm_bf16" if args.dtype == "bf16" else "gemm_f32"
    # the direct convolutions' pipe: exact f32, bf16, or bf16 at three instructions per product (a third of its peak is what they can reach)
    conv_peak = {"f32": PEAK_F32_MFMA_TFLOPS, "bf16": PEAK_BF16_MFMA_TFLOPS, "bf16x3": round(PEAK_BF16_MFMA_TFLOPS / 3.0, 1)}[args.dtype]
    conv_label = {"f32": "exact-f32 MFMA", "bf16": "bf16 operands, f32 accumulate",
                  "bf16x3": "split-bf16 operands: three bf16 MFMAs per product, f32 accumulate; peak = a third of the bf16 pipe's"}[args.dtype]
    groups = {}
    for name, recs in prof.items():
        ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in recs)
        groups[name] = dict(ms=ms, work=sum(w for _, _, w, _ in recs), launches=len(recs))

    def mfma_view(name, label):
        g = groups.get(name)
        if not g or g["ms"] <= 0:
            return None
        tf = g["work"] / (g["ms"] * 1e-3) / 1e12
        pk = conv_peak if name == "conv_direct" else peak
        return {"bound": "mfma", "kernel": label, "achieved": round(tf, 2), "peak": pk, "unit": "TFLOP/s", "frac": round(tf / pk, 4),
                "traffic": None, "launches_per_step": g["launches"] // prof_steps, "ms_per_step": round(g["ms"] / prof_steps, 4),
                "gflop_per_step": round(g["work"] / prof_steps / 1e9, 1)}

    def hbm_view(name, label):
        g = groups.get(name)
        if not g or g["ms"] <= 0:
            return None
        gbs = g["work"] / (g["ms"] * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": label, "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None, "launches_per_step": g["launches"] // prof_steps,
                "ms_per_step": round(g["ms"] / prof_steps, 4), "algorithmic_MB_per_step": round(g["work"] / prof_steps / 1e6, 1)}
    views = [v for v in (
        mfma_view("conv_direct", "conv32_direct%s_kernel / conv32_wgrad%s_kernel (masked 32->32 k x k convolutions, %s, forward + data gradient over the kept taps, weight gradient over all taps)" % (("", "", conv_label) if args.dtype == "f32" else ("_b16", "_b16", conv_label))),
        hbm_view("batchnorm", "bn_reduce_v4 / bn_apply_{fwd,bwd}_v4 (BatchNorm + residual + ELU, forward and backward)"),
        hbm_view("conv_pointwise", "conv1x1_kernel / conv1x1_wgrad_kernel (pointwise 32/64-channel convolutions)"),
        mfma_view(gemm_group, "lv_%s_kernel (im2col convolutions of the ResNet encoder and the MaskA block, linear layers)" % gemm_group),
    ) if v]
    views.sort(key=lambda v: -v["ms_per_step"])
    if views:
        out["roofline"] = views[0]
        out["roofline_other_groups"] = views[1:]
        out["kernel_groups_ms_per_step"] = round(sum(v["ms_per_step"] for v in views), 4)
        out["roofline"]["note"] = ("B = 50: every kernel of the step is a 5-35 us launch over a 5-10 MB activation, so each group sits "
                                   "far below its roofline (latency bound); group times are device times (profiled steps queued behind "
                                   "a device-side sleep), the eager step itself is host-launch bound and the hipGraph replay is not")
    else:
        out["roofline"] = {"bound": "mfma", "achieved": None, "peak": peak, "unit": "TFLOP/s", "frac": None, "traffic": None,
                           "note": "graph replay: no per-kernel events; see the eager run / profiles/ for the kernel breakdown"}
    if cpu_baseline:
        from oracle import image_vae_oracle as IO          # the checker, used here only as the timed CPU baseline
        nthreads = min(32, os.cpu_count() or 1)
        torch.set_num_threads(nthreads)
        Pd = {k: v.detach().cpu() for k, v in vae.state_dict().items()}
        xb = (probs[0].cpu() > 0.5).float()
        eps = torch.randn(B, 1, 32)
        IO.inner_step_adam(Pd, xb, 1.0, eps)
        ts = []
        for _ in range(3):
            tc = time.perf_counter()
            IO.inner_step_adam(Pd, xb, 1.0, eps)
            ts.append(time.perf_counter() - tc)
        out["cpu_baseline"] = {"value": round(B / sorted(ts)[1], 2), "unit": "img/s", "cores": nthreads, "kind": "port",
                               "sample": "median of 3 timed inner steps (image.py:300-314) at B=%d after 1 warm-up, torch CPU ATen ops "
                                         "(the reference's CPU path restated in oracle/image_vae_oracle.py)" % B}
        out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
    return out


def lstm_algorithmic_bytes(kind, T, B, H, wb, per_step_weights):
    """Algorithmic HBM bytes of one recurrence (SURVEY.md 8d accounting: every saved activation written once forward and
    read once backward, the recurrent weights read once per launch).
      forward, per timestep:  gx in (16 B per unit-row) + gate records out (16) + c, h out (4 + 4) [+ dropped h out (4) +
                              keep-mask in (1) for the decoder]                       = 40 (45) * B * H
      BPTT, per timestep:     gate records in (16) + c in (4) + bf16 dG image out (8) [+ dh_ext in (4) + mask in (1)] = 28 (33) * B * H
    per_step_weights: the launch-per-step kernels re-read W_hh (wb bytes per element) every timestep."""
    per_t = {"fwd_enc": 40, "fwd_dec": 45, "bwd_dec": 33, "bwd_enc": 28}[kind] * B * H
    w = wb * 4 * H * H
    return per_t * T + w * (T if per_step_weights else 1)


def text_rooflines(prof, steps, workload, dtype, B, T, H, peak_mfma):
    """(gemm_roof, lstm_roof, pmc whole-step GB or None) from the HIP-event records engine.PROFILE collected over `steps` steps."""
    args = argparse.Namespace(steps=steps, workload=workload, dtype=dtype)
    whole_step_pmc = None
    # live HIP-event timing of the kernel groups that make up the step, on their launch stream
    groups = {}
    for name, recs in prof.items():
        ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in recs)
        groups[name] = dict(ms=ms, work=sum(w for _, _, w, _ in recs), launches=sum(n for _, _, _, n in recs))
    gname = "gemm_" + args.dtype
    gemm = groups.get(gname, dict(ms=0.0, work=0.0, launches=0))
    gemm_tf = gemm["work"] / (gemm["ms"] * 1e-3) / 1e12 if gemm["ms"] > 0 else 0.0
    gemm_roof = {
        "bound": "mfma", "kernel": "lv_gemm_b16_t256_kernel / _t256q_kernel (256x256x64, the three vocabulary-sized products) / lv_gemm_b16_t256g_kernel (the same K loop as one grouped stream-K launch per LSTM layer: dX + [dW_ih | dW_hh], tiles summed inside the launch) / lv_gemm_b16_nt_glds_kernel (128x128x64: the two input projections)" if args.dtype == "bf16" else "lv_gemm_f32_kernel", "achieved": round(gemm_tf, 2),
        "peak": peak_mfma, "unit": "TFLOP/s", "frac": round(gemm_tf / peak_mfma, 4), "traffic": None,
        "launches_per_step": gemm["launches"] // args.steps, "ms_per_step": round(gemm["ms"] / args.steps, 4),
        "gflop_per_step": round(gemm["work"] / args.steps / 1e9, 1)}
    wb = 2.0 if args.dtype == "bf16" else 4.0
    kinds = ("fwd_enc", "fwd_dec", "bwd_dec", "bwd_enc")
    lstm_ms = lstm_bytes = 0.0
    lstm_launches = steps_total = 0
    per_kind = {}
    persistent = False
    for k in kinds:
        gk = groups.get("lstm_" + k)
        if not gk:
            continue
        Tk = T if k.endswith("enc") else T - 1
        calls = int(round(gk["work"] / Tk))
        per_step_w = gk["launches"] > calls                 # launch-per-step kernels re-read W_hh every timestep
        persistent = persistent or not per_step_w
        by = lstm_algorithmic_bytes(k, Tk, B, H, wb, per_step_w) * calls
        per_kind[k] = {"us_per_timestep": round(1e3 * gk["ms"] / gk["work"], 3), "ms_per_step": round(gk["ms"] / args.steps, 4),
                       "algorithmic_MB_per_call": round(by / calls / 1e6, 1),
                       "GBs": round(by / (gk["ms"] * 1e-3) / 1e9, 1)}
        lstm_ms += gk["ms"]; lstm_bytes += by; lstm_launches += gk["launches"]; steps_total += gk["work"]
    lstm_gbs = lstm_bytes / (lstm_ms * 1e-3) / 1e9 if lstm_ms > 0 else 0.0
    lstm_roof = {
        "bound": "hbm",
        "kernel": ("lstm_fwd_persist_k16_kernel / lstm_bwd_persist_rs16_kernel (one launch per recurrence, W_hh register-resident, 16x16x32 MFMA with the weights as the A operand, tagged-granule hand-off per timestep kept in the XCD's L2: all-gather of h forward, reduce-scatter of partial dh sums in BPTT)"
                   if persistent else "lstm_step_{fwd,bwd_elem,bwd_mm}_kernel (one launch per timestep and stage)"),
        "achieved": round(lstm_gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(lstm_gbs / PEAK_HBM_GBS, 4),
        "traffic": None, "launches_per_step": lstm_launches // args.steps, "ms_per_step": round(lstm_ms / args.steps, 4),
        "avg_launch_us": round(1e3 * lstm_ms / max(1, lstm_launches), 3),
        "timesteps_per_step": int(steps_total // args.steps),
        "us_per_timestep": round(1e3 * lstm_ms / max(1, steps_total), 3),
        "algorithmic_bytes_per_launch": round(lstm_bytes / max(1, lstm_launches)),
        "per_recurrence": per_kind,
        "note": "latency-bound chain of dependent timesteps: a persistent launch costs 12 us (forward) / 16 us (BPTT) fixed + 1.64 / "
                "1.61 us per timestep at 4 rows per XCD group (profiles/microbench/lstm_fixed_cost_probe.py; per timestep one L2 round "
                "trip for the hand-off -- 16-byte granules, every dword under its own tag --, 64 MFMAs with their fragments, the cell "
                "update and the publishing stores): us_per_timestep is the "
                "actionable number, the HBM fraction is what a perfectly overlapped version would be bound by"}
    # HBM traffic per launch from the PMC passes committed under profiles/ (a profiler cannot wrap this process)
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            pmc = json.load(fh)
        tr_ = pmc.get(args.workload, {}).get(args.dtype)
        if tr_:
            lk = tr_.get("lstm_persist_MB_per_launch" if persistent else "lstm_MB_per_launch")
            if lk is not None and lstm_ms > 0:
                lstm_roof["traffic"] = round(lk * 1e6)
            if tr_.get("gemm_MB_per_launch") is not None and gemm["ms"] > 0:
                gemm_roof["traffic"] = round(tr_["gemm_MB_per_launch"] * 1e6)
            lstm_roof["traffic_source"] = gemm_roof["traffic_source"] = pmc.get("source_short", "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950-corrected)")
            if "whole_step_GB" in tr_:
                whole_step_pmc = tr_["whole_step_GB"]
    except (OSError, ValueError, KeyError):
        pass
    return gemm_roof, lstm_roof, gemm["ms"], lstm_ms, whole_step_pmc


def text_rooflines_split(prof_lstm, steps, prof_gemm, gemm_steps, workload, dtype, B, T, H, peak_mfma):
    """The LSTM group from the events of the timed region (`steps` steps), the GEMM group from the separate untimed pass
    (`gemm_steps` steps); gemm_ms is rescaled to the timed region's step count so that the callers' per-step arithmetic holds."""
    _, lstm_roof, _, lstm_ms, pmc_gb = text_rooflines(prof_lstm, steps, workload, dtype, B, T, H, peak_mfma)
    gemm_roof, _, gemm_ms, _, _ = text_rooflines(prof_gemm, gemm_steps, workload, dtype, B, T, H, peak_mfma)
    gemm_roof["measured"] = "separate untimed pass of %d steps right behind the timed region (HIP events around every GEMM launch cost ~5 us of queue idle each)" % gemm_steps
    lstm_roof["measured"] = "live, HIP events on the launch stream inside the timed region"
    return lstm_roof, gemm_roof, gemm_ms * steps / max(1, gemm_steps), lstm_ms, pmc_gb


def side_run_text(workload, dev, steps, warmup, dtype="bf16", decoder_grads="full", cpu_leg=False):
    """A compact record of one of the other BASELINE.json text configurations, measured in the same process after the headline
    (same trainer, same kernels, its own model / pool): {value, unit, ms_per_step, dtype, workload, dominant kernel group + its
    roofline fraction}.  stress = one fixed-K inner loop of `steps` steps (BASELINE.json configs[4])."""
    from vae_lagging_encoder_amd import engine
    from vae_lagging_encoder_amd.factory import build_text_vae, synthetic_batch
    from vae_lagging_encoder_amd.trainer import AggressiveTextTrainer
    cfg = WORKLOADS[workload]
    V, ni, H, nz, B, T = (cfg[k] for k in ("V", "ni", "H", "nz", "B", "T"))
    stress = workload == "stress"
    vae = build_text_vae(V, ni, H, nz, dev, seed=783435)
    tr = AggressiveTextTrainer(vae, lr=1.0, clip=5.0, seed=783435, precision=dtype, decoder_grads=decoder_grads)
    pool = [synthetic_batch(B, T, V, seed=7000 + i).to(dev) for i in range(8 if stress else 16)]
    tr.prepare_batches(pool)
    rs = np.random.RandomState(783435)
    for _ in range(warmup):
        tr.step(pool[int(rs.randint(0, len(pool)))], 0.1)
    tr.commit()

    def run(n_steps):
        if stress:
            n = tr.inner_loop(pool, pool[0], 0.1, np_rng=rs, max_iter=10 ** 9, fixed_k=n_steps)
            assert n == n_steps
        else:
            for _ in range(n_steps):
                tr.step(pool[int(rs.randint(0, len(pool)))], 0.1)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run(steps)                                  # the timed region carries no HIP events (each record is ~5 us of queue idle)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    # kernel groups from a second, untimed pass
    prof = {}
    psteps = min(steps, 10)
    engine.PROFILE = prof
    run(psteps)
    torch.cuda.synchronize(dev)
    engine.PROFILE = None
    peak = PEAK_BF16_MFMA_TFLOPS if dtype == "bf16" else PEAK_F32_MFMA_TFLOPS
    gemm_roof, lstm_roof, gemm_ms, lstm_ms, _ = text_rooflines(prof, psteps, workload, dtype, B, T, H, peak)
    dom = gemm_roof if gemm_ms >= lstm_ms else lstm_roof
    rec = {"value": round(B * steps / dt, 2), "unit": "seq/s", "ms_per_step": round(1e3 * dt / steps, 4), "dtype": dtype, "steps": steps,
           "workload": "%s LSTM-VAE aggressive inner step, B=%d, T=%d, V=%d, ni=%d, H=%d, nz=%d%s" % (
               "yahoo" if stress else workload, B, T, V, ni, H, nz, ", one fixed-K=%d loop (stress)" % steps if stress else ""),
           "dominant_group": {"bound": dom["bound"], "frac": dom["frac"], "achieved": dom["achieved"], "unit": dom["unit"],
                              "ms_per_step": dom["ms_per_step"]},
           "gemm_tflops": gemm_roof["achieved"], "lstm_us_per_timestep": lstm_roof.get("us_per_timestep"),
           "lstm_ladder_rung": max(engine.persist_rung(tr.enc), engine.persist_rung(tr.dec))}
    rec["groups_measured"] = "separate untimed pass of %d steps behind the timed region" % psteps
    if dtype == "bf16":
        rec["parity_contract"] = {"elbo_rel": 1e-4, "rec_rel": 1e-4, "kl_rel": 1e-4,
                                  "test": {"yelp": "tests/test_gpu_parity.py (text_yelp_wide_seeded, reference-generated fixture)",
                                           "stress": "tests/test_gpu_parity.py::test_stress_config_at_full_size (the oracle's whole step on all 128 sequences)",
                                           "yahoo": "tests/test_gpu_parity.py::test_bf16_headline_path_at_headline_shape"}.get(workload)}
    if decoder_grads != "full":
        rec["decoder_grads"] = decoder_grads
    if cpu_leg:
        from oracle import text_vae_oracle as O            # the checker, here only as the timed CPU baseline
        Pc = {k: v.detach().cpu() for k, v in vae.state_dict().items() if k in O.ALL_KEYS}
        eps, m_in, m_out = O.draw_noise(B, T, ni, H, nz, seed=1)
        rec["cpu_baseline"] = cpu_text_leg(O, Pc, pool[0].cpu(), 0.1, eps, m_in, m_out, B, T, os.cpu_count() or 1, counts=(32,))
        rec["speedup_vs_cpu_baseline"] = round(rec["value"] / rec["cpu_baseline"]["value"], 1)
    del tr, vae, pool
    torch.cuda.empty_cache()
    return rec


def side_run_mixed_shapes(dev, steps=200, fixed_tokens_per_s=None):
    """The loop the reference actually runs: data/text_data.py:219-255 hands it batches of MANY lengths (sentences are bucketed by
    length; every length's last batch is a tail with B < 32), not 64 batches of one shape.  Pool: 96 batches whose lengths follow a
    Yahoo-like histogram (log-normal around 78 tokens, clipped to 20..200), six of them tails (B in 5..31); eager mode, one pass over
    the pool as warm-up (every shape's workspace gets built), then `steps` steps drawn as text.py:389 draws them.  Reports seq/s,
    tokens/s (the comparable figure: steps differ in length), the engines' workspace-cache hit rate, what the caching allocator had
    to get from the device in steady state, and the ladder rung."""
    from vae_lagging_encoder_amd import engine
    from vae_lagging_encoder_amd.factory import build_text_vae, synthetic_batch
    from vae_lagging_encoder_amd.trainer import AggressiveTextTrainer
    cfg = WORKLOADS["yahoo"]
    V, ni, H, nz = (cfg[k] for k in ("V", "ni", "H", "nz"))
    rs = np.random.RandomState(20250928)
    lens = np.clip(np.exp(rs.normal(np.log(78.0), 0.55, size=96)).astype(int), 20, 200)
    bs = [32] * 96
    for j in rs.choice(96, 6, replace=False):
        bs[j] = int(rs.randint(5, 32))
    vae = build_text_vae(V, ni, H, nz, dev, seed=783435)
    tr = AggressiveTextTrainer(vae, lr=1.0, clip=5.0, seed=783435, precision="bf16")
    pool = [synthetic_batch(b, int(t), V, seed=9000 + i).to(dev) for i, (b, t) in enumerate(zip(bs, lens))]
    tr.prepare_batches(pool)
    for x in pool:                                   # warm-up: one pass over every shape
        tr.step(x, 0.1)
    tr.commit()
    torch.cuda.synchronize(dev)
    c0 = [(e.wsc.hits, e.wsc.misses, e.wsc.evictions) for e in (tr.enc, tr.dec)]
    m0 = torch.cuda.memory_stats(dev)
    picks = [int(rs.randint(0, len(pool))) for _ in range(steps)]
    t0 = time.perf_counter()
    for i, j in enumerate(picks):
        tr.step(pool[j], 0.1)
        if (i + 1) % 15 == 0:
            tr.read_stats()                           # the loop's per-window host read (text.py:393)
    rung = tr.commit()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    m1 = torch.cuda.memory_stats(dev)
    c1 = [(e.wsc.hits, e.wsc.misses, e.wsc.evictions) for e in (tr.enc, tr.dec)]
    hits = sum(b[0] - a[0] for a, b in zip(c0, c1)); miss = sum(b[1] - a[1] for a, b in zip(c0, c1))
    seqs = sum(pool[j].shape[0] for j in picks)
    toks = sum(pool[j].shape[0] * pool[j].shape[1] for j in picks)
    rec = {"value": round(seqs / dt, 2), "unit": "seq/s", "tokens_per_s": round(toks / dt, 1), "ms_per_step": round(1e3 * dt / steps, 4),
           "steps": steps, "dtype": "bf16",
           "workload": "yahoo dims, 96 pool batches of %d distinct shapes (T %d..%d, mean %.0f; %d tails with B < 32), eager" % (
               len({tuple(x.shape) for x in pool}), int(lens.min()), int(lens.max()), float(lens.mean()), sum(1 for b in bs if b < 32)),
           "workspace_cache": {"hit_rate": round(hits / max(1, hits + miss), 4), "misses": miss,
                               "evictions": sum(b[2] - a[2] for a, b in zip(c0, c1)),
                               "resident_GB": round(sum(e.wsc.total for e in (tr.enc, tr.dec)) / 1e9, 1)},
           "steady_state_device_allocations_MB": round((m1["reserved_bytes.all.allocated"] - m0["reserved_bytes.all.allocated"]) / 1e6, 1),
           "steady_state_allocator_requests": int(m1["allocation.all.allocated"] - m0["allocation.all.allocated"]),
           "lstm_ladder_rung": rung, "recoveries": tr.recoveries}
    if fixed_tokens_per_s:
        rec["tokens_per_s_vs_fixed_shape"] = round(rec["tokens_per_s"] / fixed_tokens_per_s, 3)      # vs the headline's T = 200
    # control: the same trainer on a FIXED shape at the pool's mean length -- a step's cost per token rises as T falls (the
    # vocabulary-sized products shrink, launch and per-recurrence fixed costs do not), so this, not the T = 200 headline, is
    # what tells how much the MIXING of shapes costs
    Tm = int(round(float(lens.mean())))
    cpool = [synthetic_batch(32, Tm, V, seed=9500 + i).to(dev) for i in range(8)]
    tr.prepare_batches(cpool)
    for x in cpool:
        tr.step(x, 0.1)
    tr.commit()
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for i in range(60):
        tr.step(cpool[int(rs.randint(0, len(cpool)))], 0.1)
        if (i + 1) % 15 == 0:
            tr.read_stats()
    tr.commit()
    torch.cuda.synchronize(dev)
    d1 = time.perf_counter() - t1
    rec["fixed_shape_control"] = {"T": Tm, "B": 32, "tokens_per_s": round(60 * 32 * Tm / d1, 1), "ms_per_step": round(1e3 * d1 / 60, 4)}
    rec["tokens_per_s_vs_fixed_shape_at_mean_length"] = round(rec["tokens_per_s"] / rec["fixed_shape_control"]["tokens_per_s"], 3)
    del tr, vae, pool, cpool
    torch.cuda.empty_cache()
    return rec


def measure_dropin(V, ni, H, nz, B, pool, kl_weight, dev, steps=8, warmup=3):
    """Throughput THROUGH THE REFERENCE'S OWN BOUNDARY: the literal text.py:373-387 sequence on the drop-in modules --
    zero_grad x2, vae.loss, the per-iteration host read of text.py:381, loss.mean().backward(), clip_grad_norm_ over encoder +
    decoder parameters, enc_optimizer.step() -- instead of the fused driver every other number of this line is quoted on.
    Three variants: the default f32 arithmetic with torch's clip / SGD; vae.set_precision("bf16") with torch's; and bf16 with
    vae_lagging_encoder_amd.optim's clip_grad_norm_ / SGD (streaming launches over the flat buffers).  In all of them the backward
    leaves the .grad tensors as views of the flat gradient buffers (no clones)."""
    from vae_lagging_encoder_amd import optim as lvo
    from vae_lagging_encoder_amd.factory import build_text_vae
    res = {}
    rs = np.random.RandomState(783435)
    for label, prec, lv in (("f32_torch_optim", "f32", False), ("bf16_torch_optim", "bf16", False), ("bf16_lvae_optim", "bf16", True)):
        vae = build_text_vae(V, ni, H, nz, dev, seed=783435).set_precision(prec)
        SGD = lvo.SGD if lv else torch.optim.SGD
        clipfn = lvo.clip_grad_norm_ if lv else torch.nn.utils.clip_grad_norm_
        enc_opt, dec_opt = SGD(vae.encoder.parameters(), lr=1.0, momentum=0), SGD(vae.decoder.parameters(), lr=1.0, momentum=0)
        burn = 0.0

        def body():
            nonlocal burn
            x = pool[int(rs.randint(0, len(pool)))]
            enc_opt.zero_grad()
            dec_opt.zero_grad()
            loss, loss_rc, loss_kl = vae.loss(x, kl_weight, nsamples=1)
            burn += loss.sum().item()                    # text.py:381
            loss = loss.mean(dim=-1)
            loss.backward()
            clipfn(vae.parameters(), 5.0)
            enc_opt.step()
        n = steps if prec != "f32" else max(3, steps // 2)
        for _ in range(warmup if prec != "f32" else 2):
            body()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            body()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        res[label] = {"value": round(B * n / dt, 2), "unit": "seq/s", "ms_per_step": round(1e3 * dt / n, 4), "steps": n}
        del vae, enc_opt, dec_opt
        torch.cuda.empty_cache()
    res["note"] = ("text.py:373-387 verbatim on the drop-in modules (autograd Functions over the HIP engines, one host read per iteration as "
                   "text.py:381 has); gradients delivered as views of the flat buffers")
    return res


# ---------------------------------------------------------------------------------------------------------------------------
# Launching N > 1 ranks: a supervisor with a deadline and a ladder of exchange schedules.
#
# The first execution of this path over RCCL is the round driver's scaling run, and the driver gets one shot per N: a rank that
# hangs (or dies) must not turn into "no JSON line".  Every multi-rank run is therefore SUPERVISED: the rank processes are
# children of a supervisor that gives each attempt a deadline, asks hung ranks for their Python stacks (SIGUSR1 ->
# faulthandler), stops them, and starts the next attempt on a more conservative schedule; when every attempt has failed the
# supervisor itself prints ONE JSON line {"error": ..., "n_gpus": N, "attempts": [...]} and exits non-zero.
#   plain `python bench.py --gpus N`:  one supervisor (this process) owns all N ranks;
#   under torch.distributed.run:       each torchrun child supervises ITS OWN rank (the attempts line up by wall clock: every
#                                      supervisor gives attempt k the window [t0 + k D, t0 + (k+1) D), t0 = its own start).
LAUNCH_LADDER = [
    ("default", {},
     "persistent LSTM launches; embedding bucket and decoder exchange issued from inside the encoder backward (asynchronous)"),
    ("conservative-exchange", {"LVAE_DP_CONSERVATIVE": "1"},
     "persistent LSTM launches; every collective issued after the backward, device-synchronised on both sides (no kernel ever "
     "runs beside an RCCL kernel)"),
    ("conservative-exchange+step-kernels", {"LVAE_DP_CONSERVATIVE": "1", "LVAE_BENCH_PERSISTENT": "0"},
     "launch-per-timestep LSTM kernels; every collective issued after the backward, device-synchronised on both sides"),
]


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _done_marker(port):
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), "lvae_bench_%s_%s.done" % (port, os.environ.get("TORCHELASTIC_RUN_ID", "self")))


def supervise(n, my_ranks, base_env, deadline_s, attempts, port_for_attempt, marker, print_error, t0=None):
    """Run up to `attempts` rungs of LAUNCH_LADDER.  my_ranks: the ranks this supervisor owns.  Returns the exit code.
    An attempt succeeds when rank 0 has written the marker file (its JSON line is out) or all owned ranks returned 0."""
    import signal
    import subprocess
    t0 = time.time() if t0 is None else t0
    history = []

    def _term(signum, frame):                 # the launcher above us stops the job: take the ranks down with us (finally: below)
        raise SystemExit(128 + signum)
    try:
        signal.signal(signal.SIGTERM, _term)
    except ValueError:                          # not the main thread (tests)
        pass
    rc = 1
    for k in range(attempts):
        name, extra, what = LAUNCH_LADDER[min(k, len(LAUNCH_LADDER) - 1)]
        window_end = t0 + (k + 1) * deadline_s
        env = dict(base_env)
        env.update(extra)
        env.update(port_for_attempt(k))
        env.update(LVAE_BENCH_WORKER="1", LVAE_BENCH_ATTEMPT=str(k), LVAE_BENCH_SCHEDULE=name, LVAE_BENCH_DONE=marker,
                   LVAE_BENCH_PRIOR=json.dumps(history))
        # a collective timeout inside the ranks well before the supervisor's own deadline: the ranks then fail with a message
        env.setdefault("LVAE_DIST_TIMEOUT", str(int(max(30, min(180, deadline_s / 2)))))
        if k > 0:
            print("bench.py: attempt %d of %d on schedule '%s' (%s)" % (k + 1, attempts, name, what), file=sys.stderr, flush=True)
        procs = {}
        for r in my_ranks:
            e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
            procs[r] = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e)
        outcome, codes = None, {}
        try:
            live = dict(procs)
            while live and outcome is None:
                time.sleep(0.05)
                for r, p in list(live.items()):
                    c = p.poll()
                    if c is None:
                        continue
                    del live[r]
                    codes[r] = c
                    if c != 0:
                        rc = c if c > 0 else 128 - c
                        outcome = "rank %d exited with %d" % (r, c)
                if outcome is None and live and time.time() > window_end:
                    outcome = "timeout: ranks %s still running after %.0f s" % (sorted(live), deadline_s)
                    rc = 124
            if outcome is None:
                return 0                                        # every owned rank returned 0
            if os.path.exists(marker):
                # the JSON line is out; what failed / hung afterwards is process teardown
                print("bench.py: %s after the result line was printed; ignoring" % outcome, file=sys.stderr, flush=True)
                return 0
            print("bench.py: attempt %d ('%s') failed: %s; stopping its ranks" % (k + 1, name, outcome), file=sys.stderr, flush=True)
            for p in live.values():                             # where is it stuck?  (faulthandler in the ranks dumps all threads)
                try:
                    p.send_signal(signal.SIGUSR1)
                except OSError:
                    pass
            if live:
                time.sleep(1.5)
            for p in live.values():
                p.terminate()
            t_kill = time.time() + 5
            while any(p.poll() is None for p in live.values()) and time.time() < t_kill:
                time.sleep(0.05)
        finally:
            for p in procs.values():
                if p.poll() is None:
                    p.kill()
        history.append({"schedule": name, "outcome": outcome, "exit_codes": {str(r): c for r, c in sorted(codes.items())},
                        "ranks_alive_at_stop": sorted(set(procs) - set(codes))})
        if k + 1 < attempts and len(my_ranks) < n:
            # a supervisor per rank (torchrun): the next attempt starts when the window ends on every supervisor's clock
            while time.time() < window_end:
                if os.path.exists(marker):
                    return 0
                time.sleep(0.2)
    if os.path.exists(marker):
        return 0
    if not print_error and len(my_ranks) < n:
        # a supervisor per rank (torchrun): the launcher stops everybody as soon as ONE child reports failure -- rank 0's supervisor
        # must get its error line out first
        time.sleep(3.0)
    if print_error:
        print(json.dumps({"error": "every launch attempt failed", "metric": "aggressive-loop seqs/sec", "value": None, "unit": "seq/s",
                          "n_gpus": n, "attempts": history,
                          "ranks_alive": history[-1]["ranks_alive_at_stop"] if history else []}), flush=True)
    return rc if rc != 0 else 1


def self_launch(n, deadline_s=420.0, attempts=3):
    """`python bench.py --gpus N` without a launcher around it: start N copies of this command, one rank per GPU (rank r on
    cuda:r), with the environment `torch.distributed.run` would give them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 /
    MASTER_PORT), RCCL ("nccl") as the backend, under `supervise` (deadline per attempt, ladder of schedules, a JSON error line
    when all fail).  Rank 0 prints the one JSON line on the inherited stdout.  Returns the exit code: 0 on success, else the code
    of the rank that failed last (124 for a timeout).
    A box with fewer than N GPUs cannot run RCCL with N ranks (one communicator rank per device): the ranks then SHARE the devices
    (rank r on cuda:r % device_count) and exchange over gloo -- a functional check of the data-parallel path, said loudly on stderr and
    in the line's `config.dp_transport`; its seq/s is not a scaling number (one attempt: the ladder is about RCCL)."""
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    emu = os.environ.get("LVAE_BENCH_EMU") == "1"            # test hook, see main()
    if ndev == 0 and not emu:
        print("bench.py: no GPU visible; --gpus %d needs MI355X GPUs (no CPU fallback)" % n, file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", LVAE_BENCH_LAUNCHER="self")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # the host driver supports dmabuf IPC only (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    if emu:
        env["LVAE_DIST_BACKEND"] = "gloo"
        env["LVAE_SHARED_GPU"] = "%d ranks on the CPU emulator (test hook)" % n
        env["OMP_NUM_THREADS"] = "1"
        attempts = 1
    elif ndev < n:
        env["LVAE_DIST_BACKEND"] = "gloo"
        env["LVAE_SHARED_GPU"] = "%d ranks on %d GPU%s" % (n, ndev, "" if ndev == 1 else "s")
        attempts = 1
        print("bench.py: --gpus %d on a box with %d GPU(s): RCCL needs one device per rank, so the ranks share the device(s) and "
              "exchange over gloo -- a functional check of the data-parallel path, NOT a scaling measurement" % (n, ndev), file=sys.stderr)
    else:
        print("bench.py: launching %d ranks, one per GPU (%d visible), RCCL; %d attempt(s) of %.0f s" % (n, ndev, attempts, deadline_s),
              file=sys.stderr, flush=True)
    marker = _done_marker(_free_port())
    if os.path.exists(marker):
        os.remove(marker)
    try:
        return supervise(n, list(range(n)), env, deadline_s, attempts, lambda k: {"MASTER_PORT": str(_free_port())}, marker, True)
    finally:
        if os.path.exists(marker):
            os.remove(marker)


def supervise_own_rank(n, deadline_s=420.0, attempts=3):
    """Under torch.distributed.run (WORLD_SIZE / RANK set by the launcher): this process stays the launcher's child and runs ITS rank
    as a grandchild under `supervise`.  Attempt 0 uses the launcher's own rendezvous (MASTER_PORT, the agent's store) -- exactly a
    plain torchrun start; later attempts rendezvous on MASTER_PORT + k with rank 0 hosting the store."""
    rank = int(os.environ["RANK"])
    base_port = int(os.environ.get("MASTER_PORT", "29500"))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("LVAE_BENCH_LAUNCHER", "torch.distributed.run")
    marker = _done_marker(base_port)
    if rank == 0 and os.path.exists(marker):
        os.remove(marker)

    def port_for_attempt(k):
        if k == 0:
            return {}
        return {"MASTER_PORT": str(base_port + k), "TORCHELASTIC_USE_AGENT_STORE": "False"}
    try:
        return supervise(n, [rank], env, deadline_s, attempts, port_for_attempt, marker, rank == 0)
    finally:
        if rank == 0 and os.path.exists(marker):
            time.sleep(1.0)                # the other supervisors look at it when their rank's exit was not clean
            os.remove(marker)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed inner-loop bodies (default 20; stress: 50 = one fixed-K loop)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="yahoo", choices=sorted(WORKLOADS))
    ap.add_argument("--graph", type=int, default=0, help="replay the step as captured hipGraphs (no per-kernel events)")
    ap.add_argument("--dtype", default="bf16", choices=["f32", "bf16", "bf16x3"],
                    help="arithmetic of the large GEMMs: f32 = exact-f32 MFMA (parity path), bf16 = bf16 MFMA, f32 accumulate; bf16x3 "
                         "(--workload omniglot only) = the decoder's direct convolutions with every operand split into two bf16 numbers, "
                         "three bf16 MFMAs per product, f32 accumulate (f32-like results: holds the f32 fixtures' bounds), the rest exact f32")
    ap.add_argument("--encoder-forward", default="bf16", choices=["bf16", "f32"],
                    help="--dtype bf16 only: f32 runs the ENCODER'S FORWARD (input projection + recurrence) in exact f32 -- the KL depends "
                         "on that forward's last state alone, so ELBO, reconstruction NLL and KL all meet north_star's 1e-4 while every "
                         "gradient product and the decoder stay on the bf16 pipe; the default line carries this configuration as "
                         "`kl_exact_path`")
    ap.add_argument("--forward-operands", default="f16", choices=["f16", "bf16"],
                    help="--dtype bf16: number format of the ENCODER FORWARD's matrix-pipe operands: f16 (default: IEEE binary16, same "
                         "instructions and rates, 8x finer weight rounding -> KL within 1e-4) or bf16 (the arithmetic of rounds 1-4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-runs", action="store_true", help="skip the f32 parity run and the side runs of the other BASELINE.json configurations (profiling passes: only the timed arithmetic runs)")
    ap.add_argument("--persistent", type=int, default=1, help="forward LSTM recurrences as one persistent launch (bf16 path)")
    ap.add_argument("--overlap", default="auto", choices=["auto", "on", "off"],
                    help="decoder weight-gradient GEMMs on a side stream under the BPTT chains (auto: on for f32, off for bf16)")
    ap.add_argument("--dp-mode", default="strict", choices=["strict", "encoder_only"],
                    help="data-parallel gradient exchange: strict = encoder + decoder buffers (reference clip norm), "
                         "encoder_only = encoder buffer only (documented deviation)")
    ap.add_argument("--dp-payload", default="auto", choices=["auto", "f32", "bf16"],
                    help="wire format of the data-parallel gradient exchange (auto = bf16 for --dtype bf16, exact f32 mean for --dtype "
                         "f32; bf16 halves the bytes)")
    ap.add_argument("--dp-embedding", default="dense", choices=["auto", "rows", "dense"],
                    help="data parallel: the encoder's embedding gradient inside the dense all-reduce (default: the form the scaling "
                         "runs have always used) or as a row list (all-gather of the touched rows); auto = rows only where that is "
                         "fewer bytes on the wire (capacity * world < 2 V)")
    ap.add_argument("--micro-batches", type=int, default=1,
                    help="gradient accumulation: the step's batch as m row slices, slice i's gradient exchange under slice i+1's "
                         "computation (costs m times the recurrences at B=32: they are latency-bound; see DESIGN.md section 6)")
    ap.add_argument("--decoder-grads", default="full", choices=["full", "norm"],
                    help="full (default): every .grad left as clip_grad_norm_ leaves it; norm: in the encoder-only inner step the "
                         "decoder's two vocabulary-sized gradient tensors are reduced to their sums of squares in their producers and "
                         "never written (text.py:383-387 uses them for the clip norm alone); single GPU")
    ap.add_argument("--pool", type=int, default=None)
    ap.add_argument("--vendor-leg", default=None, choices=["yahoo", "yelp", "stress", "omniglot"], help=argparse.SUPPRESS)
    ap.add_argument("--vendor-autocast", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--no-vendor-baseline", action="store_true", help="skip the PyTorch-ROCm library yardstick (vendor_stack_baseline)")
    ap.add_argument("--launch-timeout", type=float, default=420.0,
                    help="--gpus N > 1: seconds one launch attempt may take before its ranks are stopped (stacks dumped) and the "
                         "next rung of the schedule ladder is tried; after the last the supervisor prints a JSON error line")
    ap.add_argument("--launch-attempts", type=int, default=3, help="--gpus N > 1: rungs of LAUNCH_LADDER to try (RCCL runs)")
    ap.add_argument("--force-dp", action="store_true",
                    help="--gpus 1: run the data-parallel exchange anyway, on a one-rank RCCL process group -- every collective of the "
                         "schedule executes on the real backend (all a one-GPU box can reach of RCCL); reported as `rccl_single_rank`")
    ap.add_argument("--tokens", default="uniform", choices=["uniform", "zipf"],
                    help="distribution of the synthetic token ids: uniform (SURVEY.md 8d, the default) or Zipf-like (natural text: frequent "
                         "tokens repeat hundreds of times per batch, which the embedding backward's sort / scatter feel)")
    args = ap.parse_args()
    if args.vendor_leg:
        return vendor_leg_main(args.vendor_leg, args.vendor_autocast)
    if os.environ.get("LVAE_BENCH_WATCHDOG"):
        # diagnostics: dump every thread's Python stack to stderr after this many seconds (and again every period) without exiting
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["LVAE_BENCH_WATCHDOG"]), repeat=True, exit=False)
    if args.gpus > 1 and "LVAE_BENCH_WORKER" not in os.environ:
        # this process becomes the SUPERVISOR of the rank processes (deadline per attempt, ladder of exchange schedules, a JSON error
        # line when every attempt fails) and prints nothing else itself
        if "WORLD_SIZE" not in os.environ:
            # plain `python bench.py --gpus N`: supervisor of all N ranks (one per GPU)
            sys.exit(self_launch(args.gpus, args.launch_timeout, args.launch_attempts))
        # under torch.distributed.run: supervisor of this launcher child's own rank
        sys.exit(supervise_own_rank(args.gpus, args.launch_timeout,
                                    1 if os.environ.get("LVAE_DIST_BACKEND") == "gloo" else args.launch_attempts))
    protect_stdout()
    if os.environ.get("LVAE_BENCH_WORKER"):
        import faulthandler
        import signal
        faulthandler.register(signal.SIGUSR1, all_threads=True)      # the supervisor asks a stuck rank where it is
    stress = args.workload == "stress"
    if args.steps is None:
        args.steps = 50 if stress else 20
    if args.pool is None:
        args.pool = 16 if stress else 64

    from vae_lagging_encoder_amd import dist as lvdist
    from vae_lagging_encoder_amd import engine
    from vae_lagging_encoder_amd.trainer import AggressiveTextTrainer
    from vae_lagging_encoder_amd.factory import build_text_vae as build_vae, synthetic_batch

    rank, local, world = lvdist.init_from_env(banner=args.gpus > 1 or args.force_dp, force=args.force_dp)
    if os.environ.get("LVAE_BENCH_PERSISTENT") == "0":      # a rung of LAUNCH_LADDER
        args.persistent = 0
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU, or plain `python bench.py --gpus N`)" % (args.gpus, world))
    emu = os.environ.get("LVAE_BENCH_EMU") == "1"
    if emu:
        # TEST HOOK (tests/test_bench_launch.py): the same .hip sources compiled for the host by tests/emu, on device "cpu", so that
        # the multi-rank control flow of THIS file (launch ladder, exchange, loop exit, dp_breakdown) runs on the CPU-only CI box
        # over gloo.  Not a fallback: without this variable a missing GPU is the loud exit below; its numbers mean nothing.
        import ctypes
        sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
        from build_emu import build_emu
        import install as emu_install
        from vae_lagging_encoder_amd import _lib
        emu_install.install(_lib.bind(ctypes.CDLL(build_emu())))
        torch.set_num_threads(1)
        dev = torch.device("cpu")
        args.no_side_runs = args.no_cpu_baseline = True
    else:
        assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
        dev = torch.device("cuda", local % torch.cuda.device_count())
        torch.cuda.set_device(dev)

    def dsync():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
    cfg = WORKLOADS[args.workload]
    if args.workload == "omniglot":
        return bench_omniglot(args, dev, rank, world)
    if args.dtype == "bf16x3":
        raise SystemExit("--dtype bf16x3 names the split-bf16 direct convolutions of --workload omniglot; the text workloads take f32 or bf16")
    V, ni, H, nz, B, T = (cfg[k] for k in ("V", "ni", "H", "nz", "B", "T"))

    # reference init (text.py:265-266) from the reference's default seed (text.py:54,73); same replica on every rank
    vae = build_vae(V, ni, H, nz, dev, seed=783435)
    sync = lvdist.GradSync(mode=args.dp_mode, payload=args.dp_payload, force=args.force_dp) if (world > 1 or args.force_dp) else None
    dp = sync is not None
    tr = AggressiveTextTrainer(vae, lr=1.0, clip=5.0, seed=783435, grad_sync=sync, use_graph=bool(args.graph),
                               precision=args.dtype, micro_batches=args.micro_batches, decoder_grads=args.decoder_grads,
                               encoder_forward=args.encoder_forward if args.dtype == "bf16" else None, forward_operands=args.forward_operands)
    if args.overlap != "auto":
        tr.dec.overlap = (args.overlap == "on")
    tr.enc.persistent = tr.dec.persistent = bool(args.persistent)
    if os.environ.get("LVAE_SHARED_GPU") and args.persistent:
        # ranks sharing a device (the launcher's functional-check mode): every rank's persistent launch wants all 256 CUs resident at
        # once, two of them beside each other starve each other into their bounded spins (seconds per timeout, then the ladder).  The
        # check is of the data-parallel logic, so it runs on the launch-per-timestep kernels from the start
        tr.enc.persistent = tr.dec.persistent = False
        if rank == 0:
            print("bench.py: ranks share a device: persistent LSTM launches off (%s)" % engine.PERSIST_RUNGS[2], file=sys.stderr)
    if sync is not None:
        args.dp_payload = sync.payload                      # "auto" resolved by the trainer
    pool = [synthetic_batch(B, T, V, seed=1000 * rank + i, dist=args.tokens).to(dev) for i in range(args.pool)]
    rs = np.random.RandomState(783435)
    kl_weight = 0.1                                         # text.py default kl_start
    # batch construction: the per-batch sorted token lists of the embedding backward are built with the batches (a function of
    # the token ids alone, computed once per batch as train_data_batch itself is; trainer.prepare_batches) -- the aggressive loop
    # then meets each batch many times (text.py:389)
    tr.prepare_batches(pool)
    if sync is not None and args.dp_embedding != "dense" and args.micro_batches == 1 and not args.graph:
        tr.enable_row_exchange(pool, mode=args.dp_embedding)

    def one_step():
        tr.step(pool[int(rs.randint(0, len(pool)))], kl_weight)

    prof = None
    window_stats = []

    def timed_region():
        if stress:
            # BASELINE.json configs[4]: fixed K inner steps, no data-dependent exit (text.py:371-400 with the break removed)
            n = tr.inner_loop(pool, pool[0], kl_weight, np_rng=rs, max_iter=10 ** 9, fixed_k=args.steps)
            assert n == args.steps
        else:
            for i in range(args.steps):
                # the live HIP events of the dominant group on every EVENT_EVERY-th step only: an event record is ~5 us of queue
                # idle (8 records per step around the four recurrences), and a quarter of the launches estimates their average
                if prof is not None:
                    engine.PROFILE = prof if i % EVENT_EVERY == 0 else None
                if sync is not None:
                    # ... and the exchange phases' events (8-10 records per step) on those steps only: every record is ~5 us of queue
                    # idle, i.e. ~45 us of a data-parallel step if taken everywhere (breakdown() averages over the bracketed steps)
                    sync.profile = (i % EVENT_EVERY == 0)
                one_step()
                if (i + 1) % 15 == 0:
                    window_stats.append(tr.read_stats())       # the loop's host read per 15-iteration window (text.py:393-396)
            engine.PROFILE = prof

    def warm_up():
        for _ in range(args.warmup):
            one_step()
        dsync()

    rung_before = max(engine.persist_rung(tr.enc), engine.persist_rung(tr.dec))
    warm_up()
    # The persistent LSTM launches assume that the whole GPU is resident at once and (rung 0) that a group's workgroups share an
    # XCD.  A hand-off timeout voids the step on the device; the trainer notices it at this host read, moves down its fallback
    # ladder (write-through hand-off, then the launch-per-timestep kernels) and replays -- data parallel on all ranks alike.
    rung = tr.commit()
    if rung > rung_before:
        print("bench: persistent LSTM launches timed out during warm-up; running on ladder rung %d (%s)"
              % (rung, engine.PERSIST_RUNGS[rung]), file=sys.stderr)
        warm_up()
        tr.commit()
    if dp:
        torch.distributed.barrier()
    if not args.graph and not emu:
        # the dominant kernel group (the four LSTM recurrences) is bracketed live inside the timed region; the GEMM group's events
        # (22 records per step, ~5 us of queue idle each) are taken in a separate untimed pass right after it
        prof = {}
        engine.PROFILE = prof
        engine.PROFILE_PREFIX = "lstm_"
    if sync is not None:
        sync.profile = stress                                # HIP events around the phases of the gradient exchange (text loop: every EVENT_EVERY-th step)
    tr.reset_stats()
    dsync()
    t0 = time.perf_counter()
    timed_region()
    dsync()
    if dp:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    engine.PROFILE = None
    engine.PROFILE_PREFIX = None
    stats_timed = tr.read_stats()                            # report sums of the timed region alone (the passes below add steps)
    prof_gemm, gemm_steps = None, 0
    if prof is not None:
        prof_gemm, gemm_steps = {}, min(args.steps, 5 if not stress else 3)
        engine.PROFILE, engine.PROFILE_PREFIX = prof_gemm, "gemm_"
        for _ in range(gemm_steps):
            one_step()
        dsync()
        tr.commit()
        engine.PROFILE = engine.PROFILE_PREFIX = None
    long_run = None
    if world == 1 and not dp and not args.graph and not stress and not args.no_side_runs and not emu:
        # the same region again at 10x the length, same process: 20 steps x 3.2 ms is 63 ms of timed GPU work next to a +-4 %
        # box-to-box spread -- this says whether the headline is a short-sample artefact
        n_long = 200
        dsync()
        tl0 = time.perf_counter()
        for i in range(n_long):
            one_step()
            if (i + 1) % 15 == 0:
                tr.read_stats()
        tr.commit()
        dsync()
        dtl = time.perf_counter() - tl0
        long_run = {"value": round(B * n_long / dtl, 2), "unit": "seq/s", "ms_per_step": round(1e3 * dtl / n_long, 4), "steps": n_long,
                    "note": "second pass of 200 steps in the same process (no HIP events; one host read per 15 steps as text.py:393 has)"}
    dp_breakdown = None
    if dp:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt_local, dt = dt, float(tmax.item())
        # per rank: what the compute stream spent waiting inside each phase of the exchange (exposed communication), and the
        # rank's own wall time of the timed region -- gathered so that rank 0 can print every rank's numbers
        bd = sync.breakdown()
        names = ["encoder_allreduce_issue", "decoder_reduce_scatter_wait", "decoder_allreduce_wait", "scalar_allreduce",
                 "encoder_allreduce_wait"]
        my_rung = max(engine.persist_rung(tr.enc), engine.persist_rung(tr.dec)) if args.dtype == "bf16" else -1
        mine = torch.tensor([bd.get(k, 0.0) for k in names] + [1e3 * dt_local / args.steps, float(my_rung)], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allr, mine)
        # replicas must hold the same bits after the run (every rank applied the same mean gradient): an integer checksum of the
        # encoder's parameter buffer, compared across ranks
        chk = tr.enc.flat.data.view(torch.int32).to(torch.int64).sum().reshape(1)
        chks = [torch.zeros_like(chk) for _ in range(world)]
        torch.distributed.all_gather(chks, chk)
        replicas_identical = all(int(c.item()) == int(chks[0].item()) for c in chks)
        rungs = [int(t[-1].item()) for t in allr]
        rows = [[round(float(v), 4) for v in t.tolist()[:-1]] for t in allr]
        exposed = [round(sum(r[:-1]), 4) for r in rows]
        dp_breakdown = {
            "unit": "ms per step on the compute stream (HIP events): time the step WAITED in each phase; hidden communication does not show",
            "phases": names + ["rank_ms_per_step"], "per_rank": rows, "exposed_ms_per_step_per_rank": exposed,
            "exposed_ms_per_step_max": max(exposed), "payload": sync.payload, "mode": args.dp_mode,
            "bytes_sent_per_rank_per_step": int(sync.bytes_per_step(tr.enc.flat, tr.dec.flat)) * args.micro_batches,
            "bytes_on_wire": sync.bytes_on_wire(tr.enc.flat, tr.dec.flat, "encoder", micro_batches=args.micro_batches,
                                                n_emb=(tr.enc.flat.offsets[tr.enc.flat.names[1]] // 1024 * 1024) if not args.graph else 0),
            "micro_batches": args.micro_batches,
            # every rank's rung of the persistent recurrences' fallback ladder (0 = persistent launches with the XCD-local hand-off;
            # a collective taking CUs from a persistent launch would show here as a rung > 0 on some rank)
            "lstm_ladder_rung_per_rank": rungs if args.dtype == "bf16" else None,
            "backend": torch.distributed.get_backend(),
            "replicas_identical": replicas_identical,
            "schedule": os.environ.get("LVAE_BENCH_SCHEDULE", "default") + (" (conservative)" if sync.conservative else ""),
            "encoder_bucket": "embedding gradient issued from inside the encoder backward (under dW_ih / dW_hh)" if not args.graph else "none (hipGraph split)"}
    cold = None
    if world == 1 and not args.graph and not stress and not args.no_side_runs:
        # the same step on batches it has never seen: the (token, row) sort of the embedding backward, which the aggressive loop pays
        # once per batch of the epoch (trainer.prepare_batches / first use) and the headline's pool has cached, inside every step
        fresh = [synthetic_batch(B, T, V, seed=50000 + i, dist=args.tokens).to(dev) for i in range(args.steps)]
        dsync()
        tc0 = time.perf_counter()
        for xb in fresh:
            tr.step(xb, kl_weight)
        tr.commit()
        dsync()
        cold = B * len(fresh) / (time.perf_counter() - tc0)
        del fresh

    if rank != 0:
        return
    stats = stats_timed
    seqs = world * B * args.steps
    value = seqs / dt
    out = {
        "metric": "aggressive-loop seqs/sec", "value": round(value, 2), "unit": "seq/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
        "dtype_detail": ("exact-f32 MFMA products, f32 everywhere" if args.dtype == "f32" else
                         "bf16 MFMA operands with f32 accumulation for every gradient product, the BPTT recurrences and the decoder's forward; "
                         "the ENCODER FORWARD's operands (X, W_ih, W_hh, h hand-off) are %s; master weights, cell state, gate "
                         "records, gradients and reductions f32"
                         % ("IEEE binary16 (f16: same instructions and rates, 11-bit significands -> KL within 1e-4)" if tr.enc.fwd_operands == "f16"
                            and args.encoder_forward != "f32" else ("split-bf16 / exact f32 (--encoder-forward f32)" if args.encoder_forward == "f32" else "bf16"))),
        "data": "synthetic" if args.tokens == "uniform" else "synthetic (Zipf-distributed token ids)",
        "config": {"workload": "%s LSTM-VAE aggressive inner step (fwd+bwd+clip+encoder SGD), B=%d/GPU, T=%d, V=%d, "
                               "ni=%d, H=%d, nz=%d%s" % ("yahoo" if stress else args.workload, B, T, V, ni, H, nz,
                                                          ", fixed K=%d inner steps per loop (stress)" % args.steps if stress else ""),
                   "global_batch": world * B, "seq_len": T, "parallelism": "dp%d" % world,
                   "dp_exchange": ((args.dp_mode + ("/bf16-payload" if args.dp_payload == "bf16" else "")) if world > 1 else None),
                   "dp_transport": (None if world == 1 else
                                    ("gloo, %s: functional check of the data-parallel path, not a scaling measurement" % os.environ["LVAE_SHARED_GPU"]
                                     if os.environ.get("LVAE_SHARED_GPU") else
                                     "%s, one rank per GPU" % ("RCCL over xGMI" if torch.distributed.get_backend() == "nccl" else torch.distributed.get_backend()))),
                   "launcher": os.environ.get("LVAE_BENCH_LAUNCHER", "torch.distributed.run" if world > 1 else "none"),
                   "hipgraph": bool(args.graph),
                   "clip_norm": ("vocabulary-sized tensors' sums of squares emitted by their producers + one pass over the rest"
                                 if tr._fold is not None else "one streaming pass over both flat gradients"),
                   "decoder_grads": args.decoder_grads,
                   "encoder_forward": (args.encoder_forward if args.dtype == "bf16" else "f32"),
                   "encoder_forward_operands": ("binary16 (X, W_ih, W_hh, h hand-off: 11-bit significands on the same matrix-pipe "
                                                "instructions; gradient products and BPTT bf16)" if args.dtype == "bf16" and tr.enc.fwd_operands == "f16"
                                                else ("bf16" if args.dtype == "bf16" else "f32")),
                   "lstm_ladder_rung": engine.PERSIST_RUNGS[max(engine.persist_rung(tr.enc), engine.persist_rung(tr.dec))] if args.dtype == "bf16" else None,
                   "batch_preparation": ("sorted token lists of the embedding backward built once per pool batch, with the batches"
                                         if not args.graph else "none (the captured step sorts inside the graph)")},
    }
    if not stress:
        out["mean_loss_per_seq"] = round(stats["loss_sum"] / (B * args.steps), 4)
        out["host_reads_in_timed_region"] = len(window_stats)     # one read_stats() per 15 steps, as text.py:393 reads its window
    if long_run is not None:
        out["value_200_steps"] = long_run
    if cold is not None:
        out["value_with_token_sort"] = {"value": round(cold, 2), "unit": "seq/s",
                                        "note": "every step on a batch tensor it has not seen: both embedding backwards sort their tokens "
                                                "inside the step (the loop pays that once per batch of an epoch; the headline's 64-batch pool "
                                                "has it cached)"}
    if dp_breakdown is not None:
        out["dp_breakdown"] = dp_breakdown
    if args.force_dp and world == 1:
        out["rccl_single_rank"] = ("one-rank %s process group: every collective of the data-parallel schedule executed on the real "
                                   "backend beside the engines' streams (no peer: not a scaling number)" % torch.distributed.get_backend())
    if os.environ.get("LVAE_BENCH_WORKER"):
        # which rung of the launch ladder produced this line, and what the earlier ones died of
        out["launch"] = {"attempt": int(os.environ.get("LVAE_BENCH_ATTEMPT", "0")) + 1, "schedule": os.environ.get("LVAE_BENCH_SCHEDULE", "default"),
                         "failed_attempts": json.loads(os.environ.get("LVAE_BENCH_PRIOR", "[]"))}
    # what this arithmetic is held to against the reference's CPU path (tests/test_gpu_parity.py)
    out["parity_contract"] = ({"elbo_rel": 1e-4, "rec_rel": 1e-4, "kl_rel": 1e-4, "note": "exact-f32 path: north_star's bound on all three"}
                              if args.dtype == "f32" else
                              {"elbo_rel": 1e-4, "rec_rel": 1e-4, "kl_rel": 1e-4,
                               "note": "bf16 configuration with the encoder's forward in exact f32 (the KL depends on that forward's last "
                                       "state alone): north_star's bound on all three; gradients at the bf16 configuration's bounds"}
                              if args.encoder_forward == "f32" else
                              {"elbo_rel": 1e-4, "rec_rel": 1e-4, "kl_rel": 1e-4,
                               "note": "bf16 configuration with the encoder's forward operands in binary16 (the KL depends on that forward's "
                                       "last state, which the WEIGHTS' rounding moves: 2e-4..5e-4 on bf16 operands, profiles/r05a_kl_ablation.txt): "
                                       "north_star's 1e-4 on all three against the reference fixtures at the Yahoo and Yelp shapes "
                                       "(tests/test_gpu_parity.py); kl_exact_path (split-bf16 + two-pass forward) holds 2e-5"}
                              if tr.enc.fwd_operands == "f16" else
                              {"elbo_rel": 1e-4, "rec_rel": 1e-4, "kl_rel": 1e-3,
                               "note": "bf16 forward operands (rounds 1-4): ELBO and reconstruction NLL within 1e-4, KL 2e-4..5e-4"})
    step_flops = 3 * fwd_flops(V, ni, H, nz, B, T)
    step_s = dt / args.steps
    # whole-step views SURVEY.md 8d prescribes: algorithmic bytes (56 MB/sequence at the Yahoo shape: every parameter read
    # once forward and once backward, every gradient written once and read once, saved activations written and read once,
    # logits never materialised) and algorithmic flops per sequence, against the chip's peaks; and the latency floor of the
    # four dependent recurrences (>= 1.45 us per dependent hand-off, guide price list)
    mb_per_seq = {"yahoo": 56.0, "yelp": 1430.0 / 32, "stress": 3990.0 / 128}.get(args.workload)
    gf_per_seq = step_flops / B / 1e9
    peak_mfma = PEAK_BF16_MFMA_TFLOPS if args.dtype == "bf16" else PEAK_F32_MFMA_TFLOPS
    per_gpu = value / world
    out["whole_step"] = {
        "tflops": round(step_flops / step_s / 1e12, 2), "mfma_frac": round(per_gpu * gf_per_seq / 1e3 / peak_mfma, 4),
        "hbm_GBs_algorithmic": round(per_gpu * mb_per_seq / 1e3, 1) if mb_per_seq else None,
        "hbm_frac": round(per_gpu * mb_per_seq / 1e3 / PEAK_HBM_GBS, 4) if mb_per_seq else None,
        "gflop_per_seq": round(gf_per_seq, 2), "mb_per_seq": mb_per_seq,
        "chain_floor_ms": round((4 * T - 2) * 1.45e-3, 3),
        "note": "neither roofline binds at B=32: the floor is the serial chain of 4T-2 dependent LSTM timesteps"}
    if prof:
        ev_steps = args.steps if stress else len(range(0, args.steps, EVENT_EVERY))      # steps of the timed region that carried events
        lstm_roof, gemm_roof, gemm_ms, lstm_ms, pmc_gb = text_rooflines_split(prof, ev_steps, prof_gemm, gemm_steps, args.workload, args.dtype,
                                                                             B, T, H, peak_mfma)
        lstm_ms *= args.steps / ev_steps                     # per-step arithmetic below: scaled to the whole timed region
        gemm_ms *= args.steps / ev_steps
        lstm_roof["measured"] += ", on %d of its %d steps" % (ev_steps, args.steps)
        if pmc_gb is not None:
            out["whole_step"]["hbm_GB_per_step_pmc"] = pmc_gb
        if gemm_ms >= lstm_ms:
            out["roofline"], out["roofline_secondary"] = gemm_roof, lstm_roof
        else:
            out["roofline"], out["roofline_secondary"] = lstm_roof, gemm_roof
        rest = 1e3 * step_s - (gemm_ms + lstm_ms) / args.steps
        # (the f32 configuration runs its weight-gradient GEMMs on a side stream underneath the recurrences: the two groups overlap
        # and their sum exceeds the step -- there is no "rest" to quote)
        out["rest_ms_per_step"] = round(rest, 4) if rest >= 0 else None
    else:
        out["roofline"] = {"bound": "mfma", "achieved": round(step_flops / step_s / 1e12, 2),
                           "peak": peak_mfma, "unit": "TFLOP/s",
                           "frac": round(step_flops / step_s / 1e12 / peak_mfma, 4),
                           "traffic": None, "note": "whole-step algorithmic flops (graph replay: no per-kernel events)"}

    if world == 1 and args.dtype == "bf16" and args.encoder_forward == "bf16" and not args.graph and not stress and not args.no_side_runs:
        # the bf16 configuration with the encoder's forward in exact f32 (ELBO, rec AND KL within 1e-4 of the reference CPU path:
        # tests/test_gpu_parity.py::test_bf16_with_exact_encoder_forward_holds_all_three_at_1e4), same workload, same process
        tr.enc.exact_forward = ("gx", "rec")
        for _ in range(3):
            one_step()
        tr.commit()
        dsync()
        tk0 = time.perf_counter()
        nk = max(5, args.steps // 2)
        for _ in range(nk):
            one_step()
        dsync()
        dk = time.perf_counter() - tk0
        tr.commit()
        tr.enc.exact_forward = ()
        out["kl_exact_path"] = {"value": round(B * nk / dk, 2), "unit": "seq/s", "ms_per_step": round(1e3 * dk / nk, 4), "steps": nk,
                                "parity_contract": {"elbo_rel": 1e-4, "rec_rel": 1e-4, "kl_rel": 1e-4},
                                "note": "bf16 configuration + encoder forward (input projection, recurrence) in exact f32: `--encoder-forward "
                                        "f32`; ELBO, reconstruction NLL and KL <= 1e-4 vs the reference CPU path at this shape"}
    if world == 1 and args.dtype != "f32" and not args.graph and not stress and not args.no_side_runs:
        # the exact-f32 parity path on the same workload (short run, same process) for the record
        tr.enc.precision = tr.dec.precision = "f32"
        for _ in range(2):
            one_step()
        dsync()
        tf0 = time.perf_counter()
        n32 = max(3, args.steps // 4)
        for _ in range(n32):
            one_step()
        dsync()
        d32 = time.perf_counter() - tf0
        tr.enc.precision = tr.dec.precision = args.dtype
        out["f32_parity_path"] = {"value": round(B * n32 / d32, 2), "unit": "seq/s", "ms_per_step": round(1e3 * d32 / n32, 4),
                                  "steps": n32, "note": "exact-f32 MFMA GEMMs; ELBO parity <= 1e-4 vs the reference CPU path"}
    if world == 1 and args.dtype == "bf16" and not args.graph and not stress and not args.no_side_runs:
        try:
            out["dropin_path"] = measure_dropin(V, ni, H, nz, B, pool, kl_weight, dev)
        except Exception as e:      # noqa  (must never cost the headline line)
            out["dropin_path"] = {"error": repr(e)[:300]}
    if world == 1 and args.workload == "yahoo" and args.dtype == "bf16" and not args.graph and not args.no_side_runs:
        # the other BASELINE.json GPU configurations, in front of the driver: configs[1] Yelp, configs[4] stress (one fixed-K = 50
        # loop), configs[3] Omniglot (hipGraph replay; kernel groups from an eager pass) -- compact records, ~10 s together
        side = {}
        try:
            side["mixed_shapes"] = side_run_mixed_shapes(dev, steps=200, fixed_tokens_per_s=value * T)
            side["yelp"] = side_run_text("yelp", dev, steps=10, warmup=3, cpu_leg=not args.no_cpu_baseline)
            side["stress"] = side_run_text("stress", dev, steps=50, warmup=2)
            # the headline configuration with --decoder-grads norm (the decoder's vocabulary-sized gradients reduced to their sums of
            # squares in their producers, never written: an option, not the headline -- .grad of those two tensors is then unspecified)
            side["yahoo_decoder_grads_norm"] = side_run_text("yahoo", dev, steps=20, warmup=3, decoder_grads="norm")
            oa = argparse.Namespace(**vars(args))
            oa.graph, oa.dtype, oa.steps, oa.warmup, oa.pool = 1, "f32", 20, 5, 16
            om = measure_omniglot(oa, dev, 0, 1, cpu_baseline=False, profile_eager=True)
            side["omniglot"] = {"value": om["value"], "unit": om["unit"], "ms_per_step": om["ms_per_step"], "dtype": om["dtype"],
                                "steps": om["steps"], "workload": om["config"]["workload"] + ", hipGraph replay",
                                "dominant_group": {k: om["roofline"].get(k) for k in ("bound", "frac", "achieved", "unit", "ms_per_step", "kernel")},
                                "other_groups": [{k: g.get(k) for k in ("bound", "frac", "ms_per_step")} for g in om.get("roofline_other_groups", [])]}
            # the same with the decoder's direct convolutions on split-bf16 operands (precision="bf16x3": holds the f32 fixtures' bounds)
            oa.dtype = "bf16x3"
            om3 = measure_omniglot(oa, dev, 0, 1, cpu_baseline=not args.no_cpu_baseline, profile_eager=True)
            side["omniglot_bf16x3"] = {"value": om3["value"], "unit": om3["unit"], "ms_per_step": om3["ms_per_step"], "dtype": om3["dtype"],
                                       "steps": om3["steps"], "workload": om3["config"]["workload"] + ", hipGraph replay",
                                       "dominant_group": {k: om3["roofline"].get(k) for k in ("bound", "frac", "achieved", "unit", "ms_per_step", "kernel")},
                                       "other_groups": [{k: g.get(k) for k in ("bound", "frac", "ms_per_step")} for g in om3.get("roofline_other_groups", [])]}
            if "cpu_baseline" in om3:
                side["omniglot_bf16x3"]["cpu_baseline"] = om3["cpu_baseline"]
                side["omniglot_bf16x3"]["speedup_vs_cpu_baseline"] = om3["speedup_vs_cpu_baseline"]
        except Exception as e:      # noqa  (a side run must never cost the headline line)
            side["error"] = repr(e)[:300]
        out["side_runs"] = side
    if world == 1 and not args.force_dp and not emu and not args.no_vendor_baseline and not args.no_side_runs and not args.graph:
        # the reference's own GPU path on this box (stock PyTorch-ROCm kernels), beside the headline -- yardstick only
        vs = vendor_stack_baseline(args.workload, limit_s=90.0)
        for lab in ("f32", "autocast_bf16"):
            if "value" in vs.get(lab, {}):
                vs[lab]["headline_over_this"] = round(value / vs[lab]["value"], 2)
        out["vendor_stack_baseline"] = vs
        if "side_runs" in out and "omniglot_bf16x3" in out["side_runs"]:
            vo = vendor_stack_baseline("omniglot", limit_s=90.0)
            for lab in ("f32", "autocast_bf16"):
                if "value" in vo.get(lab, {}):
                    vo[lab]["ours_over_this"] = round(out["side_runs"]["omniglot_bf16x3"]["value"] / vo[lab]["value"], 2)
            out["side_runs"]["omniglot_bf16x3"]["vendor_stack_baseline"] = vo
    if world == 1 and not args.no_cpu_baseline:
        cpu_baseline_and_elbo(out, args, vae, pool, kl_weight, V, ni, H, nz, B, T, dev, value)
    emit(out)


# ---------------------------------------------------------------------------------------------------------------------------
# Yardstick: the reference's own GPU path on this box.  The reference runs on "cuda" (text.py:62,268-279; its only shipped timing,
# plot_scripts/example.out, is a GPU log): the same Python loop body (text.py:373-387 / image.py:300-314) over stock PyTorch-ROCm
# library kernels -- MIOpen's LSTM / convolutions / BatchNorm, rocBLAS / hipBLASLt GEMMs, F.cross_entropy, autograd,
# clip_grad_norm_, optim.SGD / optim.Adam.  The op sequence is the oracle's ATen graph (validated against the imported reference
# by tests/golden/make_golden*.py) with every tensor on cuda:0.  Bench-leg use of the checker; nothing here is importable from the
# package, nothing of it is in the product path; it runs in a subprocess under a time limit (MIOpen's first-use kernel searches
# must never cost the headline line) and is untimed with respect to `value`.
def vendor_leg_main(kind, autocast, dev=None):
    if dev is None:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
    dsync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)
    amp = torch.autocast(dev.type, dtype=torch.bfloat16, enabled=bool(autocast))
    if kind != "omniglot":
        from oracle import text_vae_oracle as O
        if dev.type == "cuda":
            # nn.LSTM passes its `training` flag to the ATen op (enc_lstm.py:60 / dec_lstm.py:104 run under vae.train()); MIOpen's RNN
            # backward insists on it, the CPU op does not care (dropout 0, one layer: same numbers) -- the oracle's CPU call says False
            def lstm_train(x, w_ih, w_hh, b_ih, b_hh, h0, c0):
                out, hT, cT = torch._VF.lstm(x, (h0.unsqueeze(0), c0.unsqueeze(0)), [w_ih, w_hh, b_ih, b_hh], True, 1, 0.0, True, False, True)
                return out, (hT[0], cT[0])
            O.lstm_aten = lstm_train
        cfg = WORKLOADS[kind]
        V, ni, H, nz, B, T = (cfg[k] for k in ("V", "ni", "H", "nz", "B", "T"))
        Q = {k: v.to(dev).requires_grad_(True) for k, v in O.random_params(V, ni, H, nz, seed=783435).items()}
        enc_opt = torch.optim.SGD([Q[k] for k in O.ENC_KEYS], lr=1.0, momentum=0)      # text.py:325
        allp = [Q[k] for k in O.ALL_KEYS]
        pool = [O.synthetic_batch(B, T, V, seed=i).to(dev) for i in range(8)]
        unit, n_unit = "seq/s", B

        def step(i):
            x = pool[i % len(pool)]
            for q in allp:
                q.grad = None                                                            # zero_grad (text.py:373-374)
            eps = torch.randn(B, 1, nz, device=dev)                                      # encoder.py:77
            m_in = torch.bernoulli(torch.full((B, T - 1, ni), 0.5, device=dev))          # nn.Dropout(0.5) x2 (dec_lstm.py:81,106)
            m_out = torch.bernoulli(torch.full((B, T - 1, H), 0.5, device=dev))
            with amp:
                loss, rec, kl = O.vae_loss(Q, x, 0.1, eps, m_in, m_out, impl="aten")
            s = loss.sum().item()                                                        # text.py:381 (a host read per iteration)
            loss.mean(dim=-1).backward()
            torch.nn.utils.clip_grad_norm_(allp, 5.0)                                    # text.py:385
            enc_opt.step()                                                               # text.py:387
            return s
    else:
        from oracle import image_vae_oracle as IO
        from vae_lagging_encoder_amd.factory import build_image_vae
        B = WORKLOADS["omniglot"]["B"]
        vae = build_image_vae(torch.device("cpu"), 783435)
        P = {k: v.detach().to(dev) for k, v in vae.state_dict().items()}
        del vae
        c = IO.Ctx(P, train=True)
        probs = torch.rand(8, B, 1, 28, 28).to(dev)
        x0 = torch.bernoulli(probs[0])
        IO.vae_loss(c, x0, 1.0, torch.randn(B, 1, 32, device=dev))                      # creates the leaves (masked weights: G5)
        leaves = dict(c.leaf)
        allp = list(leaves.values())
        enc_opt = torch.optim.Adam([v for k, v in leaves.items() if k.startswith("encoder.")], lr=1e-3)   # image.py:267
        unit, n_unit = "img/s", B

        def step(i):
            x = torch.bernoulli(probs[i % 8])                                            # image.py:287,318
            for q in allp:
                q.grad = None
            cc = IO.Ctx(P, train=True)
            cc.leaf = leaves                                                             # persistent parameters (no per-step clones)
            with amp:
                loss, rec, kl = IO.vae_loss(cc, x, 1.0, torch.randn(B, 1, 32, device=dev))
            s = loss.sum().item()                                                        # image.py:308
            loss.mean(dim=-1).backward()
            torch.nn.utils.clip_grad_norm_(allp, 5.0)                                    # image.py:312
            enc_opt.step()                                                               # image.py:314
            for k, v in cc.new_stats.items():
                P[k] = v                                                                 # BatchNorm running statistics (G11)
            return s
    t_first = time.perf_counter()
    for i in range(3):
        step(i)
    dsync()
    warm = time.perf_counter() - t_first
    n = 10
    t0 = time.perf_counter()
    for i in range(n):
        last = step(3 + i)
    dsync()
    dt = time.perf_counter() - t0
    print("VENDOR " + json.dumps({"value": round(n_unit * n / dt, 2), "unit": unit, "ms_per_step": round(1e3 * dt / n, 3), "steps": n,
                                  "warmup_s": round(warm, 2), "loss_finite": bool(np.isfinite(last))}), flush=True)


def vendor_stack_baseline(kind, limit_s=150.0):
    """{"f32": {...}, "autocast_bf16": {...}}: the leg above in a subprocess per arithmetic, each under a time limit."""
    import subprocess
    res = {"what": "PyTorch-ROCm library path, yardstick only: the reference's loop body (%s) over stock ATen / MIOpen / rocBLAS kernels with "
                   "every tensor on cuda:0 -- what a user of the reference gets on this GPU without this library; 3 warm-up + 10 timed "
                   "steps in a subprocess, untimed w.r.t. the headline" % ("text.py:373-387" if kind != "omniglot" else "image.py:300-314")}
    env = dict(os.environ)
    env.setdefault("MIOPEN_FIND_MODE", "FAST")
    env.setdefault("MIOPEN_USER_DB_PATH", os.path.join(os.environ.get("TMPDIR", "/tmp"), "lvae_miopen_db"))
    env.setdefault("MIOPEN_LOG_LEVEL", "1")
    for label, ac in (("f32", 0), ("autocast_bf16", 1)):
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--vendor-leg", kind, "--vendor-autocast", str(ac)],
                               capture_output=True, text=True, timeout=limit_s, env=env, cwd=ROOT)
            line = [l for l in r.stdout.splitlines() if l.startswith("VENDOR ")]
            res[label] = json.loads(line[0][7:]) if line else {"error": ("rc %d: " % r.returncode) + r.stderr.strip()[-300:]}
        except subprocess.TimeoutExpired:
            res[label] = {"error": "no result within %.0f s (library kernel searches / compiles on a fresh box)" % limit_s}
        except Exception as e:      # noqa
            res[label] = {"error": repr(e)[:300]}
    return res


def cpu_text_leg(O, P, xs, kl_weight, es, mis, mos, Bc, T, ncpu, counts=(16, 32)):
    """The oracle's ATen path (= the reference's CPU op sequence) on one full batch: 1 warm-up, then 3 timed inner steps at each of
    the two thread counts that have won every sweep so far (oneDNN's LSTM backward degrades below 8 and above 64 threads; SURVEY.md
    8d asked for >= 3 timed steps); the MEDIAN step of the better count is the value."""
    med = {}
    warmed = False
    for nt in [n for n in counts if n <= ncpu] or [ncpu]:
        torch.set_num_threads(nt)
        if not warmed:
            O.inner_step(P, xs, kl_weight, es, mis, mos, impl="aten")              # warm-up (pages the weights in)
            warmed = True
        ts = []
        for _ in range(3):
            tc = time.perf_counter()
            O.inner_step(P, xs, kl_weight, es, mis, mos, impl="aten")
            ts.append(time.perf_counter() - tc)
        med[nt] = sorted(ts)[1]
    best = min(med, key=lambda k: med[k])
    return {"value": round(Bc / med[best], 3), "unit": "seq/s", "cores": best, "kind": "port",
            "sample": "median of 3 timed inner steps (after 1 warm-up) on a full batch of %d sequences at full length T=%d, at the "
                      "better of %s threads; torch CPU ATen ops (oneDNN LSTM) = the reference's CPU path restated in oracle/" % (
                          Bc, T, " / ".join(str(k) for k in med)),
            "median_seq_per_s_by_threads": {str(k): round(Bc / v, 3) for k, v in sorted(med.items())}, "host_cpus": ncpu}


def cpu_baseline_and_elbo(out, args, vae, pool, kl_weight, V, ni, H, nz, B, T, dev, value):
    """`cpu_baseline`: the reference's CPU op sequence (oracle 'aten' path = torch CPU ATen ops, oneDNN LSTM, validated
    bit-identical to the imported reference by tests/golden/make_golden.py) timed on this box's host cores on ONE FULL batch
    of the workload (B sequences at full length T), at the best of a thread-count sweep (oneDNN's LSTM backward degrades badly
    with hundreds of threads).  `elbo_delta`: one fused step of the timed arithmetic vs the oracle on a NON-degenerate model
    (weights 5x the reference init, wide encoder head and vocabulary projection: loss ~2.6 % above (T-1) ln V) -- at the
    reference init the loss is (T-1) ln V whatever the model computes, which cannot fail."""
    from oracle import text_vae_oracle as O            # the checker, used here only as the timed CPU baseline / ELBO check
    from vae_lagging_encoder_amd.factory import build_text_vae as build_vae
    from vae_lagging_encoder_amd.trainer import AggressiveTextTrainer
    ncpu = os.cpu_count() or 1
    P = {k: v.detach().cpu() for k, v in vae.state_dict().items() if k in O.ALL_KEYS}
    xb = pool[0].cpu()
    Bc = min(B, 32)                                       # stress: a 32-sequence slice of the 128 (per-sequence cost is the metric)
    eps, m_in, m_out = O.draw_noise(B, T, ni, H, nz, seed=1)
    xs, es, mis, mos = xb[:Bc].contiguous(), eps[:Bc].contiguous(), m_in[:Bc].contiguous(), m_out[:Bc].contiguous()
    out["cpu_baseline"] = cpu_text_leg(O, P, xs, kl_weight, es, mis, mos, Bc, T, ncpu)
    best = out["cpu_baseline"]["cores"]
    out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
    # ELBO delta on a model where the logits matter, identical batch and noise
    torch.set_num_threads(best)
    SB = 8
    Pn = O.random_params(V, ni, H, nz, seed=7, scale=0.05, head_scale=0.2)
    g = torch.Generator().manual_seed(8)
    Pn["decoder.pred_linear.weight"] = ((torch.rand(V, H, generator=g, dtype=torch.float64) * 2 - 1) * 0.3).float()
    xs, es, mis, mos = xb[:SB].contiguous(), eps[:SB].contiguous(), m_in[:SB].contiguous(), m_out[:SB].contiguous()
    r = O.inner_step(Pn, xs, 0.5, es, mis, mos, impl="aten")
    base = SB * (T - 1) * float(np.log(V))
    deltas = {}
    ef_own = getattr(args, "encoder_forward", "bf16")
    variants = [(args.dtype, args.dtype, ef_own if args.dtype == "bf16" else None)]
    if args.dtype == "bf16" and ef_own == "bf16":
        variants.append(("bf16+exact_encoder_forward", "bf16", "f32"))
        variants.append(("bf16/bf16_forward_operands", "bf16", "bf16ops"))
    if args.dtype != "f32":
        variants.append(("f32", "f32", None))
    for label, prec, ef in variants:
        vae2 = build_vae(V, ni, H, nz, dev, params=Pn)
        tr2 = AggressiveTextTrainer(vae2, lr=1.0, clip=5.0, precision=prec, encoder_forward=None if ef == "bf16ops" else ef)
        tr2.enc.fwd_operands = "bf16" if ef == "bf16ops" else getattr(args, "forward_operands", "f16")
        tr2.step(xs.to(dev), 0.5, noise=(es.to(dev), mis.to(torch.uint8).to(dev), mos.to(torch.uint8).to(dev)))
        s2 = tr2.read_stats()
        deltas[label] = {
            "elbo_rel": float("%.3e" % (abs(s2["loss_sum"] - float(r["loss"].sum())) / abs(float(r["loss"].sum())))),
            "rec_rel": float("%.3e" % (abs(s2["rec_sum"] - float(r["rec"].sum())) / abs(float(r["rec"].sum())))),
            "kl_rel": float("%.3e" % (abs(s2["kl_sum"] - float(r["kl"].sum())) / abs(float(r["kl"].sum())))),
            "grad_norm_rel": float("%.3e" % (abs(s2["norm"] - r["total_norm"]) / r["total_norm"]))}
    out["elbo_delta_vs_cpu"] = {"per_dtype": deltas, "loss_per_seq": round(float(r["loss"].mean()), 3),
                                "loss_excess_over_uniform_per_seq": round((float(r["rec"].sum()) - base) / SB, 3),
                                "sample": "%d sequences at T=%d, non-degenerate weights (scale 0.05, head 0.2, vocabulary projection 0.3)" % (SB, T)}
    out["elbo_rel_delta_vs_cpu"] = deltas[args.dtype]["elbo_rel"]


if __name__ == "__main__":
    main()
