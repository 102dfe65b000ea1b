#!/usr/bin/env python
"""bench.py -- aggressive-loop sequences/sec on the BASELINE.json metric configuration.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

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
    # BASELINE.json configs[3]: Omniglot ResNet-enc + PixelCNN-dec, 28x28 binary (parity-test case; bench line on request)
    "omniglot": dict(B=50),
}
PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md chip table
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 MFMA (same table)
PEAK_HBM_GBS = 8000.0             # HBM3E spec (6.29 TB/s measured copy)
GFLOP_PER_SEQ = {"yahoo": 39.67, "yelp": 19.75}   # SURVEY.md 8(d) / BASELINE.md section 4


def fwd_flops(V, ni, H, nz, B, T):
    return 2 * B * (T * ni * 4 * H + T * H * 4 * H + H * 2 * nz) + 2 * B * nz * H + \
        2 * B * (T - 1) * ((ni + nz) * 4 * H + H * 4 * H + H * V)


def bench_omniglot(args, dev, rank, world):
    """images/sec through the aggressive inner step of the Omniglot VAE (image.py:300-314), B=50 per GPU, replicas only
    (BatchNorm batch statistics make naive data parallelism non-equivalent: SURVEY.md 8e)."""
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
    prof = {}
    if not args.graph:
        engine.PROFILE = prof
    tr.reset_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    engine.PROFILE = None
    stats = tr.read_stats()
    value = world * B * args.steps / dt
    out = {"metric": "aggressive-loop images/sec", "value": round(value, 2), "unit": "img/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": "omniglot ResNetEncoderV2 + PixelCNNDecoderV2 aggressive inner step (fwd+bwd+clip+encoder Adam), "
                                  "B=%d/GPU, 28x28 binary, nz=32, fm=4" % B, "global_batch": world * B, "parallelism": "replicas%d" % world,
                      "hipgraph": bool(args.graph)},
           "mean_loss_per_image": round(stats["loss_sum"] / (B * args.steps), 4)}
    gname = "gemm_" + args.dtype
    recs = prof.get(gname, []) + (prof.get("gemm_f32", []) if args.dtype != "f32" else [])
    ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in recs)
    fl = sum(w for _, _, w, _ in recs)
    peak = PEAK_BF16_MFMA_TFLOPS if args.dtype == "bf16" else PEAK_F32_MFMA_TFLOPS
    tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    if recs:
        out["roofline"] = {"bound": "mfma", "kernel": "lv_gemm_%s_kernel (im2col convolutions)" % args.dtype, "achieved": round(tf, 2),
                           "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4), "traffic": None,
                           "launches_per_step": len(recs) // args.steps, "ms_per_step": round(ms / args.steps, 4),
                           "gflop_per_step": round(fl / args.steps / 1e9, 1)}
    else:
        out["roofline"] = {"bound": "mfma", "achieved": None, "peak": peak, "unit": "TFLOP/s", "frac": None, "traffic": None,
                           "note": "graph replay: no per-kernel events; see the eager run / profiles/ for the kernel breakdown"}
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import image_vae_oracle as IO          # the checker, used here only as the timed CPU baseline
        nthreads = min(64, os.cpu_count() or 1)
        torch.set_num_threads(nthreads)
        Pd = {k: v.detach().cpu() for k, v in vae.state_dict().items()}
        xb = (probs[0].cpu() > 0.5).float()
        eps = torch.randn(B, 1, 32)
        IO.inner_step_adam(Pd, xb, 1.0, eps)
        n, tcpu = 0, 0.0
        while n < 5 and tcpu < 15.0:
            tc = time.perf_counter()
            IO.inner_step_adam(Pd, xb, 1.0, eps)
            tcpu += time.perf_counter() - tc
            n += 1
        out["cpu_baseline"] = {"value": round(B * n / tcpu, 2), "unit": "img/s", "cores": nthreads, "kind": "port",
                               "sample": "%d timed inner steps at B=%d after 1 warm-up, torch CPU ATen ops (the reference's CPU path "
                                         "restated in oracle/image_vae_oracle.py)" % (n, B)}
        out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
    if rank == 0:
        print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="yahoo", choices=sorted(WORKLOADS))
    ap.add_argument("--graph", type=int, default=0, help="replay the step as captured hipGraphs (no per-kernel events)")
    ap.add_argument("--dtype", default="bf16", choices=["f32", "bf16"],
                    help="arithmetic of the large GEMMs: f32 = exact-f32 MFMA (parity path), bf16 = bf16 MFMA, f32 accumulate")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--persistent", type=int, default=1, help="forward LSTM recurrences as one persistent launch (bf16 path)")
    ap.add_argument("--overlap", default="auto", choices=["auto", "on", "off"],
                    help="decoder weight-gradient GEMMs on a side stream under the BPTT chains (auto: on for f32, off for bf16)")
    ap.add_argument("--pool", type=int, default=64)
    args = ap.parse_args()

    from vae_lagging_encoder_amd import dist as lvdist
    from vae_lagging_encoder_amd import engine
    from vae_lagging_encoder_amd.trainer import AggressiveTextTrainer
    from vae_lagging_encoder_amd.factory import build_text_vae as build_vae, synthetic_batch

    rank, local, world = lvdist.init_from_env()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d ... bench.py" % args.gpus)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    cfg = WORKLOADS[args.workload]
    if args.workload == "omniglot":
        return bench_omniglot(args, dev, rank, world)
    V, ni, H, nz, B, T = (cfg[k] for k in ("V", "ni", "H", "nz", "B", "T"))

    # reference init (text.py:265-266) from the reference's default seed (text.py:54,73); same replica on every rank
    vae = build_vae(V, ni, H, nz, dev, seed=783435)
    sync = lvdist.GradSync(mode="strict") if world > 1 else None
    tr = AggressiveTextTrainer(vae, lr=1.0, clip=5.0, seed=783435 + rank, grad_sync=sync, use_graph=bool(args.graph),
                               precision=args.dtype)
    if args.overlap != "auto":
        tr.dec.overlap = (args.overlap == "on")
    tr.enc.persistent = tr.dec.persistent = bool(args.persistent)
    pool = [synthetic_batch(B, T, V, seed=1000 * rank + i).to(dev) for i in range(args.pool)]
    rs = np.random.RandomState(783435)
    kl_weight = 0.1                                         # text.py default kl_start

    def one_step():
        tr.step(pool[int(rs.randint(0, len(pool)))], kl_weight)

    def warm_up():
        for _ in range(args.warmup):
            one_step()
        torch.cuda.synchronize(dev)

    warm_up()
    # The persistent LSTM launches need the whole GPU resident at once; if one of them reported a hand-off timeout during
    # warm-up (something else holds compute units on this box), every rank falls back to the launch-per-step kernels.
    healthy = 1
    try:
        engine.check_persistent_status(tr.enc)
        engine.check_persistent_status(tr.dec)
    except Exception as e:      # noqa
        healthy = 0
        print("bench: persistent LSTM launch unhealthy (%s); using the launch-per-step kernels" % e, file=sys.stderr)
    if world > 1:
        flag = torch.tensor([healthy], dtype=torch.int32, device=dev)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        healthy = int(flag.item())
    if not healthy:
        tr.enc.persistent = tr.dec.persistent = False
        engine.reset_persistent_status(tr.enc)
        engine.reset_persistent_status(tr.dec)
        warm_up()
    if world > 1:
        torch.distributed.barrier()
    prof = None
    if not args.graph:
        prof = {}
        engine.PROFILE = prof
    tr.reset_stats()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize(dev)
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    engine.PROFILE = None
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    stats = tr.read_stats()

    if rank != 0:
        return
    seqs = world * B * args.steps
    value = seqs / dt
    out = {
        "metric": "aggressive-loop seqs/sec", "value": round(value, 2), "unit": "seq/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "%s LSTM-VAE aggressive inner step (fwd+bwd+clip+encoder SGD), B=%d/GPU, T=%d, V=%d, "
                               "ni=%d, H=%d, nz=%d" % (args.workload, B, T, V, ni, H, nz),
                   "global_batch": world * B, "seq_len": T, "parallelism": "dp%d" % world,
                   "hipgraph": bool(args.graph)},
        "mean_loss_per_seq": round(stats["loss_sum"] / (B * args.steps), 4),
    }
    step_flops = 3 * fwd_flops(V, ni, H, nz, B, T)
    if prof:
        # live HIP-event timing of the two kernel groups that make up the step, on their launch stream
        groups = {}
        for name, recs in prof.items():
            ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in recs)
            groups[name] = dict(ms=ms, work=sum(w for _, _, w, _ in recs), launches=sum(n for _, _, _, n in recs))
        gname = "gemm_" + args.dtype
        gemm = groups.get(gname, dict(ms=0.0, work=0.0, launches=0))
        lstm_ms = sum(groups[k]["ms"] for k in ("lstm_fwd", "lstm_bwd") if k in groups)
        lstm_launches = sum(groups[k]["launches"] for k in ("lstm_fwd", "lstm_bwd") if k in groups)
        peak = PEAK_BF16_MFMA_TFLOPS if args.dtype == "bf16" else PEAK_F32_MFMA_TFLOPS
        gemm_tf = gemm["work"] / (gemm["ms"] * 1e-3) / 1e12 if gemm["ms"] > 0 else 0.0
        gemm_roof = {
            "bound": "mfma", "kernel": "lv_gemm_b16_kernel" if args.dtype == "bf16" else "lv_gemm_f32_kernel", "achieved": round(gemm_tf, 2), "peak": peak,
            "unit": "TFLOP/s", "frac": round(gemm_tf / peak, 4), "traffic": None,
            "launches_per_step": gemm["launches"] // args.steps, "ms_per_step": round(gemm["ms"] / args.steps, 4),
            "gflop_per_step": round(gemm["work"] / args.steps / 1e9, 1)}
        # LSTM recurrence.  Algorithmic HBM bytes: W_hh (4H*H, 2 B/element when the recurrent product runs on the bf16 pipe)
        # once per LAUNCH + one timestep of f32 state / gates per timestep.  Launch-per-step kernels re-read W_hh every
        # timestep; the persistent kernels (one launch per recurrence) read it once and keep it in registers.
        wb = 2.0 if args.dtype == "bf16" else 4.0
        fwd_g = groups.get("lstm_fwd", dict(work=0, launches=0, ms=0.0))      # work = timesteps
        bwd_g = groups.get("lstm_bwd", dict(work=0, launches=0, ms=0.0))
        persistent = lstm_launches > 0 and lstm_launches < fwd_g["work"] + bwd_g["work"]
        state_fwd = 4.0 * (B * H * 3 + B * 4 * H * 2)
        state_bwd = 4.0 * (B * 4 * H * 4 + B * H * 10)
        w_reads = lstm_launches if persistent else (fwd_g["work"] + bwd_g["work"])
        lstm_bytes = wb * 4 * H * H * w_reads + state_fwd * fwd_g["work"] + state_bwd * bwd_g["work"]
        lstm_gbs = lstm_bytes / (lstm_ms * 1e-3) / 1e9 if lstm_ms > 0 else 0.0
        steps_total = fwd_g["work"] + bwd_g["work"]
        lstm_roof = {
            "bound": "hbm",
            "kernel": ("lstm_{fwd,bwd}_persist_kernel (one launch per recurrence, W_hh register-resident, tagged-granule hand-off per timestep)"
                       if persistent else "lstm_step_{fwd,bwd_elem,bwd_mm}_kernel (one launch per timestep and stage)"),
            "achieved": round(lstm_gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(lstm_gbs / PEAK_HBM_GBS, 4),
            "traffic": None, "launches_per_step": lstm_launches // args.steps, "ms_per_step": round(lstm_ms / args.steps, 4),
            "avg_launch_us": round(1e3 * lstm_ms / max(1, lstm_launches), 3),
            "timesteps_per_step": int(steps_total // args.steps),
            "us_per_timestep": round(1e3 * lstm_ms / max(1, steps_total), 3),
            "note": "latency-bound chain of dependent timesteps: the HBM fraction is what a perfectly overlapped version would be bound by"}
        # HBM traffic per launch from the PMC passes committed under profiles/ (a profiler cannot wrap this process)
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
                pmc = json.load(fh)
            tr_ = pmc.get(args.workload, {}).get(args.dtype)
            if tr_:
                lstm_roof["traffic"] = round(tr_["lstm_persist_MB_per_launch" if persistent else "lstm_MB_per_launch"] * 1e6)
                gemm_roof["traffic"] = round(tr_["gemm_MB_per_launch"] * 1e6)
                lstm_roof["algorithmic_bytes_per_launch"] = round(lstm_bytes / max(1, lstm_launches))
                lstm_roof["traffic_source"] = gemm_roof["traffic_source"] = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950-corrected)"
        except (OSError, ValueError, KeyError):
            pass
        if gemm["ms"] >= lstm_ms:
            out["roofline"], out["roofline_secondary"] = gemm_roof, lstm_roof
        else:
            out["roofline"], out["roofline_secondary"] = lstm_roof, gemm_roof
        out["whole_step_tflops"] = round(step_flops / (dt / args.steps) / 1e12, 2)
    else:
        peak = PEAK_BF16_MFMA_TFLOPS if args.dtype == "bf16" else PEAK_F32_MFMA_TFLOPS
        out["roofline"] = {"bound": "mfma", "achieved": round(step_flops / (dt / args.steps) / 1e12, 2),
                           "peak": peak, "unit": "TFLOP/s",
                           "frac": round(step_flops / (dt / args.steps) / 1e12 / peak, 4),
                           "traffic": None, "note": "whole-step algorithmic flops (graph replay: no per-kernel events)"}

    if world == 1 and args.dtype != "f32" and not args.graph:
        # the exact-f32 parity path on the same workload (short run, same process) for the record
        tr.enc.precision = tr.dec.precision = "f32"
        for _ in range(2):
            one_step()
        torch.cuda.synchronize(dev)
        tf0 = time.perf_counter()
        n32 = max(3, args.steps // 4)
        for _ in range(n32):
            one_step()
        torch.cuda.synchronize(dev)
        d32 = time.perf_counter() - tf0
        tr.enc.precision = tr.dec.precision = args.dtype
        out["f32_parity_path"] = {"value": round(B * n32 / d32, 2), "unit": "seq/s", "ms_per_step": round(1e3 * d32 / n32, 4),
                                  "steps": n32, "note": "exact-f32 MFMA GEMMs; ELBO parity <= 1e-4 vs the reference CPU path"}
    if world == 1 and not args.no_cpu_baseline:
        # The reference's CPU op sequence (oracle 'aten' path = torch CPU ATen ops, oneDNN LSTM) on this box's host
        # cores, on a BOUNDED sample of the same workload: SB of the B sequences of one batch at full length T
        # (per-sequence cost is what the metric counts), <= 64 threads (oneDNN's LSTM backward degrades badly with
        # hundreds of threads: a full B=32 step took 278 s on the 256-core host), ~10-30 s of CPU work.
        from oracle import text_vae_oracle as O            # the checker, used here only as the timed CPU baseline
        nthreads = min(64, os.cpu_count() or 1)
        torch.set_num_threads(nthreads)
        SB = min(B, 8)
        P = {k: v.detach().cpu() for k, v in vae.state_dict().items() if k in O.ALL_KEYS}
        xb = pool[0].cpu()
        eps, m_in, m_out = O.draw_noise(B, T, ni, H, nz, seed=1)
        xs, es, mis, mos = xb[:SB].contiguous(), eps[:SB].contiguous(), m_in[:SB].contiguous(), m_out[:SB].contiguous()
        tw = time.perf_counter()
        O.inner_step(P, xs, kl_weight, es, mis, mos, impl="aten")              # warm-up
        warm = time.perf_counter() - tw
        n_timed, t_cpu = 0, 0.0
        while n_timed < 3 and (n_timed + 1) * warm + t_cpu < 25.0:
            tc = time.perf_counter()
            O.inner_step(P, xs, kl_weight, es, mis, mos, impl="aten")
            t_cpu += time.perf_counter() - tc
            n_timed += 1
        if n_timed == 0:
            n_timed, t_cpu = 1, warm
        out["cpu_baseline"] = {"value": round(SB * n_timed / t_cpu, 3), "unit": "seq/s", "cores": nthreads,
                               "kind": "port",
                               "sample": "%d timed inner step(s) on %d of the %d sequences of one batch at full length T=%d "
                                         "(after 1 warm-up), torch CPU ATen ops (oneDNN LSTM) = the reference's CPU path "
                                         "restated in oracle/" % (n_timed, SB, B, T)}
        # ELBO delta of the HIP path vs the oracle on the identical (sub)batch and noise (north_star: <= 1e-4 rel, f32)
        r = O.inner_step(P, xs, kl_weight, es, mis, mos, impl="aten") if n_timed == 0 else None
        r = r or O.inner_step(P, xs, kl_weight, es, mis, mos, impl="aten") if warm < 15 else None
        if r is not None:
            vae2 = build_vae(V, ni, H, nz, dev, params=P)
            tr2 = AggressiveTextTrainer(vae2, lr=1.0, clip=5.0, precision=args.dtype)
            tr2.step(xs.to(dev), kl_weight, noise=(es.to(dev), mis.to(torch.uint8).to(dev), mos.to(torch.uint8).to(dev)))
            s2 = tr2.read_stats()
            out["elbo_rel_delta_vs_cpu"] = float("%.3e" % (abs(s2["loss_sum"] - float(r["loss"].sum())) / abs(float(r["loss"].sum()))))
        out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
