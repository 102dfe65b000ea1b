"""Which rounding of the bf16 configuration's ENCODER FORWARD owns its KL error (VERDICT r4, weak #1 / next #2a)?

The KL (encoder.py:55) is a function of mu / logvar = linear(h_T) (enc_lstm.py:60-62): only the encoder's forward enters.
Its bf16 roundings: (a) the embedded rows X and (b) W_ih in the input projection Gx = X W_ih^T; (c) W_hh and (d) the
hand-off of h_{t-1} in the recurrent product.  Each is switched on ALONE here -- (a) (b) (c) by rounding that tensor to bf16
and running the exact-f32 forward on it (compared with the reference run on the unrounded weights), (d) as what is left of
the bf16 recurrence's error once (c) is taken out in quadrature -- next to the engine's own switches (exact_forward = gx / rec /
both).  Fixtures: the reference runs at the Yahoo and the Yelp shape (tests/golden/text_*_seeded.npz).

    python profiles/microbench/kl_ablation.py            (on the GPU box)
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def main():
    from helpers import load
    import test_gpu_parity as TP
    from vae_lagging_encoder_amd.trainer import AggressiveTextTrainer
    dev = torch.device("cuda:0")
    res = {}
    for name in ("text_yahoo_seeded", "text_yelp_wide_seeded"):
        fx = load(name)
        x = torch.from_numpy(fx["x"]).to(dev)
        noise = tuple(torch.from_numpy(fx[k]).to(dev) for k in ("eps", "mask_in", "mask_out"))
        kl_ref, rec_ref, loss_ref = (float(fx[k].sum()) for k in ("kl", "rec", "loss"))

        def run(exact, pre_round=(), operands="bf16", gx_round=None):
            vae = TP._seeded_full_size_vae(fx, dev)
            with torch.no_grad():
                for k in pre_round:
                    p = dict(vae.named_parameters())[k]
                    p.copy_(bf16_round(p))
            tr = AggressiveTextTrainer(vae, lr=1.0, clip=5.0, precision="bf16")
            tr.enc.exact_forward = exact
            tr.enc.fwd_operands = operands
            tr.enc.gx_round = gx_round
            tr.step(x, float(fx["kl_weight"]), noise=noise)
            st = tr.read_stats()
            return {"kl_rel": abs(st["kl_sum"] - kl_ref) / abs(kl_ref), "rec_rel": abs(st["rec_sum"] - rec_ref) / abs(rec_ref),
                    "loss_rel": abs(st["loss_sum"] - loss_ref) / abs(loss_ref), "kl_signed": (st["kl_sum"] - kl_ref) / abs(kl_ref)}
        rows = {
            "binary16 forward operands (the default since round 5: X, W_ih, W_hh, h hand-off as IEEE half)": run((), operands="f16"),
            "binary16 forward operands + Gx rounded to binary16 before the recurrence (round 6: price of a 16-bit Gx image)": run((), operands="f16", gx_round="f16"),
            "binary16 forward operands + Gx rounded to bf16 before the recurrence": run((), operands="f16", gx_round="bf16"),
            "bf16 configuration (all four roundings)": run(()),
            "exact input projection only (exact_forward = gx)": run(("gx",)),
            "exact recurrence only (exact_forward = rec)": run(("rec",)),
            "exact forward (exact_forward = gx, rec)": run(("gx", "rec")),
            "exact forward on bf16-rounded embedding table  [(a) alone]": run(("gx", "rec"), ("encoder.embed.weight",)),
            "exact forward on bf16-rounded W_ih             [(b) alone]": run(("gx", "rec"), ("encoder.lstm.weight_ih_l0",)),
            "exact forward on bf16-rounded W_hh             [(c) alone]": run(("gx", "rec"), ("encoder.lstm.weight_hh_l0",)),
            "exact gx + bf16 recurrence on bf16-rounded W_hh [(c)+(d): the same as 'exact input projection only' if pre-rounding is idempotent]":
                run(("gx",), ("encoder.lstm.weight_hh_l0",)),
        }
        res[name] = rows
        print("== %s (KL of the reference run: %.6g summed over the batch)" % (name, kl_ref))
        for k, v in rows.items():
            print("  %-110s KL %.2e (signed %+.2e)  rec %.1e  ELBO %.1e" % (k, v["kl_rel"], v["kl_signed"], v["rec_rel"], v["loss_rel"]))
        c = rows["exact forward on bf16-rounded W_hh             [(c) alone]"]["kl_signed"]
        cd = rows["exact input projection only (exact_forward = gx)"]["kl_signed"]
        print("  (d) alone, by difference of the signed errors: %+.2e" % (cd - c))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "kl_ablation.json"), "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
