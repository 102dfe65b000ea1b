"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM traffic per launch.

    python profiles/summarize_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv>

Units / corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: the counters are in KB
(bytes = value * 1024); on gfx950 FETCH_SIZE reports exactly HALF of the bytes of a wide coalesced streaming read
(128-B requests tallied at 64 B), so the read side is doubled; WRITE_SIZE is uncalibrated (reported as is).
Infinity-Cache hits appear to be counted, not excluded.
"""
import csv
import sys
from collections import defaultdict


def load(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        name = r["Kernel_Name"]
        a = acc[name]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


def main(fetch_csv, write_csv):
    f = load(fetch_csv, "FETCH_SIZE")
    w = load(write_csv, "WRITE_SIZE")
    names = sorted(set(f) | set(w), key=lambda n: -(f.get(n, [0, 0])[1] * 2 + w.get(n, [0, 0])[1]))
    print("%-100s %8s %16s %16s %16s" % ("kernel", "launches", "read_MB/launch", "write_MB/launch", "total_MB/launch"))
    print("# read = FETCH_SIZE * 1024 * 2 (gfx950 half-count correction); write = WRITE_SIZE * 1024 (uncalibrated)")
    for n in names[:30]:
        nf, vf = f.get(n, [0, 0.0])
        nw, vw = w.get(n, [0, 0.0])
        rd = vf * 1024 * 2 / max(nf, 1) / 1e6
        wr = vw * 1024 / max(nw, 1) / 1e6
        print("%-100s %8d %16.3f %16.3f %16.3f" % (n[:100], max(nf, nw), rd, wr, rd + wr))


def groups(fetch_csv, write_csv, steps=None):
    """Launch-weighted HBM MB per launch of the kernel groups bench.py reports, split by precision:
    bf16 = lstm_step_*<..., true> / lstm_*_persist_kernel + lv_gemm_b16*; f32 = lstm_step_*<..., false> + lv_gemm_f32_kernel<*, *, 2>.
    steps: inner steps the profiled command ran (timed + warm-up) -> whole-step GB over ALL kernels."""
    f = load(fetch_csv, "FETCH_SIZE")
    w = load(write_csv, "WRITE_SIZE")

    def mb(pred):
        tot, n = 0.0, 0
        for name in set(f) | set(w):
            if not pred(name):
                continue
            nf, vf = f.get(name, [0, 0.0])
            nw, vw = w.get(name, [0, 0.0])
            tot += (vf * 1024 * 2 + vw * 1024) / 1e6
            n += max(nf, nw)
        return round(tot / max(n, 1), 3), n, tot
    is_lstm = lambda n: "lstm_step_" in n
    is_pers = lambda n: "lstm_fwd_persist" in n or "lstm_bwd_persist" in n
    bf = lambda n: "true>" in n
    out = {
        "bf16": {"lstm_MB_per_launch": mb(lambda n: is_lstm(n) and bf(n))[0],
                 "lstm_persist_MB_per_launch": mb(is_pers)[0],
                 "lstm_fwd_persist_MB_per_launch": mb(lambda n: "lstm_fwd_persist" in n)[0],
                 "lstm_bwd_persist_MB_per_launch": mb(lambda n: "lstm_bwd_persist" in n)[0],
                 "gemm_MB_per_launch": mb(lambda n: "lv_gemm_b16" in n)[0]},
        "f32": {"lstm_MB_per_launch": mb(lambda n: is_lstm(n) and not bf(n))[0],
                "gemm_MB_per_launch": mb(lambda n: "lv_gemm_f32_kernel" in n and ", 2>" in n)[0]},
    }
    if steps:
        out["bf16"]["whole_step_GB"] = round(mb(lambda n: True)[2] / 1e3 / steps, 3)
    return out


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[3] == "--json":
        import json
        print(json.dumps(groups(sys.argv[1], sys.argv[2], int(sys.argv[4]) if len(sys.argv) > 4 else None), indent=1))
    else:
        main(sys.argv[1], sys.argv[2])
