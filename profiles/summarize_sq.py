"""Summarise a rocprofv3 --pmc SQ_* pass (counter_collection.csv) into per-kernel matrix-pipe utilisation and wait shares.

    python profiles/summarize_sq.py <counter_collection.csv>

MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs): the gfx94x derived-metric formula (ROCm 7.2
ships no gfx950 section, /opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots") with GRBM_GUI_ACTIVE divided by the 8
XCC instances rocprofv3 sums it over (calibration: lv_gemm_b16_nt_glds at 690 TFLOP/s = 27.6 % of the 2.5 PF peak reads 24.8 %
this way, 3.1 % without the division).  SQ_WAVE_CYCLES / SQ_WAIT_* /
SQ_ACTIVE_INST_* count quad-cycles and are reported as shares of SQ_WAVE_CYCLES (WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY
~ WAVE_CYCLES).
"""
import csv
import sys
from collections import defaultdict


def main(path):
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(lambda: defaultdict(int))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k][r["Counter_Name"]] += 1
    rows = []
    for k, c in acc.items():
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        rows.append((gui, k, max(n[k].values()),
                     100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 256 * 4) if gui else 0.0,
                     100.0 * c.get("SQ_WAIT_ANY", 0.0) / wc if wc else 0.0,
                     100.0 * c.get("SQ_WAIT_INST_ANY", 0.0) / wc if wc else 0.0,
                     100.0 * c.get("SQ_WAIT_INST_LDS", 0.0) / wc if wc else 0.0,
                     100.0 * c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc if wc else 0.0,
                     c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, c.get("SQ_LDS_IDX_ACTIVE", 0.0)) * 100.0))
    rows.sort(reverse=True)
    print("%-90s %8s %10s %9s %10s %12s %12s %10s %12s" % ("kernel", "launches", "gui_Mcyc", "MfmaUtil%", "WAIT_ANY%", "WAIT_INST%", "WAIT_LDS%", "ACTIVE%", "LDSconfl%"))
    for gui, k, ln, mu, wa, wi, wl, ac, bc in rows[:25]:
        print("%-90s %8d %10.2f %9.1f %10.1f %12.1f %12.1f %10.1f %12.1f" % (k[:90], ln, gui / 1e6, mu, wa, wi, wl, ac, bc))


if __name__ == "__main__":
    main(sys.argv[1])
