"""Summarise a rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum pass: per kernel, the L2's memory-side
read requests per launch, how many of them were destined for the memory controllers (DRAM side -- where the Infinity Cache sits: its
hits are NOT distinguished by any counter this rocprofv3 lists) and how many were 32-byte ones.

    python profiles/summarize_ea_split.py <counter_collection.csv>
"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    a = acc[r["Kernel_Name"]][r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
rows = []
for k, c in acc.items():
    n = max(v[0] for v in c.values())
    rd, dram, r32 = (c.get(x, [0, 0.0])[1] / max(n, 1) for x in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_RDREQ_32B_sum"))
    rows.append((rd, k, n, dram, r32))
print("%-96s %8s %14s %14s %10s %14s" % ("kernel", "launches", "EA reads/launch", "to DRAM side", "share", "MB at 64 B/req"))
for rd, k, n, dram, r32 in sorted(rows, reverse=True)[:24]:
    print("%-96s %8d %14.0f %14.0f %9.1f%% %14.1f" % (k[:96], n, rd, dram, 100.0 * dram / max(rd, 1.0), (rd - r32) * 64 / 1e6 + r32 * 32 / 1e6))
