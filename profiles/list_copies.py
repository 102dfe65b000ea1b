"""List the memory copies of a rocprofv3 --memory-copy-trace --kernel-trace run with the kernels dispatched just before each
(to attribute stray device-to-device copies to the code that issued them).  usage: list_copies.py results.db [max_rows]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
lim = int(sys.argv[2]) if len(sys.argv) > 2 else 60
cols = [r[1] for r in db.execute("pragma table_info(memory_copies)")]
print("# columns:", cols)
rows = list(db.execute("select * from memory_copies order by start"))
print("# copies:", len(rows))
import collections
by = collections.Counter()
ix = {c: i for i, c in enumerate(cols)}
for r in rows:
    by[(r[ix.get("name", 0)], r[ix.get("size", 0)])] += 1
for k, v in by.most_common(30):
    print(v, k)
kcols = [r[1] for r in db.execute("pragma table_info(kernels)")]
kix = {c: i for i, c in enumerate(kcols)}
ks = list(db.execute("select * from kernels order by start"))
import bisect
starts = [k[kix["start"]] for k in ks]
seen = collections.Counter()
for r in rows[len(rows) // 2:]:
    j = bisect.bisect_left(starts, r[ix["start"]])
    prev = ks[j - 1][kix["name"]][:70] if j > 0 else "-"
    nxt = ks[j][kix["name"]][:70] if j < len(ks) else "-"
    seen[(r[ix.get("size", 0)], prev, nxt)] += 1
for k, v in seen.most_common(lim):
    print(v, k)
