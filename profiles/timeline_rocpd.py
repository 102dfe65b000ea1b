"""The kernels of ONE step in dispatch order, with start offset, duration and the gap to the previous kernel, from a rocprofv3
rocpd database (…_results.db):   python profiles/timeline_rocpd.py <db> [step index counted from the end, default 2]
A step is delimited by the rng_noise_step_kernel launches (the first kernel of every fused text step)."""
import sqlite3
import sys


def main(path, back=2):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "rng_noise_step_kernel" in r[0]]
    if len(marks) < back + 1:
        raise SystemExit("not enough steps in the trace")
    a, b = marks[-back - 1], marks[-back]
    t0 = rows[a][1]
    prev_end = t0
    busy = 0
    print("# source: %s, step %d from the end: %d kernels" % (path, back, b - a))
    print("%9s %9s %8s  %s" % ("start_us", "dur_us", "gap_us", "kernel"))
    for name, st, en in rows[a:b]:
        print("%9.1f %9.1f %8.1f  %s" % ((st - t0) / 1e3, (en - st) / 1e3, (st - prev_end) / 1e3, name[:150]))
        busy += en - st
        prev_end = max(prev_end, en)
    span = rows[b][1] - t0
    print("# step span %.1f us, kernel time %.1f us, idle %.1f us" % (span / 1e3, busy / 1e3, (span - busy) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2)
