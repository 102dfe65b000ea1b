"""Summarise a rocprofv3 rocpd database (…_results.db) into the per-kernel stats table committed under profiles/.

    python profiles/summarize_rocpd.py gpurun_out/<dir>/<host>/<pid>_results.db > profiles/<name>_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print("# source: %s" % path)
    print("# total kernel time %.3f ms over %d dispatches" % (tot / 1e6, sum(r[1] for r in rows)))
    print("%-110s %8s %12s %7s %12s %12s %12s" % ("kernel", "calls", "total_ms", "pct", "avg_us", "min_us", "max_us"))
    for name, n, s, a, mn, mx in rows:
        print("%-110s %8d %12.3f %6.1f%% %12.2f %12.2f %12.2f" % (name[:110], n, s / 1e6, 100.0 * s / tot, a / 1e3, mn / 1e3, mx / 1e3))


if __name__ == "__main__":
    main(sys.argv[1])
