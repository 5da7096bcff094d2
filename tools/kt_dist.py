"""Diagnostic: distribution of a kernel's launch durations in rocprofv3 --kernel-trace results (rocpd sqlite).
usage: kt_dist.py results.db [kernel substring]"""
import sqlite3
import sys

import numpy as np


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2] if len(sys.argv) > 2 else "mcts_step"
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    kt = [t for t in tabs if t.startswith("rocpd_kernel_dispatch_")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    names = {r[0]: r[1] for r in db.execute("select id, kernel_name from %s" % sym)}
    rows = list(db.execute("select kernel_id, start, end from %s order by start" % kt))
    d = np.array([e - s for kid, s, e in rows if pat in names[kid]]) / 1e3
    n = len(d)
    print("%s: %d launches, mean %.1f us, percentiles 5/25/50/75/95/99: %s" % (pat, n, d.mean(), np.percentile(d, [5, 25, 50, 75, 95, 99]).round(1)))
    g = n // 3
    print("  third generation by tenth: %s" % [round(float(d[2 * g + i * g // 10:2 * g + (i + 1) * g // 10].mean()), 1) for i in range(10)])


if __name__ == "__main__":
    main()
