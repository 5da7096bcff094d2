#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2 `rocpd` sqlite) result as the per-kernel table
`rocprofv3 --kernel-trace --stats` describes: calls, total, average, share.
usage: rocpd_summary.py results.db [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    unit = "us"
    # rocpd stores nanoseconds; the view reports microseconds when durations are divided -- detect by magnitude
    lines = ["| kernel | calls | total (%s) | average (%s) | share %% |" % (unit, unit), "|---|---:|---:|---:|---:|"]
    for name, calls, total, avg, pct in rows:
        lines.append("| `%s` | %d | %.1f | %.3f | %.2f |" % (name.split("(")[0], calls, total, avg, pct))
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        with open(sys.argv[2], "a") as f:
            f.write(text)
    print(text)


if __name__ == "__main__":
    main()
