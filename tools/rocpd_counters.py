#!/usr/bin/env python3
"""Per-kernel averages of the counters of a rocprofv3 --pmc pass (ROCm 7.2 rocpd sqlite).
usage: rocpd_counters.py results.db [kernel-substring]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
want = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
kcol = "kernel_name" if "kernel_name" in cols else "name"
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
disp = collections.defaultdict(set)
for k, did, c, v in db.execute("select %s, dispatch_id, counter_name, value from counters_collection" % kcol):
    k = k.split("(")[0]
    if want and want not in k:
        continue
    acc[k][c] += v
    disp[(k, c)].add(did)
for k in acc:
    print(k)
    for c in sorted(acc[k]):
        n = max(len(disp[(k, c)]), 1)
        print("   %-28s %16.1f per launch (%d launches)" % (c, acc[k][c] / n, n))
