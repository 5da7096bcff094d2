#!/usr/bin/env python3
"""Diagnostic: per-kernel averages of one rocprofv3 --pmc pass (rocpd sqlite output).
usage: pmc_pass.py <dir given to rocprofv3 -d>"""
import collections
import glob
import os
import sqlite3
import sys

f = glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)
db = sqlite3.connect(f[0])
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
kcol = "kernel_name" if "kernel_name" in cols else "name"
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for k, did, c, v in db.execute("select %s, dispatch_id, counter_name, value from counters_collection" % kcol):
    k = k.split("(")[0]
    acc[k][c] += v
    disp[k].add(did)
for k in sorted(acc, key=lambda k: -len(disp[k])):
    n = len(disp[k])
    print("%-40s launches %6d  " % (k[:40], n) + "  ".join("%s %.4g" % (c, acc[k][c] / n) for c in sorted(acc[k])))
