#!/usr/bin/env python3
"""Diagnostic: from a rocprofv3 --kernel-trace database of a fused training run, the duration of the search launches
(co_k_mcts_step) of the LAST generation per pool stream, by progress through the generation (twentieths of the pool's
iterations) -- where in a generation the search is slow.  usage: k3_by_iteration.py results.db"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
if "kernels" in tabs:
    rows = list(db.execute("select name, stream_id, queue_id, start, end from kernels order by start"))
else:  # raw rocpd schema
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(db.execute("select s.kernel_name, d.stream_id, d.queue_id, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)))
streams = defaultdict(list)
for n, st, q, s, e in rows:
    streams[(st, q)].append((n, s, e))
pools = [k for k, v in streams.items() if sum("mcts_step" in n for n, _, _ in v) > 100]
for k in pools:
    v = [(s, e) for n, s, e in streams[k] if "mcts_step" in n]
    gaps = [(v[j + 1][0] - v[j][1], j) for j in range(len(v) - 1)]
    cut = max(gaps)[1] + 1
    v = v[cut:]
    n = len(v)
    t0, t1 = v[0][0], v[-1][1]
    print("pool stream %s: %d search launches in %.1f ms, search time %.1f ms" % (k, n, (t1 - t0) / 1e6, sum(e - s for s, e in v) / 1e6))
    line = []
    for b in range(20):
        seg = v[b * n // 20:(b + 1) * n // 20]
        if seg:
            d = [(e - s) / 1e3 for s, e in seg]
            line.append("%3d-%3d%%: mean %6.1f us  max %6.1f  (period %6.1f)" % (5 * b, 5 * b + 5, sum(d) / len(d), max(d), (seg[-1][1] - seg[0][0]) / 1e3 / len(seg)))
    print("\n".join("   " + x for x in line))
