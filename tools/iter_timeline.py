#!/usr/bin/env python3
"""Diagnostic: from a rocprofv3 --kernel-trace database of a fused training run, the iterations of the LAST generation
per pool (stream): period from one search launch to the next, and the time of its priors / search / network kernels,
binned by the network launch's duration (a proxy for the batch size) -- where the thin tail of a generation spends its
time.  usage: iter_timeline.py results.db"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, stream_id, queue_id, start, end from kernels order by start"))
streams = defaultdict(list)
for n, st, q, s, e in rows:
    streams[(st, q)].append((n, s, e))
pools = [k for k, v in streams.items() if sum("mcts_step" in n for n, _, _ in v) > 100]
print("pool streams:", pools)
bins = [60, 100, 150, 250, 400, 600, 1e9]
for k in pools:
    v = streams[k]
    k3 = [i for i, (n, _, _) in enumerate(v) if "mcts_step" in n]
    # last generation: after the largest gap between consecutive search launches
    gaps = [(v[k3[j + 1]][1] - v[k3[j]][2], j) for j in range(len(k3) - 1)]
    cut = max(gaps)[1] + 1
    k3 = k3[cut:]
    agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0, 0.0])
    for a, b in zip(k3[:-1], k3[1:]):
        seg = v[a:b]  # search kernel, then network kernel(s), then next priors
        period = (v[b][1] - v[a][1]) / 1e3
        t_k3 = (v[a][2] - v[a][1]) / 1e3
        nn = [(e - s) / 1e3 for n, s, e in seg if "forward" in n]
        pr = [(e - s) / 1e3 for n, s, e in seg if "priors" in n]
        t_nn = sum(nn)
        busy = sum((e - s) / 1e3 for n, s, e in seg)
        key = next(i for i, hi in enumerate(bins) if max(nn or [0]) < hi)
        g = agg[key]
        g[0] += 1
        g[1] += period
        g[2] += t_k3
        g[3] += t_nn
        g[4] += sum(pr)
        g[5] += period - busy
    tot = sum(g[1] for g in agg.values())
    print("stream %s: %d iterations, %.1f ms" % (k, len(k3) - 1, tot / 1e3))
    print("  network launch   iterations   share of wall   mean period   search   network   priors   stream idle (us)")
    lo = 0
    for i, hi in enumerate(bins):
        if i in agg:
            g = agg[i]
            n = g[0]
            print("  %4d-%-6s us   %8d   %10.1f %%   %11.1f   %6.1f   %7.1f   %6.1f   %6.1f" %
                  (lo, "inf" if hi > 1e8 else int(hi), n, 100 * g[1] / tot, g[1] / n, g[2] / n, g[3] / n, g[4] / n, g[5] / n))
        lo = hi
