#!/usr/bin/env python3
"""Diagnostic: in the thin iterations (network launch under 60 us) of the last generation of a rocprofv3 --kernel-trace
database, per pool stream: the mean gap in front of the search launch and in front of the network launch, and what else
ran on the stream in between.  usage: gap_probe.py results.db"""
import sqlite3
import sys
from collections import defaultdict, Counter

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, stream_id, queue_id, start, end from kernels order by start"))
streams = defaultdict(list)
for n, st, q, s, e in rows:
    streams[(st, q)].append((n.split("(")[0], s, e))
for k, v in streams.items():
    k3 = [i for i, (n, _, _) in enumerate(v) if "mcts_step" in n]
    if len(k3) < 100:
        print("stream", k, "kernels:", Counter(n for n, _, _ in v).most_common(6))
        continue
    gaps = [(v[k3[j + 1]][1] - v[k3[j]][2], j) for j in range(len(k3) - 1)]
    cut = max(gaps)[1] + 1
    k3 = k3[cut:]
    g_k3, g_nn, n, others = 0.0, 0.0, 0, Counter()
    for a, b in zip(k3[:-1], k3[1:]):
        seg = v[a:b]
        nn = [(s, e) for nm, s, e in seg if "forward" in nm]
        if not nn or max(e - s for s, e in nn) > 60e3:
            continue
        n += 1
        g_nn += (nn[0][0] - v[a][2]) / 1e3          # search end -> first network kernel start
        g_k3 += (v[b][1] - max(e for _, _, e in seg)) / 1e3  # last kernel end -> next search start
        for nm, s, e in seg[1:]:
            if "forward" not in nm:
                others[nm] += 1
    print("stream %s: %d thin iterations; gap search -> network %.1f us, network -> next search %.1f us; other kernels in between: %s"
          % (k, n, g_nn / max(n, 1), g_k3 / max(n, 1), dict(others)))
