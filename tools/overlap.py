#!/usr/bin/env python3
"""Diagnostic: from a rocprofv3 --kernel-trace result (rocpd sqlite), how much of the search kernel's
time runs while a network kernel of the OTHER pool is executing (co-residency of K3 beside K5/K6).
usage: overlap.py results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = [t for t in tabs if "kernels" == t or t.endswith("kernels")]
cols = None
for t in ("kernels", "rocpd_kernel_dispatch"):
    if t in tabs:
        cols = [r[1] for r in db.execute("pragma table_info(%s)" % t)]
        src = t
        break
if cols is None:
    print(tabs)
    raise SystemExit("no kernel table")
namec = "name" if "name" in cols else "kernel_name"
rows = list(db.execute("select %s, start, end from %s order by start" % (namec, src)))
t0 = rows[0][1]
k3 = [(s - t0, e - t0) for n, s, e in rows if "mcts_step" in n]
pr = [(s - t0, e - t0) for n, s, e in rows if "priors" in n]
nn = [(s - t0, e - t0) for n, s, e in rows if "forward" in n]


def union(iv):
    iv = sorted(iv)
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def total(iv):
    return sum(e - s for s, e in iv)


def inter(a, b):
    i = j = 0
    t = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if e > s:
            t += e - s
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return t


un, uk = union(nn), union(k3 + pr)
span = rows[-1][2] - t0
print("span %.1f ms; network busy (union) %.1f ms; search+priors busy (union) %.1f ms; both at once %.1f ms; GPU idle %.1f ms" %
      (span / 1e6, total(un) / 1e6, total(uk) / 1e6, inter(un, uk) / 1e6, (span - total(union(nn + k3 + pr))) / 1e6))
print("search kernel: %d launches, mean %.1f us; priors mean %.1f us; network: %d launches, mean %.1f us" %
      (len(k3), total(k3) / max(len(k3), 1) / 1e3, total(pr) / max(len(pr), 1) / 1e3, len(nn), total(nn) / max(len(nn), 1) / 1e3))

# distribution of the network launches by duration: how much of the span the thin iterations take
import collections
buckets = collections.OrderedDict((b, [0, 0.0]) for b in (100, 150, 200, 300, 500, 800, 1200, 10**9))
for s, e in nn:
    d = (e - s) / 1e3
    for b in buckets:
        if d < b:
            buckets[b][0] += 1
            buckets[b][1] += d
            break
print("network launches by duration (us): " + ", ".join("<%s: %d launches, %.1f ms" % ("inf" if b > 10**8 else b, n, t / 1e3) for b, (n, t) in buckets.items()))
