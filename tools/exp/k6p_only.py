#!/usr/bin/env python3
"""Experiment helper: nothing but the pixel-major f16x3 kernel (K6p) on one batch, for rocprofv3 --pmc passes and A/B of
variants.  usage (GPU box): python tools/exp/k6p_only.py [lib.so] [rows] [reps]; tools/exp/k6p_only.py --pmc dir sums the
counters of a rocprofv3 counter_collection.csv per kernel."""
import ctypes as C
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--pmc":
    import sqlite3

    acc = {}
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*.db"), recursive=True):
        db = sqlite3.connect(f)
        cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
        kcol = "kernel_name" if "kernel_name" in cols else "name"
        seen = {}
        for k, did, c, v in db.execute("select %s, dispatch_id, counter_name, value from counters_collection" % kcol):
            a = acc.setdefault((k[:44], c), [0.0, set()])
            a[0] += v
            a[1].add(did)
    for (k, c), (v, n) in sorted(acc.items()):
        print("%-46s %-28s %14.0f per launch (%d launches)" % (k, c, v / len(n), len(n)))
    sys.exit(0)

from corintho_ai_amd import NET_RESCNN4_H3, Trainer, _lib, nets  # noqa: E402

lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else None
rest = [a for a in sys.argv[1:] if not a.endswith(".so")]
rows = int(rest[0]) if rest else 32768
reps = int(rest[1]) if len(rest) > 1 else 50
L = _lib.declare(C.CDLL(os.path.abspath(lib))) if lib else None
rng = np.random.default_rng(0)
st = np.zeros((rows, 70), np.float32)
st[:, :64] = rng.integers(0, 2, (rows, 64))
st[:, 64:] = rng.integers(0, 5, (rows, 6)) * 0.25
t = Trainer(max(rows // 16, 64), "", 1, 50, 16, 1.0, 0.25, 0, 1, False, stagger=False, _cdll=L)
t.set_net(NET_RESCNN4_H3, nets.init_rescnn4(0, bn_noise=True))
for _ in range(2):
    ms = t.net_bench(st, reps=reps)
print("%s: %d rows %.4f ms per launch" % (lib or "product library", rows, ms))
t.close()
