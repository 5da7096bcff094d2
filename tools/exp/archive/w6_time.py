#!/usr/bin/env python3
"""Experiment: kind 7 (Winograd) -- error against float64 on one weight set and launch times."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from corintho_ai_amd import Trainer, nets  # noqa: E402
from tests import ref_nets
import _wino_lib  # noqa: E402

L = _wino_lib.load()

rng = np.random.default_rng(5)
kinds = [int(k) for k in sys.argv[1:]] or [7]


def states(n):
    s = np.zeros((n, 70), np.float32)
    s[:, :64] = rng.integers(0, 2, (n, 64))
    s[:, 64:] = rng.integers(0, 5, (n, 6)) * 0.25
    return s


t = Trainer(8192, "", 1, 50, 16, 1.0, 0.25, 0, 1, False, _cdll=L)
st = states(777)
w = nets.trained_like_rescnn4(1)
want = ref_nets.rescnn4_forward_f64(w, st)
out = []
for kind in kinds:
    t.set_net(kind, w)
    ev, pr = t.net_forward(st)
    out.append("kind %d: err value %.2e policy %.2e" % (kind, np.max(np.abs(ev - want[0])), np.max(np.abs(pr - want[1]))))
    for rows in (1024, 8192, 65536):
        ms = t.net_bench(states(rows), reps=10)
        out.append("%d rows %.3f ms" % (rows, ms))
print("; ".join(out), flush=True)
