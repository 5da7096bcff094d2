#!/usr/bin/env python3
"""Experiment: the Winograd kernel (kind 7) against float64 and beside kinds 2 / 5; launch times."""
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


def states(n):
    s = np.zeros((n, 70), np.float32)
    s[:, :64] = rng.integers(0, 2, (n, 64))
    s[:, 64:] = rng.integers(0, 5, (n, 6)) * 0.25
    return s


t = Trainer(8192, "", 1, 50, 16, 1.0, 0.25, 0, 1, False, _cdll=L)
st = states(777)
for name, w in (("init", nets.init_rescnn4(0)), ("bn-noise", nets.init_rescnn4(3, bn_noise=True)),
                ("trained-like-0", nets.trained_like_rescnn4(0)), ("trained-like-1", nets.trained_like_rescnn4(1))):
    want = ref_nets.rescnn4_forward_f64(w, st)
    line = name
    for kind in (2, 5, 7):
        t.set_net(kind, w)
        ev, pr = t.net_forward(st)
        line += "  kind %d: value %.2e policy %.2e" % (kind, np.max(np.abs(ev - want[0])), np.max(np.abs(pr - want[1])))
    print(line, flush=True)
w = nets.init_rescnn4(0)
for rows in (1024, 8192, 16384, 65536, 131072):
    s = states(rows)
    for kind in (5, 7):
        t.set_net(kind, w)
        ms = t.net_bench(s, reps=10)
        print("rows %6d kind %d: %.3f ms  %.1f Mrows/s" % (rows, kind, ms, rows / ms / 1e3), flush=True)
