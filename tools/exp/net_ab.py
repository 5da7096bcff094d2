#!/usr/bin/env python3
"""Experiment helper: the same network kind through two builds of the library -- kernel-only ms per evaluation and whether
the outputs agree bit for bit.  usage (GPU box): python tools/exp/net_ab.py libA.so libB.so [mlp12x100h3|rescnn4h3|...] [rows]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from corintho_ai_amd import Trainer, _lib, nets  # noqa: E402

kind_name = sys.argv[3] if len(sys.argv) > 3 else "mlp12x100h3"
rows = int(sys.argv[4]) if len(sys.argv) > 4 else 20000
kind = {"mlp12x100": 1, "mlp12x100x3": 4, "rescnn4": 2, "rescnn4x3": 3, "rescnn4h3": 8, "mlp12x100h3": 9, "rescnn4x6": 5, "mlp12x100x6": 6}[kind_name]
w = nets.init_mlp12x100(0) if kind_name.startswith("mlp") else nets.init_rescnn4(0, bn_noise=True)
rng = np.random.default_rng(0)
st = np.zeros((rows, 70), np.float32)
st[:, :64] = rng.integers(0, 2, (rows, 64))
st[:, 64:] = rng.integers(0, 5, (rows, 6)) * 0.25
out = []
for lib in sys.argv[1:3]:
    L = _lib.declare(C.CDLL(os.path.abspath(lib)))
    t = Trainer((rows + 15) // 16, "", 1, 50, 16, 1.0, 0.25, 0, 1, False, stagger=False, _cdll=L)
    t.set_net(kind, w)
    out.append(t.net_forward(st))
    for _ in range(2):
        ms = t.net_bench(st, reps=200)
    print("%s: %s, %d rows %.4f ms" % (lib, kind_name, rows, ms), flush=True)
    t.close()
same = np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
print("outputs bit-identical:", same)
sys.exit(0 if same else 1)
