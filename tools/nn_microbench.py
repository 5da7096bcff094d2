#!/usr/bin/env python3
"""Diagnostic: time one network kernel in isolation on a fixed batch (rows), optionally from
a library built with extra -D flags (experiments).  usage: nn_microbench.py kind rows [flags...]"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from corintho_ai_amd import Trainer, _lib, build, nets  # noqa: E402

kind = int(sys.argv[1])
rows = int(sys.argv[2])
flags = sys.argv[3:]
L = None
if flags:
    out = os.path.join(ROOT, "gpurun_out", "libcorintho_hip_exp.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call([build.hipcc()] + build.FLAGS + flags + ["-o", out] +
                          [os.path.join(build.CSRC, s) for s in build.SOURCES])
    L = _lib.declare(C.CDLL(out))
G = (rows + 15) // 16
t = Trainer(G, "", 1, 50, 16, 1.0, 0.25, 0, 1, False, _cdll=L)
w = nets.init_mlp12x100(0) if kind in (1, 4, 6) else nets.init_rescnn4(0)
t.set_net(kind, w)
rng = np.random.default_rng(0)
s = np.zeros((rows, 70), np.float32)
s[:, :64] = rng.integers(0, 2, (rows, 64))
s[:, 64:] = rng.integers(0, 5, (rows, 6)) * 0.25
ms = t.net_bench(s, reps=int(os.environ.get("NN_REPS", "20")))
flop = (nets.rescnn4_flop_per_row() if kind not in (1, 4, 6) else 253400.0) * rows
print("kind %d rows %d flags %s: %.3f ms per launch, %.1f TFLOP/s algorithmic" % (kind, rows, flags, ms, flop / ms / 1e9))
