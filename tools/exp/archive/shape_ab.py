#!/usr/bin/env python3
"""Experiment: two prebuilt engine libraries, one network kind: are the outputs bit-identical, and what does a launch take
under sustained back-to-back launches (the chip's clock settles only after seconds of load)?
usage: shape_ab.py libA.so libB.so [kind] [rows] [reps]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from corintho_ai_amd import Trainer, _lib, nets  # noqa: E402

libs = sys.argv[1:3]
kind = int(sys.argv[3]) if len(sys.argv) > 3 else 8
rows = int(sys.argv[4]) if len(sys.argv) > 4 else 24576
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3000
rng = np.random.default_rng(0)
s = np.zeros((rows, 70), np.float32)
s[:, :64] = rng.integers(0, 2, (rows, 64))
s[:, 64:] = rng.integers(0, 5, (rows, 6)) * 0.25
w = nets.init_rescnn4(3, bn_noise=True)
outs = []
for path in libs:
    L = _lib.declare(C.CDLL(os.path.abspath(path)))
    t = Trainer((rows + 15) // 16, "", 1, 50, 16, 1.0, 0.25, 0, 1, False, _cdll=L)
    t.set_net(kind, w)
    res = []
    for n in (rows, 4096, 2048, 100):  # throughput, small-batch and thin paths
        ev, pr = t.net_forward(s[:n])
        res.append((ev.copy(), pr.copy()))
    outs.append(res)
    for _ in range(2):
        print("%-24s %d rows: %.4f ms per launch (%d back-to-back)" % (os.path.basename(path), rows, t.net_bench(s, reps=reps), reps))
for i, n in enumerate((rows, 4096, 2048, 100)):
    same = outs[0][i][0].tobytes() == outs[1][i][0].tobytes() and outs[0][i][1].tobytes() == outs[1][i][1].tobytes()
    d = max(float(np.max(np.abs(outs[0][i][0] - outs[1][i][0]))), float(np.max(np.abs(outs[0][i][1] - outs[1][i][1]))))
    print("%6d rows: outputs %s (max abs difference %.3g)" % (n, "bit-identical" if same else "DIFFER", d))
# within one library: a row's result must not depend on the batch it came in
for k, res in enumerate(outs):
    ok = all(res[0][0][:n].tobytes() == res[i][0].tobytes() and res[0][1][:n].tobytes() == res[i][1].tobytes()
             for i, n in ((1, 4096), (2, 2048), (3, 100)))
    print("%-24s batch-invariant: %s" % (os.path.basename(libs[k]), ok))
