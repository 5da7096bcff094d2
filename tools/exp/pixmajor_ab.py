#!/usr/bin/env python3
"""Experiment: the pixel-major f16x3 throughput kernel (K6p, nn_rescnn.hip co_k_rescnn_forward_h3p) against the
(position, pixel)-column kernel it replaces.  The product library holds the pixel-major kernel only (round 5: no
environment switch, no dead kernel in the shipped library); the other one comes from a diagnostic build,
    tools/build_variant.sh colmajor -DCO_RESCNN_PIXMAJOR_DEFAULT=0
Prints kernel-only ms per evaluation for several batch sizes and checks that the two kernels give the same bits.
usage (GPU box): python tools/exp/pixmajor_ab.py build_ab/colmajor.so [product.so]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from corintho_ai_amd import NET_RESCNN4_H3, Trainer, _lib, nets  # noqa: E402

LIBS = {"0": _lib.declare(C.CDLL(os.path.abspath(sys.argv[1]))),
        "1": _lib.declare(C.CDLL(os.path.abspath(sys.argv[2]))) if len(sys.argv) > 2 else None}

rng = np.random.default_rng(0)
R = 65536
st = np.zeros((R, 70), np.float32)
st[:, :64] = rng.integers(0, 2, (R, 64))
st[:, 64:] = rng.integers(0, 5, (R, 6)) * 0.25
w = nets.init_rescnn4(0, bn_noise=True)
out = {}
for pm in ("0", "1"):
    t = Trainer(R // 16, "", 1, 50, 16, 1.0, 0.25, 0, 1, False, stagger=False, _cdll=LIBS[pm])
    t.set_net(NET_RESCNN4_H3, w)
    ev, pr = t.net_forward(st[:20000])
    out[pm] = (ev, pr)
    for rows in (8192, 10000, 12288, 14000, 16384, 20000, 32768, 65536):
        for _ in range(2):
            ms = t.net_bench(st[:rows], reps=200)
        print("pixel-major %s: %6d rows %.4f ms" % (pm, rows, ms), flush=True)
    t.close()
same = np.array_equal(out["0"][0], out["1"][0]) and np.array_equal(out["0"][1], out["1"][1])
print("outputs of 20000 rows bit-identical between the two kernels:", same)
print("max |dv| %.3g, max |dp| %.3g" % (np.max(np.abs(out["0"][0] - out["1"][0])), np.max(np.abs(out["0"][1] - out["1"][1]))))
sys.exit(0 if same else 1)
