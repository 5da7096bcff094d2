#!/usr/bin/env python3
"""Diagnostic: build the engine with -DCO_PROF into a separate library and print where the
cycles of a split-precision residual-CNN kernel go (usage: prof_nn.py [rows] [NET_RESCNN4_H3 | _X3 | _X6]) (stamps of wave 0 of every workgroup).  Never used
by the product."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import corintho_ai_amd as CA  # noqa: E402
from corintho_ai_amd import Trainer, _lib, build, nets  # noqa: E402

out = os.path.join(ROOT, "gpurun_out", "libcorintho_hip_prof.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
cmd = [build.hipcc()] + build.FLAGS + ["-DCO_PROF"] + sys.argv[3:] + ["-o", out] + [os.path.join(build.CSRC, s) for s in build.SOURCES]
subprocess.check_call(cmd)
L = _lib.declare(C.CDLL(out))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
t = Trainer(rows // 16, "", 1, 50, 16, 1.0, 0.25, 0, 1, False, stagger=False, _cdll=L)
kind = sys.argv[2] if len(sys.argv) > 2 else "NET_RESCNN4_H3"
t.set_net(getattr(CA, kind), nets.init_rescnn4(0))
rng = np.random.default_rng(0)
st = np.zeros((rows, 70), np.float32)
st[:, :64] = rng.integers(0, 2, (rows, 64))
st[:, 64:] = rng.integers(0, 5, (rows, 6)) * 0.25
ms = t.net_bench(st, reps=int(os.environ.get("NN_REPS", "20")))
p = (C.c_ulonglong * 12)()
L.ca_net_prof.argtypes = [C.POINTER(C.c_ulonglong)]
assert L.ca_net_prof(p) == 0
v = [int(x) for x in p]
n = max(v[7], 1)
names = ["input+pack", "stem conv", "epilogue+pack (9x)", "64ch convs (8x)", "1x1 heads", "dense heads+softmax"]
print("%s, %d rows: %.3f ms per launch; %d workgroup passes stamped" % (kind, rows, ms, n))
for i, nm in enumerate(names):
    print("  %-24s %9.0f cycles  %5.1f%%" % (nm, v[i] / n, 100.0 * v[i] / max(v[6], 1)))
print("  %-24s %9.0f cycles in %.1f us: in-kernel clock %.2f GHz" % ("whole workgroup", v[6] / n, v[8] / n / 100.0, v[6] / max(v[8], 1) * 0.1))
if v[9] + v[10] + v[11]:
    print("  pixel-major kernel, inside the convolutions: waited for the weight DMA %.0f, at the tap barrier %.0f, multiplied %.0f cycles"
          % (v[9] / n, v[10] / n, v[11] / n))
if os.environ.get("NN_TRACE"):
    tr = (C.c_uint * 160)()
    L.ca_net_trace.argtypes = [C.POINTER(C.c_uint)]
    assert L.ca_net_trace(tr) == 0
    T = np.array(list(tr), dtype=np.int64).reshape(8, 20)
    t0 = T[:, 0].min()
    print("  workgroup 0, one trunk convolution: core cycles since the first wave reached tap 0's barrier")
    print("  tap   " + "".join("   wave %d: arrived, left " % w for w in range(8)))
    for tap in range(9):
        print("  %d     " % tap + "".join("   %9d %9d     " % ((T[w, 2 * tap] - t0) & 0xFFFFFFFF, (T[w, 2 * tap + 1] - t0) & 0xFFFFFFFF) for w in range(8)))
    print("  end   " + "".join("   %9d               " % ((T[w, 18] - t0) & 0xFFFFFFFF) for w in range(8)))
    last = [(int(np.argmax([(T[w, 2 * tap] - t0) & 0xFFFFFFFF for w in range(8)]))) for tap in range(9)]
    print("  the wave that arrived last at each tap's barrier:", last)
