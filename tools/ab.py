#!/usr/bin/env python3
"""Diagnostic A/B: time whole generations with two prebuilt engine libraries in the same process
on the same GPU, alternating, so that box-to-box variance cancels.
usage: ab.py libA.so libB.so [libC.so ...] [net] [games] [reps] [pools]
A library may carry environment settings that apply while ITS trainer is created and its network set (switches read
there): lib.so:VAR=VAL[,VAR2=VAL2]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from corintho_ai_amd import NET_MLP12X100, NET_RESCNN4, NET_RESCNN4_X3, Trainer, _lib, nets  # noqa: E402

libs = [a for a in sys.argv[1:] if ".so" in a]
rest = [a for a in sys.argv[1:] if ".so" not in a]
net = rest[0] if len(rest) > 0 else "rescnn4x3"
G = int(rest[1]) if len(rest) > 1 else 4096
reps = int(rest[2]) if len(rest) > 2 else 3
pools = int(rest[3]) if len(rest) > 3 else 1
kind = {"mlp12x100": NET_MLP12X100, "mlp12x100x3": 4, "rescnn4": NET_RESCNN4, "rescnn4x3": NET_RESCNN4_X3, "rescnn4h3": 8, "mlp12x100h3": 9,
        "rescnn4x6": 5, "mlp12x100x6": 6}[net]
w = nets.init_mlp12x100(0) if net.startswith("mlp12x100") else nets.init_rescnn4(0)
if os.environ.get("AB_TRAINED") and net.startswith("mlp12x100"):  # the reference's last checkpoint instead of random init: narrow, deep trees
    import numpy as np
    w = np.load(os.path.join(ROOT, "tests", "golden", "trained_last.npz"))["weights"]
ts = []
for spec in libs:
    path, _, envs = spec.partition(":")
    for kv in filter(None, envs.split(",")):
        os.environ[kv.split("=")[0]] = kv.split("=")[1]
    L = _lib.declare(C.CDLL(os.path.abspath(path)))
    kw = {"step_budget": int(os.environ["AB_BUDGET"])} if "AB_BUDGET" in os.environ else {}  # lib.so:AB_BUDGET=96 (a library that knows the field)
    t = Trainer(G, "", 12345, 400, 16, 1.0, 0.25, 0, 1, False, stagger=False, pools=pools, _cdll=L, **kw)
    t.set_net(kind, w)
    t.reset(1)
    t.run()
    ts.append(t)
    for kv in filter(None, envs.split(",")):
        del os.environ[kv.split("=")[0]]
acc = [[0.0, 0.0, 0.0] for _ in libs]
extra = ["" for _ in libs]
for r in range(reps):
    for i, t in enumerate(ts):
        t.reset(100 + r)
        t0 = time.perf_counter()
        t.run()
        dt = time.perf_counter() - t0
        st = t.stats()
        acc[i][0] += dt * 1e3
        acc[i][1] += st["mcts_ms"]
        acc[i][2] += st["nn_ms"]
        acc[i].append(st.get("nn_rows_evaluated", 0) / max(st.get("nn_rows", 1), 1))
        extra[i] = "iterations %d, steps cut %d, last budget %d" % (st["iterations"], st.get("steps_cut", 0), st.get("step_budget_last", 0))
for path, a, x in zip(libs, acc, extra):
    print("%-44s wall %.1f ms  search %.1f ms  network %.1f ms   (%d games, %s, %d pool(s), mean of %d; rows evaluated %.4f of the requested; %s)" %
          (os.path.basename(path), a[0] / reps, a[1] / reps, a[2] / reps, G, net, pools, reps, sum(a[3:]) / max(len(a[3:]), 1), x))
