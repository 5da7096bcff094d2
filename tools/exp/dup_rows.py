#!/usr/bin/env python3
"""Experiment: how many of an iteration's network requests are duplicates of another row of the same batch?
Host-driven protocol (the rows are visible), 4096 games x 400 simulations, rescnn4x6."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from corintho_ai_amd import NET_RESCNN4_X6, Trainer, nets  # noqa: E402

G, SPE = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 16
t = Trainer(G, "", 7, 400, SPE, 1.0, 0.25, 0, 1, False, stagger=("--stagger" in sys.argv))
t.set_net(NET_RESCNN4_X6, nets.init_rescnn4(0))
states = np.zeros((G * SPE, 70), np.float32)
evals = np.zeros((G * SPE,), np.float32)
probs = np.zeros((G * SPE, 96), np.float32)
tot = uniq = it = 0
n = 0
hist = []
while not t.doIteration(evals, probs, -1):
    n = t.num_requests(-1)
    t.writeRequests(states, -1)
    if n == 0:
        continue
    u = len(np.unique(states[:n].view(np.dtype((np.void, 280))).ravel()))
    tot += n
    uniq += u
    it += 1
    if it % 50 == 1:
        hist.append((it, n, u))
    t.net_forward(states[:n], out_evals=evals, out_probs=probs)
print("iterations %d, rows %d, unique within their batch %d (%.1f %%)" % (it, tot, uniq, 100.0 * uniq / max(tot, 1)))
print("every 50th iteration (iteration, rows, unique):", hist)
