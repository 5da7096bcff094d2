#!/usr/bin/env python3
"""Experiment: are two games of the same pairing and the same match seed, played with network evaluations that differ
in the last bits, the same game?  The reference's tournament seeds its matches from a default-constructed generator
in addMatch order, so every worker file of rating/round.py re-uses the same ~114 seeds; its TFLite evaluations are
batched, i.e. a row's bits depend on what else is in the batch.  Here: the pairing at seed positions 0..N-1 with the
four float32-class network kinds (different arithmetic, errors ~1e-6): per seed position, do the four games agree?
usage (GPU box): python tools/exp/seed_corr.py [a b] [N]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from corintho_ai_amd import Tourney  # noqa: E402
from tools.exp.ref_rows import PLAYER_MODEL, REF_ROWS, weights  # noqa: E402

a, b = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 0)
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1140
W = weights()
KINDS = {"fp32": 1, "bf16x6": 6, "f16x3": 9, "bf16x3": 4}
res = {}
for name, kind in KINDS.items():
    t = Tourney(1, "")
    for p in (a, b):
        t.addPlayer(p, PLAYER_MODEL[p], 1600, 16, 3.0, 0.25, False)
    for _ in range(N):
        t.addMatch(a, b, False)
    t.set_exact_offsets(True)
    for p in (a, b):
        t.set_net(PLAYER_MODEL[p], kind, W[PLAYER_MODEL[p]])
    assert t.run()
    res[name] = np.array([t.match_score(i) for i in range(N)])
    plies = np.array([t.match_info(i)["plies"] for i in range(N)])
    print("%-7s first player wins %.3f, draws %.3f, mean plies %.1f" % (name, np.mean(res[name] == 1.0), np.mean(res[name] == 0.5), plies.mean()))
    t.close()
names = list(KINDS)
M = np.stack([res[n] for n in names])
print("pairing %d %d, %d seed positions; reference row: %s" % (a, b, N, REF_ROWS[(a, b)]))
for i in range(len(names)):
    for j in range(i + 1, len(names)):
        same = np.mean(M[i] == M[j])
        p1, p2 = np.mean(M[i] == 1.0), np.mean(M[j] == 1.0)
        indep = p1 * p2 + (1 - p1) * (1 - p2)
        print("%-7s vs %-7s: same result at the same seed %.3f (independent games would agree %.3f)" % (names[i], names[j], same, indep))
wins = (M == 1.0).astype(float)
per_seed = wins.mean(axis=0)
p = wins.mean()
print("variance of the per-seed mean over the %d kinds: %.4f; binomial (independent) %.4f; fully correlated %.4f"
      % (len(names), per_seed.var(), p * (1 - p) / len(names), p * (1 - p)))
dr = (M == 0.5)
print("seed positions with a draw in at least one kind: %d; in all kinds: %d" % (int(dr.any(axis=0).sum()), int(dr.all(axis=0).sum())))
