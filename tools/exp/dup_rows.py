#!/usr/bin/env python3
"""Experiment: how many of a generation's network requests ask for a position that was asked for before?
Host-driven protocol (the rows are visible).  Counts, over a whole generation:
  batch-unique   rows that are the first of their position within their own batch
  new            rows whose position was never requested before in the generation (any game, any earlier batch)
  own-game       repeated rows whose earlier request came from the SAME game (the two players' trees of a game search
                 overlapping positions: players_[2] share nothing, selfplayer.h:86-111)
usage: python tools/exp/dup_rows.py [games] [--stagger] [--net rescnn4x6|mlp12x100x6] [--trained] [--sims N]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from corintho_ai_amd import NET_MLP12X100_X6, NET_RESCNN4_X6, Trainer, nets  # noqa: E402


def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


G = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4096
SPE = 16
SIMS = int(arg("--sims", "400"))
net = arg("--net", "rescnn4x6")
t = Trainer(G, "", 7, SIMS, SPE, 1.0, 0.25, 0, 1, False, stagger=("--stagger" in sys.argv), trace=False)
if net == "rescnn4x6":
    t.set_net(NET_RESCNN4_X6, nets.init_rescnn4(0))
elif "--trained" in sys.argv:
    t.set_net(NET_MLP12X100_X6, np.load(os.path.join(ROOT, "tests", "golden", "trained_last.npz"))["weights"])
else:
    t.set_net(NET_MLP12X100_X6, nets.init_mlp12x100(0))
states = np.zeros((G * SPE, 70), np.float32)
evals = np.zeros((G * SPE,), np.float32)
probs = np.zeros((G * SPE, 96), np.float32)
tot = uniq_batch = it = 0
hist = []
all_h, all_g = [], []
rng = np.random.default_rng(1)
MULT = rng.integers(1, 2 ** 63, size=35, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
SMALL = G <= 256


def row_hash(rows):
    """64-bit hash of each 70-float row (35 words of 64 bits, odd multipliers, one finaliser)"""
    v = rows.view(np.uint64).reshape(rows.shape[0], 35)
    with np.errstate(over="ignore"):
        h = (v * MULT).sum(axis=1, dtype=np.uint64)
        h ^= h >> np.uint64(31)
        h *= np.uint64(0x9E3779B97F4A7C15)
        h ^= h >> np.uint64(29)
    return h


while not t.doIteration(evals, probs, -1):
    n = t.num_requests(-1)
    t.writeRequests(states, -1)
    if n == 0:
        continue
    h = row_hash(states[:n])
    u = len(np.unique(h))
    tot += n
    uniq_batch += u
    all_h.append(h)
    if SMALL:  # rows come game-major: row -> game through the per-game request counts
        counts = np.array([t.game_info(g)["n_pending"] for g in range(G)])
        all_g.append(np.repeat(np.arange(G, dtype=np.uint64), counts))
    it += 1
    if it % 50 == 1:
        hist.append((it, n, u))
    t.net_forward(states[:n], out_evals=evals, out_probs=probs)
H = np.concatenate(all_h)
new = len(np.unique(H))
print("%s%s, %d games x %d sims, %s: iterations %d, rows %d, first of their position in their batch %d (%.1f %%), "
      "distinct positions in the generation %d (%.1f %% of the rows)"
      % (net, " trained" if "--trained" in sys.argv else "", G, SIMS, "stagger" if "--stagger" in sys.argv else "no stagger", it, tot,
         uniq_batch, 100.0 * uniq_batch / max(tot, 1), new, 100.0 * new / max(tot, 1)))
if SMALL:
    Gs = np.concatenate(all_g)
    with np.errstate(over="ignore"):
        per_game = len(np.unique(H * np.uint64(1000003) + Gs))
    print("distinct (game, position) pairs %d (%.1f %% of the rows): the rest are positions a game asks for more than once "
          "(its two players' trees overlap)" % (per_game, 100.0 * per_game / tot))
print("every 50th iteration (iteration, rows, unique in batch):", hist)
