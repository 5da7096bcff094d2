#!/usr/bin/env python3
"""Replay rows of the reference's rating tournament (corintho_ai/rating/results.txt) on the MI355X in several
tournament SHAPES, to find out what those rows measure:

  homog   N matches of the one pairing (a, b), the reference's offset table (tourney.cpp:55-62, quirk 10)
  exact   the same, every match reading its own rows (ca_tourney_set_exact_offsets)
  pairs   the match-file shape of rating/round.py:206-214: (a, b), (b, a), (a, b), ... with the reference's table
  blind   exact offsets, but the network replaced by zero weights (uniform priors, value 0): what a search
          that learns nothing from its evaluations scores

usage (GPU box): python tools/exp/ref_rows.py [N] [shape ...]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from corintho_ai_amd import NET_MLP12X100_X6, Tourney  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
# player id -> model id (rating/tourney/players.txt), all `1600 16 3.0 0.25 0`; 96 = the random player
PLAYER_MODEL = {0: 93, 1: 92, 46: 47, 89: 4, 90: 3}
# rows of rating/results.txt: (first player, second player): (wins, draws, losses) of the FIRST player
REF_ROWS = {
    (0, 1): (393, 46, 865), (1, 0): (368, 9, 927),
    (0, 89): (625, 0, 283), (89, 0): (69, 17, 822),
    (0, 46): (706, 13, 772), (46, 0): (546, 0, 945),
    (46, 90): (627, 3, 402), (90, 46): (107, 7, 916),
    (1, 46): (689, 12, 762), (46, 1): (486, 6, 971),
    (0, 90): (586, 6, 354), (90, 0): (70, 3, 873),
    (89, 90): (460, 25, 742), (90, 89): (243, 0, 983),
    (0, 96): (76, 0, 0), (96, 0): (0, 0, 76),
    (46, 96): (77, 0, 0), (96, 46): (0, 0, 76),
    (90, 96): (76, 0, 0), (96, 90): (0, 0, 76),
    (89, 96): (76, 0, 0), (96, 89): (0, 0, 76),
}

# the six ordered pairs of the five checkpoints that the 14 rows above leave out (round 3, after the fact: with them
# every pairing among the committed checkpoints has been replayed).  REF_ROWS_SET=extra selects them.
EXTRA_ROWS = {
    (1, 89): (580, 15, 327), (89, 1): (59, 13, 850),
    (1, 90): (536, 8, 422), (90, 1): (67, 3, 896),
    (46, 89): (493, 6, 494), (89, 46): (74, 2, 916),
}
if os.environ.get("REF_ROWS_SET") == "extra":
    REF_ROWS = EXTRA_ROWS


def weights():
    w = {}
    for tag, mid in (("early", 3), ("middle", 47), ("last", 93)):
        w[mid] = np.load(os.path.join(GOLDEN, "trained_%s.npz" % tag))["weights"]
    d = np.load(os.path.join(GOLDEN, "ref_models.npz"))
    for k in d.files:
        w[int(k.split("_")[1])] = d[k]
    return w


def play(a, b, n, shape, W, kind=NET_MLP12X100_X6):
    t = Tourney(1, "")
    for p in (a, b):
        if p == 96:
            t.addPlayer(96, -1, 1600, 16, 3.0, 0.25, True)
        else:
            t.addPlayer(p, PLAYER_MODEL[p], 1600, 16, 3.0, 0.25, False)
    if shape == "pairs":
        for _ in range(n):
            t.addMatch(a, b, False)
            t.addMatch(b, a, False)
    else:
        for _ in range(n):
            t.addMatch(a, b, False)
    if shape in ("exact", "blind"):
        t.set_exact_offsets(True)
    for p in (a, b):
        if p != 96:
            w = W[PLAYER_MODEL[p]]
            t.set_net(PLAYER_MODEL[p], kind, np.zeros_like(w) if shape == "blind" else w)
    t0 = time.time()
    assert t.run()
    dt = time.time() - t0
    m = t.num_matches()
    sc = np.array([t.match_score(i) for i in range(m)])
    t.close()
    if shape == "pairs":
        ab, ba = sc[0::2], sc[1::2]
        return {(a, b): wdl(ab), (b, a): wdl(ba)}, dt
    return {(a, b): wdl(sc)}, dt


def wdl(sc):
    return int(np.sum(sc == 1.0)), int(np.sum(sc == 0.5)), int(np.sum(sc == 0.0))


def z_two_sample(x, y):
    """per-outcome two-sample z of the win fraction and the chi-square (2 dof) of the W/D/L table"""
    nx, ny = sum(x), sum(y)
    px, py = x[0] / nx, y[0] / ny
    p = (x[0] + y[0]) / (nx + ny)
    z = (px - py) / max(np.sqrt(p * (1 - p) * (1 / nx + 1 / ny)), 1e-12)
    chi = 0.0
    for k in range(3):
        tot = x[k] + y[k]
        if tot == 0:
            continue
        ex, ey = tot * nx / (nx + ny), tot * ny / (nx + ny)
        chi += (x[k] - ex) ** 2 / ex + (y[k] - ey) ** 2 / ey
    return z, chi


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    shapes = sys.argv[2:] or ["homog", "exact", "pairs", "blind"]
    W = weights()
    out = {}
    for shape in shapes:
        done = set()
        for (a, b), ref in REF_ROWS.items():
            if (a, b) in done:
                continue
            if shape == "blind" and 96 in (a, b):
                continue
            res, dt = play(a, b, n, shape, W)
            for key, got in res.items():
                done.add(key)
                r = REF_ROWS[key]
                z, chi = z_two_sample(got, r)
                out["%s %d %d" % (shape, key[0], key[1])] = {"got": got, "ref": r, "z_win": z, "chi2": chi, "seconds": dt}
                print("%-6s %2d %2d  here %5d/%4d/%5d = %.3f %.3f   ref %4d/%3d/%4d = %.3f %.3f   z %+5.1f chi2 %6.1f  (%.1f s)"
                      % (shape, key[0], key[1], got[0], got[1], got[2], got[0] / sum(got), got[1] / sum(got), r[0], r[1], r[2],
                         r[0] / sum(r), r[1] / sum(r), z, chi, dt), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "ref_rows_%s%d.json" % (os.environ.get("REF_ROWS_SET", ""), n)), "w"), indent=1)


if __name__ == "__main__":
    main()
