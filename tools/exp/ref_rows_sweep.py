#!/usr/bin/env python3
"""Experiment: is the one row of rating/results.txt the engine does not reproduce (`1 0`) explained by a search setting
that differed when the reference played its tournament?  The same pairings with the Dirichlet noise off and with other
exploration constants, every match reading its own rows.  usage (GPU box): python tools/exp/ref_rows_sweep.py [N]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from corintho_ai_amd import NET_MLP12X100_X6, Tourney  # noqa: E402
from tools.exp.ref_rows import PLAYER_MODEL, REF_ROWS, weights, wdl, z_two_sample  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
W = weights()
PAIRS = [(0, 1), (1, 0), (0, 89), (89, 0), (0, 46), (46, 0), (1, 46), (46, 1)]
SETTINGS = [("players.txt: c_puct 3.0, eps 0.25", 3.0, 0.25), ("no noise: eps 0", 3.0, 0.0), ("c_puct 1.0", 1.0, 0.25),
            ("c_puct 2.0", 2.0, 0.25), ("c_puct 4.0", 4.0, 0.25), ("eps 0.5", 3.0, 0.5)]
for name, cp, eps in SETTINGS:
    zs = []
    for a, b in PAIRS:
        t = Tourney(1, "")
        for p in (a, b):
            t.addPlayer(p, PLAYER_MODEL[p], 1600, 16, cp, eps, False)
        for _ in range(N):
            t.addMatch(a, b, False)
        t.set_exact_offsets(True)
        for p in (a, b):
            t.set_net(PLAYER_MODEL[p], NET_MLP12X100_X6, W[PLAYER_MODEL[p]])
        assert t.run()
        got = wdl(np.array([t.match_score(i) for i in range(N)]))
        t.close()
        z, chi = z_two_sample(got, REF_ROWS[(a, b)])
        zs.append(z)
        r = REF_ROWS[(a, b)]
        print("%-36s %2d %2d  here %.3f (draws %.3f)  ref %.3f (draws %.3f)  z %+5.1f" %
              (name, a, b, got[0] / N, got[1] / N, r[0] / sum(r), r[1] / sum(r), z), flush=True)
    print("%-36s sum z^2 over %d rows = %.1f" % (name, len(zs), float(np.sum(np.array(zs) ** 2))), flush=True)
