#!/usr/bin/env python3
"""Experiment: were the result-changing quirks of the committed search in the code that produced the
reference's rating/results.txt?  Rows of it replayed with engine builds in which ONE quirk is replaced by its evident
intent (quirk 1: propagateTerminal looks at the child for a draw; quirk 6: a drawn child's exploration term is
divided by n + 1 like the others).  A build that misses rows the product reproduces shows the quirk was there.
The product sources carry no experiment switch: `--build 1|6` (build container) copies corintho_ai_amd/csrc to
build_ab/ablate_q<k>/, rewrites the one expression there and builds build_ab/libcorintho_hip_q<k>.so.
usage: python tools/exp/quirk_ablation.py --build 1|6          (here)
       python tools/exp/quirk_ablation.py lib.so [N]            (GPU box)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

Q1_ANCHOR = "      co_store_unit(A, path_slot[d], co_slot_set_result(s, has_draw ? CO_DEDUCED_DRAW : CO_DEDUCED_LOSS));\n"
Q1_CODE = """      has_draw = 0; /* ablation: the evident intent -- a drawn CHILD */
      for (int base = 0; base < n; base += CO_WAVE) {
        LV(int, dr);
        FOR_LANES {
          int e = base + lane;
          L(dr) = 0;
          if (e < n) L(dr) = co_res_drawn(co_slot_result(A[pb + 2 + e]));
        }
        if (WAVE_BALLOT(dr)) has_draw = 1;
      }
"""
Q6_ANCHOR = "          uc = searchable ? uc : CO_NEG_INF;\n"
Q6_CODE = "          uc = drawn ? (float)b : uv; /* ablation: the drawn child's term divided like the others */\n"


def build_ablated(which):
    import shutil
    import subprocess

    from corintho_ai_amd import build

    src = os.path.join(ROOT, "build_ab", "ablate_q%d" % which)
    shutil.rmtree(src, ignore_errors=True)
    shutil.copytree(build.CSRC, src)
    path = os.path.join(src, "mcts.h")
    text = open(path).read()
    anchor, code = (Q1_ANCHOR, Q1_CODE) if which == 1 else (Q6_ANCHOR, Q6_CODE)
    assert text.count(anchor) == 1, "mcts.h no longer holds the expression this experiment rewrites"
    open(path, "w").write(text.replace(anchor, code + anchor))
    out = os.path.join(ROOT, "build_ab", "libcorintho_hip_q%d.so" % which)
    subprocess.check_call([build.hipcc()] + build.FLAGS + ["-I", os.path.join(ROOT, "include"), "-o", out] +
                          [os.path.join(src, f) for f in build.SOURCES])
    return out


if sys.argv[1] == "--build":
    print(build_ablated(int(sys.argv[2])))
    sys.exit(0)
from corintho_ai_amd import NET_MLP12X100_X6, Tourney, _lib  # noqa: E402
from tools.exp.ref_rows import PLAYER_MODEL, REF_ROWS, weights, wdl, z_two_sample  # noqa: E402

lib = _lib.declare(C.CDLL(os.path.abspath(sys.argv[1])))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
W = weights()
PAIRS = [(0, 1), (1, 0), (0, 89), (89, 0), (0, 46), (46, 0), (46, 90), (90, 46), (89, 90), (90, 89)]
zs = []
tot = np.zeros(3)
for a, b in PAIRS:
    t = Tourney(1, "", _cdll=lib)
    for p in (a, b):
        t.addPlayer(p, PLAYER_MODEL[p], 1600, 16, 3.0, 0.25, False)
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
    tot += got
    r = REF_ROWS[(a, b)]
    print("%-14s %2d %2d  here %.3f (draws %.4f)  ref %.3f (draws %.4f)  z %+5.1f" %
          (os.path.basename(sys.argv[1]), a, b, got[0] / N, got[1] / N, r[0] / sum(r), r[1] / sum(r), z), flush=True)
ref = np.sum([REF_ROWS[p] for p in PAIRS], axis=0)
print("%-14s sum z^2 over %d rows = %.1f; draws %.4f here, %.4f in the reference" %
      (os.path.basename(sys.argv[1]), len(zs), float(np.sum(np.array(zs) ** 2)), tot[1] / tot.sum(), ref[1] / ref.sum()), flush=True)
