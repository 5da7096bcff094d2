#!/usr/bin/env python3
"""The search layer's pin as a POPULATION: every row of the reference's rating tournament
(corintho_ai/rating/results.txt) between the 25 checkpoints committed as data -- 600 ordered pairs, 659 454 reference
games -- replayed on the MI355X at the production setting of rating/tourney/players.txt (1600 simulations per move,
16 per evaluation, c_puct 3.0, epsilon 0.25, testing), every match reading the evaluations of its own requests
(tests/test_reference_results.py explains why), N matches per row.

The players were fixed before any of their rows was replayed (tools/gen_trained_golden.py population_players: the five
of round 3, the next-strongest two, 18 drawn with a fixed seed).  Output: gpurun_out/ref_population_<N>[_part].json and a
markdown table; tools/ref_population.py --report file.json ... merges parts and prints the population statistics
(sum of z^2, the count of |z| > 2 / 3 against the normal expectation, the rows by first mover).

usage (GPU box): python tools/ref_population.py N [part nparts] [--net x6|h3]
       (anywhere): python tools/ref_population.py --report gpurun_out/ref_population_2000_*.json > profiles/r04_reference_rows_population.md
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def checkpoints():
    """model id -> weights (engine layout), the 25 checkpoints committed as data"""
    w = {}
    for tag, mid in (("early", 3), ("middle", 47), ("last", 93)):
        w[mid] = np.load(os.path.join(GOLDEN, "trained_%s.npz" % tag))["weights"]
    for name in ("ref_models.npz", "ref_models_pop.npz"):
        d = np.load(os.path.join(GOLDEN, name))
        for k in d.files:
            w[int(k.split("_")[1])] = d[k]
    return w


def reference_rows():
    d = json.load(open(os.path.join(GOLDEN, "ref_results_pop.json")))
    return d["players"], {(r[0], r[1]): tuple(r[2:]) for r in d["rows"]}


def wdl(sc):
    sc = np.asarray(sc)
    return int(np.sum(sc == 1.0)), int(np.sum(sc == 0.5)), int(np.sum(sc == 0.0))


def z_win(x, y):
    """two-sample z of the first player's win fraction (pooled variance)"""
    nx, ny = sum(x), sum(y)
    p = (x[0] + y[0]) / (nx + ny)
    return (x[0] / nx - y[0] / ny) / max(np.sqrt(p * (1 - p) * (1 / nx + 1 / ny)), 1e-12)


def play_row(a, b, n, W, kind):
    from corintho_ai_amd import Tourney

    t = Tourney(1, "")
    for p in (a, b):
        t.addPlayer(p, 93 - p, 1600, 16, 3.0, 0.25, False)
    for _ in range(n):
        t.addMatch(a, b, False)
    t.set_exact_offsets(True)
    for p in (a, b):
        t.set_net(93 - p, kind, W[93 - p])
    assert t.run()
    got = wdl([t.match_score(i) for i in range(n)])
    t.close()
    return got


def run(n, part, nparts, kind_name):
    import corintho_ai_amd as CA

    kind = {"x6": CA.NET_MLP12X100_X6, "h3": CA.NET_MLP12X100_H3}[kind_name]
    W = checkpoints()
    players, rows = reference_rows()
    keys = sorted(rows)
    keys = [k for i, k in enumerate(keys) if i % nparts == part]
    out = {"matches_per_row": n, "net": "mlp12x100" + kind_name, "rows": []}
    path = os.path.join(ROOT, "gpurun_out", "ref_population_%d_%dof%d.json" % (n, part, nparts))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    t0 = time.time()
    for i, (a, b) in enumerate(keys):
        got = play_row(a, b, n, W, kind)
        ref = rows[(a, b)]
        out["rows"].append({"a": a, "b": b, "here": got, "ref": ref, "z": z_win(got, ref)})
        if i % 10 == 9 or i + 1 == len(keys):
            json.dump(out, open(path, "w"))
            print("%d of %d rows, %.0f s: last %d %d here %.3f ref %.3f z %+.2f" %
                  (i + 1, len(keys), time.time() - t0, a, b, got[0] / n, ref[0] / sum(ref), out["rows"][-1]["z"]), flush=True)


def report(paths):
    from scipy.stats import chi2, norm

    rows, n, net = [], None, None
    for p in paths:
        d = json.load(open(p))
        rows += d["rows"]
        n, net = d["matches_per_row"], d["net"]
    rows.sort(key=lambda r: (r["a"], r["b"]))
    z = np.array([r["z"] for r in rows])
    k = len(rows)
    print("# The reference's tournament rows as a population (round 4)\n")
    print("%d rows of `rating/results.txt` between the 25 committed checkpoints, %d matches per row here (%s, every match "
          "reading its own rows), %d reference games.\n" % (k, n, net, sum(sum(r["ref"]) for r in rows)))
    s2 = float(np.sum(z ** 2))
    print("| statistic | value | expected under independence |\n|---|---|---|")
    print("| sum of z^2 over %d rows | %.1f | %d +- %.0f (p = %.2g) |" % (k, s2, k, np.sqrt(2 * k), chi2.sf(s2, k)))
    for thr in (2.0, 3.0, 4.0):
        print("| rows with |z| > %.0f | %d | %.1f |" % (thr, int(np.sum(np.abs(z) > thr)), k * 2 * norm.sf(thr)))
    print("| mean z | %+.3f | 0 +- %.3f |" % (z.mean(), 1 / np.sqrt(k)))
    here = np.sum([r["here"] for r in rows], axis=0)
    ref = np.sum([r["ref"] for r in rows], axis=0)
    print("| pooled draws | %.4f here, %.4f in the reference | |" % (here[1] / here.sum(), ref[1] / ref.sum()))
    # ---- what kind of disagreement is it?  (a) A real strength difference of a pairing on this side would show in BOTH
    # of its rows with opposite signs (if a is too strong here it wins more moving first, and b wins less moving first against
    # it): z(a, b) and -z(b, a) would correlate.  Sampling noise of the reference's rows -- they are not independent games:
    # tests/test_reference_results.py, the seeding of rating/round.py's match files -- leaves them uncorrelated.
    zmap = {(r["a"], r["b"]): r["z"] for r in rows}
    pairs = [(zmap[(a, b)], -zmap[(b, a)]) for (a, b) in zmap if a < b and (b, a) in zmap]
    if len(pairs) > 10:
        pa = np.array(pairs)
        rho = float(np.corrcoef(pa[:, 0], pa[:, 1])[0, 1])
        print("| correlation of z(a, b) with -z(b, a) over %d pairings | %+.3f | 0 +- %.3f if the excess is sampling noise; "
              "near +1 if pairings differed in strength |" % (len(pairs), rho, 1 / np.sqrt(len(pairs))))
    # (b) a checkpoint that plays differently here (import, evaluation) would shift all its rows one way
    phi = s2 / k
    worst = 0.0
    for pl in sorted({r["a"] for r in rows}):
        zz = np.array([r["z"] for r in rows if r["a"] == pl] + [-r["z"] for r in rows if r["b"] == pl])
        worst = max(worst, abs(zz.mean()) / np.sqrt(phi / len(zz)))
    print("| largest |mean signed z| of one checkpoint over all its rows, in units of its standard error at the observed "
          "dispersion | %.2f | < 3.3 for 25 checkpoints |" % worst)
    # (c) is the excess the same for draws?  expected draws of a reference row from this side's draw rate of the row
    qs = np.array([(r["here"][1] + 0.5) / (sum(r["here"]) + 1.0) for r in rows])
    nr = np.array([sum(r["ref"]) for r in rows], float)
    dr = np.array([r["ref"][1] for r in rows], float)
    var = nr * qs * (1 - qs) * (1 + nr / n)
    print("| dispersion of the reference's DRAW counts about this side's draw rates (Pearson chi^2 / rows) | %.2f | 1 |"
          % float(np.mean((dr - nr * qs) ** 2 / var)))
    print("| dispersion of the win counts (sum z^2 / rows) | %.2f | 1 |" % phi)
    print("\n## By first mover\n\n| first mover (player: checkpoint) | rows | sum z^2 | mean z | max abs z |\n|---|---|---|---|---|")
    for a in sorted({r["a"] for r in rows}):
        za = np.array([r["z"] for r in rows if r["a"] == a])
        print("| %d: model_%d | %d | %.1f | %+.2f | %.1f |" % (a, 93 - a, len(za), float(np.sum(za ** 2)), za.mean(), np.max(np.abs(za))))
    print("\n## By second mover\n\n| second mover | rows | sum z^2 | mean z |\n|---|---|---|---|")
    for b in sorted({r["b"] for r in rows}):
        zb = np.array([r["z"] for r in rows if r["b"] == b])
        print("| %d: model_%d | %d | %.1f | %+.2f |" % (b, 93 - b, len(zb), float(np.sum(zb ** 2)), zb.mean()))
    print("\n## Rows beyond 2.5 sigma\n\n| row | here W/D/L | reference W/D/L | here | reference | z |\n|---|---|---|---|---|---|")
    for r in sorted(rows, key=lambda r: -abs(r["z"])):
        if abs(r["z"]) <= 2.5:
            break
        print("| %d %d | %d/%d/%d | %d/%d/%d | %.3f | %.3f | %+.2f |" % (r["a"], r["b"], *r["here"], *r["ref"], r["here"][0] / sum(r["here"]),
                                                                   r["ref"][0] / sum(r["ref"]), r["z"]))
    print("\n## Every row\n\n| row | here W/D/L | reference W/D/L | z |\n|---|---|---|---|")
    for r in rows:
        print("| %d %d | %d/%d/%d | %d/%d/%d | %+.2f |" % (r["a"], r["b"], *r["here"], *r["ref"], r["z"]))


if __name__ == "__main__":
    if sys.argv[1] == "--report":
        report(sys.argv[2:])
    else:
        a = [x for x in sys.argv[1:] if not x.startswith("--")]
        net = sys.argv[sys.argv.index("--net") + 1] if "--net" in sys.argv else "x6"
        a = [x for x in a if x not in ("x6", "h3")]
        run(int(a[0]), int(a[1]) if len(a) > 1 else 0, int(a[2]) if len(a) > 2 else 1, net)
