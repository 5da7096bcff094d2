#!/usr/bin/env python3
"""Does the reference's tournament driver explain the scatter of rating/results.txt about this engine's rates?

tests/test_reference_results.py replays 600 rows of results.txt at 2000 independent matches each: no bias, but the
reference's rows scatter 4.3 x wider than binomial samples would (sum z^2 / rows), a few of them 8-14 sigma out
(`51 92`, `10 92`, `89 4`, ...).  DESIGN.md section 2 attributes that to how the reference drew its sample:
  * rating/round.py:131-192 pops the 10 000 pairs of a round from a heap ordered by the posterior variance of their
    scores, :196-214 deals them out to the workers' match files (pair j to file j % T, two lines per pair) and writes
    the number of PAIRS on the first line;
  * rating/tourney.pyx:94-99 reads that many LINES -- the first half of a file;
  * cpp/src/tourney.cpp:82-88 seeds match L of a worker with the L-th output of a default-constructed std::mt19937
    (cpp/include/tourney.h:43).
So a game is a function of (first player, second player, line L), L < ~10 000 / T, and a pairing whose variance rank is
similar from round to round plays the SAME games again and again.  Here that process is simulated -- the scheduler exactly
as round.py runs it, 1000 rounds of 10 000 games (results.txt holds exactly 10 000 000), every (pair, colour, line)
an outcome drawn ONCE from the row's true rates (results.txt's own fractions stand in for them) -- and the simulated
rows are scored against those true rates the way the replay scores the reference's rows.  A control draws every game
afresh.  No engine, no GPU: this is about the reference's sampling, not about the search.

usage (build container, reads /root/reference): python tools/replica_sim.py [workers ...] [fidelities 0.x ...] > profiles/r05_outlier_rows.md
"""
import heapq
import sys

import numpy as np

RESULTS = "/root/reference/corintho_ai/rating/results.txt"
NAMED = [(51, 92), (10, 92), (89, 4), (3, 25)]  # the rows the replay found furthest out (tests/test_reference_results.py)


def get_variance(n1, m1, n2, m2):  # round.py:78-86
    return (n1 + 1) * (m1 + 1) / ((n1 + m1 + 2) ** 2 * (n1 + m1 + 3)) + (n2 + 1) * (m2 + 1) / ((n2 + m2 + 2) ** 2 * (n2 + m2 + 3))


def simulate(truth, players, T, rounds, games, rng, replicas=True, fidelity=1.0):
    """-> {(a, b): [wins, draws, losses]} after `rounds` rounds"""
    n = {k: [0.0, 0.0] for k in truth}   # round.py:108-127: [wins + draws / 2, losses + draws / 2] of the FIRST player
    wdl = {k: [0, 0, 0] for k in truth}
    fixed = {}                           # (a, b, line) -> outcome, drawn once
    pairs = [(a, b) for i, a in enumerate(players) for b in players[i + 1:] if (a, b) in truth and (b, a) in truth]
    for _ in range(rounds):
        heap = []
        for (a, b) in pairs:             # round.py:131-176
            n1, m1 = n[(a, b)]
            n2, m2 = n[(b, a)]
            heap.append((-get_variance(n1, m1, n2, m2), ((a, b), (n1, m1, n2, m2))))
        heapq.heapify(heap)
        matches = []
        for _ in range(games):           # round.py:181-194
            _, item = heapq.heappop(heap)
            matches.append(item[0])
            n1, m1, n2, m2 = item[1]
            n1 += n1 / (n1 + m1) if n1 + m1 > 0 else 0.5
            m1 += m1 / (n1 + m1) if n1 + m1 > 0 else 0.5
            n2 += n2 / (n2 + m2) if n2 + m2 > 0 else 0.5
            m2 += m2 / (n2 + m2) if n2 + m2 > 0 else 0.5
            heapq.heappush(heap, (-get_variance(n1, m1, n2, m2), (item[0], (n1, m1, n2, m2))))
        for i in range(T):               # round.py:196-214, tourney.pyx:94-99
            mine = matches[i::T]
            lines = [x for (a, b) in mine for x in ((a, b), (b, a))][:len(mine)]
            for L, (a, b) in enumerate(lines):
                key = (a, b, L)
                if replicas and key in fixed and (fidelity >= 1.0 or rng.random() < fidelity):
                    o = fixed[key]  # the same pairing at the same seed: the same game (with probability `fidelity`)
                else:
                    o = int(rng.choice(3, p=truth[(a, b)]))
                    fixed[key] = o
                wdl[(a, b)][o] += 1
                if o == 0:
                    n[(a, b)][0] += 1
                elif o == 1:
                    n[(a, b)][0] += 0.5
                    n[(a, b)][1] += 0.5
                else:
                    n[(a, b)][1] += 1
    return wdl


def score(wdl, truth, label, ref_rows):
    zs, ns, worst = [], [], []
    for k, (w, d, l) in wdl.items():
        m = w + d + l
        if m < 50:
            continue
        p = truth[k][0]
        z = (w / m - p) / np.sqrt(p * (1 - p) / m)
        zs.append(z)
        ns.append(m)
        worst.append((abs(z), k, m, w / m, p))
    zs, ns = np.array(zs), np.array(ns)
    worst.sort(reverse=True)
    big = ns >= np.percentile(ns, 95)
    print("| %s | %d | %.2f | %d | %d | %.1f | %.2f | %.0f / %.0f |" %
          (label, len(zs), float(np.mean(zs ** 2)), int(np.sum(np.abs(zs) > 3)), int(np.sum(np.abs(zs) > 6)), float(np.max(np.abs(zs))),
           float(np.mean(zs[big] ** 2)), float(np.median(ns)), float(np.max(ns))))
    return worst


def main():
    workers = [int(a) for a in sys.argv[1:] if "." not in a] or [88]
    fidelities = [float(a) for a in sys.argv[1:] if "." in a]
    rows = {}
    for l in open(RESULTS):
        f = l.split()
        if len(f) == 5:
            rows[(int(f[0]), int(f[1]))] = tuple(int(x) for x in f[2:])
    players = sorted({a for a, _ in rows} | {b for _, b in rows})
    truth = {k: np.array([v[0] + 0.5, v[1] + 0.5, v[2] + 0.5]) / (sum(v) + 1.5) for k, v in rows.items()}
    tot = sum(sum(v) for v in rows.values())
    print("# The reference's tournament schedule, simulated: do seed replicas explain the scatter of `results.txt`?\n")
    print("`tools/replica_sim.py` (docstring: the mechanism and the reference lines).  %d players, %d rows, %d games in "
          "`results.txt`; simulated: %d rounds of 10 000 games with the scheduler of `rating/round.py`, every (first player, second "
          "player, line of the worker's file) ONE outcome drawn from the row's own fractions; rows scored against those fractions "
          "as `tests/test_reference_results.py` scores the reference's rows against the engine's.\n" % (len(players), len(rows), tot, tot // 10000))
    print("| simulation | rows | sum z^2 / rows | rows beyond 3 sigma | beyond 6 sigma | largest abs z | sum z^2 / rows, the 5 %% of rows with the most games | games per row: median / most |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|")
    rng = np.random.default_rng(20261004)
    rounds = tot // 10000
    worst_by = {}
    for T in workers:
        w = simulate(truth, players, T, rounds, 10000, rng, True)
        worst_by[T] = (score(w, truth, "replicas, %d workers (%d lines read per file)" % (T, -(-10000 // T)), rows), w)
    for q in fidelities:
        wq = simulate(truth, players, workers[0], rounds, 10000, rng, True, q)
        score(wq, truth, "%d workers, a replayed (pairing, colour, line) comes out as before with probability %.2f" % (workers[0], q), rows)
    w0 = simulate(truth, players, workers[0], rounds, 10000, rng, False)
    score(w0, truth, "control: every game drawn afresh", rows)
    zs = []
    for k, v in rows.items():
        m = sum(v)
        if m >= 50:
            zs.append(m)
    print("\nThe replay on the MI355X (round 4, 600 rows at 2000 matches): sum z^2 / rows = 4.3, 66 rows beyond 3 sigma, "
          "7 beyond 6, the largest 14.5 (`51 92`, 799 reference games).  `results.txt` itself: median %d games per row, most %d.\n" % (int(np.median(zs)), max(zs)))
    T = workers[0]
    worst, w = worst_by[T]
    print("## The rows furthest out in the simulation (%d workers)\n" % T)
    print("| row | simulated games | simulated first-player wins | true rate | abs z |\n|---|---:|---:|---:|---:|")
    for az, k, m, r, p in worst[:10]:
        print("| `%d %d` | %d | %.3f | %.3f | %.1f |" % (k[0], k[1], m, r, p, az))
    print("\n## The rows the replay named, in the simulation\n")
    print("| row | reference games | simulated games | distinct games among them (lines of a worker's file) |\n|---|---:|---:|---:|")
    for k in NAMED:
        if k in w:
            print("| `%d %d` | %d | %d | <= %d |" % (k[0], k[1], sum(rows[k]), sum(w[k]), -(-10000 // T)))


if __name__ == "__main__":
    main()
