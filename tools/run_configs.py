#!/usr/bin/env python3
"""Run the other BASELINE.json configurations once each in fused mode and print one JSON
line per configuration (games/s, device time split, arena high-water mark).
  cfg1: 64 games, 50 sims (plumbing)        cfg4: 4096 games, 1600 sims + Dirichlet noise
  cfg5: arena, 1024 two-model games, testing=True, 400 sims
  tourney: 1024 matches between 4 players of 2 models (+ a random player), fused"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from corintho_ai_amd import NET_MLP12X100, NET_RESCNN4_H3, NET_RESCNN4_X3, NET_RESCNN4_X6, Tourney, Trainer, nets  # noqa: E402

which = sys.argv[1:] or ["cfg1", "cfg4", "cfg5", "tourney", "compat"]
net_kind, net_name = ((NET_RESCNN4_X3, "rescnn4x3") if "x3" in which else (NET_RESCNN4_X6, "rescnn4x6") if "x6" in which
                      else (NET_RESCNN4_H3, "rescnn4h3"))
w0, w1 = nets.init_rescnn4(0), nets.init_rescnn4(1)


def run(name, G, S, testing=False, reps=1):
    t = Trainer(G, "", 12345, S, 16, 1.0, 0.25, 0, 1, testing, stagger=False)
    t.set_net(net_kind, w0, slot=0)
    if testing:
        t.set_net(net_kind, w1, slot=1)
    t.run()  # warm
    best = None
    for r in range(reps):
        t.reset(100 + r)
        t0 = time.perf_counter()
        assert t.run()
        dt = time.perf_counter() - t0
        st = t.stats()
        rec = {"config": name, "games": G, "sims": S, "testing": testing, "net": net_name, "seconds": dt,
               "games_per_s": G / dt, "iterations": st["iterations"], "searches": st["searches"], "evals": st["evals"],
               "plies_per_game": st["plies"] / G, "device_ms": {k: st[k] for k in ("mcts_ms", "nn_ms", "pack_ms")},
               "peak_arena_units_per_tree": st["peak_arena_units"], "score": t.score()}
        if best is None or rec["games_per_s"] > best["games_per_s"]:
            best = rec
    print(json.dumps(best), flush=True)
    t.close()


if "cfg1" in which:
    run("cfg1: 64 games x 50 sims", 64, 50, reps=2)
if "cfg4" in which:
    run("cfg4: 4096 games x 1600 sims + Dirichlet (deep-tree stress)", 4096, 1600)
if "cfg5" in which:
    run("cfg5: arena, 1024 two-model games, greedy 400 sims", 1024, 400, testing=True)


def run_tourney(n_matches=1024):
    """round robin of 4 searching players (2 models, two search budgets) + games against a random player"""
    def make():
        t = Tourney(1, "")
        t.addPlayer(0, 0, 400, 16, 1.0, 0.25, False)
        t.addPlayer(1, 1, 400, 16, 1.0, 0.25, False)
        t.addPlayer(2, 0, 100, 16, 1.0, 0.25, False)
        t.addPlayer(3, 1, 100, 16, 1.0, 0.25, False)
        t.addPlayer(4, -1, 0, 0, 1.0, 0.25, True)
        pairs = [(a, b) for a in range(4) for b in range(4) if a != b] + [(0, 4), (4, 1)]
        for i in range(n_matches):
            t.addMatch(*pairs[i % len(pairs)], False)
        t.set_net(0, net_kind, w0)
        t.set_net(1, net_kind, w1)
        return t

    t = make()
    t.run()  # warm (kernels, allocations)
    t.close()
    t = make()
    t0 = time.perf_counter()
    assert t.run()
    dt = time.perf_counter() - t0
    st = t.stats()
    score = sum(t.match_score(i) for i in range(n_matches)) / n_matches
    print(json.dumps({"config": "tourney: %d matches, 5 players, 2 models + random" % n_matches, "matches": n_matches,
                      "net": net_name, "seconds": dt, "matches_per_s": n_matches / dt, "iterations": st["iterations"],
                      "searches": st["searches"], "evals": st["evals"], "plies_per_match": st["plies"] / n_matches,
                      "mean_first_player_score": score}), flush=True)
    t.close()


if "tourney" in which:
    run_tourney()


def run_compat(G=4096, S=400):
    """the reference protocol (compat mode): request rows come back to the host every iteration,
    the caller evaluates them (here: on the same GPU through ca_trainer_net_forward, i.e. another
    host round trip) and hands evaluations in -- the PCIe-inclusive rate of DESIGN.md section 6"""
    import numpy as np

    t = Trainer(G, "", 12345, S, 16, 1.0, 0.25, 0, 1, False, stagger=False)
    t.set_net(net_kind, w0)
    cap = G * 16
    evals = np.zeros(cap, np.float32)
    probs = np.zeros((cap, 96), np.float32)
    gs = np.zeros((cap, 70), np.float32)
    pinned = t.pin(evals, probs, gs) if "nopin" not in which else False  # the arrays of main.pyx:132-134, allocated once
    t0 = time.perf_counter()
    iters = 0
    while not t.doIteration(evals, probs, -1):
        n = t.num_requests(-1)
        t.writeRequests(gs, -1)
        t.net_forward(gs[:n], out_evals=evals, out_probs=probs)
        iters += 1
    dt = time.perf_counter() - t0
    print(json.dumps({"config": "compat: %d games x %d sims, host-driven protocol, network via net_forward" % (G, S),
                      "games": G, "sims": S, "net": net_name, "pinned_host_arrays": bool(pinned), "seconds": dt, "games_per_s": G / dt,
                      "iterations": iters}),
          flush=True)
    t.close()


if "compat" in which:
    run_compat()
