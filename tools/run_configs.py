#!/usr/bin/env python3
"""Run the other BASELINE.json configurations once each in fused mode and print one JSON
line per configuration (games/s, device time split, arena high-water mark, roofline objects of both kernel families).
  cfg1: 64 games, 50 sims (plumbing)        cfg4: 4096 games, 1600 sims + Dirichlet noise (toml/train.toml:2-18)
  cfg5: arena, 1024 two-model games, testing=True, 400 sims (main.pyx:329-349)
  tourney: 1024 matches between 4 players of 2 models (+ a random player), fused
  compat: the reference protocol, host-driven (PCIe-inclusive; never `value`)
bench.py imports `measure_all` for its `detail.configs`; as a script: run_configs.py [cfg1 cfg4 cfg5 tourney compat] [x3|x6]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from corintho_ai_amd import NET_RESCNN4_H3, NET_RESCNN4_X3, NET_RESCNN4_X6, Tourney, Trainer, nets  # noqa: E402

BYTES_PER_SIM = 3200.0  # SURVEY 8d (bench.py uses the same figure)
HBM_PEAK_GBS = 8000.0
PEAK_TFLOPS = 2500.0  # dense fp16 / bf16 MFMA
ISSUED = {"rescnn4h3": 2.09, "rescnn4x6": 6.0, "rescnn4x3": 3.0}
KINDS = {"rescnn4h3": NET_RESCNN4_H3, "rescnn4x6": NET_RESCNN4_X6, "rescnn4x3": NET_RESCNN4_X3}
ALL = ("cfg1", "cfg4", "cfg5", "tourney", "compat")


def rooflines(st, wall_s, net_name):
    """roofline objects of a finished run from the engine's own statistics: device time per kernel family is estimated
    from the HIP-event durations of the timed launches (engine.hip run_pools), as in bench.py"""
    flop = nets.rescnn4_flop_per_row()
    rows = st.get("nn_rows_evaluated", 0) or st.get("evals", 0)
    timed = st["nn_ms"] > 0 and st["mcts_ms"] > 0  # (the tournament's fused loop carries no events: wall level only)
    nn_s, mc_s = (st["nn_ms"] * 1e-3, st["mcts_ms"] * 1e-3) if timed else (wall_s, wall_s)
    a_n = rows * flop / nn_s / 1e12
    a_s = st["searches"] * BYTES_PER_SIM / mc_s / 1e9
    rn = {"kernel": "network (K6p / K6h3 small and thin paths)" if net_name == "rescnn4h3" else "network", "bound": "mfma",
          "achieved": a_n, "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": a_n / PEAK_TFLOPS, "traffic": None,
          "issued_frac": ISSUED.get(net_name, 1.0) * a_n / PEAK_TFLOPS,
          "wall": {"achieved": rows * flop / wall_s / 1e12, "frac": rows * flop / wall_s / 1e12 / PEAK_TFLOPS}}
    rs = {"kernel": "co_k_mcts_step", "bound": "hbm", "achieved": a_s, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a_s / HBM_PEAK_GBS,
          "traffic": None, "wall": {"achieved": st["searches"] * BYTES_PER_SIM / wall_s / 1e9,
                                    "frac": st["searches"] * BYTES_PER_SIM / wall_s / 1e9 / HBM_PEAK_GBS}}
    if not timed:
        rn["note"] = rs["note"] = "wall level: this mode's launches carry no HIP events"
    return {"roofline": rn if (nn_s >= mc_s and rows > 0) else rs, "roofline_network": rn, "roofline_search": rs}


def run_trainer(name, G, S, net_name, w0, w1, testing=False, reps=1, device=0):
    t = Trainer(G, "", 12345, S, 16, 1.0, 0.25, 0, 1, testing, stagger=False, device=device)
    t.set_net(KINDS[net_name], w0, slot=0)
    if testing:
        t.set_net(KINDS[net_name], w1, slot=1)
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
        rec.update(rooflines(st, dt, net_name))
        if best is None or rec["games_per_s"] > best["games_per_s"]:
            best = rec
    t.close()
    return best


def run_tourney(net_name, w0, w1, n_matches=1024, device=0):
    """round robin of 4 searching players (2 models, two search budgets) + games against a random player"""
    def make():
        t = Tourney(1, "", device=device)
        t.addPlayer(0, 0, 400, 16, 1.0, 0.25, False)
        t.addPlayer(1, 1, 400, 16, 1.0, 0.25, False)
        t.addPlayer(2, 0, 100, 16, 1.0, 0.25, False)
        t.addPlayer(3, 1, 100, 16, 1.0, 0.25, False)
        t.addPlayer(4, -1, 0, 0, 1.0, 0.25, True)
        pairs = [(a, b) for a in range(4) for b in range(4) if a != b] + [(0, 4), (4, 1)]
        for i in range(n_matches):
            t.addMatch(*pairs[i % len(pairs)], False)
        t.set_net(0, KINDS[net_name], w0)
        t.set_net(1, KINDS[net_name], w1)
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
    rec = {"config": "tourney: %d matches, 5 players, 2 models + random" % n_matches, "matches": n_matches,
           "net": net_name, "seconds": dt, "matches_per_s": n_matches / dt, "iterations": st["iterations"],
           "searches": st["searches"], "evals": st["evals"], "plies_per_match": st["plies"] / n_matches,
           "mean_first_player_score": score, "device_ms": {k: st[k] for k in ("mcts_ms", "nn_ms", "pack_ms")}}
    rec.update(rooflines(st, dt, net_name))
    t.close()
    return rec


def run_compat(net_name, w0, G=4096, S=400, pin=True, device=0):
    """the reference protocol (compat mode): request rows come back to the host every iteration,
    the caller evaluates them (here: on the same GPU through ca_trainer_net_forward, i.e. another
    host round trip) and hands evaluations in -- the PCIe-inclusive rate of DESIGN.md section 6"""
    import numpy as np

    t = Trainer(G, "", 12345, S, 16, 1.0, 0.25, 0, 1, False, stagger=False, device=device)
    t.set_net(KINDS[net_name], w0)
    cap = G * 16
    evals = np.zeros(cap, np.float32)
    probs = np.zeros((cap, 96), np.float32)
    gs = np.zeros((cap, 70), np.float32)
    pinned = t.pin(evals, probs, gs) if pin else False  # the arrays of main.pyx:132-134, allocated once
    t0 = time.perf_counter()
    iters = 0
    rows = 0
    while not t.doIteration(evals, probs, -1):
        n = t.num_requests(-1)
        t.writeRequests(gs, -1)
        t.net_forward(gs[:n], out_evals=evals, out_probs=probs)
        iters += 1
        rows += n
    dt = time.perf_counter() - t0
    st = t.stats()
    flop = nets.rescnn4_flop_per_row()
    rec = {"config": "compat: %d games x %d sims, host-driven protocol, network via net_forward" % (G, S),
           "games": G, "sims": S, "net": net_name, "pinned_host_arrays": bool(pinned), "seconds": dt, "games_per_s": G / dt,
           "iterations": iters, "request_rows": rows,
           # host-driven: only wall-level figures mean anything (every launch stands between two PCIe transfers)
           "roofline": {"kernel": "network via ca_trainer_net_forward", "bound": "mfma", "achieved": rows * flop / dt / 1e12,
                        "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": rows * flop / dt / 1e12 / PEAK_TFLOPS, "traffic": None,
                        "note": "wall level, PCIe-inclusive"},
           "roofline_search": {"kernel": "co_k_mcts_step", "bound": "hbm", "achieved": st["searches"] * BYTES_PER_SIM / dt / 1e9,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": st["searches"] * BYTES_PER_SIM / dt / 1e9 / HBM_PEAK_GBS,
                               "traffic": None, "note": "wall level, PCIe-inclusive"}}
    t.close()
    return rec


def measure_all(which=ALL, net_name="rescnn4h3", device=0, emit=None):
    """-> {name: record}; `emit(record)` is called as each configuration finishes"""
    w0, w1 = nets.init_rescnn4(0), nets.init_rescnn4(1)
    out = {}

    def done(key, rec):
        out[key] = rec
        if emit:
            emit(rec)

    if "cfg1" in which:
        done("cfg1", run_trainer("cfg1: 64 games x 50 sims", 64, 50, net_name, w0, w1, reps=2, device=device))
    if "cfg4" in which:
        done("cfg4", run_trainer("cfg4: 4096 games x 1600 sims + Dirichlet (deep-tree stress)", 4096, 1600, net_name, w0, w1, device=device))
    if "cfg5" in which:
        done("cfg5", run_trainer("cfg5: arena, 1024 two-model games, greedy 400 sims", 1024, 400, net_name, w0, w1, testing=True, device=device))
    if "tourney" in which:
        done("tourney", run_tourney(net_name, w0, w1, device=device))
    if "compat" in which:
        done("compat", run_compat(net_name, w0, pin="nopin" not in which, device=device))
    return out


if __name__ == "__main__":
    args = sys.argv[1:]
    which = [a for a in args if a in ALL or a == "nopin"] or list(ALL)
    if which == ["nopin"]:
        which = list(ALL) + ["nopin"]
    net = "rescnn4x3" if "x3" in args else "rescnn4x6" if "x6" in args else "rescnn4h3"
    measure_all(which, net, emit=lambda rec: print(json.dumps(rec), flush=True))
