#!/usr/bin/env python3
"""Golden vectors on TRAINED weights (build container only: needs the reference's checkpoints under
/root/reference/corintho_ai/rating/tflite_models/).  For three checkpoints -- an early one, a middle one
and the last -- commits tests/golden/trained_<name>.npz with
    weights      the engine's flat float32 layout, imported by corintho_ai_amd/tflite_import.py
                 (data of the reference's checkpoint files; no code)
    states       256 positions met in self-play (tests/golden/net_vectors.npz)
    value_f64, policy_f64   the stored TFLite graph evaluated in float64 (tflite_import.tflite_forward_np
                 on float64 copies): the yardstick, independent of the engine's BatchNorm handling
so that the network kernels are checked on trained-scale weights on the GPU box, where the reference
tree does not exist (tests/test_trained_golden.py).

usage: python tools/gen_trained_golden.py
"""
import glob
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from corintho_ai_amd import nets  # noqa: E402
from tests import ref_nets
from corintho_ai_amd import tflite_import as TI  # noqa: E402

REF = "/root/reference/corintho_ai/rating/tflite_models"
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    paths = sorted(glob.glob(os.path.join(REF, "model_*.tflite")), key=lambda p: int(re.findall(r"model_(\d+)", p)[0]))
    if not paths:
        raise SystemExit("reference checkpoints not mounted")
    states = np.load(os.path.join(OUT, "net_vectors.npz"))["states"]
    for tag, path in (("early", paths[3]), ("middle", paths[len(paths) // 2]), ("last", paths[-1])):
        m = TI.read_tflite(path)
        roles = TI.output_roles(m)
        w = TI.mlp12x100_from_tflite(path)
        g = TI.tflite_forward_np(m, states, dtype=np.float64)
        v64, p64 = g[roles["value"]][:, 0], g[roles["policy"]]
        # the import is faithful: the engine's layout evaluated in float64 is the stored graph in float64
        v2, p2 = ref_nets.mlp12x100_forward_f64(w, states)
        assert np.max(np.abs(v2 - v64)) < 2e-6 and np.max(np.abs(p2 - p64)) < 2e-6, (np.max(np.abs(v2 - v64)), np.max(np.abs(p2 - p64)))
        name = os.path.basename(path)
        np.savez_compressed(os.path.join(OUT, "trained_%s.npz" % tag), weights=w, states=states, value_f64=v64, policy_f64=p64,
                            checkpoint=np.array(name))
        ent = -(p64 * np.log(p64 + 1e-300)).sum(axis=1).mean()
        print("%-7s %-16s value in [%.3f, %.3f], mean policy entropy %.2f nats, max prior %.3f" %
              (tag, name, v64.min(), v64.max(), ent, p64.max(axis=1).mean()))


def extra_models(ids=(92, 4)):
    """weights only (the engine's flat layout) of further checkpoints, for the replay of rating/results.txt rows
    (tests/test_reference_results.py): tests/golden/ref_models.npz, keys model_<id>"""
    out = {}
    for i in ids:
        path = os.path.join(REF, "model_%d.tflite" % i)
        w = TI.mlp12x100_from_tflite(path)
        m = TI.read_tflite(path)
        roles = TI.output_roles(m)
        states = np.load(os.path.join(OUT, "net_vectors.npz"))["states"][:64]
        g = TI.tflite_forward_np(m, states, dtype=np.float64)
        v2, p2 = ref_nets.mlp12x100_forward_f64(w, states)
        assert np.max(np.abs(v2 - g[roles["value"]][:, 0])) < 2e-6 and np.max(np.abs(p2 - g[roles["policy"]])) < 2e-6
        out["model_%d" % i] = w
    np.savez_compressed(os.path.join(OUT, "ref_models.npz"), **out)
    print("ref_models.npz:", sorted(out))


# The population of round 4 (tests/test_reference_results.py, tools/ref_population.py): players of
# rating/tourney/players.txt chosen before any of their rows was replayed -- the next-strongest two (2, 3: does the
# oddity of player 1's first-mover rows repeat for them?) and 18 drawn with a fixed seed from the searching players
# 4 .. 93 that were not committed yet.  Player p <= 93 plays checkpoint model_(93 - p).
POPULATION_SEED = 20261004


def population_players():
    rng = np.random.default_rng(POPULATION_SEED)
    cand = [p for p in range(4, 94) if p not in (46, 89, 90)]
    return [2, 3] + sorted(int(x) for x in rng.choice(cand, 18, replace=False))


def population_models():
    """tests/golden/ref_models_pop.npz: weights (the engine's flat layout) of the population's checkpoints"""
    out = {}
    states = np.load(os.path.join(OUT, "net_vectors.npz"))["states"][:64]
    for p in population_players():
        i = 93 - p
        path = os.path.join(REF, "model_%d.tflite" % i)
        w = TI.mlp12x100_from_tflite(path)
        m = TI.read_tflite(path)
        roles = TI.output_roles(m)
        g = TI.tflite_forward_np(m, states, dtype=np.float64)
        v2, p2 = ref_nets.mlp12x100_forward_f64(w, states)
        assert np.max(np.abs(v2 - g[roles["value"]][:, 0])) < 2e-6 and np.max(np.abs(p2 - g[roles["policy"]])) < 2e-6
        out["model_%d" % i] = w
    np.savez_compressed(os.path.join(OUT, "ref_models_pop.npz"), **out)
    print("ref_models_pop.npz: players", population_players(), "models", sorted(int(k.split("_")[1]) for k in out))
    # the reference's rows between the 25 committed players (data of rating/results.txt: first player, second player,
    # the first player's wins, draws, losses), all 600 ordered pairs
    import json

    players = sorted([0, 1, 46, 89, 90] + population_players())
    rows = []
    for line in open("/root/reference/corintho_ai/rating/results.txt"):
        f = line.split()
        if len(f) == 5 and int(f[0]) in players and int(f[1]) in players:
            rows.append([int(x) for x in f])
    assert len(rows) == len(players) * (len(players) - 1)
    json.dump({"source": "corintho_ai/rating/results.txt (rows between the listed players of rating/tourney/players.txt; "
                         "player p <= 93 plays model_(93 - p) with `1600 16 3.0 0.25 0`)",
               "players": players, "rows": rows}, open(os.path.join(OUT, "ref_results_pop.json"), "w"))
    print("ref_results_pop.json: %d rows, %d games" % (len(rows), sum(sum(r[2:]) for r in rows)))


if __name__ == "__main__":
    if "--population" in sys.argv:
        population_models()
    else:
        main()
        extra_models()
