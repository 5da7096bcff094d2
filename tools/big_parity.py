#!/usr/bin/env python3
"""One-off parity run at the bench's full size: a fused generation on the GPU, then EVERY game
replayed on the CPU oracle (fed by the same device network through ca_trainer_net_forward) and
compared bit for bit: sample tensors (state, policy, outcome), score, mate length.
usage: big_parity.py [games] [sims] [net: mlp12x100|mlp12x100x3|mlp12x100x6|mlp12x100h3|rescnn4|rescnn4x3|rescnn4x6|rescnn4h3] [seed] [resident slots] [trained]
(trained: the mlp12x100 kinds with the reference's last checkpoint, tests/golden/trained_last.npz -- narrow, deep trees)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import corintho_ai_amd as CA  # noqa: E402
from corintho_ai_amd import Trainer, nets  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests import harness as H  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 400
net = sys.argv[3] if len(sys.argv) > 3 else "rescnn4x6"
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 12345
resident = int(sys.argv[5]) if len(sys.argv) > 5 else -1
KINDS = {"mlp12x100": "NET_MLP12X100", "mlp12x100x3": "NET_MLP12X100_X3", "mlp12x100x6": "NET_MLP12X100_X6",
         "rescnn4": "NET_RESCNN4", "rescnn4x3": "NET_RESCNN4_X3", "rescnn4x6": "NET_RESCNN4_X6", "rescnn4h3": "NET_RESCNN4_H3",
         "mlp12x100h3": "NET_MLP12X100_H3"}
kind = getattr(CA, KINDS[net])
w = nets.init_mlp12x100(0) if net.startswith("mlp12x100") else nets.init_rescnn4(0)
trained = len(sys.argv) > 6 and sys.argv[6] == "trained"
if trained:
    w = np.load(os.path.join(ROOT, "tests", "golden", "trained_last.npz"))["weights"]
spe = 16
t = Trainer(G, "", seed, S, spe, 1.0, 0.25, 0, 1, False, stagger=False, resident=resident)
t.set_net(kind, w)
t0 = time.perf_counter()
assert t.run()
t_gpu = time.perf_counter() - t0
sp, oc = t.export_samples()
o = O.Trainer(G, seed=seed, max_searches=S, searches_per_eval=spe, num_threads=int(os.environ.get("CORINTHO_CPU_THREADS", "16")))
o.set_stagger(False)
t0 = time.perf_counter()
cap = t.stats()["resident_slots"] * spe  # rows one device evaluation takes


def fw(states):
    parts = [t.net_forward(states[i:i + cap]) for i in range(0, states.shape[0], cap)]
    return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])


r = H.play_generation(o, G, spe, fw)
t_cpu = time.perf_counter() - t0
ogs, oev, opr = H.get_samples(o)
n = t.num_samples()
ok = (ogs.shape[0] == n * 8 and ogs[0::8].tobytes() == sp[:, :70].tobytes() and opr[0::8].tobytes() == sp[:, 70:].tobytes()
      and oev[0::8].tobytes() == oc.tobytes() and o.score() == t.score() and o.avg_mate_length() == t.avg_mate_length())
st = t.stats()
print("%d games x %d sims/move on %d slots, %s%s, seed %d: GPU generation %.2f s (%d of %d request rows evaluated); oracle replay of all "
      "games %.1f s (%d host iterations)" % (G, S, st["resident_slots"], net, " (the reference's last checkpoint)" if trained else "", seed, t_gpu, st["nn_rows_evaluated"], st["nn_rows"], t_cpu,
                                             r["iterations"]))
print("plies %d, simulations %d, leaf evaluations %d, samples %d, score %.6f" % (st["plies"], st["searches"], st["evals"], n, t.score()))
print("BIT-EXACT: every (state[70], policy[96], outcome) row, score and mate length agree" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
