#!/usr/bin/env python3
"""Experiment: the two-term fp16 split ("f16x3", kinds 8 / 9) against float64, beside fp32 MFMA, bf16x6 and bf16x3,
on every weight set of tests/test_net_precision.py and the three reference checkpoints; kernel times at 16 k / 32 k rows."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import corintho_ai_amd as CA  # noqa: E402
from tests import ref_nets
from corintho_ai_amd import Trainer, nets  # noqa: E402


def states(n, seed):
    rng = np.random.default_rng(seed)
    s = np.zeros((n, 70), np.float32)
    s[:, :64] = rng.integers(0, 2, (n, 64))
    s[:, 64:] = rng.integers(0, 5, (n, 6)) * 0.25
    return s


t = Trainer(2048, "", 1, 50, 16, 1.0, 0.25, 0, 1, False, arena_units=4096)
S = states(777, 5)
G = os.path.join(ROOT, "tests", "golden")
real = np.load(os.path.join(G, "net_vectors.npz"))["states"]
cnn = [("init", nets.init_rescnn4(0)), ("bn-noise", nets.init_rescnn4(3, bn_noise=True)), ("trained-like-0", nets.trained_like_rescnn4(0)),
       ("trained-like-1", nets.trained_like_rescnn4(1))]
mlp = [("init", nets.init_mlp12x100(0)), ("bn-noise", nets.init_mlp12x100(7, bn_noise=True)), ("trained-like-0", nets.trained_like_mlp12x100(0)),
       ("trained-like-1", nets.trained_like_mlp12x100(1))]
for tag in ("early", "middle", "last"):
    mlp.append(("checkpoint-" + tag, np.load(os.path.join(G, "trained_%s.npz" % tag))["weights"]))
d = np.load(os.path.join(G, "ref_models.npz"))
for k in d.files:
    mlp.append((k, d[k]))
for label, sets, f64, kinds in (("rescnn4", cnn, ref_nets.rescnn4_forward_f64, ("NET_RESCNN4", "NET_RESCNN4_X6", "NET_RESCNN4_H3", "NET_RESCNN4_X3")),
                                ("mlp12x100", mlp, ref_nets.mlp12x100_forward_f64, ("NET_MLP12X100", "NET_MLP12X100_X6", "NET_MLP12X100_H3", "NET_MLP12X100_X3"))):
    for name, w in sets:
        for st_name, st in (("synthetic", S), ("self-play", real)):
            want = f64(w, st)
            out = []
            for k in kinds:
                t.set_net(getattr(CA, k), w)
                ev, pr = t.net_forward(st)
                out.append((float(np.max(np.abs(ev.astype(np.float64) - want[0]))), float(np.max(np.abs(pr.astype(np.float64) - want[1])))))
            print("%-9s %-17s %-9s value: fp32 %.2e x6 %.2e h3 %.2e x3 %.2e | policy: fp32 %.2e x6 %.2e h3 %.2e x3 %.2e"
                  % (label, name, st_name, out[0][0], out[1][0], out[2][0], out[3][0], out[0][1], out[1][1], out[2][1], out[3][1]), flush=True)
big = states(32768, 11)
for k in ("NET_RESCNN4_X6", "NET_RESCNN4_H3", "NET_RESCNN4_X3", "NET_MLP12X100_X6", "NET_MLP12X100_H3", "NET_MLP12X100_X3"):
    t.set_net(getattr(CA, k), nets.init_rescnn4(0) if "RESCNN" in k else nets.init_mlp12x100(0))
    print("%-18s ms per launch: 16384 rows %.4f, 32768 rows %.4f" % (k, t.net_bench(big[:16384], 20), t.net_bench(big, 20)), flush=True)
