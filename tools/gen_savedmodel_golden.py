#!/usr/bin/env python3
"""Golden vector from the reference's Keras SavedModel (build container only: needs /root/reference/corintho_ai/model).
Commits tests/golden/savedmodel.npz with
    weights      the engine's flat float32 layout WITH EXPLICIT BatchNorm (gamma, beta, moving mean and variance as Keras
                 holds them), read by corintho_ai_amd/savedmodel_import.py from variables/variables.{index,data-*}
                 (data of the reference's model files; no code)
    states       256 positions met in self-play (tests/golden/net_vectors.npz)
    value_f64, policy_f64   the network of wrapper.py:256-271 on those weights in float64 (tests/ref_nets.py): the yardstick
    bn_stats     per layer: min / max of gamma, beta, mean, variance -- what makes this the one fixture of reference data
                 that exercises the unfolded BatchNorm path (the TFLite checkpoints carry it folded)
usage: python tools/gen_savedmodel_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from corintho_ai_amd import savedmodel_import as SI  # noqa: E402
from tests import ref_nets  # noqa: E402

REF = "/root/reference/corintho_ai/model"
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference model directory not mounted")
    w = SI.mlp12x100_from_savedmodel(REF)
    states = np.load(os.path.join(OUT, "net_vectors.npz"))["states"]
    v, p = ref_nets.mlp12x100_forward_f64(w, states)
    stats = []
    off = 0
    n_in = 70
    for _ in range(12):
        off += n_in * 100 + 100
        g, b, m, var = (w[off + 100 * i:off + 100 * (i + 1)] for i in range(4))
        stats.append([g.min(), g.max(), b.min(), b.max(), m.min(), m.max(), var.min(), var.max()])
        off += 400
        n_in = 100
    np.savez_compressed(os.path.join(OUT, "savedmodel.npz"), weights=w, states=states, value_f64=v, policy_f64=p,
                        bn_stats=np.array(stats, np.float32), source=np.array("corintho_ai/model (Keras SavedModel, TensorBundle)"))
    print("savedmodel.npz: %d weights, value in [%.3f, %.3f], max prior %.3f, BatchNorm variance in [%.3g, %.3g]"
          % (w.size, v.min(), v.max(), p.max(), np.array(stats)[:, 6].min(), np.array(stats)[:, 7].max()))


if __name__ == "__main__":
    main()
