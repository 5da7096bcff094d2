#!/usr/bin/env python3
"""Every checkpoint of the reference's rating tournament (rating/tflite_models/model_*.tflite, 95 files) through the
f16x3 MLP kernel (K5h3), against the stored TFLite graph evaluated in float64.

  build container:  python tools/all_checkpoints_h3.py --prepare
        imports the 95 checkpoints (corintho_ai_amd/tflite_import.py), evaluates each stored graph in float64 on the 256
        positions met in self-play of tests/golden/net_vectors.npz and writes build_ab/all_models.npz (50 MB: travels to
        the GPU box with the snapshot, stays out of the history);
  GPU box:          python tools/all_checkpoints_h3.py > profiles/r04_all_checkpoints_f16x3.md
        sets every checkpoint as mlp12x100h3 (the range guard of nn.h applies: weights are checked at set_net, activations
        by the kernel), mlp12x100x6 and fp32 MFMA, and prints the worst absolute error of value and policy per checkpoint
        and arithmetic, the largest folded weight and whether any activation left the fp16 range.
"""
import glob
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PATH = os.path.join(ROOT, "build_ab", "all_models.npz")


def prepare():
    from corintho_ai_amd import tflite_import as TI

    ref = "/root/reference/corintho_ai/rating/tflite_models"
    paths = sorted(glob.glob(os.path.join(ref, "model_*.tflite")), key=lambda p: int(re.findall(r"model_(\d+)", p)[0]))
    if os.path.exists(os.path.join(ref, "first_run.tflite")):
        paths.append(os.path.join(ref, "first_run.tflite"))  # the 95th file: id -1
    states = np.load(os.path.join(ROOT, "tests", "golden", "net_vectors.npz"))["states"][:256]
    ids, W, V, P = [], [], [], []
    for path in paths:
        m = TI.read_tflite(path)
        roles = TI.output_roles(m)
        g = TI.tflite_forward_np(m, states, dtype=np.float64)
        ids.append(int(re.findall(r"model_(\d+)", path)[0]) if "model_" in path else -1)
        W.append(TI.mlp12x100_from_tflite(path))
        V.append(g[roles["value"]][:, 0])
        P.append(g[roles["policy"]])
    os.makedirs(os.path.dirname(PATH), exist_ok=True)
    np.savez(PATH, ids=np.array(ids), weights=np.array(W, np.float32), states=states, v64=np.array(V), p64=np.array(P))
    print("%d checkpoints -> %s" % (len(ids), PATH))


def folded_max(w):
    """largest |weight| after the float64 BatchNorm fold of nn_mlp_split.hip (layer l's affine folded into layer l + 1)"""
    p, in_dim, a_prev, worst = 0, 70, None, 0.0
    for _ in range(12):
        K = w[p:p + in_dim * 100].reshape(in_dim, 100).astype(np.float64)
        ga, va = w[p + in_dim * 100 + 100:p + in_dim * 100 + 200], w[p + in_dim * 100 + 400:p + in_dim * 100 + 500]
        if a_prev is not None:
            K = K * a_prev[:, None]
        worst = max(worst, float(np.abs(K).max()))
        a_prev = (ga.astype(np.float64) / np.sqrt(va.astype(np.float64) + 1e-3)).astype(np.float32).astype(np.float64)
        p += in_dim * 100 + 500
        in_dim = 100
    for n_out in (1, 96):
        K = w[p:p + 100 * n_out].reshape(100, n_out).astype(np.float64) * a_prev[:, None]
        worst = max(worst, float(np.abs(K).max()))
        p += 100 * n_out + n_out
    return worst


def run():
    from corintho_ai_amd import NET_MLP12X100, NET_MLP12X100_H3, NET_MLP12X100_X6, Trainer

    d = np.load(PATH)
    states = d["states"]
    t = Trainer(64, "", 1, 50, 16, 1.0, 0.25, 0, 1, False, stagger=False)
    kinds = (("f16x3", NET_MLP12X100_H3), ("bf16x6", NET_MLP12X100_X6), ("fp32 MFMA", NET_MLP12X100))
    print("# All %d reference checkpoints through the MLP kernels (positions met in self-play, 256 rows)\n" % len(d["ids"]))
    print("Worst absolute error against the stored TFLite graph evaluated in float64 (`tools/all_checkpoints_h3.py`); value after tanh, "
          "policy after softmax -- the outputs the interface exposes (the contract of BASELINE.json: 1e-4).\n")
    print("| checkpoint | largest folded weight | f16x3 value / policy | bf16x6 value / policy | fp32 MFMA value / policy | f16x3 / fp32 |")
    print("|---|---|---|---|---|---|")
    worst = {k: [0.0, 0.0] for k, _ in kinds}
    ratio_max = 0.0
    for i, mid in enumerate(d["ids"]):
        w = d["weights"][i]
        row = []
        for name, kind in kinds:
            t.set_net(kind, w)  # (a weight beyond fp16's range would raise here for f16x3)
            ev, pr = t.net_forward(states)  # (an activation beyond it would raise here)
            ev_err, pr_err = float(np.max(np.abs(ev - d["v64"][i]))), float(np.max(np.abs(pr - d["p64"][i])))
            worst[name][0], worst[name][1] = max(worst[name][0], ev_err), max(worst[name][1], pr_err)
            row.append((ev_err, pr_err))
        ratio = max(row[0][0], row[0][1]) / max(row[2][0], row[2][1], 1e-12)
        ratio_max = max(ratio_max, ratio)
        print("| %s | %.1f | %.1e / %.1e | %.1e / %.1e | %.1e / %.1e | %.2f |" % ("model_%d" % mid if mid >= 0 else "first_run", folded_max(w), *row[0], *row[1], *row[2], ratio))
    print("\nWorst over the %d checkpoints: " % len(d["ids"]) +
          "; ".join("%s %.1e / %.1e" % (k, *worst[k]) for k, _ in kinds) +
          "; f16x3 at most %.2f x the fp32-MFMA kernel's own error.  No checkpoint raised the fp16 range guard (weights at set_net, "
          "activations in the kernel): the largest folded weight is far inside 65504." % ratio_max)


if __name__ == "__main__":
    prepare() if "--prepare" in sys.argv else run()
