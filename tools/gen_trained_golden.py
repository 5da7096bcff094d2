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
        v2, p2 = nets.mlp12x100_forward_f64(w, states)
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
        v2, p2 = nets.mlp12x100_forward_f64(w, states)
        assert np.max(np.abs(v2 - g[roles["value"]][:, 0])) < 2e-6 and np.max(np.abs(p2 - g[roles["policy"]])) < 2e-6
        out["model_%d" % i] = w
    np.savez_compressed(os.path.join(OUT, "ref_models.npz"), **out)
    print("ref_models.npz:", sorted(out))


if __name__ == "__main__":
    main()
    extra_models()
