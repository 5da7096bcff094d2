"""Weight import from TFLite checkpoints (corintho_ai_amd/tflite_import.py).
Synthetic files written by tests/tflite_writer.py in the shape of the reference's checkpoints
are imported and evaluated; in the container where the reference is mounted its real
checkpoints (rating/tflite_models/model_*.tflite) are imported too."""
import glob
import os

import numpy as np
import pytest

from corintho_ai_amd import nets

from tests import ref_nets
from corintho_ai_amd import tflite_import as TI
from tests import tflite_writer as TW
from tests.engines import ENGINES, make_trainer

REF_MODELS = sorted(glob.glob("/root/reference/corintho_ai/rating/tflite_models/model_*.tflite"))


def _states(n, seed=0):
    rng = np.random.default_rng(seed)
    s = np.zeros((n, 70), np.float32)
    s[:, :64] = rng.integers(0, 2, (n, 64))
    s[:, 64:] = rng.integers(0, 5, (n, 6)) * 0.25
    return s


@pytest.mark.parametrize("with_bias", [True, False])
def test_import_of_a_synthetic_checkpoint(with_bias):
    w = nets.init_mlp12x100(seed=4, bn_noise=with_bias)
    blob = TW.write_mlp_tflite(*TW.fold_keras_mlp(w), with_bias=with_bias)
    m = TI.read_tflite(blob)
    assert [op["code"] for op in m["ops"]].count(TI.OP_FULLY_CONNECTED) == 14
    roles = TI.output_roles(m)
    assert m["outputs"] == [roles["policy"], roles["value"]]
    imported = TI.mlp12x100_from_tflite(blob)
    assert imported.size == nets.MLP_NUM_WEIGHTS
    s = _states(300)
    # the imported flat weights, the stored graph and the un-folded Keras weights are one function
    ev_i, pr_i = ref_nets.mlp12x100_forward_np(imported, s)
    g = TI.tflite_forward_np(m, s)
    ev_k, pr_k = ref_nets.mlp12x100_forward_np(w, s)
    assert np.max(np.abs(ev_i - g[roles["value"]][:, 0])) < 1e-6
    assert np.max(np.abs(pr_i - g[roles["policy"]])) < 1e-6
    assert np.max(np.abs(ev_i - ev_k)) < 1e-4 and np.max(np.abs(pr_i - pr_k)) < 1e-4


def test_malformed_files_are_rejected():
    with pytest.raises(TI.TFLiteFormatError):
        TI.read_tflite(b"\0" * 64)
    w = nets.init_mlp12x100(seed=1)
    layers, vh, ph = TW.fold_keras_mlp(w)
    with pytest.raises(TI.TFLiteFormatError, match="layer 11"):
        TI.mlp12x100_from_tflite(TW.write_mlp_tflite(layers[:11], vh, ph))


@pytest.mark.parametrize("engine", ENGINES)
def test_imported_checkpoint_on_the_engine(engine):
    """the fused network kernel on imported weights == the stored TFLite graph (<= 1e-4)"""
    w = nets.init_mlp12x100(seed=9, bn_noise=True)
    blob = TW.write_mlp_tflite(*TW.fold_keras_mlp(w))
    m = TI.read_tflite(blob)
    roles = TI.output_roles(m)
    t = make_trainer(engine, 16, "", 1, 50, 16, 1.0, 0.25, 0, 1, False)
    t.set_net(1, TI.mlp12x100_from_tflite(blob))
    s = _states(200, seed=2)
    ev, pr = t.net_forward(s)
    g = TI.tflite_forward_np(m, s)
    assert np.max(np.abs(ev - g[roles["value"]][:, 0])) < 1e-4
    assert np.max(np.abs(pr - g[roles["policy"]])) < 1e-4


@pytest.mark.skipif(not REF_MODELS, reason="reference checkpoints are only mounted in the build container")
def test_reference_checkpoints_import():
    s = _states(128, seed=3)
    for path in (REF_MODELS[0], REF_MODELS[len(REF_MODELS) // 2], REF_MODELS[-1]):
        m = TI.read_tflite(path)
        roles = TI.output_roles(m)
        w = TI.mlp12x100_from_tflite(path)
        ev, pr = ref_nets.mlp12x100_forward_np(w, s)
        g = TI.tflite_forward_np(m, s)
        assert np.max(np.abs(ev - g[roles["value"]][:, 0])) < 1e-4, path  # BLAS summation order differs
        assert np.max(np.abs(pr - g[roles["policy"]])) < 1e-4, path
        assert m["outputs"] == [roles["policy"], roles["value"]]  # tourney.pyx:153-154: 0 policy, 1 value
    # and through the engine's network kernel (emulation build here)
    t = make_trainer("emu", 16, "", 1, 50, 16, 1.0, 0.25, 0, 1, False)
    t.set_net(1, w)
    ev, pr = t.net_forward(s)
    assert np.max(np.abs(ev - g[roles["value"]][:, 0])) < 1e-4
    assert np.max(np.abs(pr - g[roles["policy"]])) < 1e-4


@pytest.mark.skipif(not REF_MODELS, reason="reference checkpoints are only mounted in the build container")
def test_generation_with_a_trained_reference_checkpoint_replays_on_the_oracle():
    """a trained network plays differently from a random one (sharper priors, solved lines):
    a fused generation driven by the last reference checkpoint is replayed move for move"""
    from oracle import oracle as O
    from tests import harness as H

    G, S_, spe = 12, 120, 16
    w = TI.mlp12x100_from_tflite(REF_MODELS[-1])
    f = make_trainer("emu", G, "", 4242, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False)
    f.set_net(1, w)
    assert f.run()
    o = O.Trainer(G, seed=4242, max_searches=S_, searches_per_eval=spe)
    o.set_stagger(False)
    H.play_generation(o, G, spe, lambda s: f.net_forward(s))
    for x, y in zip(H.get_samples(f), H.get_samples(o)):
        assert x.tobytes() == y.tobytes()
    assert f.score() == o.score() and f.avg_mate_length() == o.avg_mate_length()
