"""sample files as the reference writes and reads them (main.pyx:189-219)"""
import numpy as np
import pytest

from corintho_ai_amd import nets, samples_io
from oracle import oracle as O
from tests import harness as H
from tests.engines import ENGINES, make_trainer


def test_save_and_load_roundtrip(tmp_path):
    t = make_trainer("emu", 6, "", 3, 16, 4, 1.0, 0.25, 0, 1, False)
    H.play_generation(t, 6, 4, H.hash_net)
    gs, ev, pr = samples_io.get_samples(t)
    samples_io.save_samples(str(tmp_path / "gen_0"), gs, ev, pr)
    for name in ("game_states.npz", "evaluation_labels.npz", "probability_labels.npz"):
        assert (tmp_path / "gen_0" / name).exists()
    a, b, c = samples_io.load_samples(str(tmp_path / "gen_0"))
    assert a.tobytes() == gs.tobytes() and b.tobytes() == ev.tobytes() and c.tobytes() == pr.tobytes()
    with np.load(str(tmp_path / "gen_0" / "game_states.npz")) as z:
        assert list(z.keys()) == ["arr_0"]


@pytest.mark.parametrize("engine", ENGINES)
def test_generation_to_training_hand_off(engine, tmp_path):
    """SURVEY 8f row 2 end to end: a fused generation on the engine -> get_samples (main.pyx:189-198) ->
    the three npz files (main.pyx:200-204) -> read back as the training step reads them; every array equals
    what the oracle's writeSamples gives for the same games.  Then the replay window of main.pyx:206-217:
    earlier generations are read, and -- as in the reference, whose np.concatenate results are dropped --
    the arrays handed to training are the current generation's alone."""
    G, S_, spe = 24, 40, 8
    w = nets.init_mlp12x100(seed=1, bn_noise=True)
    folders = []
    per_gen = []
    for gen, seed in enumerate((11, 12, 13)):
        t = make_trainer(engine, G, "", seed, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False)
        t.set_net(1, w)
        assert t.run()
        folder = str(tmp_path / ("gen_%d" % gen))
        got = samples_io.samples_for_training(t, folder, old_training_samples=list(folders))
        o = O.Trainer(G, seed=seed, max_searches=S_, searches_per_eval=spe)
        o.set_stagger(False)
        H.play_generation(o, G, spe, lambda st: t.net_forward(st))
        want = H.get_samples(o)
        for x, y in zip(got, want):
            assert x.dtype == np.float32 and x.shape == y.shape and x.tobytes() == y.tobytes()
        for x, y in zip(samples_io.load_samples(folder), want):
            assert x.tobytes() == y.tobytes()
        mixed = samples_io.samples_for_training(t, folder, old_training_samples=list(folders), mix_old=True)
        assert mixed[0].shape[0] == want[0].shape[0] + sum(p[0].shape[0] for p in per_gen)
        assert mixed[2][want[2].shape[0]:].tobytes() == b"".join(p[2].tobytes() for p in per_gen)
        folders.append(folder)
        per_gen.append(want)
