"""sample files as the reference writes and reads them (main.pyx:189-219)"""
import numpy as np

from corintho_ai_amd import samples_io
from tests import harness as H
from tests.engines import make_trainer


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
