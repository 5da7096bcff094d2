"""Per-game text logs of the device engine (Trainer's num_logged, trainer.cpp:243-250) against the oracle's, byte for
byte: the search kernel records the numbers of every move choice, engine.hip prints them (csrc/logfmt.h).  The oracle's
own log is pinned in tests/test_oracle_reference_tests.py::test_per_game_text_logs."""
import os

import numpy as np
import pytest

from corintho_ai_amd import NET_MLP12X100, nets
from oracle import oracle as O
from tests import harness as H
from tests.engines import ENGINES, make_trainer


def _files(folder):
    return {n: open(os.path.join(folder, n), "rb").read() for n in sorted(os.listdir(folder))}


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("G,S_,spe,eps,logged,testing", [(5, 60, 8, 0.25, 3, False), (4, 120, 16, 0.0, 4, False),
                                                       (3, 40, 4, 0.25, 7, False), (4, 50, 8, 0.25, 2, True)])
def test_logs_match_the_oracle(engine, tmp_path, G, S_, spe, eps, logged, testing):
    """compat protocol, stand-in network on the host: the same evaluations reach both sides"""
    a, b = tmp_path / "engine", tmp_path / "oracle"
    a.mkdir()
    b.mkdir()
    t = make_trainer(engine, G, str(a), 77, S_, spe, 1.0, eps, logged, 1, testing, stagger=False)
    o = O.Trainer(G, str(b), 77, S_, spe, 1.0, eps, logged, 1, testing)
    o.set_stagger(False)
    other = lambda s: H.hash_net(s * 0.5)  # a second stand-in model for the arena
    by_player = (H.hash_net, other) if testing else None
    H.play_generation(t, G, spe, H.hash_net, nets_by_player=by_player)
    H.play_generation(o, G, spe, H.hash_net, nets_by_player=by_player)
    fa, fb = _files(a), _files(b)
    assert list(fa) == list(fb) == ["game_%d.txt" % i for i in range(min(logged, G))]
    for name in fa:
        assert fa[name] == fb[name], "%s differs from the oracle's log" % name
        assert fa[name].count(b"TURN ") >= 2 and (b"WON!" in fa[name] or b"DRAWN." in fa[name])


@pytest.mark.parametrize("engine", ENGINES)
def test_logs_of_a_fused_generation(engine, tmp_path):
    """fused mode (network and search on the device, two pools, recycled slots): the logged games are the first ones
    of the generation whatever the slot pool; the oracle replays with the device network's outputs"""
    G, S_, spe, logged = 12, 50, 8, 3
    a, b = tmp_path / "engine", tmp_path / "oracle"
    a.mkdir()
    b.mkdir()
    w = nets.init_mlp12x100(2)
    t = make_trainer(engine, G, str(a), 5, S_, spe, 1.0, 0.25, logged, 1, False, stagger=False, resident=8, pools=2)
    t.set_net(NET_MLP12X100, w)
    assert t.run()
    o = O.Trainer(G, str(b), 5, S_, spe, 1.0, 0.25, logged, 1, False)
    o.set_stagger(False)
    cap = t.stats()["resident_slots"] * spe  # rows one device evaluation takes

    def fw(s):
        parts = [t.net_forward(s[i:i + cap]) for i in range(0, s.shape[0], cap)]
        return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])

    H.play_generation(o, G, spe, fw)
    fa, fb = _files(a), _files(b)
    assert list(fa) == list(fb) == ["game_%d.txt" % i for i in range(logged)]
    for name in fa:
        assert fa[name] == fb[name], "%s differs from the oracle's log" % name


def test_logging_is_refused_once_the_games_run(tmp_path):
    from tests.emu import emulib

    L = emulib.load()
    t = make_trainer("emu", 2, "", 1, 20, 4, 1.0, 0.25, 0, 1, False, stagger=False)
    ev = np.zeros(2 * 4, np.float32)
    pr = np.zeros((2 * 4, 96), np.float32)
    t.doIteration(ev, pr)
    assert L.ca_trainer_set_logging(t._t, str(tmp_path).encode(), 1) != 0
    # an unwritable folder is skipped without an error, as the reference's ofstream is
    t2 = make_trainer("emu", 2, str(tmp_path / "missing"), 1, 20, 4, 1.0, 0.25, 2, 1, False, stagger=False)
    H.play_generation(t2, 2, 4, H.hash_net)
    assert not (tmp_path / "missing").exists()


def test_a_shard_logs_the_games_it_owns(tmp_path):
    """multi-GPU sharding (ca_config.game_base / total_games): the logged games are the first num_logged of the
    GENERATION; each shard writes the ones it holds, under their generation-wide names"""
    G, S_, spe, logged = 6, 40, 8, 4
    a, b = tmp_path / "engine", tmp_path / "oracle"
    a.mkdir()
    b.mkdir()
    o = O.Trainer(G, str(b), 9, S_, spe, 1.0, 0.25, logged, 1, False)
    o.set_stagger(False)
    H.play_generation(o, G, spe, H.hash_net)
    for base in (0, 3):
        t = make_trainer("emu", 3, str(a), 9, S_, spe, 1.0, 0.25, logged, 1, False, stagger=False, game_base=base, total_games=G)
        H.play_generation(t, 3, spe, H.hash_net)
    fa, fb = _files(a), _files(b)
    assert list(fa) == list(fb) == ["game_%d.txt" % i for i in range(logged)]
    assert fa == fb


# ---- Match logs of a tournament (Tourney::addMatch(..., logging = true), tourney.cpp:83-96; match.cpp:78-190)
T_PLAYERS = [(0, 0, 40, 8, 1.0, 0.25, False), (1, 1, 24, 4, 1.5, 0.25, False), (2, 0, 16, 16, 1.0, 0.0, False),
             (3, -1, 0, 0, 1.0, 0.25, True)]
T_MATCHES = [(0, 1, True), (1, 0, False), (2, 1, True), (0, 3, True), (3, 1, True), (0, 2, False), (2, 3, True)]


def _tourney(factory):
    t = factory()
    for p in T_PLAYERS:
        t.addPlayer(*p)
    for a, b, lg in T_MATCHES:
        t.addMatch(a, b, lg)
    return t


@pytest.mark.parametrize("engine", ENGINES)
def test_match_logs_match_the_oracle(engine, tmp_path):
    from corintho_ai_amd.tourney import Tourney
    from tests.engines import cdll

    a, b = tmp_path / "engine", tmp_path / "oracle"
    a.mkdir()
    b.mkdir()
    e = _tourney(lambda: Tourney(1, str(a), _cdll=cdll(engine)))
    o = _tourney(lambda: O.Tourney(1, str(b)))
    nets_by_model = {0: lambda s: H.hash_net(s, 11), 1: lambda s: H.hash_net(s, 22)}
    rows = sum(T_PLAYERS[x][3] + T_PLAYERS[y][3] for x, y, _ in T_MATCHES)
    H.play_tourney(e, [-1, 0, 1], nets_by_model, rows)
    H.play_tourney(o, [-1, 0, 1], nets_by_model, rows)
    fa, fb = _files(a), _files(b)
    want = sorted("match_%d_%d_%d.txt" % (x, y, i) for i, (x, y, lg) in enumerate(T_MATCHES) if lg)
    assert list(fa) == list(fb) == want
    for name in fa:
        assert fa[name] == fb[name], "%s differs from the oracle's log" % name
    # a match against the random player logs the searcher's turns only, and every chosen move
    txt = fa["match_0_3_3.txt"].decode()
    assert txt.count("CHOSE MOVE") > txt.count("TURN ") > 0 and "PLAYER 2 TO PLAY" not in txt


@pytest.mark.parametrize("engine", ENGINES)
def test_match_logs_of_a_fused_tournament(engine, tmp_path):
    """networks on the device (ca_tourney_run); the oracle replays with the same kernels' outputs"""
    from corintho_ai_amd.tourney import Tourney
    from tests.engines import cdll

    a, b = tmp_path / "engine", tmp_path / "oracle"
    a.mkdir()
    b.mkdir()
    weights = {0: nets.init_mlp12x100(seed=5, bn_noise=True), 1: nets.init_mlp12x100(seed=6, bn_noise=True)}
    f = _tourney(lambda: Tourney(1, str(a), _cdll=cdll(engine)))
    for mid, w in weights.items():
        f.set_net(mid, 1, w)
    assert f.run()
    evaluators = {}
    for mid, w in weights.items():
        t = make_trainer(engine, 64, "", 1, 50, 16, 1.0, 0.25, 0, 1, False)
        t.set_net(1, w)
        evaluators[mid] = t
    o = _tourney(lambda: O.Tourney(1, str(b)))
    rows = sum(T_PLAYERS[x][3] + T_PLAYERS[y][3] for x, y, _ in T_MATCHES)
    H.play_tourney(o, [-1, 0, 1], {mid: (lambda s, t=t: t.net_forward(s)) for mid, t in evaluators.items()}, rows)
    fa, fb = _files(a), _files(b)
    assert list(fa) == list(fb) and len(fa) == sum(lg for _, _, lg in T_MATCHES)
    assert fa == fb
