"""Match / Tourney: the reference's tournament interface
(corintho_ai/cpp/include/{match,tourney}.h, rating/tourney.pyx) on the oracle, the emulation
build and -- with -m gpu -- the MI355X build; bit-exact per-ply traces, request rows, scores."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from corintho_ai_amd import Tourney, _lib
from oracle import oracle as O
from tests import harness as H
from tests.engines import ENGINES, cdll

HERE = os.path.dirname(os.path.abspath(__file__))


def _uid_lib():
    so = os.path.join(HERE, "cxx", "libuid_check.so")
    src = os.path.join(HERE, "cxx", "uid_check.cpp")
    if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, src])
    L = C.CDLL(so)
    L.uid_draws.argtypes = [C.c_uint32, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]
    return L


def test_uniform_int_distribution_restatement_matches_the_cxx_library():
    """match.cpp:199-200 draws the random player's move with std::uniform_int_distribution on the
    match's std::mt19937; the restatement must consume the stream exactly as libstdc++ does"""
    L = _uid_lib()
    rng = np.random.default_rng(1)
    for seed in (5489, 1, 0xDEADBEEF):
        ns = np.concatenate([rng.integers(1, 97, 4000), [1, 1, 2, 96, 3, 1]]).astype(np.int32)
        want = np.zeros(ns.size, np.int32)
        nxt = C.c_uint32()
        L.uid_draws(seed, ns.ctypes.data_as(C.POINTER(C.c_int32)), ns.size, want.ctypes.data_as(C.POINTER(C.c_int32)),
                    C.byref(nxt))
        g = O.MT19937(seed)
        got = [O.uniform_below(g, int(n)) for n in ns]
        assert got == [int(x) for x in want]
        assert g() == nxt.value


# (player_id, model_id, max_searches, spe, c_puct, epsilon, random)
PLAYERS_A = [(0, 0, 40, 8, 1.0, 0.25, False), (1, 1, 24, 4, 1.5, 0.25, False), (2, 0, 16, 16, 1.0, 0.0, False),
             (3, -1, 0, 0, 1.0, 0.25, True)]
MATCHES_A = [(0, 1), (1, 0), (2, 1), (0, 3), (3, 1), (0, 2), (1, 2), (2, 3)]
# one network for everybody: every unfinished match then waits for the same id, the case in which
# the reference's offset table (tourney.cpp:55-62) and its writeRequests order agree
PLAYERS_B = [(0, 0, 32, 8, 1.0, 0.25, False), (1, 0, 48, 8, 3.0, 0.25, False)]
MATCHES_B = [(0, 1), (1, 0), (0, 0), (1, 1), (0, 1)]


def _build(factory, players, matches):
    t = factory()
    for p in players:
        t.addPlayer(*p)
    for a, b in matches:
        t.addMatch(a, b, False)
    return t


def _rows(players, matches):
    spe = {p[0]: p[3] for p in players}
    return max(1, sum(spe[a] + spe[b] for a, b in matches))


def _nets():
    return {0: lambda s: H.hash_net(s, 11), 1: lambda s: H.hash_net(s, 22)}


@pytest.mark.parametrize("players,matches", [(PLAYERS_A, MATCHES_A), (PLAYERS_B, MATCHES_B)], ids=["mixed", "one_model"])
def test_oracle_tourney_plays_to_the_end(players, matches):
    t = _build(lambda: O.Tourney(2, "", trace=True), players, matches)
    model_ids = sorted({p[1] for p in players})
    r = H.play_tourney(t, model_ids, _nets(), _rows(players, matches))
    assert t.all_done() and r["rounds"] > 3
    for i in range(t.num_matches()):
        assert t.match_info(i)["done"] == 1
        assert t.match_score(i) in (0.0, 0.5, 1.0)
        assert t.match_info(i)["result"] in (1, 2, 3)  # loss / draw / win for the first player


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("players,matches", [(PLAYERS_A, MATCHES_A), (PLAYERS_B, MATCHES_B)], ids=["mixed", "one_model"])
def test_tourney_matches_oracle(engine, players, matches, tmp_path):
    """same players, same pairings, same stand-in networks: every evaluation request of every
    round, every per-ply trace (incl. the random player's draws), results and the scores file"""
    model_ids = sorted({p[1] for p in players})
    rows = _rows(players, matches)
    e = _build(lambda: Tourney(1, "", trace=True, _cdll=cdll(engine)), players, matches)
    o = _build(lambda: O.Tourney(2, "", trace=True), players, matches)
    ra = H.play_tourney(e, model_ids, _nets(), rows, record=True)
    rb = H.play_tourney(o, model_ids, _nets(), rows, record=True)
    assert ra["rounds"] == rb["rounds"]
    assert [(a[0], a[1].tobytes()) for a in ra["log"]] == [(b[0], b[1].tobytes()) for b in rb["log"]]
    for i in range(len(matches)):
        assert np.array_equal(e.trace(i), o.trace(i)), "per-ply trace of match %d" % i
        assert e.match_info(i)["result"] == o.match_info(i)["result"]
        assert e.match_score(i) == o.match_score(i)
        assert (e.match_info(i)["player1"], e.match_info(i)["player2"]) == matches[i]
    c, st = o.counters(), e.stats()
    assert (st["searches"], st["evals"], st["nodes"], st["plies"]) == (c["searches"], c["leaf_evals"],
                                                                       c["nodes_created"], c["plies"])
    fa, fb = tmp_path / "a.txt", tmp_path / "b.txt"
    e.writeScores(fa)
    o.writeScores(fb)
    assert fa.read_text() == fb.read_text() and len(fa.read_text().splitlines()) == len(matches)


@pytest.mark.parametrize("engine", ENGINES)
def test_tourney_argument_errors(engine):
    t = Tourney(1, "", _cdll=cdll(engine))
    t.addPlayer(0, 0, 8, 4, 1.0, 0.25, False)
    t.addPlayer(1, -1, 0, 0, 1.0, 0.25, True)
    with pytest.raises(_lib.EngineError, match="unknown player"):
        t.addMatch(0, 7, False)
    with pytest.raises(_lib.EngineError, match="random"):
        t.addMatch(1, 1, False)
    with pytest.raises(_lib.EngineError, match="without matches"):
        t.all_done()


@pytest.mark.parametrize("engine", ENGINES)
def test_fused_tourney_replays_on_the_oracle(engine):
    """networks on the device (ca_tourney_set_net / ca_tourney_run): the matches are the ones the
    reference protocol plays when the driver evaluates the same networks"""
    from corintho_ai_amd import nets
    from tests.engines import make_trainer

    players = [(0, 0, 40, 8, 1.0, 0.25, False), (1, 1, 32, 8, 1.0, 0.25, False), (2, 1, 24, 4, 2.0, 0.25, False),
               (3, -1, 0, 0, 1.0, 0.25, True)]
    matches = [(0, 1), (1, 0), (2, 0), (0, 3), (3, 2), (1, 2)]
    weights = {0: nets.init_mlp12x100(seed=5, bn_noise=True), 1: nets.init_mlp12x100(seed=6, bn_noise=True)}
    f = _build(lambda: Tourney(1, "", trace=True, _cdll=cdll(engine)), players, matches)
    for mid, w in weights.items():
        f.set_net(mid, 1, w)
    assert not f.run(max_rounds=3)
    assert f.run()
    # the same networks evaluated for the oracle by the engine's kernels
    evaluators = {}
    for mid, w in weights.items():
        t = make_trainer(engine, 64, "", 1, 50, 16, 1.0, 0.25, 0, 1, False)
        t.set_net(1, w)
        evaluators[mid] = t
    o = _build(lambda: O.Tourney(2, "", trace=True), players, matches)
    H.play_tourney(o, [-1, 0, 1], {mid: (lambda s, t=t: t.net_forward(s)) for mid, t in evaluators.items()},
                   _rows(players, matches))
    for i in range(len(matches)):
        assert np.array_equal(f.trace(i), o.trace(i)), "per-ply trace of match %d" % i
        assert f.match_score(i) == o.match_score(i)


def test_pairing_files_are_read_like_the_reference_driver(tmp_path):
    from corintho_ai_amd.tourney import read_pairings

    (tmp_path / "players.txt").write_text("3\n0 1600 16 1.0 0.25 0\n4 800 8 1.5 0.1 0\n-1 0 0 1.0 0.25 1\n")
    (tmp_path / "matches.txt").write_text("2\n0 1 0\n2 0 1\n")
    players, matches = read_pairings(tmp_path / "players.txt", tmp_path / "matches.txt")
    assert players == [(0, 0, 1600, 16, 1.0, 0.25, False), (1, 4, 800, 8, 1.5, 0.1, False), (2, -1, 0, 0, 1.0, 0.25, True)]
    assert matches == [(0, 1, False), (2, 0, True)]


def _few_searches_tourney(factory):
    """tests/cpp/tourney_test.cpp:30-69 (TourneyTest.FewSearches): six searching players with their
    own model ids (2, 3, 4 simulations per move; 1 or all of them per evaluation), a random player
    with id 0, every ordered pairing"""
    t = factory()
    counter = 1
    for max_searches in (2, 3, 4):
        for spe in (1, max_searches):
            t.addPlayer(counter, counter, max_searches, spe, 1.0, 0.25, False)
            counter += 1
    t.addPlayer(0, 0, 1, 1, 0.0, 0.0, True)
    n = 0
    for i in range(counter):
        for j in range(counter):
            if i != j:
                t.addMatch(i, j, False)
                n += 1
    return t, counter, n


@pytest.mark.parametrize("engine", ENGINES)
def test_reference_tourney_few_searches(engine):
    nets_by_model = {m: (lambda s, m=m: H.hash_net(s, 100 + m)) for m in range(1, 7)}
    e, counter, n = _few_searches_tourney(lambda: Tourney(1, "", trace=True, _cdll=cdll(engine)))
    o, _, _ = _few_searches_tourney(lambda: O.Tourney(1, "", trace=True))
    assert n == 42 and e.num_matches() == 42
    # model id 0 belongs to the random player: no requests, but its matches move (tourney_test.cpp:52-66)
    ra = H.play_tourney(e, list(range(counter)), nets_by_model, 4 * 12, record=True)
    rb = H.play_tourney(o, list(range(counter)), nets_by_model, 4 * 12, record=True)
    assert e.all_done() and o.all_done()
    assert [(a[0], a[1].tobytes()) for a in ra["log"]] == [(b[0], b[1].tobytes()) for b in rb["log"]]
    for m, rows in ra["log"]:
        assert np.all((rows >= 0.0) & (rows <= 1.0))
    for i in range(n):
        assert np.array_equal(e.trace(i), o.trace(i)), "match %d" % i
        assert e.match_score(i) == o.match_score(i)


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("max_searches,spe", [(2, 1), (2, 2), (5, 3), (9, 9), (16, 1), (16, 16)])
@pytest.mark.parametrize("random_id", [0, 1, 2])
def test_reference_match_few_searches(engine, max_searches, spe, random_id):
    """tests/cpp/match_test.cpp:28-76 (MatchTest.FewSearches): one match, either side (or neither)
    random; between iterations the side to move has 1..searches_per_eval requests with rows in [0, 1]"""
    def build(factory):
        t = factory()
        t.addPlayer(0, 0, max_searches, spe, 1.0, 0.25, random_id == 0)
        t.addPlayer(1, 1, max_searches, spe, 1.0, 0.25, random_id == 1)
        t.addMatch(0, 1, False)
        return t

    e = build(lambda: Tourney(1, "", trace=True, _cdll=cdll(engine)))
    o = build(lambda: O.Tourney(1, "", trace=True))
    evals = np.zeros(spe, np.float32)
    probs = np.zeros((spe, 96), np.float32)
    gs = np.zeros((spe, 70), np.float32)
    for t in (e, o):
        guard = 0
        while not t.all_done():
            for mid in (0, 1):
                n = t.num_requests(mid)
                assert 0 <= n <= spe
                if n:
                    t.writeRequests(gs, mid)
                    assert np.all((gs[:n] >= 0.0) & (gs[:n] <= 1.0))
                    evals[:n], probs[:n] = H.hash_net(gs[:n], 5 + mid)
                t.doIteration(evals, probs, mid)
            guard += 1
            assert guard < 100000
    assert np.array_equal(e.trace(0), o.trace(0))
    assert e.match_score(0) == o.match_score(0)
