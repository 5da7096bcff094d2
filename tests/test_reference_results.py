"""A statistical pin against results the REFERENCE itself holds: its rating tournament
(corintho_ai/rating/results.txt: `player1 player2 wins draws losses`, first player's view;
players in rating/tourney/players.txt) contains one pairing that can be re-played exactly at
negligible cost -- player 95 = the last checkpoint (model_93) searching with max_searches = 1,
searches_per_eval = 1, c_puct 3, epsilon 0.25 (`93 1 1 3.0 0.25 0`) against player 96 = the uniformly
random player (`-1 1600 16 3.0 0.25 1`):

    95 96  276 0 913        model_93 @ 1 search moves first: wins 23.2 % of 1189 games
    96 95  846 0 343        the random player moves first:   model_93 @ 1 search wins 28.8 %

(it rates BELOW the random player, gd_ratings.txt: -210 against 0 -- with one search `chooseMove`
falls through to `chooseHighProbMove`, whose `int32 max_prob` makes it play the last legal move,
SURVEY 8a quirk 2).  Re-playing it exercises, end to end and against numbers the reference produced:
the checkpoint import, the rules and terminal detection, `Match` with a random side
(`std::uniform_int_distribution` on the match's mt19937), the one-search path of `TrainMC` with
quirk 2, and the result attribution.  It is the only reference-held evidence about anything above
the rule layer; it is statistical, not bit-exact.  Weights: tests/golden/trained_last.npz (model_93,
committed as data)."""
import os

import numpy as np
import pytest

from corintho_ai_amd import NET_MLP12X100, Tourney, nets
from oracle import oracle as O
from tests import harness as H
from tests.conftest import REFERENCE
from tests.engines import ENGINES, cdll

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# rating/results.txt rows `95 96 ...` and `96 95 ...`
REF_FIRST = (276, 0, 913)    # model_93 @ 1 search moves first: its wins, draws, losses
REF_SECOND = (846, 0, 343)   # the random player moves first: ITS wins, draws, losses


def test_the_quoted_rows_are_the_reference_rows():
    path = os.path.join(REFERENCE, "corintho_ai/rating/results.txt")
    if not os.path.exists(path):
        pytest.skip("reference tree not mounted")
    rows = {tuple(map(int, l.split()[:2])): tuple(map(int, l.split()[2:])) for l in open(path) if l.strip()}
    assert rows[(95, 96)] == REF_FIRST and rows[(96, 95)] == REF_SECOND
    players = open(os.path.join(REFERENCE, "corintho_ai/rating/tourney/players.txt")).read().split("\n")
    assert players[1 + 95] == "93 1 1 3.0 0.25 0" and players[1 + 96] == "-1 1600 16 3.0 0.25 1"


def _play(factory, n_each, fused_weights=None):
    t = factory()
    t.addPlayer(0, 0, 1, 1, 3.0, 0.25, False)   # model_93, one search per move
    t.addPlayer(1, -1, 1600, 16, 3.0, 0.25, True)  # the random player
    for _ in range(n_each):
        t.addMatch(0, 1, False)
    for _ in range(n_each):
        t.addMatch(1, 0, False)
    return t


def _rates(scores, n_each):
    first = np.array(scores[:n_each])      # model moved first: score = model's result
    second = np.array(scores[n_each:])     # random moved first: score = random's result
    return float(np.mean(first == 1.0)), float(np.mean(second == 0.0)), float(np.mean(np.array(scores) == 0.5))


def _check(p_first, p_second, p_draw, n_each):
    for got, (w, d, l), name in ((p_first, (REF_FIRST[0], 0, REF_FIRST[2]), "moving first"),
                                 (p_second, (REF_SECOND[2], 0, REF_SECOND[0]), "moving second")):
        n_ref = w + l
        p_ref = w / n_ref
        sigma = np.sqrt(p_ref * (1 - p_ref) * (1.0 / n_ref + 1.0 / n_each))
        assert abs(got - p_ref) < 4.5 * sigma, "%s: %.3f here, %.3f in the reference (sigma %.3f)" % (name, got, p_ref, sigma)
    assert p_draw < 0.01  # the reference saw no draw in 2378 games
    assert p_first < 0.35 and p_second < 0.40  # it loses to the random player either way


def test_oracle_reproduces_the_reference_rates():
    w = np.load(os.path.join(GOLDEN, "trained_last.npz"))["weights"]
    n_each = 1500
    t = _play(lambda: O.Tourney(8, ""), n_each)
    H.play_tourney(t, [-1, 0], {0: lambda s: nets.mlp12x100_forward_np(w, s)}, rows=2 * n_each)
    scores = [t.match_score(i) for i in range(2 * n_each)]
    p1, p2, pd = _rates(scores, n_each)
    print("oracle: model_93 @ 1 search wins %.3f moving first (reference %.3f), %.3f moving second (reference %.3f), draws %.3f"
          % (p1, REF_FIRST[0] / sum(REF_FIRST), p2, REF_SECOND[2] / sum(REF_SECOND), pd))
    _check(p1, p2, pd, n_each)


@pytest.mark.parametrize("engine", ENGINES)
def test_engine_reproduces_the_reference_rates(engine):
    """the same tournament on the engine with the network on the device (fused); the MI355X plays 6000 matches"""
    w = np.load(os.path.join(GOLDEN, "trained_last.npz"))["weights"]
    n_each = 3000 if engine == "hip" else 400
    t = _play(lambda: Tourney(1, "", _cdll=cdll(engine)), n_each)
    t.set_net(0, NET_MLP12X100, w)
    assert t.run()
    scores = [t.match_score(i) for i in range(2 * n_each)]
    p1, p2, pd = _rates(scores, n_each)
    print("%s: model_93 @ 1 search wins %.3f moving first, %.3f moving second, draws %.3f" % (engine, p1, p2, pd))
    _check(p1, p2, pd, n_each)
