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
quirk 2, and the result attribution.  Weights: tests/golden/trained_last.npz (model_93, committed as data).

Independent games give 19.9 % / 26.95 % (10^5 oracle games: +2.9 sigma / +1.5 sigma of the reference's sampling
error over 1189 games each, joint p = 0.006) -- but the reference's games of this pairing are NOT independent:
every `Tourney` object seeds its matches from a default-constructed mt19937 in addMatch order (tourney.h:43,
tourney.cpp:86), rating/round.py gives each of its `cpu_count() - 8` worker processes a match file
`a b / b a / ...` of which the first `len` lines are read (round.py:206-214, tourney.pyx:96-99), and these two
players never look at a network output -- so a game of this pairing is a function of its POSITION in the match
file alone, and the 2 x 1189 games are the same few positions again and again.  results.txt holds exactly
10 000 000 games = 1000 rounds of the default 10 000; with the 88 workers of a 96-vCPU machine a file has 114
lines = 57 positions per colour, and the 57 games at those positions score 22.8 % / 29.8 %
(test_one_search_pairing_at_the_reference_seed_positions) against the reference's 23.2 % / 28.8 %.
(The rows between searching players behave as independent samples -- see the real-search test below -- because
there a game depends on float32 network outputs, which a batched TFLite evaluation does not reproduce bit for
bit from one batch composition to the next.)"""
import os

import numpy as np
import pytest

from corintho_ai_amd import NET_MLP12X100, Tourney, nets

from tests import ref_nets
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


def test_one_search_pairing_at_the_reference_seed_positions():
    """the pairing as rating/round.py lays it out: `95 96` at the even match positions, `96 95` at the odd ones, the
    first 114 positions (10 000 games over 88 worker processes): the oracle's 57 + 57 games, each a function of its
    position only, against the reference's rates.  Deterministic on this side; the reference's multiplicities per
    position are unknown, so the comparison is to 2 points, not exact."""
    w = np.load(os.path.join(GOLDEN, "trained_last.npz"))["weights"]
    pairs = 57
    t = O.Tourney(8, "")
    t.addPlayer(0, 0, 1, 1, 3.0, 0.25, False)
    t.addPlayer(1, -1, 1600, 16, 3.0, 0.25, True)
    for _ in range(pairs):
        t.addMatch(0, 1, False)
        t.addMatch(1, 0, False)
    H.play_tourney(t, [-1, 0], {0: lambda s: ref_nets.mlp12x100_forward_np(w, s)}, rows=2 * pairs)
    sc = np.array([t.match_score(i) for i in range(2 * pairs)])
    p_first = float(np.mean(sc[0::2] == 1.0))
    p_second = float(np.mean(sc[1::2] == 0.0))
    ref_first, ref_second = REF_FIRST[0] / sum(REF_FIRST), REF_SECOND[2] / sum(REF_SECOND)
    print("first 57 positions per colour: %.3f / %.3f, reference %.3f / %.3f" % (p_first, p_second, ref_first, ref_second))
    assert abs(p_first - ref_first) < 0.02 and abs(p_second - ref_second) < 0.02
    assert not np.any(sc == 0.5)  # and no draw among them, as in the reference's 2378 games


def test_oracle_reproduces_the_reference_rates():
    w = np.load(os.path.join(GOLDEN, "trained_last.npz"))["weights"]
    n_each = 1500
    t = _play(lambda: O.Tourney(8, ""), n_each)
    H.play_tourney(t, [-1, 0], {0: lambda s: ref_nets.mlp12x100_forward_np(w, s)}, rows=2 * n_each)
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


# ---------------------------------------------------------------------------------------------------------------
# Real searches: rows of rating/results.txt between players at the PRODUCTION setting (players.txt: every
# checkpoint `1600 16 3.0 0.25 0`: 1600 simulations per move, 16 per evaluation, c_puct 3, epsilon 0.25, testing).
# Chosen a priori: the two best checkpoints both ways, the best against an early one, a middle against an early
# one, best/second-best against the middle one, best against another early one, two early ones -- 14 rows with
# 900-1500 reference games each, first-player win rates from 7 % to 69 % -- and 8 rows against the random player.
# Weights: the five checkpoints as committed data (tests/golden/trained_*.npz, ref_models.npz).
#
# What reproduces them (measured on the MI355X, 2000 matches per row, tools/exp/ref_rows.py):
#   * every match reading the evaluations of ITS OWN requests: all 14 rows within 2.2 sigma of the reference's
#     win rate (sum of z^2 = 14.5 over 14 rows), pooled draw rate 0.85 % against 0.88 %;
#   * the offset table of the committed tourney.cpp:55-62 (match i reads where match i - 1's
#     requests would start, i.e. most matches read the rows of OTHER matches): off by 10-22 sigma -- with other
#     positions' evaluations every checkpoint plays alike (0.40-0.48 whatever the pairing).
# So results.txt was produced by matches that read their own rows, and the search layer restated in oracle/ and
# built in csrc/ -- PUCT arithmetic, prior quantisation, Dirichlet noise, the solver, move choice, tree reuse --
# reproduces the reference's own tournament statistics at the production setting.  The engine keeps the committed
# table by default (parity with the committed source, tests/test_tourney.py); `set_exact_offsets` is the switch.
PLAYER_MODEL = {0: 93, 1: 92, 46: 47, 89: 4, 90: 3}  # rating/tourney/players.txt: player id -> checkpoint
REF_SEARCH_ROWS = {  # (first player, second player): the first player's (wins, draws, losses) in rating/results.txt
    (0, 1): (393, 46, 865), (1, 0): (368, 9, 927),
    (0, 89): (625, 0, 283), (89, 0): (69, 17, 822),
    (0, 46): (706, 13, 772), (46, 0): (546, 0, 945),
    (46, 90): (627, 3, 402), (90, 46): (107, 7, 916),
    (1, 46): (689, 12, 762), (46, 1): (486, 6, 971),
    (0, 90): (586, 6, 354), (90, 0): (70, 3, 873),
    (89, 90): (460, 25, 742), (90, 89): (243, 0, 983),
}
REF_RANDOM_ROWS = {  # against player 96, the random player: the searcher never lost or drew
    (0, 96): (76, 0, 0), (96, 0): (0, 0, 76), (46, 96): (77, 0, 0), (96, 46): (0, 0, 76),
    (90, 96): (76, 0, 0), (96, 90): (0, 0, 76), (89, 96): (76, 0, 0), (96, 89): (0, 0, 76),
}


def test_the_real_search_rows_are_the_reference_rows():
    path = os.path.join(REFERENCE, "corintho_ai/rating/results.txt")
    if not os.path.exists(path):
        pytest.skip("reference tree not mounted")
    rows = {tuple(map(int, l.split()[:2])): tuple(map(int, l.split()[2:])) for l in open(path) if l.strip()}
    for k, v in {**REF_SEARCH_ROWS, **REF_RANDOM_ROWS}.items():
        assert rows[k] == v, k
    players = open(os.path.join(REFERENCE, "corintho_ai/rating/tourney/players.txt")).read().split("\n")
    for pid, mid in PLAYER_MODEL.items():
        assert players[1 + pid] == "%d 1600 16 3.0 0.25 0" % mid


def _checkpoints():
    w = {}
    for tag, mid in (("early", 3), ("middle", 47), ("last", 93)):
        w[mid] = np.load(os.path.join(GOLDEN, "trained_%s.npz" % tag))["weights"]
    d = np.load(os.path.join(GOLDEN, "ref_models.npz"))
    for k in d.files:
        w[int(k.split("_")[1])] = d[k]
    return w


def _add_players(t, a, b, sims=1600, spe=16):
    for p in (a, b):
        if p == 96:
            t.addPlayer(96, -1, 1600, 16, 3.0, 0.25, True)
        else:
            t.addPlayer(p, PLAYER_MODEL[p], sims, spe, 3.0, 0.25, False)


def _wdl(sc):
    sc = np.asarray(sc)
    return int(np.sum(sc == 1.0)), int(np.sum(sc == 0.5)), int(np.sum(sc == 0.0))


def _z_win(x, y):
    nx, ny = sum(x), sum(y)
    p = (x[0] + y[0]) / (nx + ny)
    return (x[0] / nx - y[0] / ny) / max(np.sqrt(p * (1 - p) * (1 / nx + 1 / ny)), 1e-12)


def _chi2_sf(x, k):
    from scipy.stats import chi2

    return float(chi2.sf(x, k))


@pytest.mark.parametrize("engine", ["emu"])
def test_exact_offsets_switch_equals_oracle(engine):
    """the diagnostic switch on the CPU: engine (emulation build) and oracle agree bit for bit with it on, in a
    tournament whose matches would read other matches' rows through the reference's table"""
    W = _checkpoints()
    a, b, n = 0, 89, 10
    e = Tourney(1, "", trace=True, _cdll=cdll(engine))
    o = O.Tourney(4, "", trace=True)
    for t in (e, o):
        _add_players(t, a, b, sims=40, spe=8)
        for i in range(n):
            t.addMatch(*((a, b) if i % 3 else (b, a)), False)
        t.set_exact_offsets(True)
    nets_by_model = {m: (lambda s, m=m: ref_nets.mlp12x100_forward_np(W[m], s)) for m in (PLAYER_MODEL[a], PLAYER_MODEL[b])}
    ids = sorted(nets_by_model)
    H.play_tourney(o, ids, nets_by_model, rows=n * 8)
    H.play_tourney(e, ids, nets_by_model, rows=n * 8)
    for i in range(n):
        assert e.match_score(i) == o.match_score(i)
        assert np.array_equal(e.trace(i), o.trace(i)), i
    # and the table of tourney.cpp:55-62 gives different games from the same seeds (the switch does something)
    q = O.Tourney(4, "", trace=True)
    _add_players(q, a, b, sims=40, spe=8)
    for i in range(n):
        q.addMatch(*((a, b) if i % 3 else (b, a)), False)
    H.play_tourney(q, ids, nets_by_model, rows=n * 8)
    assert any(not np.array_equal(q.trace(i), o.trace(i)) for i in range(n))


@pytest.mark.gpu
def test_engine_reproduces_the_real_search_rows():
    """14 + 8 rows of the reference's tournament replayed at its production setting on the MI355X, 2000 matches per
    row (the network at float32-equivalent split precision on the device), and the first 64 matches of every
    real-search row replayed on the oracle bit for bit (every request row evaluated by the same device kernel), so
    that oracle and engine are pinned by the same reference-held numbers."""
    from corintho_ai_amd import NET_MLP12X100_X6, Trainer

    W = _checkpoints()
    n, n_oracle = 2000, 64
    # one evaluator for the oracle's replays: the same kernels, rows are batch-independent
    ev = Trainer(n_oracle, "", 1, 1600, 16, 3.0, 0.25, 0, 1, True, stagger=False, arena_units=4096)
    table = []
    zs = []
    tot_here, tot_ref = np.zeros(3), np.zeros(3)
    for (a, b), ref in {**REF_SEARCH_ROWS, **REF_RANDOM_ROWS}.items():
        t = Tourney(1, "", trace=(96 not in (a, b)))
        _add_players(t, a, b)
        for _ in range(n):
            t.addMatch(a, b, False)
        t.set_exact_offsets(True)
        models = sorted({PLAYER_MODEL[p] for p in (a, b) if p != 96})
        for m in models:
            t.set_net(m, NET_MLP12X100_X6, W[m])
        assert t.run()
        got = _wdl([t.match_score(i) for i in range(n)])
        if 96 in (a, b):
            searcher_fails = got[1] + (got[2] if a != 96 else got[0])
            table.append("%2d %2d  here %4d/%3d/%4d  reference %3d/%d/%3d" % ((a, b) + got + ref))
            assert searcher_fails <= n // 200, "the searcher lost or drew %d of %d games against the random player" % (searcher_fails, n)
            t.close()
            continue
        z = _z_win(got, ref)
        zs.append(z)
        tot_here += got
        tot_ref += ref
        table.append("%2d %2d  here %4d/%3d/%4d = %.3f  reference %3d/%2d/%3d = %.3f  z %+.2f"
                     % ((a, b) + got + (got[0] / n,) + ref + (ref[0] / sum(ref), z)))
        # the oracle plays the first matches of the same tournament (same seeds: addMatch order)
        o = O.Tourney(16, "", trace=True)
        _add_players(o, a, b)
        for _ in range(n_oracle):
            o.addMatch(a, b, False)
        o.set_exact_offsets(True)
        for slot, m in enumerate(models):
            ev.set_net(NET_MLP12X100_X6, W[m], slot=slot)
        fw = {m: (lambda s, slot=slot: ev.net_forward(s, slot=slot)) for slot, m in enumerate(models)}
        H.play_tourney(o, models, fw, rows=n_oracle * 16)
        for i in range(n_oracle):
            assert o.match_score(i) == t.match_score(i), (a, b, i)
            assert np.array_equal(o.trace(i), t.trace(i)), (a, b, i)
        t.close()
    print("\n".join(table))
    zs = np.array(zs)
    joint = float(np.sum(zs ** 2))
    p_joint = _chi2_sf(joint, len(zs))
    d_here, d_ref = tot_here[1] / tot_here.sum(), tot_ref[1] / tot_ref.sum()
    z_draw = (d_here - d_ref) / np.sqrt(d_ref * (1 - d_ref) * (1 / tot_here.sum() + 1 / tot_ref.sum()))
    print("win rates: max |z| %.2f, sum z^2 = %.1f over %d rows (p = %.3f); draws %.4f here, %.4f in the reference (z %+.2f)"
          % (np.max(np.abs(zs)), joint, len(zs), p_joint, d_here, d_ref, z_draw))
    # The bounds are those of the POPULATION of rows (below: the reference's rows scatter about 4 x wider than independent
    # binomial samples -- its tournament replays seeds -- with no bias); these 14 happen to sit at sum z^2 = 14.7 at 2000
    # matches and 24.6 at 8000 (round 3), with `1 0` at -3.8 sigma there.  Not asserted at the level of independent samples.
    assert np.all(np.abs(zs) < 6.0)
    assert joint < 7.0 * len(zs) and abs(float(np.mean(zs))) < 3.5 * 2.1 / np.sqrt(len(zs))
    assert abs(z_draw) < 3.5 * 1.9


@pytest.mark.gpu
def test_the_committed_offset_table_does_not_reproduce_them():
    """control: the same tournaments through the table of tourney.cpp:55-62 (the engine's default, as the committed
    source) -- matches search on other matches' evaluations, checkpoints play alike, and the rows with a clear
    favourite are missed by more than 8 sigma"""
    from corintho_ai_amd import NET_MLP12X100_X6

    W = _checkpoints()
    n = 1000
    for (a, b) in ((89, 0), (90, 46)):
        t = Tourney(1, "")
        _add_players(t, a, b)
        for _ in range(n):
            t.addMatch(a, b, False)
        for m in {PLAYER_MODEL[a], PLAYER_MODEL[b]}:
            t.set_net(m, NET_MLP12X100_X6, W[m])
        assert t.run()
        got = _wdl([t.match_score(i) for i in range(n)])
        z = _z_win(got, REF_SEARCH_ROWS[(a, b)])
        print("%d %d through the committed table: %s against %s, z %+.1f" % (a, b, got, REF_SEARCH_ROWS[(a, b)], z))
        assert abs(z) > 8.0
        t.close()


# ---------------------------------------------------------------------------------------------------------------
# The rows as a POPULATION (round 4).  25 of the reference's checkpoints are committed as data (the five above, the
# next-strongest two, 18 drawn with a fixed seed before any row was replayed: tools/gen_trained_golden.py
# population_players) with the 600 rows of results.txt between them (tests/golden/ref_results_pop.json).  All 600 were
# replayed at 2000 matches (tools/ref_population.py; profiles/r04_reference_rows_population.{md,json}).  What they show:
#   * no bias: mean z = -0.06 over 600 rows; no checkpoint plays stronger or weaker here than there (largest mean signed
#     z of a checkpoint over its 48 rows: 2.1 standard errors);
#   * but the rows scatter 4.3 times wider than two independent binomial samples would (sum z^2 = 2591 over 600 rows;
#     robustly, median z^2 = 1.65 where 0.455 is expected: 3.6 x), the reference's DRAW counts by the same factor (3.5 x);
#   * the scatter belongs to the ROW, not to the pairing: z(a, b) and -z(b, a) are uncorrelated (+0.015 over 300
#     pairings; a strength difference of a pairing would move both rows) -- so round 3's "player 1 moving first" rows were
#     three draws from this distribution, not a property of that player (its 24 first-mover rows: mean z -0.09);
#   * and the reference's own rows contradict each other where they deviate most: against model_1 (the second-weakest
#     checkpoint) every first mover wins 87-95 % here and 85-98 % there -- except the two rows with the MOST reference
#     games of that column, `51 92` (74.6 % of 799) and `10 92` (79.1 % of 642; z = +14.5 and +11.3).  More games make an
#     independent sample more accurate; here they mark the rows rating/round.py kept scheduling because their variance
#     estimate stayed high (:131-192), and it puts the pairs it pops first at the head of every worker's match file
#     (:196-214), where matches get the first outputs of a default-constructed mt19937 (tourney.cpp:86): the same seeds
#     for the same pairing round after round.  The one-search pairing above shows what that does when no float enters a
#     game (exact replicas, reproduced by the oracle at those positions); with a network in the loop the replicas are
#     partial and show as over-dispersion.
# The rows therefore pin the search layer as a population: location exactly (no bias, no checkpoint- or pairing-level
# effect), dispersion at the level the reference's own sampling explains -- and wrong search parameters move the
# LOCATION by 10 sigma and more (profiles/r03_reference_rows_sweep.md).  The test below replays all 156 rows between 13
# of the 25 players (fixed in advance: round 3's five, players 2 and 3, the first six of the seeded draw) at 1000 matches
# and asserts those properties; the rows beyond 6 sigma in the 2000-match run are named, not tuned away.
POP_PLAYERS = [0, 1, 2, 3, 46, 89, 90, 4, 7, 10, 22, 25, 33]
POP_KNOWN_OUTLIERS = {(89, 4): -8.3, (3, 25): +6.8}  # |z| > 6 at 2000 matches (of the seven such rows among all 600, those between POP_PLAYERS)


def _population_weights():
    W = _checkpoints()
    d = np.load(os.path.join(GOLDEN, "ref_models_pop.npz"))
    for k in d.files:
        W[int(k.split("_")[1])] = d[k]
    return W


def _population_rows():
    import json

    d = json.load(open(os.path.join(GOLDEN, "ref_results_pop.json")))
    return {(r[0], r[1]): tuple(r[2:]) for r in d["rows"] if r[0] in POP_PLAYERS and r[1] in POP_PLAYERS}


def test_the_population_rows_are_the_reference_rows():
    path = os.path.join(REFERENCE, "corintho_ai/rating/results.txt")
    if not os.path.exists(path):
        pytest.skip("reference tree not mounted")
    import json

    rows = {tuple(map(int, l.split()[:2])): tuple(map(int, l.split()[2:])) for l in open(path) if l.strip()}
    d = json.load(open(os.path.join(GOLDEN, "ref_results_pop.json")))
    assert len(d["rows"]) == 600 and len(d["players"]) == 25
    for a, b, w, dr, l in d["rows"]:
        assert rows[(a, b)] == (w, dr, l)
    players = open(os.path.join(REFERENCE, "corintho_ai/rating/tourney/players.txt")).read().split("\n")
    for p in d["players"]:
        assert players[1 + p] == "%d 1600 16 3.0 0.25 0" % (93 - p)
    assert len(_population_rows()) == 156


@pytest.mark.gpu
def test_engine_reproduces_the_reference_rows_as_a_population():
    from corintho_ai_amd import NET_MLP12X100_X6

    W = _population_weights()
    ref = _population_rows()
    n = 1000
    zs, zh, table = {}, [], []
    tot_here, tot_ref = np.zeros(3), np.zeros(3)
    def play(key):
        a, b = key
        t = Tourney(1, "")
        for p in (a, b):
            t.addPlayer(p, 93 - p, 1600, 16, 3.0, 0.25, False)
        for _ in range(n):
            t.addMatch(a, b, False)
        t.set_exact_offsets(True)
        for p in (a, b):
            t.set_net(93 - p, NET_MLP12X100_X6, W[93 - p])
        assert t.run()
        sc = [t.match_score(i) for i in range(n)]
        t.close()
        return sc

    # a row is a tournament of its own (two networks), as long as its longest game whatever its size: four at a time, each on
    # its own streams (the engine keeps no state outside a trainer; ctypes releases the interpreter lock during a call)
    from concurrent.futures import ThreadPoolExecutor

    keys = sorted(ref)
    with ThreadPoolExecutor(4) as ex:
        scores = dict(zip(keys, ex.map(play, keys)))
    for (a, b) in keys:
        r, sc = ref[(a, b)], scores[(a, b)]
        got = _wdl(sc)
        zs[(a, b)] = _z_win(got, r)
        zh.append(_z_win(_wdl(sc[: n // 2]), _wdl(sc[n // 2:])))  # this side against itself: two halves of fresh seeds
        tot_here += got
        tot_ref += r
        table.append("%2d %2d  here %4d/%3d/%4d = %.3f  reference %4d/%2d/%4d = %.3f  z %+.2f" % ((a, b) + got + (got[0] / n,) + r + (r[0] / sum(r), zs[(a, b)])))
    print("\n".join(table))
    z = np.array([zs[k] for k in sorted(zs)])
    zh = np.array(zh)
    k = len(z)
    phi_mean, phi_med = float(np.mean(z ** 2)), float(np.median(z ** 2) / 0.455)
    pairs = np.array([(zs[(a, b)], -zs[(b, a)]) for (a, b) in zs if a < b])
    rho = float(np.corrcoef(pairs[:, 0], pairs[:, 1])[0, 1])
    by_first = {p: float(np.mean([zs[(a, b)] for (a, b) in zs if a == p])) for p in (0, 1, 2, 3)}
    shift = {p: float(np.mean([zs[(a, b)] for (a, b) in zs if a == p] + [-zs[(a, b)] for (a, b) in zs if b == p])) for p in POP_PLAYERS}
    d_here, d_ref = tot_here[1] / tot_here.sum(), tot_ref[1] / tot_ref.sum()
    print("%d rows x %d matches: mean z %+.3f, median z %+.3f; dispersion mean z^2 %.2f, robust %.2f; this side against itself %.2f; "
          "rows of a pairing: correlation %+.3f over %d pairings; first movers 0-3: %s; draws %.4f here, %.4f in the reference"
          % (k, n, z.mean(), np.median(z), phi_mean, phi_med, float(np.mean(zh ** 2)), rho, len(pairs),
             {p: round(v, 2) for p, v in by_first.items()}, d_here, d_ref))
    # (1) this side's games are independent samples: two halves of a row agree like binomial samples
    assert 0.7 < float(np.mean(zh ** 2)) < 1.4
    # (2) location: no bias over the population, for no checkpoint, and not for "a strong checkpoint moving first"
    se = np.sqrt(phi_mean)
    assert abs(z.mean()) < 3.5 * se / np.sqrt(k) and abs(np.median(z)) < 0.6
    for p, v in shift.items():
        assert abs(v) < 3.5 * se / np.sqrt(2 * (len(POP_PLAYERS) - 1)), (p, v)
    for p, v in by_first.items():
        assert abs(v) < 3.5 * se / np.sqrt(len(POP_PLAYERS) - 1), (p, v)
    # (3) the excess scatter is the rows' (the reference's sampling), not the pairings' (a difference in playing strength)
    assert abs(rho) < 3.0 / np.sqrt(len(pairs))
    # (4) and it is the size the 600-row replay found (3.6 robust / 4.3 mean at 2000 matches; a little less at 1000): a
    #     search that differed from the reference's would add to it (c_puct 2 or 4 instead of 3: 12 and more per row)
    assert 1.5 < phi_med < 5.5 and phi_mean < 7.0
    # (5) the rows beyond 6 sigma are the named ones, on the side they were found
    for key, zz in zs.items():
        if abs(zz) > 6.0:
            assert key in POP_KNOWN_OUTLIERS and zz * POP_KNOWN_OUTLIERS[key] > 0, (key, zz)
    # (6) draws: the pooled rate agrees
    z_draw = (d_here - d_ref) / np.sqrt(d_ref * (1 - d_ref) * (1 / tot_here.sum() + 1 / tot_ref.sum()) * 3.5)
    assert abs(z_draw) < 3.5
