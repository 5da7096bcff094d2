"""Resident-slot pool (ca_config.resident): a Trainer of G games played on R <= G slots.  The reference keeps the
trees of all its games and staggers their starts to bound the memory (trainer.cpp:184-186, 243-255); here a slot
whose game ends takes the next game, seeded from the Trainer stream by game index, and the finished game's tree
memory is given back.  A game's sequence of operations depends on its own generator and on the network's rows only
(training mode), so every per-game result -- samples, traces, results, the score -- must equal the run with all
games resident, bit for bit, and the oracle's."""
import numpy as np
import pytest

from corintho_ai_amd import nets
from oracle import oracle as O
from tests import harness as H
from tests.engines import ENGINES, make_trainer


def _generation(engine, G, S_, spe, seed, resident, pools=0, trace=True, w=None, kind=1):
    t = make_trainer(engine, G, "", seed, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False, trace=trace, resident=resident,
                     pools=pools)
    t.set_net(kind, w if w is not None else nets.init_mlp12x100(seed=3, bn_noise=True))
    assert t.run()
    return t


def _digest(t, G):
    gs, ev, pr = H.get_samples(t)
    info = [tuple(sorted(t.game_info(g).items())) for g in range(G)]
    return gs.tobytes(), ev.tobytes(), pr.tobytes(), t.score(), t.avg_mate_length(), t.num_samples(), info


@pytest.mark.parametrize("engine", ENGINES)
def test_recycled_slots_equal_all_games_resident(engine):
    G, S_, spe, seed = 21, 40, 8, 77
    full = _generation(engine, G, S_, spe, seed, resident=-1)
    assert full.stats()["resident_slots"] == G
    want = _digest(full, G)
    traces = [full.trace(g) for g in range(G)]
    for R, pools in ((1, 1), (4, 1), (6, 2), (20, 3)):
        t = _generation(engine, G, S_, spe, seed, resident=R, pools=pools)
        st = t.stats()
        assert st["resident_slots"] == R
        assert _digest(t, G) == want, (R, pools)
        for g in range(G):
            assert np.array_equal(t.trace(g), traces[g]), (R, g)
        assert st["searches"] == full.stats()["searches"] and st["evals"] == full.stats()["evals"]
        assert st["nn_rows"] == full.stats()["nn_rows"]
        # the same pool again with another seed, then the first seed: no state of a generation survives a reset
        t.reset(seed + 1)
        assert t.run()
        assert _digest(t, G) != want
        t.reset(seed)
        assert t.run()
        assert _digest(t, G) == want


@pytest.mark.parametrize("engine", ENGINES)
def test_recycled_pool_matches_oracle_and_the_reference_protocol(engine):
    """host-driven protocol (doIteration / num_requests / writeRequests) on 5 slots for 17 games: every game equals
    the oracle's game of the same index (the request ORDER is by slot, a game's rows are its own)"""
    G, S_, spe, seed = 17, 50, 16, 4242
    t = make_trainer(engine, G, "", seed, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False, trace=True, resident=5)
    H.play_generation(t, 5, spe, H.hash_net)
    o = O.Trainer(G, seed=seed, max_searches=S_, searches_per_eval=spe)
    o.enable_trace()
    o.set_stagger(False)
    H.play_generation(o, G, spe, H.hash_net)
    for g in range(G):
        assert np.array_equal(t.trace(g), o.trace(g)), g
    a, b = H.get_samples(t), H.get_samples(o)
    assert all(x.tobytes() == y.tobytes() for x, y in zip(a, b))
    assert t.score() == o.score() and t.avg_mate_length() == o.avg_mate_length()
    assert t.num_samples() == o.num_samples()


@pytest.mark.parametrize("engine", ENGINES)
def test_sharded_recycled_pool(engine):
    """a shard (game_base / total_games) on recycled slots: seeds and colours stay on the global index"""
    S_, spe, seed = 30, 8, 9
    whole = _generation(engine, 12, S_, spe, seed, resident=-1, trace=False)
    sp_all, oc_all = whole.export_samples()
    counts = [whole.game_info(g)["n_samples"] for g in range(12)]
    t = make_trainer(engine, 7, "", seed, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False, game_base=5, total_games=12, resident=2)
    t.set_net(1, nets.init_mlp12x100(seed=3, bn_noise=True))
    assert t.run()
    sp, oc = t.export_samples()
    lo = sum(counts[:5])
    assert sp.tobytes() == sp_all[lo:].tobytes() and oc.tobytes() == oc_all[lo:].tobytes()


def test_iteration_cap_leaves_unstarted_games_untouched():
    t = make_trainer("emu", 9, "", 5, 30, 8, 1.0, 0.25, 0, 1, False, stagger=False, resident=2)
    t.set_net(1, nets.init_mlp12x100(seed=3, bn_noise=True))
    assert not t.run(max_iterations=6)
    infos = [t.game_info(g) for g in range(9)]
    assert all(i["done"] == 0 for i in infos) and all(i["plies"] == 0 for i in infos[2:])
    assert t.run()
    assert all(t.game_info(g)["done"] == 1 for g in range(9))


@pytest.mark.gpu
def test_16384_games_on_4096_slots_equal_the_unrecycled_run():
    from corintho_ai_amd import NET_MLP12X100_X6

    G, S_, spe, seed = 16384, 400, 16, 12345
    w = nets.init_mlp12x100(0)
    a = _generation("hip", G, S_, spe, seed, resident=-1, trace=False, w=w, kind=NET_MLP12X100_X6)
    b = _generation("hip", G, S_, spe, seed, resident=4096, trace=False, w=w, kind=NET_MLP12X100_X6)
    assert a.stats()["resident_slots"] == G and b.stats()["resident_slots"] == 4096
    ga, gb = H.get_samples(a), H.get_samples(b)
    assert all(x.tobytes() == y.tobytes() for x, y in zip(ga, gb))
    assert a.score() == b.score() and a.num_samples() == b.num_samples()
    assert b.stats()["peak_arena_units"] <= a.stats()["peak_arena_units"] * 1.5


@pytest.mark.gpu
def test_reference_production_setting_25000_games_1600_simulations():
    """corintho_ai/toml/train.toml:2-18: 25 000 games, 1600 simulations per move, 16 per evaluation, c_puct 3.0,
    epsilon 0.25 -- through the reference's constructor arguments alone (resident = automatic).  The reference
    bounds its memory by the staggered start; the engine by the slots that fit in 5/8 of the device memory.  Every
    game finishes, the sample invariants hold, and the first 64 games equal the oracle's bit for bit."""
    import ctypes as C

    from corintho_ai_amd import NET_MLP12X100_X6

    G, S_, spe, seed = 25000, 1600, 16, 2023
    t = make_trainer("hip", G, "", seed, S_, spe, 3.0, 0.25, 0, 1, False)  # stagger as the reference's default
    st0 = t.stats()
    R = st0["resident_slots"]
    assert R < G
    t.set_net(NET_MLP12X100_X6, nets.init_mlp12x100(0))
    assert t.run()
    st = t.stats()
    units = st["peak_arena_units"]
    print("25000 x 1600: %d resident slots, arena %.1f GB, high water %d units per tree" % (R, R * 2 * 899360 * 16 / 1e9, units))
    assert R * 2 * (899200 + 160) * 16 < 200e9
    infos = [t.game_info(g) for g in range(0, G, 37)]
    assert all(i["done"] == 1 and i["error"] == 0 and 0 < i["n_samples"] <= 40 for i in infos)
    n = t.num_samples()
    assert 12 * G < n < 30 * G
    sp, oc = t.export_samples()
    counts = [t.game_info(g)["n_samples"] for g in range(64)]
    m = sum(counts)
    o = O.Trainer(64, seed=seed, max_searches=S_, searches_per_eval=spe, c_puct=3.0, epsilon=0.25, num_threads=16, game_base=0,
                  total_games=G)
    o.set_stagger(False)
    H.play_generation(o, 64, spe, lambda s: t.net_forward(s))
    ogs, oev, opr = H.get_samples(o)
    assert ogs.shape[0] == m * 8
    assert ogs[0::8].tobytes() == sp[:m, :70].tobytes()
    assert opr[0::8].tobytes() == sp[:m, 70:].tobytes()
    assert oev[0::8].tobytes() == oc[:m].tobytes()


# ---------------------------------------------------------------------------------------------------------------
def test_the_pools_share_one_cache_table():
    """round 4: ONE table for all pools -- a position another pool evaluated in an earlier iteration is not evaluated
    again (what both reach in the same iteration still is).  The counts themselves depend on how the waves interleave (two
    that claim for one position in the same instant both evaluate it), on the thread count and on the pools' order within an
    iteration, so only what must hold is asserted: the cache serves rows, more pools lose a little to the
    one iteration of lag, and even three pools on one table beat what a table per pool evaluated (31 443 with two pools)."""
    w = nets.init_mlp12x100(seed=3, bn_noise=True)
    got = {}
    for pools in (1, 2, 3):
        t = make_trainer("emu", 48, "", 5, 60, 8, 1.0, 0.25, 0, 1, False, stagger=False, pools=pools, eval_cache=18)
        t.set_net(1, w)
        assert t.run()
        st = t.stats()
        got[pools] = st["nn_rows_evaluated"]
        assert st["nn_rows"] == 35584
        t.close()
    # what must hold whatever the interleaving (no strict chain between pool counts -- two or three pools
    # differ by a few rows either way depending on the thread count)
    assert all(v < 31443 for v in got.values()), got
    assert got[1] < 0.88 * 35584, got
    assert got[1] <= min(got[2], got[3]) + 64, got  # one pool has no lag to lose rows to


# Evaluation cache (ca_config.eval_cache): a request row whose position was evaluated earlier in the generation gets the
# stored outputs.  A row's outputs are a function of the row, so nothing a game sees changes: every result equals the
# uncached run's bit for bit, with fewer rows through the network kernel.
@pytest.mark.parametrize("engine", ENGINES)
def test_evaluation_cache_changes_nothing_but_the_rows_evaluated(engine):
    G, S_, spe, seed = 24, 60, 8, 31
    runs = {}
    # 18: a table of 2^18 entries per pool (an explicit size switches the cache on for any network; automatic = only for
    # networks whose rows are expensive); 6: 64 entries -- full at once, emptied again and again
    for cache in (False, 18, 6):
        for R, pools in ((-1, 1), (-1, 2), (7, 2), (-1, 3)):
            t = make_trainer(engine, G, "", seed, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False, trace=True, resident=R, pools=pools,
                             eval_cache=cache)
            t.set_net(1, nets.init_mlp12x100(seed=3, bn_noise=True))
            assert t.run()
            st = t.stats()
            runs[(cache, R, pools)] = (_digest(t, G), [t.trace(g).tobytes() for g in range(G)], st)
            if cache == 18:
                assert 0 < st["nn_rows_evaluated"] < st["nn_rows"] == st["evals"]
            elif cache:
                assert 0 < st["nn_rows_evaluated"] <= st["nn_rows"] == st["evals"]
            else:
                assert st["nn_rows_evaluated"] == st["nn_rows"] == st["evals"]
            # a second generation in the same pool starts from an empty table and is a function of its seed only
            t.reset(seed)
            assert t.run()
            assert _digest(t, G) == runs[(cache, R, pools)][0]
            if cache == 18:
                assert t.stats()["nn_rows_evaluated"] < t.stats()["nn_rows"]
    ref = runs[(False, -1, 1)]
    for k, v in runs.items():
        assert v[0] == ref[0] and v[1] == ref[1], k
    saved = 1.0 - runs[(18, -1, 1)][2]["nn_rows_evaluated"] / runs[(18, -1, 1)][2]["nn_rows"]
    print("rows served by the cache: %.1f %%" % (100 * saved))
    assert saved > 0.03


@pytest.mark.parametrize("engine", ENGINES)
def test_evaluation_cache_with_a_tiny_table_and_an_iteration_cap(engine, monkeypatch):
    """continuing a capped run keeps the table; the oracle agrees with the cached engine game for game"""
    G, S_, spe, seed = 10, 40, 8, 5
    t = make_trainer(engine, G, "", seed, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False, trace=True, eval_cache=12)
    t.set_net(1, nets.init_mlp12x100(seed=3, bn_noise=True))
    assert not t.run(max_iterations=9)
    assert t.run()
    o = O.Trainer(G, seed=seed, max_searches=S_, searches_per_eval=spe)
    o.enable_trace()
    o.set_stagger(False)
    H.play_generation(o, G, spe, lambda s: t.net_forward(s))
    for g in range(G):
        assert np.array_equal(t.trace(g), o.trace(g)), g
    assert all(x.tobytes() == y.tobytes() for x, y in zip(H.get_samples(t), H.get_samples(o)))


@pytest.mark.parametrize("engine", ENGINES)
def test_a_new_network_in_mid_generation_is_not_served_by_the_old_ones_entries(engine):
    """set_net between two capped runs: the rows queued from then on are evaluated by the NEW network -- the table's
    entries were filled by the old one and must not serve it.  The uncached engine is the yardstick; an engine whose
    table survived the change returns the old network's outputs for every position met before."""
    G, S_, spe, seed = 12, 40, 8, 77
    w0, w1 = nets.init_mlp12x100(seed=3, bn_noise=True), nets.init_mlp12x100(seed=4, bn_noise=True)
    out = {}
    for cache in (False, 14):
        t = make_trainer(engine, G, "", seed, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False, trace=True, eval_cache=cache)
        t.set_net(1, w0)
        assert not t.run(max_iterations=25)
        t.set_net(1, w1)
        assert t.run()
        out[cache] = (_digest(t, G), [t.trace(g).tobytes() for g in range(G)], t.stats())
    assert out[14][0] == out[False][0] and out[14][1] == out[False][1]
    assert out[14][2]["nn_rows_evaluated"] < out[14][2]["nn_rows"]  # (the cache did serve rows)
    # and the change did matter: the same generation under the first network alone differs
    t = make_trainer(engine, G, "", seed, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False, trace=True, eval_cache=False)
    t.set_net(1, w0)
    assert t.run()
    assert _digest(t, G) != out[False][0]


def test_eval_cache_argument_is_explicit():
    from corintho_ai_amd.trainer import _eval_cache_cfg

    assert _eval_cache_cfg(True) == 0 and _eval_cache_cfg(False) == -1 and _eval_cache_cfg(None) == -1
    assert _eval_cache_cfg(12) == 12
    for bad in (0, 1, 5, 31, "on", 2.5):
        with pytest.raises(ValueError):
            _eval_cache_cfg(bad)


@pytest.mark.gpu
def test_a_network_that_toggles_the_evaluation_cache_is_refused_in_mid_generation():
    """automatic cache: off for the 0.25 MFLOP MLP, on for the residual CNN.  Replacing one by the other between two capped
    runs would free (or create) the tables the pending leaves point into: refused, and the generation goes on as it was"""
    from corintho_ai_amd import NET_MLP12X100_H3, NET_RESCNN4_H3

    G, S_, spe = 64, 40, 8
    t = make_trainer("hip", G, "", 5, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False)
    t.set_net(NET_MLP12X100_H3, nets.init_mlp12x100(0))
    assert not t.run(max_iterations=10)
    with pytest.raises(RuntimeError, match="evaluation cache"):
        t.set_net(NET_RESCNN4_H3, nets.init_rescnn4(0))
    assert t.run()  # ... with the network it had
    ref = make_trainer("hip", G, "", 5, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False)
    ref.set_net(NET_MLP12X100_H3, nets.init_mlp12x100(0))
    assert ref.run()
    assert _digest(t, G) == _digest(ref, G)
    # between generations the change is allowed
    t.reset(6)
    t.set_net(NET_RESCNN4_H3, nets.init_rescnn4(0))
    assert t.run() and 0 < t.stats()["nn_rows_evaluated"] < t.stats()["nn_rows"]


@pytest.mark.parametrize("engine", ENGINES)
def test_more_logged_games_than_resident_slots_is_an_error(engine, tmp_path):
    """a logged game must start in its own slot: asking for more log files than the pool has slots would write fewer
    files than the reference does -- refused instead of clamped"""
    with pytest.raises(RuntimeError, match="resident"):
        make_trainer(engine, 12, str(tmp_path), 3, 30, 8, 1.0, 0.25, 6, 1, False, stagger=False, resident=4)
    t = make_trainer(engine, 12, str(tmp_path), 3, 30, 8, 1.0, 0.25, 4, 1, False, stagger=False, resident=4)  # 4 on 4: fine
    t.close()


# Step budget (ca_config.step_budget, round 6): a game's step stops selecting after so many PUCT scans and goes on in the
# next iteration with its queued leaves held back until the batch is complete.  The game performs the reference's
# sequence of operations spread over more iterations: everything per game equals the unlimited run -- and the oracle.
@pytest.mark.parametrize("engine", ENGINES)
def test_step_budget_changes_nothing_but_the_iterations(engine):
    G, S_, spe, seed = 24, 60, 8, 31
    w = nets.init_mlp12x100(seed=3, bn_noise=True)
    runs = {}
    for budget, R, pools, cache in ((-1, -1, 1, False), (1, -1, 1, False), (3, -1, 2, 18), (7, 7, 2, 6), (12, -1, 3, False), (40, -1, 1, 18),
                                     (0, -1, 3, 18), (-17, -1, 2, False)):  # 0: automatic (the default); -17: automatic, 17/16 of the mean
        t = make_trainer(engine, G, "", seed, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False, trace=True, resident=R, pools=pools,
                         eval_cache=cache, step_budget=budget)
        t.set_net(1, w)
        assert t.run()
        st = t.stats()
        runs[budget] = (_digest(t, G), [t.trace(g).tobytes() for g in range(G)], st)
        # every leaf that was queued was submitted exactly once
        assert st["nn_rows"] == st["evals"]
        # a second generation in the same pool: nothing of a held step survives a reset
        t.reset(seed)
        assert t.run()
        assert _digest(t, G) == runs[budget][0]
        t.close()
    ref = runs[-1]
    for budget, (dig, traces, st) in runs.items():
        assert dig == ref[0], budget
        assert traces == ref[1], budget
        assert st["searches"] == ref[2]["searches"] and st["evals"] == ref[2]["evals"], budget
    # a budget of one scan cuts every step after its first simulation: many more iterations, the same games
    assert runs[1][2]["iterations"] > 1.5 * ref[2]["iterations"]  # (a group of four simulations passes between two looks at the budget)
    assert runs[40][2]["iterations"] >= ref[2]["iterations"]
    # the automatic budget has a floor of 24 scans, which no step of this size reaches; steps of 16 simulations at 160 per
    # move do: a budget just above the pool's mean is met by many of them
    big = {}
    for budget in (-1, -17, 0):
        t = make_trainer(engine, 6, "", seed, 160, 16, 1.0, 0.25, 0, 1, False, stagger=False, trace=True, pools=1, step_budget=budget)
        t.set_net(1, w)
        assert t.run()
        big[budget] = (_digest(t, 6), [t.trace(g).tobytes() for g in range(6)], t.stats()["iterations"])
        t.close()
    assert big[-17][:2] == big[-1][:2] and big[0][:2] == big[-1][:2]
    assert big[-17][2] > big[-1][2] and big[0][2] >= big[-1][2]
    # ... and the oracle plays those games (the network's rows through the engine's own forward)
    t = make_trainer(engine, G, "", seed, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False, trace=True, step_budget=3)
    t.set_net(1, w)
    assert t.run()
    o = O.Trainer(G, seed=seed, max_searches=S_, searches_per_eval=spe)
    o.enable_trace()
    o.set_stagger(False)
    H.play_generation(o, G, spe, lambda s: t.net_forward(s))
    for g in range(G):
        assert np.array_equal(t.trace(g), o.trace(g)), g
    assert all(x.tobytes() == y.tobytes() for x, y in zip(H.get_samples(t), H.get_samples(o)))
