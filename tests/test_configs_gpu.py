"""BASELINE.json configs 2, 3, 4 and 5 at FULL size on one MI355X.

configs[1] -- the configuration `bench.py` times by default: 4096 games, 400 sims/move, the residual CNN at f16x3
(NET_RESCNN4_H3), evaluation cache on, two pools -- is replayed WHOLE on the CPU oracle (all 4096 games, fed by the same
device network: samples, score, mate length bit for bit; the reference's counterpart is tests/cpp/trainer_test.cpp:21-136,
which runs the one code path there is), and so is the bench's `recycled` variant (16 384 games on the 4 096 slots).
tests/test_engine_parity.py::test_full_size_generation_properties adds the size-independent properties, determinism and
sharding for the same configuration.  Configs 3-5 are checked through size-independent properties plus an oracle replay
of a slice (cfg3: 32 games of the shard; cfg4: a 512-game generation whole and 512 games of the full-size one; cfg5:
all 1024 games), bit for bit -- at the bench's arithmetic (f16x3) and at bf16x6.

  cfg3  32 768 games sharded over 8 GPUs: ONE shard of it on this GPU -- rank 7 of 8, games
        [28 672, 32 768), seeds / parity / colours on the global index (trainer.cpp:243-255)
  cfg4  4096 games, 1600 sims/move, Dirichlet noise 0.25 (toml/train.toml:2-18: the reference's
        production setting): deep trees, the search-tree arena's high-water mark
  cfg5  arena: 1024 two-model games, testing = true, 400 sims/move, fused on the device
"""
import numpy as np
import pytest

from corintho_ai_amd import NET_RESCNN4_H3, NET_RESCNN4_X6, nets
from oracle import oracle as O
from tests import harness as H
from tests.engines import make_trainer

pytestmark = pytest.mark.gpu

NREPLAY = 32


def _replay_first_games(t, G_total, base, sims, spe, seed, n=NREPLAY, eps=0.25):
    """games [base, base + n) on the CPU oracle (a slice of a Trainer of G_total games), evaluated by
    the device network of trainer `t`; -> the oracle trainer"""
    o = O.Trainer(n, seed=seed, max_searches=sims, searches_per_eval=spe, epsilon=eps, num_threads=16, game_base=base,
                  total_games=G_total)
    o.set_stagger(False)
    H.play_generation(o, n, spe, lambda st: t.net_forward(st))
    return o


def _replay_whole_generation(t, G, sims, spe, seed, c_puct=1.0, eps=0.25):
    """EVERY game of trainer t's generation on the CPU oracle, evaluated by t's device network (in pieces of the rows one
    device evaluation takes): sample tensors, score and mate length must be the same bytes (tools/big_parity.py as a test)"""
    o = O.Trainer(G, seed=seed, max_searches=sims, searches_per_eval=spe, c_puct=c_puct, epsilon=eps, num_threads=16)
    o.set_stagger(False)
    cap = t.stats()["resident_slots"] * spe

    def fw(states):
        parts = [t.net_forward(states[i:i + cap]) for i in range(0, states.shape[0], cap)]
        return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])

    H.play_generation(o, G, spe, fw)
    sp, oc = t.export_samples()
    ogs, oev, opr = H.get_samples(o)
    n = t.num_samples()
    assert ogs.shape[0] == n * 8 == sp.shape[0] * 8
    assert ogs[0::8].tobytes() == sp[:, :70].tobytes(), "states differ from the oracle"
    assert opr[0::8].tobytes() == sp[:, 70:].tobytes(), "policies differ from the oracle"
    assert oev[0::8].tobytes() == oc.tobytes(), "outcomes differ from the oracle"
    assert o.score() == t.score() and o.avg_mate_length() == t.avg_mate_length()
    return o


@pytest.mark.parametrize("games,slots,pools", [(4096, -1, 0), (4096, -1, 2), (16384, 4096, 0)], ids=["default", "two_pools", "recycled"])
def test_cfg2_the_bench_default_whole_on_the_oracle(games, slots, pools):
    """what `python bench.py` times (its `two_pools` record of rounds 1-4, its `recycled` variant): NET_RESCNN4_H3,
    evaluation cache on, the engine's default of three pools, 400 sims/move, seed 12345 -- every game replayed on the oracle"""
    S_, spe, seed = 400, 16, 12345
    t = make_trainer("hip", games, "", seed, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False, resident=slots, pools=pools)
    t.set_net(NET_RESCNN4_H3, nets.init_rescnn4(0))
    assert t.run()
    st = t.stats()
    assert st["resident_slots"] == 4096 and st["pools"] == (pools or 3)
    assert st["nn_rows"] == st["evals"] > 0
    assert 0 < st["nn_rows_evaluated"] < 0.9 * st["nn_rows"]  # the evaluation cache is on and serves rows
    _replay_whole_generation(t, games, S_, spe, seed)
    print("cfg2 (%d games on %d slots): %d plies, %d simulations, %d of %d request rows evaluated, score %.6f"
          % (games, st["resident_slots"], st["plies"], st["searches"], st["nn_rows_evaluated"], st["nn_rows"], t.score()))


def test_trained_checkpoint_generation_whole_on_the_oracle():
    """the regime the step budget was made for (round 6): the reference's last checkpoint guiding 4096 games -- narrow deep
    trees, in late plies every second simulation terminal, ~100 000 steps of the generation stopped at their budget and
    resumed -- with every game replayed on the oracle (`detail.trained_checkpoint` of the bench line times this workload)"""
    import os

    import numpy as np

    from corintho_ai_amd import NET_MLP12X100_H3

    games, S_, spe, seed = 4096, 400, 16, 12345
    w = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trained_last.npz"))["weights"]
    t = make_trainer("hip", games, "", seed, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False)
    t.set_net(NET_MLP12X100_H3, w)
    assert t.run()
    st = t.stats()
    assert st["pools"] == 3 and st["nn_rows"] == st["evals"] > 0
    assert st["steps_cut"] > 10000 and st["step_budget_last"] > 0  # the budget is on, and it bites
    _replay_whole_generation(t, games, S_, spe, seed)
    print("trained checkpoint, %d games: %d plies, %d simulations, %d iterations, %d steps stopped at their budget (last budget %d us)"
          % (games, st["plies"], st["searches"], st["iterations"], st["steps_cut"], st["step_budget_last"]))


def test_cfg4_a_512_game_generation_whole_on_the_oracle():
    """cfg4's setting (1600 sims/move, Dirichlet noise) at the bench's arithmetic, 512 games, every game on the oracle"""
    G, S_, spe, seed = 512, 1600, 16, 4321
    t = make_trainer("hip", G, "", seed, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False)
    t.set_net(NET_RESCNN4_H3, nets.init_rescnn4(0))
    assert t.run()
    _replay_whole_generation(t, G, S_, spe, seed)


def _check_training_generation(t, G):
    infos = [t.game_info(g) for g in range(0, G, 61)]
    assert all(i["done"] == 1 and i["error"] == 0 and 0 < i["n_samples"] <= 40 for i in infos)
    gs, ev, pr = H.get_samples(t)
    n = t.num_samples()
    assert gs.shape == (n * 8, 70) and 12 * G < n < 24 * G
    H.check_sample_properties(gs, ev, pr)
    assert 0.0 <= t.score() <= 1.0
    return gs, ev, pr


KINDS = [pytest.param(NET_RESCNN4_H3, id="f16x3"), pytest.param(NET_RESCNN4_X6, id="bf16x6")]  # the bench's arithmetic first


@pytest.mark.parametrize("kind", KINDS)
def test_cfg3_one_shard_of_the_32768_game_generation(kind):
    G, TOTAL, BASE, S_, spe, seed = 4096, 32768, 28672, 400, 16, 12345
    w = nets.init_rescnn4(0)
    t = make_trainer("hip", G, "", seed, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False, game_base=BASE, total_games=TOTAL)
    t.set_net(kind, w)
    assert t.run()
    _check_training_generation(t, G)
    sp_all, oc_all = t.export_samples()
    counts = [t.game_info(g)["n_samples"] for g in range(NREPLAY)]
    m = sum(counts)
    o = _replay_first_games(t, TOTAL, BASE, S_, spe, seed)
    ogs, oev, opr = H.get_samples(o)
    assert ogs.shape[0] == m * 8
    assert ogs[0::8].tobytes() == sp_all[:m, :70].tobytes()
    assert opr[0::8].tobytes() == sp_all[:m, 70:].tobytes()
    assert oev[0::8].tobytes() == oc_all[:m].tobytes()
    for g in range(NREPLAY):
        assert t.game_info(g)["result"] == o.game_result(g)
    # the shard's seeds are those of the global indices: the same 32 local games of shard 0 differ
    t0 = make_trainer("hip", NREPLAY, "", seed, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False, game_base=0, total_games=TOTAL)
    t0.set_net(kind, w)
    assert t0.run()
    assert t0.export_samples()[0].tobytes() != sp_all[:m].tobytes()


def test_cfg4_1600_simulations_with_dirichlet_noise():
    """full size at the bench's arithmetic; the first 512 games replayed on the oracle (round 4: 32 at bf16x6)"""
    G, S_, spe, seed = 4096, 1600, 16, 12345
    N4 = 512
    w = nets.init_rescnn4(0)
    t = make_trainer("hip", G, "", seed, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False)
    t.set_net(NET_RESCNN4_H3, w)
    assert t.run()
    _check_training_generation(t, G)
    st = t.stats()
    # deep-tree stress: the arena's high-water mark stays below the default capacity
    # ((1600 * 14 + 64) * 40 units per tree, engine.hip init) -- an overflow would have raised
    cap = (S_ * 14 + 64) * 40
    assert 0 < st["peak_arena_units"] < cap, st["peak_arena_units"]
    assert st["searches"] > G * S_ * 8  # ~18 plies per game; tree reuse and solved roots make a ply cheaper than 1600 new simulations
    sp_all, oc_all = t.export_samples()
    m = sum(t.game_info(g)["n_samples"] for g in range(N4))
    o = _replay_first_games(t, G, 0, S_, spe, seed, n=N4)
    ogs, oev, opr = H.get_samples(o)
    assert ogs.shape[0] == m * 8
    assert ogs[0::8].tobytes() == sp_all[:m, :70].tobytes()
    assert opr[0::8].tobytes() == sp_all[:m, 70:].tobytes()
    assert oev[0::8].tobytes() == oc_all[:m].tobytes()
    print("cfg4: peak arena units per tree %d of %d; %.1f plies, %.0f evaluations per game"
          % (st["peak_arena_units"], cap, st["plies"] / G, st["evals"] / G))


@pytest.mark.parametrize("kind", KINDS)
def test_cfg5_arena_1024_two_model_games(kind):
    G, S_, spe, seed = 1024, 400, 16, 77
    wa, wb = nets.init_rescnn4(0), nets.init_rescnn4(1)
    t = make_trainer("hip", G, "", seed, S_, spe, 1.0, 0.25, 0, 1, True, trace=True)
    t.set_net(kind, wa, slot=0)  # best model
    t.set_net(kind, wb, slot=1)  # new model
    assert t.run()
    infos = [t.game_info(g) for g in range(G)]
    assert all(i["done"] == 1 and i["error"] == 0 for i in infos)
    assert all(i["result"] in (1, 2, 3) for i in infos)  # loss / draw / win of the first player (util.h:57-64)
    assert t.num_samples() == 0  # testing_ disables samples (trainmc.cpp:114)
    s = t.score()
    assert 0.0 <= s <= 1.0
    # score = mean over games of the new model's result, colours alternating with the game index
    # (trainer.cpp:59-68): recompute it from the per-game results
    pts = {1: 0.0, 2: 0.5, 3: 1.0}
    mine = sum(pts[i["result"]] if g % 2 == 0 else 1.0 - pts[i["result"]] for g, i in enumerate(infos)) / G
    assert abs(mine - s) < 1e-5
    # ALL 1024 games on the oracle, driven by the reference loop (main.pyx:142-168) with the two device networks.
    # (A subset cannot be replayed in arena mode: when a model's batch comes back empty the loop flips to_play and
    # calls doIteration with the arrays of the LAST prediction (main.pyx:150-154), so the first evaluations a side
    # consumes after a hand-over are whatever lies at its offset -- rows of other games.  The engine reproduces
    # that; a game's trajectory therefore depends on the whole batch.  DESIGN.md, quirk 12.)
    o = O.Trainer(G, seed=seed, max_searches=S_, searches_per_eval=spe, testing=True, num_threads=16)
    o.enable_trace()
    nets2 = (lambda st: t.net_forward(st, slot=1), lambda st: t.net_forward(st, slot=0))  # to_play 0 -> new model
    H.play_generation(o, G, spe, None, nets_by_player=nets2)
    for g in range(G):
        assert np.array_equal(t.trace(g), o.trace(g)), "game %d" % g
        assert infos[g]["result"] == o.game_result(g)
    assert t.score() == o.score()


@pytest.mark.parametrize("pools,bits,seed", [(2, 14, 1), (3, 14, 2), (3, 17, 3)], ids=["2_pools_2^14", "3_pools_2^14", "3_pools_2^17"])
def test_shared_cache_table_under_real_concurrency(pools, bits, seed):
    """The pools share one evaluation-cache table through relaxed atomics, per-pool progress words and an emptying protocol
    (mcts.h co_cache_resolve, EvalCache::guard_from).  The emulation build runs the pools of an iteration
    one after the other and cannot see a race; here the streams really run side by side, with tables so small that they
    are emptied hundreds of times per generation -- and every game must still replay on the oracle bit for bit.
    (tools/exp/cache_stress.py as a test)"""
    G = 1024
    t = make_trainer("hip", G, "", seed, 400, 16, 1.0, 0.25, 0, 1, False, stagger=False, pools=pools, eval_cache=bits)
    t.set_net(NET_RESCNN4_H3, nets.init_rescnn4(0))
    assert t.run()
    st = t.stats()
    assert st["pools"] == pools and 0 < st["nn_rows_evaluated"] < st["nn_rows"]
    _replay_whole_generation(t, G, 400, 16, seed)
