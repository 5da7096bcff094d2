"""Parity of the engine (kernel source: corintho_ai_amd/csrc) with the CPU oracle,
through the C ABI, on the same seeded inputs.  Bit-exact: request rows of every
iteration, per-ply traces (root children: move, visits, evaluation bits, result),
chosen moves, the three sample tensors, score and mate length.
Runs on the emulation build here and on the MI355X with -m gpu."""
import ctypes as C

import numpy as np
import pytest

from corintho_ai_amd import _lib, nets

from tests import ref_nets
from oracle import oracle as O
from tests import harness as H
from tests import ref_scenarios as S
from tests.engines import ENGINES, cdll, make_trainer


def _meta(pieces, to_play):
    m = 0
    for i, p in enumerate(pieces):
        m |= int(p) << (3 * i)
    return m | (int(to_play) << 18)


class EngineGame:
    """Game value type whose rule calls go through ca_rules_* (one wavefront each)."""

    def __init__(self, L, board=0, pieces=(4, 4, 4, 4, 4, 4), to_play=0):
        self.L, self.board, self.pieces, self.to_play = L, int(board), list(pieces), int(to_play)

    def copy(self):
        return EngineGame(self.L, self.board, self.pieces, self.to_play)

    def legal_moves(self):
        b = (C.c_uint64 * 1)(self.board)
        m = (C.c_uint32 * 1)(_meta(self.pieces, self.to_play))
        out = (C.c_uint32 * 3)()
        ln = (C.c_int32 * 1)()
        _lib.check(self.L, self.L.ca_rules_legal_moves(0, b, m, 1, out, ln))
        mask = int(out[0]) | (int(out[1]) << 32) | (int(out[2]) << 64)
        return [bool(mask >> i & 1) for i in range(96)], bool(ln[0])

    def _apply(self, move):
        b = (C.c_uint64 * 1)(self.board)
        m = (C.c_uint32 * 1)(_meta(self.pieces, self.to_play))
        mv = (C.c_int32 * 1)(move)
        st = np.zeros(70, np.float32)
        _lib.check(self.L, self.L.ca_rules_do_move(0, b, m, mv, 1, st.ctypes.data_as(_lib.f32p)))
        return int(b[0]), int(m[0]), st

    def do_move(self, move):
        b, m, _ = self._apply(move)
        self.board = b
        self.pieces = [(m >> (3 * i)) & 7 for i in range(6)]
        self.to_play = (m >> 18) & 1

    def state(self):
        return self._apply(-1)[2]

    def terminal_result(self):
        lm, lines = self.legal_moves()
        if any(lm):
            return 0
        return 1 if lines else 2


def backend(L):
    class B:
        new_game = staticmethod(lambda: EngineGame(L))

        @staticmethod
        def game_from_arrays(board64, to_play, pieces):
            b = 0
            for i, v in enumerate(board64):
                if v:
                    b |= 1 << i
            return EngineGame(L, b, pieces, to_play)

        encode_place = staticmethod(O.encode_place)
        encode_move = staticmethod(O.encode_move)
        decode_move = staticmethod(O.decode_move)

    return B


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("scenario", [s for s in S.RULE_SCENARIOS if s.__name__ != "move_codec"],
                         ids=lambda f: f.__name__)
def test_reference_rule_scenarios(engine, scenario):
    scenario(backend(cdll(engine)))


@pytest.mark.parametrize("engine", ENGINES)
def test_legal_masks_random_corpus(engine):
    """bit-exact masks / is_lines / states on positions from random playouts"""
    L = cdll(engine)
    rng = np.random.default_rng(7)
    boards, metas, want_mask, want_lines, want_state = [], [], [], [], []
    n_games = 400 if engine == "hip" else 120
    for _ in range(n_games):
        g = O.Game()
        while True:
            mask, lines = g.legal_mask()
            boards.append(g.board)
            metas.append(_meta(g.pieces, g.to_play))
            want_mask.append(mask)
            want_lines.append(lines)
            want_state.append(g.state())
            moves = [i for i in range(96) if mask >> i & 1]
            if not moves:
                break
            g.do_move(int(rng.choice(moves)))
    n = len(boards)
    b = np.array(boards, np.uint64)
    m = np.array(metas, np.uint32)
    out = np.zeros((n, 3), np.uint32)
    ln = np.zeros(n, np.int32)
    _lib.check(L, L.ca_rules_legal_moves(0, b.ctypes.data_as(_lib.u64p), m.ctypes.data_as(_lib.u32p), n,
                                         out.ctypes.data_as(_lib.u32p), ln.ctypes.data_as(_lib.i32p)))
    got = [int(out[i, 0]) | (int(out[i, 1]) << 32) | (int(out[i, 2]) << 64) for i in range(n)]
    assert got == want_mask
    assert [bool(x) for x in ln] == want_lines
    st = np.zeros((n, 70), np.float32)
    mv = np.full(n, -1, np.int32)
    b2, m2 = b.copy(), m.copy()
    _lib.check(L, L.ca_rules_do_move(0, b2.ctypes.data_as(_lib.u64p), m2.ctypes.data_as(_lib.u32p),
                                     mv.ctypes.data_as(_lib.i32p), n, st.ctypes.data_as(_lib.f32p)))
    assert np.array_equal(st, np.array(want_state))
    # the four-positions-per-wavefront rule layer of the search (rules.h co_legal_moves_rows), ragged tail included
    for cut in (n, n - 1, n - 2, n - 3):
        b3, m3 = b[:cut].copy(), m[:cut].copy()
        out3 = np.zeros((cut, 3), np.uint32)
        _lib.check(L, L.ca_rules_rows(0, b3.ctypes.data_as(_lib.u64p), m3.ctypes.data_as(_lib.u32p),
                                      mv[:cut].copy().ctypes.data_as(_lib.i32p), cut, out3.ctypes.data_as(_lib.u32p)))
        assert np.array_equal(out3, out[:cut]) and np.array_equal(b3, b[:cut]) and np.array_equal(m3, m[:cut])


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("chunk", [1, 17, 48, 64])
def test_mt19937_stream(engine, chunk):
    L = cdll(engine)
    n = 2000
    out = np.zeros(n, np.uint32)
    for seed in (5489, 12345, 0xFFFFFFFF):
        _lib.check(L, L.ca_rng_draw(0, seed, n, chunk, out.ctypes.data_as(_lib.u32p)))
        g = O.MT19937(seed)
        assert [int(x) for x in out] == [g() for _ in range(n)]


def fp_expected(x):
    """the reference's float/double expressions, in numpy (IEEE, no contraction)"""
    f32, f64 = np.float32, np.float64
    c_puct, visits, ev, p9, denom, cv, s, eps = [x[:, i].astype(f32) for i in range(8)]
    v_sqrt = (c_puct.astype(f64) * np.sqrt(visits.astype(f64))).astype(f32)
    prob = (p9 * denom).astype(f32)
    pv = (prob * v_sqrt).astype(f32)
    a = (-1.0 * ev.astype(f64)) / cv.astype(f64)
    b = pv.astype(f64) / (cv.astype(f64) + 1.0)
    one_minus = (f32(1) - eps).astype(f32)
    out = np.zeros_like(x)
    out[:, 0] = v_sqrt
    out[:, 1] = (a + b).astype(f32)
    out[:, 2] = pv
    out[:, 3] = (1.0 / s.astype(f64) * one_minus.astype(f64)).astype(f32)
    out[:, 4] = (1.0 / s.astype(f64) * eps.astype(f64)).astype(f32)
    out[:, 5] = (f32(511.0) / s).astype(f32)
    out[:, 6] = (1.0 / np.trunc(cv).astype(f32).astype(f64)).astype(f32)
    xq = (p9 * out[:, 5]).astype(f32)
    out[:, 7] = np.floor(xq.astype(f64) + 0.5).astype(f32)  # lround, x >= 0
    return out


@pytest.mark.parametrize("engine", ENGINES)
def test_floating_point_contract(engine):
    """sqrt/div/mixed-precision expressions of the PUCT and prior code must round
    exactly as on the reference's x86-64 build (no FMA contraction, IEEE divide)"""
    L = cdll(engine)
    rng = np.random.default_rng(3)
    n = 200000 if engine == "hip" else 20000
    x = np.zeros((n, 8), np.float32)
    x[:, 0] = rng.choice([1.0, 3.0, 0.7, 2.5], n)
    x[:, 1] = rng.integers(1, 32767, n)
    x[:, 2] = rng.normal(0, 30, n)
    x[:, 3] = rng.integers(1, 512, n)
    x[:, 4] = (1.0 / rng.integers(1, 20000, n)).astype(np.float32)
    x[:, 5] = rng.integers(1, 3000, n)
    x[:, 6] = rng.uniform(1e-3, 60, n)
    x[:, 7] = rng.choice([0.25, 0.0, 0.1, 1.0], n)
    out = np.zeros((n, 8), np.float32)
    _lib.check(L, L.ca_fp_probe(0, x.ctypes.data_as(_lib.f32p), n, out.ctypes.data_as(_lib.f32p)))
    want = fp_expected(x)
    for col in range(8):
        assert np.array_equal(out[:, col].view(np.uint32), want[:, col].view(np.uint32)), "column %d" % col


def run_pair(engine, G, S_, spe, eps=0.25, c_puct=1.0, seed=12345, net=H.hash_net, stagger=True):
    t = make_trainer(engine, G, "", seed, S_, spe, c_puct, eps, 0, 1, False, trace=True, stagger=stagger)
    o = O.Trainer(G, seed=seed, max_searches=S_, searches_per_eval=spe, c_puct=c_puct, epsilon=eps, num_threads=4)
    o.enable_trace()
    o.set_stagger(stagger)
    ra = H.play_generation(t, G, spe, net, record=True)
    rb = H.play_generation(o, G, spe, net, record=True)
    assert ra["iterations"] == rb["iterations"]
    for i, (a, b) in enumerate(zip(ra["log"], rb["log"])):
        assert a[1].shape == b[1].shape, "request count differs at iteration %d" % i
        assert a[1].tobytes() == b[1].tobytes(), "request rows differ at iteration %d" % i
    for g in range(G):
        assert np.array_equal(t.trace(g), o.trace(g)), "per-ply trace of game %d" % g
        assert t.game_info(g)["result"] == o.game_result(g)
    sa, sb = H.get_samples(t), H.get_samples(o)
    for x, y in zip(sa, sb):
        assert x.shape == y.shape and x.tobytes() == y.tobytes()
    assert t.num_samples() == o.num_samples()
    assert t.score() == o.score()
    assert t.avg_mate_length() == o.avg_mate_length()
    c = o.counters()
    st = t.stats()
    assert (st["searches"], st["evals"], st["nodes"], st["plies"]) == (c["searches"], c["leaf_evals"],
                                                                       c["nodes_created"], c["plies"])
    H.check_sample_properties(*sa)
    return t, o


CASES = [
    # G, sims, spe, eps
    (64, 50, 16, 0.25),   # BASELINE config 1 shape
    (16, 50, 1, 0.25),    # one request per iteration
    (8, 200, 16, 0.0),    # deterministic selection
    (6, 400, 16, 0.25),   # BASELINE sims/move
    (40, 8, 4, 0.25),     # staggered start, tiny trees
    (5, 1, 1, 0.25), (5, 2, 1, 0.25), (5, 3, 2, 0.25),  # selfplayer_test.cpp FewSearches corner
    (6, 120, 40, 0.25),   # more leaves pending than one round of the step's staged records (16) and two backup batches hold
]


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("G,S_,spe,eps", CASES)
def test_selfplay_matches_oracle(engine, G, S_, spe, eps):
    run_pair(engine, G, S_, spe, eps)


@pytest.mark.parametrize("engine", ENGINES)
def test_selfplay_c_puct_3_no_stagger(engine):
    run_pair(engine, 12, 64, 16, eps=0.25, c_puct=3.0, seed=99, stagger=False)  # train.toml c_puct


@pytest.mark.parametrize("engine", ENGINES)
def test_arena_mode_matches_oracle(engine):
    """testing=True, two models, per-game parity (trainer.cpp:205-235, main.pyx:329-349)"""
    G, S_, spe = 10, 40, 8
    t = make_trainer(engine, G, "", 5, S_, spe, 1.0, 0.25, 0, 1, True, trace=True)
    o = O.Trainer(G, seed=5, max_searches=S_, searches_per_eval=spe, testing=True)
    o.enable_trace()
    nets2 = (lambda s: H.hash_net(s, 1), lambda s: H.hash_net(s, 2))
    ra = H.play_generation(t, G, spe, None, nets_by_player=nets2, record=True)
    rb = H.play_generation(o, G, spe, None, nets_by_player=nets2, record=True)
    assert ra["iterations"] == rb["iterations"]
    assert [(a[0], a[1].tobytes()) for a in ra["log"]] == [(b[0], b[1].tobytes()) for b in rb["log"]]
    for g in range(G):
        assert np.array_equal(t.trace(g), o.trace(g))
    assert t.score() == o.score()
    assert t.num_samples() == 0


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("testing", [False, True], ids=["train", "arena"])
def test_write_scores_file_matches_oracle(engine, testing, tmp_path):
    """Trainer::writeScores (trainer.cpp:115-162): the six-line win/draw/loss summary, byte for byte"""
    G, S_, spe = 14, 24, 8
    t = make_trainer(engine, G, "", 21, S_, spe, 1.0, 0.25, 0, 1, testing)
    o = O.Trainer(G, seed=21, max_searches=S_, searches_per_eval=spe, testing=testing)
    for tr in (t, o):
        if testing:
            H.play_generation(tr, G, spe, None, nets_by_player=(lambda s: H.hash_net(s, 1), lambda s: H.hash_net(s, 2)))
        else:
            H.play_generation(tr, G, spe, H.hash_net)
    fa, fb = str(tmp_path / "engine.txt"), str(tmp_path / "oracle.txt")
    t.writeScores(fa)
    o.writeScores(fb)
    a, b = open(fa, "rb").read(), open(fb, "rb").read()
    assert a == b
    assert a.count(b"\n") == 6 and a.startswith(b"First player wins: ") and b"Second player losses: " in a
    assert t.score() == o.score()


def test_oracle_slice_equals_the_full_trainer():
    """test infrastructure: games [5, 12) of an oracle Trainer(12) as a slice play the same games
    (the GPU shard tests replay a shard of a 32768-game generation this way)"""
    G, S_, spe = 12, 24, 8
    for stagger in (False,):  # (a staggered slice has no request until its first game is released: the
        # reference's play loop, main.pyx:161-163, would raise)
        whole = O.Trainer(G, seed=77, max_searches=S_, searches_per_eval=spe)
        whole.enable_trace()
        whole.set_stagger(stagger)
        H.play_generation(whole, G, spe, H.hash_net)
        part = O.Trainer(7, seed=77, max_searches=S_, searches_per_eval=spe, game_base=5, total_games=G)
        part.enable_trace()
        part.set_stagger(stagger)
        H.play_generation(part, 7, spe, H.hash_net)
        for g in range(7):
            assert np.array_equal(part.trace(g), whole.trace(5 + g))
        n5 = sum(whole.game_num_samples(g) for g in range(5))
        for x, y in zip(H.get_samples(part), H.get_samples(whole)):
            assert x.tobytes() == y[n5 * 8:].tobytes()


@pytest.mark.parametrize("engine", ENGINES)
def test_sharded_trainers_equal_one_trainer(engine):
    """games [base, base+n) of a sharded generation replay the same games (seeds and
    colours follow the GLOBAL index, trainer.cpp:243-255)"""
    G, S_, spe = 12, 24, 8
    whole = make_trainer(engine, G, "", 77, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False)
    H.play_generation(whole, G, spe, H.hash_net)
    sp_all, oc_all = whole.export_samples()
    parts = []
    for base, n in ((0, 5), (5, 7)):
        t = make_trainer(engine, n, "", 77, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False, game_base=base,
                         total_games=G)
        H.play_generation(t, n, spe, H.hash_net)
        parts.append(t.export_samples())
    sp = np.concatenate([p[0] for p in parts])
    oc = np.concatenate([p[1] for p in parts])
    assert sp.tobytes() == sp_all.tobytes() and oc.tobytes() == oc_all.tobytes()
    from corintho_ai_amd import expand_samples

    a = expand_samples(sp, oc, _cdll=cdll(engine))
    b = H.get_samples(whole)
    for x, y in zip(a, b):
        assert x.tobytes() == y.tobytes()


@pytest.mark.parametrize("engine", ENGINES)
def test_arena_overflow_is_reported(engine):
    t = make_trainer(engine, 2, "", 1, 64, 8, 1.0, 0.25, 0, 1, False, arena_units=300)
    with pytest.raises(_lib.EngineError, match="arena"):
        H.play_generation(t, 2, 8, H.hash_net)


@pytest.mark.parametrize("engine", ENGINES)
def test_fused_mode_equals_compat_mode_with_same_network(engine):
    """the device-resident loop (ca_trainer_run) must play exactly the games the
    reference protocol plays when the caller evaluates the same network"""
    G, S_, spe = 16, 48, 8
    w = nets.init_mlp12x100(seed=0, bn_noise=True)
    fused = make_trainer(engine, G, "", 2024, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False)
    fused.set_net(1, w)
    assert fused.run()
    compat = make_trainer(engine, G, "", 2024, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False)
    compat.set_net(1, w)
    H.play_generation(compat, G, spe, lambda s: compat.net_forward(s))
    for x, y in zip(H.get_samples(fused), H.get_samples(compat)):
        assert x.tobytes() == y.tobytes()
    assert fused.score() == compat.score()
    # and the oracle, fed the same network outputs, agrees with both
    o = O.Trainer(G, seed=2024, max_searches=S_, searches_per_eval=spe)
    o.set_stagger(False)
    H.play_generation(o, G, spe, lambda s: compat.net_forward(s))
    for x, y in zip(H.get_samples(fused), H.get_samples(o)):
        assert x.tobytes() == y.tobytes()


@pytest.mark.parametrize("engine", ENGINES)
def test_fused_pools_do_not_change_any_game(engine):
    """fused training split into independent pools on separate streams (ca_config.pools):
    samples, score and per-game evaluation counts are those of the single-pool run, for
    even and ragged splits, staggered starts and a run resumed after an iteration cap"""
    G, S_, spe = 13, 40, 8
    w = nets.init_mlp12x100(seed=3, bn_noise=True)
    runs = []
    for pools in (1, 2, 3, 4):
        t = make_trainer(engine, G, "", 77, S_, spe, 1.0, 0.25, 0, 1, False, pools=pools)
        t.set_net(1, w)
        if pools == 3:
            assert not t.run(max_iterations=11)
            assert not t.run(max_iterations=5)
        assert t.run()
        runs.append((H.get_samples(t), t.score(), t.stats()["nn_rows"]))
    for r in runs[1:]:
        for x, y in zip(r[0], runs[0][0]):
            assert x.tobytes() == y.tobytes()
        assert r[1] == runs[0][1]
        assert r[2] == runs[0][2]


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("pools,base,total", [(2, 0, 300), (1, 150, 300), (2, 150, 450)])
def test_fused_staggered_start_with_late_pools(engine, pools, base, total):
    """The staggered start (trainer.cpp:184-186) releases game i at iteration i / max(G / S, 1): a
    pool (or a shard with game_base > 0) whose first game starts after more than 17 host polls
    (136 iterations) has running games and no request rows until then.  That is not the
    "No requests during training" condition of main.pyx:161-163 (which looks at ALL games)."""
    G = total - base if pools == 1 else 300
    S_, spe = total, 16  # stagger_div = max(total / S, 1) = 1
    w = nets.init_mlp12x100(seed=0)
    t = make_trainer(engine, G, "", 3, S_, spe, 1.0, 0.25, 0, 1, False, pools=pools, game_base=base,
                     total_games=base + G)
    t.set_net(1, w)
    first_late = base + (G // 2 if pools == 2 else 0)  # release iteration of the last pool's first game
    assert first_late > 136
    assert not t.run(max_iterations=first_late + S_ // spe + 8)  # the old idle count raised after ~136 iterations
    info = t.game_info(G // 2 if pools == 2 else 0)
    assert info["error"] == 0 and info["plies"] >= 1  # that game was released and has moved


@pytest.mark.parametrize("engine", ENGINES)
def test_mlp_matches_float32_restatement(engine):
    """policy/value within 1e-4 (fp32) of the numpy restatement of wrapper.py:256-271"""
    G, spe = 64, 16
    t = make_trainer(engine, G, "", 1, 50, spe, 1.0, 0.25, 0, 1, False)
    rng = np.random.default_rng(5)
    n = 777
    states = np.zeros((n, 70), np.float32)
    states[:, :64] = rng.integers(0, 2, (n, 64))
    states[:, 64:] = rng.integers(0, 5, (n, 6)) * 0.25
    for seed, noise in ((0, False), (1, True)):
        w = nets.init_mlp12x100(seed=seed, bn_noise=noise)
        t.set_net(1, w)
        ev, pr = t.net_forward(states)
        ev0, pr0 = ref_nets.mlp12x100_forward_np(w, states)
        assert np.max(np.abs(ev - ev0)) < 1e-4
        assert np.max(np.abs(pr - pr0)) < 1e-4
        assert np.all(np.abs(pr.sum(axis=1) - 1) < 1e-5)
        # a row's result must not depend on its batch (SURVEY 8e invariant)
        ev1, pr1 = t.net_forward(states[5:6])
        assert ev1[0] == ev[5] and np.array_equal(pr1[0], pr[5])


@pytest.mark.gpu
def test_mlp_bf16x3_within_tolerance_of_float32():
    """the reference network with its dense layers at bf16x3 split precision (and BatchNorm folded
    into the next layer): within the 1e-4 contract of the numpy restatement, batch independent,
    and a fused generation replays on the oracle"""
    from corintho_ai_amd import NET_MLP12X100_X3

    t = make_trainer("hip", 64, "", 1, 50, 16, 1.0, 0.25, 0, 1, False)
    rng = np.random.default_rng(21)
    n = 1000
    states = np.zeros((n, 70), np.float32)
    states[:, :64] = rng.integers(0, 2, (n, 64))
    states[:, 64:] = rng.integers(0, 5, (n, 6)) * 0.25
    for seed, noise in ((0, False), (1, True), (7, True)):
        w = nets.init_mlp12x100(seed=seed, bn_noise=noise)
        t.set_net(NET_MLP12X100_X3, w)
        ev, pr = t.net_forward(states)
        ev0, pr0 = ref_nets.mlp12x100_forward_np(w, states)
        assert np.max(np.abs(ev - ev0)) < 1e-4, np.max(np.abs(ev - ev0))
        assert np.max(np.abs(pr - pr0)) < 1e-4, np.max(np.abs(pr - pr0))
        assert np.all(np.abs(pr.sum(axis=1) - 1) < 1e-5)
        ev1, pr1 = t.net_forward(states[130:131])
        assert ev1[0] == ev[130] and np.array_equal(pr1[0], pr[130])
    G, S_, spe = 24, 40, 8
    f = make_trainer("hip", G, "", 77, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False)
    f.set_net(NET_MLP12X100_X3, nets.init_mlp12x100(seed=2, bn_noise=True))
    assert f.run()
    o = O.Trainer(G, seed=77, max_searches=S_, searches_per_eval=spe)
    o.set_stagger(False)
    H.play_generation(o, G, spe, lambda s: f.net_forward(s))
    for x, y in zip(H.get_samples(f), H.get_samples(o)):
        assert x.tobytes() == y.tobytes()


@pytest.mark.gpu
def test_rescnn4_matches_float32_restatement():
    """north-star network: policy/value within 1e-4 (fp32) of the torch-CPU restatement
    of the specification in corintho_ai_amd/nets.py; rows independent of their batch"""
    from corintho_ai_amd import NET_RESCNN4

    t = make_trainer("hip", 64, "", 1, 50, 16, 1.0, 0.25, 0, 1, False)
    rng = np.random.default_rng(11)
    n = 203
    states = np.zeros((n, 70), np.float32)
    states[:, :64] = rng.integers(0, 2, (n, 64))
    states[:, 64:] = rng.integers(0, 5, (n, 6)) * 0.25
    for seed, noise in ((0, False), (3, True)):
        w = nets.init_rescnn4(seed=seed, bn_noise=noise)
        t.set_net(NET_RESCNN4, w)
        ev, pr = t.net_forward(states)
        ev0, pr0 = ref_nets.rescnn4_forward_ref(w, states)
        assert np.max(np.abs(ev - ev0)) < 1e-4, np.max(np.abs(ev - ev0))
        assert np.max(np.abs(pr - pr0)) < 1e-4, np.max(np.abs(pr - pr0))
        assert np.all(np.abs(pr.sum(axis=1) - 1) < 1e-5)
        ev1, pr1 = t.net_forward(states[7:8])
        assert ev1[0] == ev[7] and np.array_equal(pr1[0], pr[7])


@pytest.mark.gpu
def test_fused_rescnn4_generation_replays_on_the_oracle():
    from corintho_ai_amd import NET_RESCNN4

    G, S_, spe = 24, 40, 8
    w = nets.init_rescnn4(seed=0, bn_noise=True)
    f = make_trainer("hip", G, "", 31, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False)
    f.set_net(NET_RESCNN4, w)
    assert f.run()
    o = O.Trainer(G, seed=31, max_searches=S_, searches_per_eval=spe)
    o.set_stagger(False)
    H.play_generation(o, G, spe, lambda s: f.net_forward(s))
    for x, y in zip(H.get_samples(f), H.get_samples(o)):
        assert x.tobytes() == y.tobytes()
    assert f.score() == o.score()


@pytest.mark.gpu
def test_rescnn4_bf16x3_within_tolerance_of_float32():
    """split-precision convolutions (bf16 MFMA x3, fp32 accumulate): still within the
    1e-4 contract of the float32 restatement, and batch independent"""
    from corintho_ai_amd import NET_RESCNN4_X3

    t = make_trainer("hip", 64, "", 1, 50, 16, 1.0, 0.25, 0, 1, False)
    rng = np.random.default_rng(12)
    n = 333
    states = np.zeros((n, 70), np.float32)
    states[:, :64] = rng.integers(0, 2, (n, 64))
    states[:, 64:] = rng.integers(0, 5, (n, 6)) * 0.25
    for seed, noise in ((0, False), (3, True)):
        w = nets.init_rescnn4(seed=seed, bn_noise=noise)
        t.set_net(NET_RESCNN4_X3, w)
        ev, pr = t.net_forward(states)
        ev0, pr0 = ref_nets.rescnn4_forward_ref(w, states)
        assert np.max(np.abs(ev - ev0)) < 1e-4, np.max(np.abs(ev - ev0))
        assert np.max(np.abs(pr - pr0)) < 1e-4, np.max(np.abs(pr - pr0))
        ev1, pr1 = t.net_forward(states[9:10])
        assert ev1[0] == ev[9] and np.array_equal(pr1[0], pr[9])


@pytest.mark.gpu
def test_fused_rescnn4_bf16x3_generation_replays_on_the_oracle():
    from corintho_ai_amd import NET_RESCNN4_X3

    G, S_, spe = 24, 40, 8
    w = nets.init_rescnn4(seed=0, bn_noise=True)
    f = make_trainer("hip", G, "", 32, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False)
    f.set_net(NET_RESCNN4_X3, w)
    assert f.run()
    o = O.Trainer(G, seed=32, max_searches=S_, searches_per_eval=spe)
    o.set_stagger(False)
    H.play_generation(o, G, spe, lambda s: f.net_forward(s))
    for x, y in zip(H.get_samples(f), H.get_samples(o)):
        assert x.tobytes() == y.tobytes()


@pytest.mark.parametrize("engine", ENGINES)
def test_fused_arena_matches_oracle(engine):
    """arena (testing=True) kept on the device: two networks, model switching as main.pyx:150-154,
    offsets recomputed at entry for the model to move (trainer.cpp:208-215) -- same games, same
    score as the oracle driven by the reference loop with the same two networks"""
    G, S_, spe = 12, 32, 8
    wa, wb = nets.init_mlp12x100(seed=4, bn_noise=True), nets.init_mlp12x100(seed=5, bn_noise=True)
    f = make_trainer(engine, G, "", 9, S_, spe, 1.0, 0.25, 0, 1, True, trace=True)
    f.set_net(1, wa, slot=0)  # best model
    f.set_net(1, wb, slot=1)  # new model
    assert f.run()
    o = O.Trainer(G, seed=9, max_searches=S_, searches_per_eval=spe, testing=True)
    o.enable_trace()
    nets2 = (lambda s: f.net_forward(s, slot=1), lambda s: f.net_forward(s, slot=0))  # to_play 0 -> new model
    H.play_generation(o, G, spe, None, nets_by_player=nets2)
    for g in range(G):
        assert np.array_equal(f.trace(g), o.trace(g)), "game %d" % g
    assert f.score() == o.score()


@pytest.mark.gpu
@pytest.mark.parametrize("kind_name", ["NET_RESCNN4_H3", "NET_RESCNN4_X6"])  # H3 = the bench default (f16x3); X6 = float32-equivalent
def test_full_size_generation_properties(kind_name):
    """BASELINE configs[1] at full size (4096 games, 400 sims/move, residual CNN, fused): too big
    for the oracle, so checked through size-independent properties -- every game finished, the
    reference's sample invariants (ranges, probability sums, the 7 symmetry copies being
    permutations, selfplayer_test.cpp:63-142), alternating outcome labels, determinism of a
    re-run, a 256-game shard reproducing its slice, and the first 64 games of the same generation
    replayed on the oracle (fed by the same device network) bit for bit.  (The whole generation on the oracle:
    tests/test_configs_gpu.py::test_cfg2_the_bench_default_whole_on_the_oracle.)"""
    import corintho_ai_amd as CA

    NET_RESCNN4_X3 = getattr(CA, kind_name)  # (the kind under test)
    G, S_, spe = 4096, 400, 16
    w = nets.init_rescnn4(0)
    t = make_trainer("hip", G, "", 12345, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False)
    t.set_net(NET_RESCNN4_X3, w)
    assert t.run()
    st = t.stats()
    assert st["nn_rows"] == st["evals"] > 0  # every evaluation consumed was one request row
    assert 0 < st["nn_rows_evaluated"] < st["nn_rows"]  # ... and the evaluation cache served some of them (on by default here)
    infos = [t.game_info(g) for g in range(0, G, 97)]
    assert all(i["done"] == 1 and i["error"] == 0 and 0 < i["n_samples"] <= 40 for i in infos)
    gs, ev, pr = H.get_samples(t)
    n = t.num_samples()
    assert gs.shape == (n * 8, 70) and 13 * G < n < 22 * G
    H.check_sample_properties(gs, ev, pr)
    assert 0.0 <= t.score() <= 1.0
    digest = (gs.tobytes(), ev.tobytes(), pr.tobytes(), t.score())
    # same seed again in the same pool: identical generation
    t.reset(12345)
    assert t.run()
    gs2, ev2, pr2 = H.get_samples(t)
    assert (gs2.tobytes(), ev2.tobytes(), pr2.tobytes(), t.score()) == digest
    # games [512, 768) of that generation as their own shard, then on the oracle with the same network
    sp_all, oc_all = t.export_samples()
    counts = [t.game_info(g)["n_samples"] for g in range(768)]
    lo, hi = sum(counts[:512]), sum(counts[:768])
    shard = make_trainer("hip", 256, "", 12345, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False, game_base=512,
                         total_games=G)
    shard.set_net(NET_RESCNN4_X3, w)
    assert shard.run()
    sp, oc = shard.export_samples()
    assert sp.tobytes() == sp_all[lo:hi].tobytes() and oc.tobytes() == oc_all[lo:hi].tobytes()
    # the first 64 games on the CPU oracle, evaluated by the same device network
    o = O.Trainer(64, seed=12345, max_searches=S_, searches_per_eval=spe, num_threads=8)
    o.set_stagger(False)
    H.play_generation(o, 64, spe, lambda st: t.net_forward(st))
    ogs, oev, opr = H.get_samples(o)
    m = sum(counts[:64])
    assert ogs.shape[0] == m * 8
    assert ogs[0::8].tobytes() == sp_all[:m, :70].tobytes()
    assert opr[0::8].tobytes() == sp_all[:m, 70:].tobytes()
    assert oev[0::8].tobytes() == oc_all[:m].tobytes()
    # the network rows the device evaluated for those 64 games = the oracle's evaluation count (a 64-game shard of the
    # same generation, whose games are the same games)
    first = make_trainer("hip", 64, "", 12345, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False, game_base=0, total_games=G)
    first.set_net(NET_RESCNN4_X3, w)
    assert first.run()
    oc_ = o.counters()
    fs = first.stats()
    assert fs["nn_rows"] == fs["evals"] == oc_["leaf_evals"] and fs["searches"] == oc_["searches"]
    assert first.export_samples()[0].tobytes() == sp_all[:m].tobytes()


@pytest.mark.parametrize("engine", ENGINES)
def test_arena_overflow_is_reported_by_the_fused_loop(engine):
    """a tree that outgrows its arena stops that game with an error; the device loop must end
    and raise (not spin, not drop the error) -- for one pool and for several"""
    w = nets.init_mlp12x100(seed=0)
    for pools in (1, 2):
        t = make_trainer(engine, 6, "", 1, 64, 8, 1.0, 0.25, 0, 1, False, arena_units=300, stagger=False, pools=pools)
        t.set_net(1, w)
        with pytest.raises(_lib.EngineError, match="arena"):
            t.run()
