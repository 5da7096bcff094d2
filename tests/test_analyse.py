"""`DockerMC` (dockermc.h:13-51, dockermc.cpp, trainmc.cpp:38-45) -- the web app's
single-position search -- as a batch on the engine: N positions, one wavefront each.  Every position
is checked against the oracle's DockerMC restatement driven by the loop of docker/choose_move.pyx
(:88-133 search, :199-221 result) with the same stand-in network: identical request rows in every
iteration, identical move, done / drawn flags, legal moves, node count and root evaluation."""
import numpy as np
import pytest

from corintho_ai_amd import nets
from corintho_ai_amd.analyse import Analyser
from oracle import oracle as O
from tests import harness as H
from tests.engines import ENGINES, cdll


def _positions(n, seed, min_plies=0, max_plies=24):
    """positions met in random playouts, as DockerMC constructor arguments"""
    rng = np.random.default_rng(seed)
    boards, to_play, pieces = [], [], []
    while len(boards) < n:
        g = O.Game()
        plies = int(rng.integers(min_plies, max_plies + 1))
        ok = True
        for _ in range(plies):
            mask, _ = g.legal_mask()
            moves = [i for i in range(96) if mask >> i & 1]
            if not moves:
                ok = False
                break
            g.do_move(int(rng.choice(moves)))
        if not ok and rng.random() < 0.7:
            continue  # keep a few terminal positions (pre-result path), drop most
        boards.append([(g.board >> i) & 1 for i in range(64)])
        to_play.append(g.to_play)
        pieces.append(list(g.pieces))
    return np.array(boards, np.int32), np.array(to_play, np.int32), np.array(pieces, np.int32)


def _oracle_result(board, tp, pc, seed, S_, spe, net):
    """docker/choose_move.pyx:155-221 with a fixed search budget"""
    mc = O.DockerMC(int(seed), S_, spe, 1.0, 0.25, board, int(tp), pc)
    if mc.done():
        return {"pre-result": "draw" if mc.drawn() else "win"}, []
    evals = np.zeros(spe, np.float32)
    probs = np.zeros((spe, 96), np.float32)
    gs = np.zeros((spe, 70), np.float32)
    log = []
    while not mc.doIteration(evals, probs):
        n = mc.num_requests()
        if n == 0:
            break
        mc.writeRequests(gs)
        e, p = net(gs[:n])
        evals[:n] = e
        probs[:n] = p
        log.append(gs[:n].copy())
    move = mc.chooseMove()
    done = mc.done()
    nodes = mc.num_nodes()
    ev = mc.eval()
    return {"move": move, "is_done": done, "has_won": bool(done and not mc.drawn()),
            "legal_moves": [] if done else [i for i in range(96) if mc.getLegalMoves()[i]],
            "nodes_searched": nodes, "eval_sum": ev}, log


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("S_,spe", [(48, 8), (200, 16), (7, 1)])
def test_batched_analysis_matches_dockermc(engine, S_, spe):
    n = 40 if engine == "hip" else 24
    boards, tp, pc = _positions(n, seed=S_)
    seeds = np.arange(n, dtype=np.int32) * 7 + 3
    a = Analyser(boards, tp, pc, seeds, S_, spe, _cdll=cdll(engine))
    cap = n * spe
    evals = np.zeros(cap, np.float32)
    probs = np.zeros((cap, 96), np.float32)
    gs = np.zeros((cap, 70), np.float32)
    rows_by_iter = []
    while not a.doIteration(evals, probs):
        k = a.num_requests()
        assert k > 0
        a.writeRequests(gs)
        e, p = H.hash_net(gs[:k])
        evals[:k] = e
        probs[:k] = p
        rows_by_iter.append(gs[:k].copy())
    got = a.results()
    all_rows = set()
    for r in rows_by_iter:
        all_rows.update(x.tobytes() for x in r)
    n_pre = 0
    for i in range(n):
        want, log = _oracle_result(boards[i], tp[i], pc[i], seeds[i], S_, spe, H.hash_net)
        if "pre-result" in want:
            n_pre += 1
            assert got[i] == want
            continue
        g = dict(got[i])
        g.pop("evaluation")
        assert g == want, "position %d" % i
        for r in log:  # every row the single-position search asked for was asked for by the batch
            assert all(x.tobytes() in all_rows for x in r)
    assert n_pre < n


@pytest.mark.parametrize("engine", ENGINES)
def test_fused_analysis_equals_host_driven(engine):
    """network on the device (set_net + run) = the host-driven protocol with the same network"""
    n, S_, spe = 20, 64, 8
    boards, tp, pc = _positions(n, seed=5, min_plies=2)
    seeds = np.arange(n, dtype=np.int32) + 100
    w = nets.init_mlp12x100(seed=2, bn_noise=True)
    f = Analyser(boards, tp, pc, seeds, S_, spe, _cdll=cdll(engine))
    f.set_net(1, w)
    assert f.run()
    h = Analyser(boards, tp, pc, seeds, S_, spe, _cdll=cdll(engine))
    h.set_net(1, w)
    cap = n * spe
    evals, probs, gs = np.zeros(cap, np.float32), np.zeros((cap, 96), np.float32), np.zeros((cap, 70), np.float32)
    while not h.doIteration(evals, probs):
        k = h.num_requests()
        h.writeRequests(gs)
        e, p = h.net_forward(gs[:k])
        evals[:k] = e
        probs[:k] = p
    assert f.results() == h.results()
    for i in range(n):
        want, _ = _oracle_result(boards[i], tp[i], pc[i], seeds[i], S_, spe, lambda s: h.net_forward(s))
        g = dict(f.results()[i])
        g.pop("evaluation", None)
        assert g == want


@pytest.mark.parametrize("engine", ENGINES)
def test_choose_move_after_a_time_limit(engine):
    """docker/choose_move.pyx:110-117 leaves its loop on a TIME limit and calls chooseMove on the search as it
    stands (:199): ca_trainer_finish after k iterations = the oracle's DockerMC::chooseMove after k iterations,
    the evaluations still pending never received.  Positions whose search ended earlier keep their result."""
    n, S_, spe = 24, 400, 8
    boards, tp, pc = _positions(n, seed=11, min_plies=1)
    seeds = np.arange(n, dtype=np.int32) * 5 + 1
    for k in (1, 2, 7):
        a = Analyser(boards, tp, pc, seeds, S_, spe, _cdll=cdll(engine))
        cap = n * spe
        evals, probs, gs = np.zeros(cap, np.float32), np.zeros((cap, 96), np.float32), np.zeros((cap, 70), np.float32)
        for _ in range(k):
            if a.doIteration(evals, probs):
                break
            m = a.num_requests()
            a.writeRequests(gs)
            evals[:m], probs[:m] = H.hash_net(gs[:m])
        a.finish()
        got = a.results()
        for i in range(n):
            mc = O.DockerMC(int(seeds[i]), S_, spe, 1.0, 0.25, boards[i], int(tp[i]), pc[i])
            if mc.done():
                assert "pre-result" in got[i]
                continue
            e1, p1, g1 = np.zeros(spe, np.float32), np.zeros((spe, 96), np.float32), np.zeros((spe, 70), np.float32)
            for _ in range(k):
                if mc.doIteration(e1, p1):
                    break
                m = mc.num_requests()
                if m == 0:
                    break
                mc.writeRequests(g1)
                e1[:m], p1[:m] = H.hash_net(g1[:m])
            move = mc.chooseMove()
            assert got[i]["move"] == move, (k, i)
            assert got[i]["is_done"] == mc.done() and got[i]["nodes_searched"] == mc.num_nodes()
            assert got[i]["eval_sum"] == mc.eval()
        a.close()
