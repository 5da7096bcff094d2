#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ (F1 legal masks,
F2 search traces, F3 samples, F4 Trainer-level counts, plus network logits).

The generator is the CPU oracle (oracle/), which is itself pinned against the reference's own
tests and the reference probe outputs of BASELINE.md (tests/test_oracle_reference_tests.py);
the network vectors come from the float32 restatements in corintho_ai_amd/nets.py.  The files
freeze those outputs so that the oracle, the emulation build and the MI355X build are all
checked against the same bytes (tests/test_golden.py), also on a box without the reference.

usage: python tools/gen_golden.py            (rewrites tests/golden/*.npz)
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from corintho_ai_amd import nets  # noqa: E402
from tests import ref_nets
from oracle import oracle as O  # noqa: E402
from tests import harness as H  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

# (name, G, sims, spe, eps, c_puct, seed, testing, stagger)
SELFPLAY_CASES = [
    ("selfplay_g8_s50", 8, 50, 16, 0.25, 1.0, 12345, False, True),
    ("selfplay_g4_s400", 4, 400, 16, 0.25, 1.0, 2024, False, True),
    ("selfplay_g12_s64_cpuct3", 12, 64, 16, 0.25, 3.0, 99, False, False),
    ("selfplay_g6_s30_eps0_spe1", 6, 30, 1, 0.0, 1.0, 7, False, True),
    ("arena_g10_s40", 10, 40, 8, 0.25, 1.0, 5, True, True),
]


def meta_of(pieces, to_play):
    m = 0
    for i, p in enumerate(pieces):
        m |= int(p) << (3 * i)
    return m | (int(to_play) << 18)


def rules_corpus(n_games=160, seed=2718):
    rng = np.random.default_rng(seed)
    boards, metas, masks, lines, states, moves, boards2, metas2 = [], [], [], [], [], [], [], []
    for _ in range(n_games):
        g = O.Game()
        while True:
            mask, ln = g.legal_mask()
            legal = [i for i in range(96) if mask >> i & 1]
            boards.append(g.board)
            metas.append(meta_of(g.pieces, g.to_play))
            masks.append([mask & 0xFFFFFFFF, (mask >> 32) & 0xFFFFFFFF, (mask >> 64) & 0xFFFFFFFF])
            lines.append(1 if ln else 0)
            states.append(g.state())
            if not legal:
                moves.append(-1)
                boards2.append(g.board)
                metas2.append(meta_of(g.pieces, g.to_play))
                break
            mv = int(rng.choice(legal))
            g.do_move(mv)
            moves.append(mv)
            boards2.append(g.board)
            metas2.append(meta_of(g.pieces, g.to_play))
    return dict(boards=np.array(boards, np.uint64), metas=np.array(metas, np.uint32),
                masks=np.array(masks, np.uint32), is_lines=np.array(lines, np.uint8),
                states_packed=np.packbits((np.array(states, np.float32)[:, :64] != 0).astype(np.uint8), axis=1),
                state_scalars=np.array(states, np.float32)[:, 64:],
                moves=np.array(moves, np.int32), boards_after=np.array(boards2, np.uint64),
                metas_after=np.array(metas2, np.uint32))


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def selfplay_case(G, sims, spe, eps, c_puct, seed, testing, stagger, trainer_factory=None):
    """plays one generation and returns everything the golden file holds; `trainer_factory`
    lets tests/test_golden.py run the same recipe on another engine"""
    if trainer_factory is None:
        t = O.Trainer(G, seed=seed, max_searches=sims, searches_per_eval=spe, c_puct=c_puct, epsilon=eps,
                      testing=testing, num_threads=4)
        t.enable_trace()
        t.set_stagger(stagger)
        result = t.game_result
    else:
        t = trainer_factory(G, seed, sims, spe, c_puct, eps, testing, stagger)
        result = lambda g: t.game_info(g)["result"]  # noqa: E731
    if testing:
        nets2 = (lambda s: H.hash_net(s, 1), lambda s: H.hash_net(s, 2))
        r = H.play_generation(t, G, spe, None, nets_by_player=nets2, record=True)
    else:
        r = H.play_generation(t, G, spe, H.hash_net, record=True)
    counts = np.array([a[1].shape[0] for a in r["log"]], np.int32)
    who = np.array([a[0] for a in r["log"]], np.int8)
    rows = hashlib.sha256()
    for a in r["log"]:
        rows.update(a[1].tobytes())
    traces = [np.asarray(t.trace(g)) for g in range(G)]
    gs, ev, pr = H.get_samples(t)
    return dict(
        iterations=np.int64(r["iterations"]), request_counts=counts, request_model=who,
        request_rows_sha256=np.array(rows.hexdigest()),
        trace_lengths=np.array([x.size for x in traces], np.int64),
        traces=np.concatenate([x.ravel() for x in traces]) if traces else np.zeros(0, np.int64),
        results=np.array([result(g) for g in range(G)], np.int32),
        num_samples=np.int64(t.num_samples()), score=np.float32(t.score()),
        avg_mate_length=np.float32(t.avg_mate_length()),
        samples_sha256=np.array(sha(gs, ev, pr)),
        # the first 8 plies' samples in full (identity symmetry only), for a readable diff
        head_states=gs[0:64:8].copy(), head_evals=ev[0:64:8].copy(), head_probs=pr[0:64:8].copy())


def net_vectors():
    """states met in play (first rows of a generation) and the float32 restatements' outputs"""
    t = O.Trainer(4, seed=1, max_searches=40, searches_per_eval=8)
    r = H.play_generation(t, 4, 8, H.hash_net, record=True)
    states = np.concatenate([a[1] for a in r["log"]])[:256]
    out = {"states": states}
    for name, w, f in (("mlp_seed0", nets.init_mlp12x100(0), ref_nets.mlp12x100_forward_np),
                       ("mlp_seed1_noise", nets.init_mlp12x100(1, bn_noise=True), ref_nets.mlp12x100_forward_np),
                       ("rescnn4_seed0", nets.init_rescnn4(0), ref_nets.rescnn4_forward_ref),
                       ("rescnn4_seed3_noise", nets.init_rescnn4(3, bn_noise=True), ref_nets.rescnn4_forward_ref)):
        ev, pr = f(w, states)
        out[name + "_value"] = ev.astype(np.float32)
        out[name + "_policy"] = pr.astype(np.float32)
        out[name + "_weights_sha256"] = np.array(sha(w))
    return out


NET_INITS = {
    "mlp_seed0": lambda: nets.init_mlp12x100(0),
    "mlp_seed1_noise": lambda: nets.init_mlp12x100(1, bn_noise=True),
    "rescnn4_seed0": lambda: nets.init_rescnn4(0),
    "rescnn4_seed3_noise": lambda: nets.init_rescnn4(3, bn_noise=True),
}


def main():
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "rules_corpus.npz"), **rules_corpus())
    for name, *cfg in SELFPLAY_CASES:
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **selfplay_case(*cfg))
    np.savez_compressed(os.path.join(OUT, "net_vectors.npz"), **net_vectors())
    for f in sorted(os.listdir(OUT)):
        print("%8d  %s" % (os.path.getsize(os.path.join(OUT, f)), f))


if __name__ == "__main__":
    main()
