"""Shared test harness: the reference's Python play loop restated
(corintho_ai/python/main.pyx:123-187 play_games, :189-198 get_samples), a few
deterministic stand-in networks, and the reference's sample property checks
(tests/cpp/selfplayer_test.cpp:22-145, trainer_test.cpp:21-136).

Works with any object exposing the reference Trainer surface
(num_requests / writeRequests / doIteration / num_samples / writeSamples /
score): the CPU oracle's and the HIP engine's.
"""
import numpy as np

GS, NM, NSYM = 70, 96, 8


# ----------------------------------------------------------------- fake nets
def uniform_net(states):
    """value 0, uniform priors -- the net of the BASELINE.md probe runs."""
    n = states.shape[0]
    return np.zeros(n, np.float32), np.full((n, NM), 1.0 / NM, np.float32)


def _mix(h):
    h = (h ^ (h >> np.uint64(33))) * np.uint64(0xFF51AFD7ED558CCD)
    h = (h ^ (h >> np.uint64(33))) * np.uint64(0xC4CEB9FE1A85EC53)
    return h ^ (h >> np.uint64(33))


def hash_net(states, salt=0):
    """Deterministic, batch-independent stand-in network: value in (-1, 1) and
    96 positive priors, both functions of the 70-float row only (integer
    hashing, then exact float32 conversions)."""
    n = states.shape[0]
    with np.errstate(over="ignore"):
        q = np.rint(states.astype(np.float64) * 4.0).astype(np.uint64)  # entries are k/4
        h = np.full(n, np.uint64(0x9E3779B97F4A7C15) + np.uint64(salt), dtype=np.uint64)
        for j in range(GS):
            h = _mix(h ^ (q[:, j] + np.uint64(j * 0x100000001B3 + 1)))
        ev = ((h >> np.uint64(40)).astype(np.float64) / float(1 << 24) * 2.0 - 1.0).astype(np.float32)
        pr = np.empty((n, NM), np.float32)
        for m in range(NM):
            hm = _mix(h + np.uint64((m + 1) * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF))
            pr[:, m] = (((hm >> np.uint64(40)).astype(np.float64) + 1.0) / float((1 << 24) + 1)).astype(np.float32)
    return ev, pr


def random_net_factory(seed=12345):
    """eval ~ U(-1,1), probs ~ U(0,1) normalised -- the kind of stand-in the
    reference tests use (selfplayer_test.cpp:31-55).  NOT state-determined."""
    rng = np.random.default_rng(seed)

    def net(states):
        n = states.shape[0]
        ev = rng.uniform(-1, 1, n).astype(np.float32)
        pr = rng.uniform(0, 1, (n, NM)).astype(np.float32)
        pr /= pr.sum(axis=1, keepdims=True)
        return ev, pr

    return net


# ------------------------------------------------------------------ play loop
def play_generation(trainer, num_games, searches_per_eval, net, nets_by_player=None, record=False, max_iters=10**7,
                    on_iteration=None):
    """main.pyx:123-187.  `net(states)->(evals, probs)`.  For the arena
    (`nets_by_player` = (new_model_net, best_model_net)) to_play starts at 0 and
    flips when the active model has no request (main.pyx:150-154)."""
    cap = num_games * searches_per_eval
    evals = np.zeros(cap, np.float32)
    probs = np.zeros((cap, NM), np.float32)
    game_states = np.zeros((cap, GS), np.float32)
    to_play = -1 if nets_by_player is None else 0
    log = []
    iters = 0
    evals_done = 0
    while True:
        res = trainer.doIteration(evals, probs, to_play)
        iters += 1
        if res:
            break
        assert iters < max_iters, "play loop did not terminate"
        n = trainer.num_requests(to_play)
        if n == 0:
            if to_play != -1:
                to_play = 1 - to_play
                continue
            raise RuntimeError("No requests during training")
        trainer.writeRequests(game_states, to_play)
        f = net if nets_by_player is None else nets_by_player[to_play]
        e, p = f(game_states[:n])
        evals[:n] = e
        probs[:n] = p
        evals_done += 1
        if record:
            log.append((to_play, game_states[:n].copy()))
        if on_iteration is not None:
            on_iteration(iters, to_play, n, game_states[:n], evals[:n], probs[:n])
    return {"iterations": iters, "evals_done": evals_done, "log": log}


def get_samples(trainer):
    """main.pyx:189-198"""
    n = trainer.num_samples()
    gs = np.zeros((n * NSYM, GS), np.float32)
    ev = np.zeros(n * NSYM, np.float32)
    pr = np.zeros((n * NSYM, NM), np.float32)
    if n:
        trainer.writeSamples(gs, ev, pr)
    return gs, ev, pr


# ---------------------------------------------------- reference sample checks
def check_sample_properties(gs, ev, pr, max_per_game=None):
    """selfplayer_test.cpp:63-142 / trainer_test.cpp:60-134"""
    n = ev.shape[0] // NSYM
    assert np.all((gs >= 0.0) & (gs <= 1.0))
    assert np.all((ev >= -1.0) & (ev <= 1.0))
    assert np.all((pr >= 0.0) & (pr <= 1.0))
    s = pr.sum(axis=1)
    assert np.all((s >= 0.99) & (s <= 1.01))
    g = gs.reshape(n, NSYM, GS)
    p = pr.reshape(n, NSYM, NM)
    g0 = -np.sort(-g[:, 0, :], axis=1)
    p0 = -np.sort(-p[:, 0, :], axis=1)
    for k in range(1, NSYM):
        assert np.all(np.abs(-np.sort(-g[:, k, :], axis=1) - g0) < 1e-6)
        assert np.all(np.abs(-np.sort(-p[:, k, :], axis=1) - p0) < 1e-6)
    # every symmetry copy carries the same label
    e = ev.reshape(n, NSYM)
    assert np.all(e == e[:, :1])


# ------------------------------------------------------------- tournament loop
def play_tourney(tourney, model_ids, nets_by_model, rows, record=False, max_rounds=10**6):
    """rating/tourney.pyx:113-160 play_games: for every model id in turn, evaluate that model's
    requests and iterate its matches.  The three arrays are allocated once with `rows` rows
    (tourney.pyx:118-120) and kept, so reads through the reference's offset table are defined."""
    evals = np.zeros(rows, np.float32)
    probs = np.zeros((rows, NM), np.float32)
    game_states = np.zeros((rows, GS), np.float32)
    log = []
    rounds = 0
    while not tourney.all_done():
        for mid in model_ids:
            n = tourney.num_requests(mid) if mid >= 0 else 0
            if n > 0:
                tourney.writeRequests(game_states, mid)
                e, p = nets_by_model[mid](game_states[:n])
                evals[:n] = e
                probs[:n] = p
                if record:
                    log.append((mid, game_states[:n].copy()))
            elif record:
                log.append((mid, np.zeros((0, GS), np.float32)))
            tourney.doIteration(evals, probs, mid)
        rounds += 1
        assert rounds < max_rounds, "tournament did not terminate"
    return {"rounds": rounds, "log": log}
