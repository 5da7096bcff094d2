"""Host-side mirror of the reference's Trainer for the MI355X engine.

Same names, argument meaning and protocol as the Cython-declared C++ class
(corintho_ai/python/main.pyx:17-38, corintho_ai/cpp/include/trainer.h:17-81):

    t = Trainer(num_games, log_folder, seed, max_searches, searches_per_eval,
                c_puct, epsilon, num_logged, num_threads, testing)
    while not t.doIteration(evals, probs, to_play):
        n = t.num_requests(to_play); t.writeRequests(game_states, to_play)
        evals[:n], probs[:n] = model(game_states[:n])
    t.num_samples(); t.writeSamples(gs, ev, pr); t.score(); t.avg_mate_length()

plus the fused mode (`set_net`, `run`) in which the network runs on the GPU.
Buffers are caller-owned C-contiguous float32 numpy arrays, as in main.pyx:132-134.
"""
import ctypes as C

import numpy as np

from . import _lib

GAME_STATE_SIZE = 70
NUM_MOVES = 96
NUM_SYMMETRIES = 8

NET_MLP12X100 = 1
NET_RESCNN4 = 2
NET_RESCNN4_X3 = 3
NET_MLP12X100_X3 = 4
NET_RESCNN4_X6 = 5    # float32-equivalent: three bf16 terms per operand, six MFMA products
NET_MLP12X100_X6 = 6
NET_RESCNN4_H3 = 8    # two fp16 terms per operand (22 significand bits), three MFMA products
NET_MLP12X100_H3 = 9


def _f32(a, what):
    if not isinstance(a, np.ndarray) or a.dtype != np.float32 or not a.flags["C_CONTIGUOUS"]:
        raise TypeError("%s must be a C-contiguous float32 numpy array" % what)
    return a.ctypes.data_as(_lib.f32p)


def _eval_cache_cfg(eval_cache):
    """ca_config.eval_cache from the Python argument: True -> 0 (on where it pays, table sized automatically),
    False / None -> -1 (off), an int n in 6..30 -> a table of 2**n entries.  Anything else is refused
    (0 and 1 as plain ints would silently mean something else on each side of the C ABI)."""
    if eval_cache is True:
        return 0
    if eval_cache is False or eval_cache is None:
        return -1
    if isinstance(eval_cache, (int, np.integer)) and 6 <= int(eval_cache) <= 30:
        return int(eval_cache)
    raise ValueError("eval_cache must be True, False or a table size log2 in 6..30, not %r" % (eval_cache,))


class Trainer:
    def __init__(self, num_games, log_folder="", seed=0, max_searches=1600, searches_per_eval=16, c_puct=1.0,
                 epsilon=0.25, num_logged=0, num_threads=1, testing=False, *, device=0, stagger=True, arena_units=0,
                 trace=False, game_base=0, total_games=0, pools=0, analyse=False, resident=0, eval_cache=True, step_budget=0, _cdll=None):
        self._L = _cdll if _cdll is not None else _lib.load()
        self._t = C.c_void_p()
        cfg = _lib.CaConfig(num_games=num_games, seed=int(seed) & 0x7FFFFFFF if seed >= 0 else int(seed),
                            max_searches=max_searches, searches_per_eval=searches_per_eval, c_puct=c_puct,
                            epsilon=epsilon, num_logged=0, num_threads=num_threads, testing=int(bool(testing)),
                            device=device, no_stagger=int(not stagger), arena_units=arena_units, trace=int(bool(trace)),
                            game_base=game_base, total_games=total_games, pools=pools, analyse=int(bool(analyse)),
                            resident=int(resident), eval_cache=_eval_cache_cfg(eval_cache), step_budget=int(step_budget))
        self.num_games = num_games
        self.searches_per_eval = searches_per_eval
        self.testing = bool(testing)
        _lib.check(self._L, self._L.ca_trainer_create(C.byref(cfg), C.byref(self._t)))
        if num_logged:  # trainer.cpp:243-250: the first num_logged games write log_folder/game_<i>.txt
            _lib.check(self._L, self._L.ca_trainer_set_logging(self._t, str(log_folder).encode(), int(num_logged)))

    def close(self):
        if getattr(self, "_t", None) and self._t.value:
            self.unpin()  # registrations of arrays this object still holds, before the engine goes
            self._L.ca_trainer_destroy(self._t)
            self._t = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reference surface (main.pyx:30-38) ----
    def num_requests(self, to_play=-1):
        out = C.c_int32()
        _lib.check(self._L, self._L.ca_trainer_num_requests(self._t, to_play, C.byref(out)))
        return out.value

    def num_samples(self):
        out = C.c_int32()
        _lib.check(self._L, self._L.ca_trainer_num_samples(self._t, C.byref(out)))
        return out.value

    def score(self):
        out = C.c_float()
        _lib.check(self._L, self._L.ca_trainer_score(self._t, C.byref(out)))
        return out.value

    def avg_mate_length(self):
        out = C.c_float()
        _lib.check(self._L, self._L.ca_trainer_avg_mate_length(self._t, C.byref(out)))
        return out.value

    def writeRequests(self, game_states, to_play=-1):
        _lib.check(self._L, self._L.ca_trainer_write_requests(self._t, _f32(game_states, "game_states"), to_play))

    def writeSamples(self, game_states, eval_samples, prob_samples):
        _lib.check(self._L, self._L.ca_trainer_write_samples(self._t, _f32(game_states, "game_states"),
                                                             _f32(eval_samples, "eval_samples"),
                                                             _f32(prob_samples, "prob_samples")))

    def writeScores(self, file):
        if isinstance(file, str):
            file = file.encode()
        _lib.check(self._L, self._L.ca_trainer_write_scores(self._t, file))

    def doIteration(self, evaluations, probabilities, to_play=-1):
        done = C.c_int32()
        _lib.check(self._L, self._L.ca_trainer_do_iteration(self._t, _f32(evaluations, "evaluations"),
                                                            _f32(probabilities, "probabilities"), to_play,
                                                            C.byref(done)))
        return bool(done.value)

    # ---- fused mode ----
    def set_net(self, kind, weights, slot=0):
        w = np.ascontiguousarray(weights, dtype=np.float32)
        _lib.check(self._L, self._L.ca_trainer_set_net(self._t, slot, kind, _f32(w, "weights"), w.size))

    def run(self, max_iterations=0):
        done = C.c_int32()
        _lib.check(self._L, self._L.ca_trainer_run(self._t, max_iterations, C.byref(done)))
        return bool(done.value)

    def pin(self, *arrays):
        """Page-lock caller arrays (the evals / probs / game_states of the play loop, main.pyx:132-134) for
        direct DMA.  They must outlive the trainer or be passed to unpin() first.  -> all pinned?"""
        ok = True
        pinned = self.__dict__.setdefault("_pinned", {})
        for a in arrays:
            _f32(a, "array")
            got = C.c_int32()
            _lib.check(self._L, self._L.ca_trainer_pin_host(self._t, C.c_void_p(a.ctypes.data), a.nbytes, C.byref(got)))
            if got.value:
                # a registration outlives the caller's own reference only if we keep one: a dropped array's
                # address could be handed to a new one, which the engine would take for the registered pages
                pinned[a.ctypes.data] = a
            ok = ok and bool(got.value)
        return ok

    def unpin(self, *arrays):
        """unpin the given arrays; with no argument, every array pinned through this object"""
        pinned = self.__dict__.setdefault("_pinned", {})
        addrs = [a.ctypes.data for a in arrays] if arrays else list(pinned)
        for addr in addrs:
            _lib.check(self._L, self._L.ca_trainer_unpin_host(self._t, C.c_void_p(addr)))
            pinned.pop(addr, None)

    def net_forward(self, states, slot=0, out_evals=None, out_probs=None):
        """evaluate `states` with the network of `slot`; out_evals / out_probs: caller arrays to fill (their
        first len(states) rows) instead of new ones"""
        s = np.ascontiguousarray(states, dtype=np.float32)
        if s.ndim != 2 or s.shape[1] != GAME_STATE_SIZE:
            raise ValueError("net_forward: states must be [n][%d]" % GAME_STATE_SIZE)
        n = s.shape[0]
        ev = np.zeros(n, np.float32) if out_evals is None else out_evals
        pr = np.zeros((n, NUM_MOVES), np.float32) if out_probs is None else out_probs
        if ev.size < n:
            raise ValueError("net_forward: out_evals holds %d values, %d rows to write" % (ev.size, n))
        if pr.ndim != 2 or pr.shape[1] != NUM_MOVES or pr.shape[0] < n:
            raise ValueError("net_forward: out_probs must be [>= %d][%d], got %s" % (n, NUM_MOVES, pr.shape))
        if n == 0:
            return ev, pr
        _lib.check(self._L, self._L.ca_trainer_net_forward(self._t, slot, _f32(s, "states"), n, _f32(ev, "ev"),
                                                           _f32(pr, "pr")))
        return ev, pr

    def net_bench(self, states, reps=20, slot=0):
        """kernel-only milliseconds per evaluation of `states` (diagnostics)"""
        s = np.ascontiguousarray(states, dtype=np.float32)
        ms = C.c_float()
        _lib.check(self._L, self._L.ca_trainer_net_bench(self._t, slot, _f32(s, "states"), s.shape[0], reps, C.byref(ms)))
        return ms.value

    def export_samples(self):
        n = self.num_samples()
        sp = np.zeros((n, GAME_STATE_SIZE + NUM_MOVES), np.float32)
        oc = np.zeros(n, np.float32)
        if n:
            _lib.check(self._L, self._L.ca_trainer_export_samples(self._t, _f32(sp, "sp"), _f32(oc, "oc")))
        return sp, oc

    def reset(self, seed):
        """new generation in the same device pool (no reallocation)"""
        _lib.check(self._L, self._L.ca_trainer_reset(self._t, int(seed)))

    def pack_samples_device(self, d_state_policy_ptr, d_outcome_ptr, cap_rows):
        """pack un-augmented samples into caller-owned device memory; returns rows"""
        n = C.c_int32()
        _lib.check(self._L, self._L.ca_trainer_pack_samples_device(self._t, C.c_void_p(d_state_policy_ptr),
                                                                   C.c_void_p(d_outcome_ptr), cap_rows, C.byref(n)))
        return n.value

    # ---- analysis mode (N x DockerMC, dockermc.h:13-51) ----
    def set_positions(self, boards, to_play, pieces, seeds):
        b = np.ascontiguousarray(boards, dtype=np.int32).reshape(self.num_games, 64)
        tp = np.ascontiguousarray(to_play, dtype=np.int32).reshape(self.num_games)
        pc = np.ascontiguousarray(pieces, dtype=np.int32).reshape(self.num_games, 6)
        sd = np.ascontiguousarray(seeds, dtype=np.int32).reshape(self.num_games)
        p = lambda a: a.ctypes.data_as(_lib.i32p)  # noqa: E731
        _lib.check(self._L, self._L.ca_trainer_set_positions(self._t, p(b), p(tp), p(pc), p(sd)))

    def finish(self):
        """analysis mode: every position whose search has not ended chooses its move on its tree as it stands
        (DockerMC::chooseMove after the time limit of choose_move.pyx:110-117)"""
        _lib.check(self._L, self._L.ca_trainer_finish(self._t))

    def analysis(self):
        out = np.zeros((self.num_games, 8), np.int32)
        _lib.check(self._L, self._L.ca_trainer_analysis(self._t, out.ctypes.data_as(_lib.i32p)))
        return out

    # ---- introspection ----
    def stats(self):
        s = _lib.CaStats()
        _lib.check(self._L, self._L.ca_trainer_stats(self._t, C.byref(s)))
        return {k: getattr(s, k) for k, _ in s._fields_}

    def game_info(self, g):
        out = (C.c_int32 * 8)()
        _lib.check(self._L, self._L.ca_trainer_game_info(self._t, g, out))
        keys = ("to_play", "done", "result", "n_samples", "n_pending", "error", "mate_turn", "plies")
        return dict(zip(keys, list(out)))

    def trace(self, g):
        n = C.c_int32()
        _lib.check(self._L, self._L.ca_trainer_trace(self._t, g, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), np.int32)
        _lib.check(self._L, self._L.ca_trainer_trace(self._t, g, out.ctypes.data_as(_lib.i32p), n.value, C.byref(n)))
        return out[: n.value]


def expand_samples(state_policy, outcome, device=0, _cdll=None):
    """x8 symmetry expansion of gathered (state, policy, outcome) rows ->
    the three arrays of Trainer::writeSamples (trainer.cpp:103-113)."""
    L = _cdll if _cdll is not None else _lib.load()
    sp = np.ascontiguousarray(state_policy, dtype=np.float32)
    oc = np.ascontiguousarray(outcome, dtype=np.float32)
    n = sp.shape[0]
    gs = np.zeros((n * NUM_SYMMETRIES, GAME_STATE_SIZE), np.float32)
    ev = np.zeros(n * NUM_SYMMETRIES, np.float32)
    pr = np.zeros((n * NUM_SYMMETRIES, NUM_MOVES), np.float32)
    if n:
        _lib.check(L, L.ca_expand_samples(device, _f32(sp, "sp"), _f32(oc, "oc"), n, _f32(gs, "gs"), _f32(ev, "ev"),
                                          _f32(pr, "pr")))
    return gs, ev, pr
