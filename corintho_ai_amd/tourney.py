"""Host-side mirror of the reference's Tourney for the MI355X engine.

Same names, argument meaning and protocol as the Cython-declared C++ class
(corintho_ai/rating/tourney.pyx:15-31, corintho_ai/cpp/include/tourney.h:13-46):

    t = Tourney(num_threads, log_folder)
    t.addPlayer(player_id, model_id, max_searches, searches_per_eval, c_puct, epsilon, random)
    t.addMatch(player1, player2, logging)
    while not t.all_done():
        for id in model_ids:                      # tourney.pyx:127-159
            n = t.num_requests(id) if id >= 0 else 0
            if n: t.writeRequests(game_states, id); evals, probs = model[id](game_states[:n])
            t.doIteration(evals, probs, id)
    t.writeScores(file)
"""
import ctypes as C

import numpy as np

from . import _lib
from .trainer import _f32


class Tourney:
    def __init__(self, num_threads=1, log_folder="", *, device=0, arena_units=0, trace=False, _cdll=None):
        self._L = _cdll if _cdll is not None else _lib.load()
        self._t = C.c_void_p()
        _lib.check(self._L, self._L.ca_tourney_create(device, arena_units, int(bool(trace)), C.byref(self._t)))
        if log_folder:  # tourney.h:46: matches added with logging=True write <log_folder>/match_<p1>_<p2>_<index>.txt
            _lib.check(self._L, self._L.ca_tourney_set_log_folder(self._t, str(log_folder).encode()))

    def close(self):
        if getattr(self, "_t", None) and self._t.value:
            self._L.ca_tourney_destroy(self._t)
            self._t = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- the reference surface
    def addPlayer(self, player_id, model_id, max_searches=1600, searches_per_eval=16, c_puct=1.0, epsilon=0.25,
                  random=False):
        _lib.check(self._L, self._L.ca_tourney_add_player(self._t, player_id, model_id, max_searches, searches_per_eval,
                                                          c_puct, epsilon, int(bool(random))))

    def addMatch(self, player1, player2, logging=False):
        _lib.check(self._L, self._L.ca_tourney_add_match(self._t, player1, player2, int(bool(logging))))

    def all_done(self):
        out = C.c_int32()
        _lib.check(self._L, self._L.ca_tourney_all_done(self._t, C.byref(out)))
        return bool(out.value)

    def num_requests(self, id):
        out = C.c_int32()
        _lib.check(self._L, self._L.ca_tourney_num_requests(self._t, id, C.byref(out)))
        return out.value

    def writeRequests(self, game_states, id):
        _lib.check(self._L, self._L.ca_tourney_write_requests(self._t, _f32(game_states, "game_states"), id))

    def doIteration(self, evaluations, probabilities, id):
        rows = min(evaluations.shape[0], probabilities.shape[0])
        _lib.check(self._L, self._L.ca_tourney_do_iteration(self._t, _f32(evaluations, "evaluations"),
                                                            _f32(probabilities, "probabilities"), rows, id))

    def writeScores(self, filename):
        _lib.check(self._L, self._L.ca_tourney_write_scores(self._t, str(filename).encode()))

    # ---- fused mode (not in the reference): the networks run on the GPU too
    def set_net(self, model_id, kind, weights):
        w = np.ascontiguousarray(weights, dtype=np.float32)
        _lib.check(self._L, self._L.ca_tourney_set_net(self._t, model_id, kind, _f32(w, "weights"), w.size))

    def set_exact_offsets(self, on=True):
        """diagnostic: matches read their own rows instead of the reference's offset table (tourney.cpp:55-62)"""
        _lib.check(self._L, self._L.ca_tourney_set_exact_offsets(self._t, int(bool(on))))

    def run(self, max_rounds=0):
        done = C.c_int32()
        _lib.check(self._L, self._L.ca_tourney_run(self._t, max_rounds, C.byref(done)))
        return bool(done.value)

    # ---- introspection
    def num_matches(self):
        out = C.c_int32()
        _lib.check(self._L, self._L.ca_tourney_num_matches(self._t, C.byref(out)))
        return out.value

    def match_info(self, i):
        out = (C.c_int32 * 8)()
        _lib.check(self._L, self._L.ca_tourney_match_info(self._t, i, out))
        keys = ("player1", "player2", "done", "result", "to_play", "n_pending", "plies", "error")
        return dict(zip(keys, list(out)))

    def match_score(self, i):
        out = C.c_float()
        _lib.check(self._L, self._L.ca_tourney_match_score(self._t, i, C.byref(out)))
        return out.value

    def trace(self, i):
        n = C.c_int32()
        _lib.check(self._L, self._L.ca_tourney_trace(self._t, i, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), np.int32)
        _lib.check(self._L, self._L.ca_tourney_trace(self._t, i, out.ctypes.data_as(_lib.i32p), n.value, C.byref(n)))
        return out[:n.value]

    def stats(self):
        s = _lib.CaStats()
        _lib.check(self._L, self._L.ca_tourney_stats(self._t, C.byref(s)))
        return {k: getattr(s, k) for k, _ in s._fields_}


def read_pairings(player_file, match_file):
    """The two text files of rating/tourney.pyx:63-111 (get_tourney):
    players: first line the count, then per player `model_id max_searches searches_per_eval c_puct epsilon random`
             (player ids are the line numbers);
    matches: first line the count, then per match `player1 player2 logging`."""
    players, matches = [], []
    with open(player_file) as f:
        n = int(f.readline())
        for pid in range(n):
            x = f.readline().split()
            players.append((pid, int(x[0]), int(x[1]), int(x[2]), float(x[3]), float(x[4]), float(x[5]) == 1.0))
    with open(match_file) as f:
        n = int(f.readline())
        for _ in range(n):
            a, b, lg = (int(v) for v in f.readline().split())
            matches.append((a, b, lg == 1))
    return players, matches


def run(model_paths, player_file, match_file, log_folder, num_threads=1, *, device=0):
    """rating/tourney.pyx:179-end `run`, with the whole tournament on the GPU: `model_paths[i]` is
    the TFLite checkpoint of model id i (imported by tflite_import), pairings come from the two
    text files, the result goes to `<log_folder>/scores.txt`."""
    import os

    from .tflite_import import mlp12x100_from_tflite
    from .trainer import NET_MLP12X100

    players, matches = read_pairings(player_file, match_file)
    t = Tourney(num_threads, log_folder, device=device)
    for p in players:
        t.addPlayer(*p)
    for a, b, lg in matches:
        t.addMatch(a, b, lg)
    for mid in sorted({p[1] for p in players if p[1] >= 0}):
        t.set_net(mid, NET_MLP12X100, mlp12x100_from_tflite(model_paths[mid]))
    t.run()
    os.makedirs(log_folder, exist_ok=True)
    t.writeScores(os.path.join(log_folder, "scores.txt"))
    return t
