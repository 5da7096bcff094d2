"""ctypes binding of the CPU oracle (oracle/libcorintho_oracle.so).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this.  The product (corintho_ai_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcorintho_oracle.so")

GAME_STATE_SIZE = 70
NUM_MOVES = 96
NUM_SYMMETRIES = 8


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("corintho_oracle.c", "corintho_oracle.h", "tables.inc")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    f32p = C.POINTER(C.c_float)
    i8p = C.POINTER(C.c_int8)
    u32p = C.POINTER(C.c_uint32)
    i32p = C.POINTER(C.c_int32)
    L.co_legal_moves.argtypes = [C.c_uint64, i8p, C.c_int, u32p]
    L.co_legal_moves.restype = C.c_int
    L.co_do_move.argtypes = [C.POINTER(C.c_uint64), i8p, C.POINTER(C.c_int), C.c_int]
    L.co_write_game_state.argtypes = [C.c_uint64, i8p, C.c_int, f32p]
    L.co_terminal_result.argtypes = [C.c_uint64, i8p, C.c_int]
    L.co_terminal_result.restype = C.c_int
    L.co_decode_move.argtypes = [C.c_int, C.POINTER(C.c_int)]
    L.co_encode_place.argtypes = [C.c_int] * 3
    L.co_encode_place.restype = C.c_int
    L.co_encode_move.argtypes = [C.c_int] * 4
    L.co_encode_move.restype = C.c_int
    L.co_line_breakers.restype = u32p
    L.co_gamma_samples.restype = f32p
    L.co_space_symmetries.restype = i32p
    L.co_move_symmetries.restype = i32p
    L.co_mt_create.argtypes = [C.c_uint32]
    L.co_mt_create.restype = C.c_void_p
    L.co_mt_next.argtypes = [C.c_void_p]
    L.co_mt_next.restype = C.c_uint32
    L.co_mt_destroy.argtypes = [C.c_void_p]
    L.co_trainer_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int]
    L.co_trainer_create.restype = C.c_void_p
    L.co_trainer_create_slice.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                                          C.c_int]
    L.co_trainer_create_slice.restype = C.c_void_p
    L.co_trainer_write_scores.argtypes = [C.c_void_p, C.c_char_p]
    L.co_trainer_write_scores.restype = C.c_int
    L.co_trainer_destroy.argtypes = [C.c_void_p]
    L.co_trainer_num_requests.argtypes = [C.c_void_p, C.c_int]
    L.co_trainer_num_requests.restype = C.c_int
    L.co_trainer_num_samples.argtypes = [C.c_void_p]
    L.co_trainer_num_samples.restype = C.c_int
    L.co_trainer_score.argtypes = [C.c_void_p]
    L.co_trainer_score.restype = C.c_float
    L.co_trainer_avg_mate_length.argtypes = [C.c_void_p]
    L.co_trainer_avg_mate_length.restype = C.c_float
    L.co_trainer_write_requests.argtypes = [C.c_void_p, f32p, C.c_int]
    L.co_trainer_write_samples.argtypes = [C.c_void_p, f32p, f32p, f32p]
    L.co_trainer_do_iteration.argtypes = [C.c_void_p, f32p, f32p, C.c_int]
    L.co_trainer_do_iteration.restype = C.c_int
    L.co_trainer_set_stagger.argtypes = [C.c_void_p, C.c_int]
    for name in ("game_result", "game_to_play", "game_num_requests", "game_num_samples", "game_done"):
        fn = getattr(L, "co_trainer_" + name)
        fn.argtypes = [C.c_void_p, C.c_int]
        fn.restype = C.c_int
    L.co_trainer_enable_trace.argtypes = [C.c_void_p, C.c_int]
    L.co_trainer_set_logging.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.co_trainer_set_logging.restype = C.c_int
    L.co_trainer_trace.argtypes = [C.c_void_p, C.c_int, i32p, C.c_int]
    L.co_trainer_trace.restype = C.c_int
    L.co_trainer_counters.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    vp = C.c_void_p
    L.co_dockermc_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, i32p, C.c_int, i32p]
    L.co_dockermc_create.restype = vp
    L.co_dockermc_destroy.argtypes = [vp]
    L.co_dockermc_eval.argtypes = [vp]
    L.co_dockermc_eval.restype = C.c_float
    for name in ("num_requests", "num_nodes", "done", "drawn", "choose_move"):
        fn = getattr(L, "co_dockermc_" + name)
        fn.argtypes = [vp]
        fn.restype = C.c_int
    L.co_dockermc_write_requests.argtypes = [vp, f32p]
    L.co_dockermc_get_legal_moves.argtypes = [vp, i32p]
    L.co_dockermc_do_iteration.argtypes = [vp, f32p, f32p]
    L.co_dockermc_do_iteration.restype = C.c_int
    L.co_tourney_create.argtypes = [C.c_int]
    L.co_tourney_create.restype = vp
    L.co_tourney_destroy.argtypes = [vp]
    L.co_tourney_add_player.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int]
    L.co_tourney_add_player.restype = C.c_int
    L.co_tourney_add_match.argtypes = [vp, C.c_int, C.c_int]
    L.co_tourney_add_match.restype = C.c_int
    L.co_tourney_add_match_logged.argtypes = [vp, C.c_int, C.c_int, C.c_char_p]
    L.co_tourney_add_match_logged.restype = C.c_int
    L.co_tourney_all_done.argtypes = [vp]
    L.co_tourney_all_done.restype = C.c_int
    L.co_tourney_num_requests.argtypes = [vp, C.c_int]
    L.co_tourney_num_requests.restype = C.c_int
    L.co_tourney_write_requests.argtypes = [vp, f32p, C.c_int]
    L.co_tourney_do_iteration.argtypes = [vp, f32p, f32p, C.c_int]
    L.co_tourney_do_iteration.restype = None
    L.co_tourney_set_exact_offsets.argtypes = [vp, C.c_int]
    L.co_tourney_set_exact_offsets.restype = None
    L.co_tourney_write_scores.argtypes = [vp, C.c_char_p]
    L.co_tourney_write_scores.restype = C.c_int
    L.co_tourney_num_matches.argtypes = [vp]
    L.co_tourney_num_matches.restype = C.c_int
    for name in ("match_done", "match_result", "match_to_play", "match_num_requests"):
        fn = getattr(L, "co_tourney_" + name)
        fn.argtypes = [vp, C.c_int]
        fn.restype = C.c_int
    L.co_tourney_match_score.argtypes = [vp, C.c_int]
    L.co_tourney_match_score.restype = C.c_float
    L.co_tourney_enable_trace.argtypes = [vp, C.c_int]
    L.co_tourney_trace.argtypes = [vp, C.c_int, i32p, C.c_int]
    L.co_tourney_trace.restype = C.c_int
    L.co_tourney_counters.argtypes = [vp, C.POINTER(C.c_int64)]
    L.co_uniform_below.argtypes = [vp, C.c_uint32]
    L.co_uniform_below.restype = C.c_uint32
    _lib = L
    return L


def _f32(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


# ------------------------------------------------------------------ rules
class Game:
    """Value-type mirror of the reference's Game (game.h:19-136)."""

    def __init__(self, board=0, pieces=(4, 4, 4, 4, 4, 4), to_play=0):
        self.board = int(board)
        self.pieces = np.array(pieces, dtype=np.int8)
        self.to_play = int(to_play)

    def copy(self):
        return Game(self.board, self.pieces.copy(), self.to_play)

    def _p(self):
        return self.pieces.ctypes.data_as(C.POINTER(C.c_int8))

    def legal_moves(self):
        """-> (list of 96 bools, is_lines)"""
        m = (C.c_uint32 * 3)()
        lines = lib().co_legal_moves(C.c_uint64(self.board), self._p(), self.to_play, m)
        mask = int(m[0]) | (int(m[1]) << 32) | (int(m[2]) << 64)
        return [bool(mask >> i & 1) for i in range(96)], bool(lines)

    def legal_mask(self):
        m = (C.c_uint32 * 3)()
        lines = lib().co_legal_moves(C.c_uint64(self.board), self._p(), self.to_play, m)
        return int(m[0]) | (int(m[1]) << 32) | (int(m[2]) << 64), bool(lines)

    def do_move(self, move_id):
        b = C.c_uint64(self.board)
        tp = C.c_int(self.to_play)
        lib().co_do_move(C.byref(b), self._p(), C.byref(tp), int(move_id))
        self.board = int(b.value)
        self.to_play = int(tp.value)

    def state(self):
        out = np.zeros(GAME_STATE_SIZE, dtype=np.float32)
        lib().co_write_game_state(C.c_uint64(self.board), self._p(), self.to_play, _f32(out))
        return out

    def terminal_result(self):
        return lib().co_terminal_result(C.c_uint64(self.board), self._p(), self.to_play)

    # reference Game(int board[64], to_play, pieces[6]) ctor (game.cpp:13-26)
    @staticmethod
    def from_arrays(board64, to_play, pieces):
        b = 0
        for i, v in enumerate(board64):
            if v:
                b |= 1 << i
        return Game(b, pieces, to_play)


def encode_place(row, col, piece):
    return lib().co_encode_place(row, col, piece)


def encode_move(r0, c0, r1, c1):
    return lib().co_encode_move(r0, c0, r1, c1)


def decode_move(move_id):
    out = (C.c_int * 6)()
    lib().co_decode_move(move_id, out)
    return tuple(out)


def move_str(move_id):
    """Reference print format (move.cpp:56-78): 'Ba4', 'a4R' ..."""
    is_place, piece, r0, c0, r1, c1 = decode_move(move_id)
    col = "abcd"
    if is_place:
        return "BCA"[piece] + col[c1] + str(4 - r1)
    s = col[c0] + str(4 - r0)
    if c1 < c0:
        return s + "L"
    if c1 > c0:
        return s + "R"
    if r1 < r0:
        return s + "U"
    return s + "D"


class MT19937:
    def __init__(self, seed):
        self._g = lib().co_mt_create(C.c_uint32(seed & 0xFFFFFFFF))

    def __call__(self):
        return int(lib().co_mt_next(self._g))

    def __del__(self):
        try:
            lib().co_mt_destroy(self._g)
        except Exception:
            pass


# ---------------------------------------------------------------- Trainer
class Trainer:
    """Same surface as the reference Trainer (trainer.h:22-53, main.pyx:17-38)."""

    def __init__(self, num_games, log_folder="", seed=0, max_searches=1600, searches_per_eval=16, c_puct=1.0,
                 epsilon=0.25, num_logged=0, num_threads=1, testing=False, *, game_base=0, total_games=0):
        """game_base / total_games: games [game_base, game_base + num_games) of a Trainer of total_games
        games (co_trainer_create_slice), for replaying one shard of a sharded generation"""
        self.num_games = num_games
        self.searches_per_eval = searches_per_eval
        self._t = lib().co_trainer_create_slice(total_games or num_games, game_base, num_games, seed, max_searches,
                                                searches_per_eval, c_puct, epsilon, num_threads, int(bool(testing)))
        if not self._t:
            raise ValueError("bad Trainer arguments")
        if num_logged:  # the reference's constructor arguments: the first num_logged games write <log_folder>/game_<i>.txt
            self.set_logging(log_folder, min(num_logged, num_games))

    def __del__(self):
        try:
            if self._t:
                lib().co_trainer_destroy(self._t)
                self._t = None
        except Exception:
            pass

    def set_logging(self, log_folder, num_logged):
        """per-game text logs of the first num_logged games (trainer.cpp:243-250); -> files opened"""
        n = lib().co_trainer_set_logging(self._t, str(log_folder).encode(), int(num_logged))
        if n < 0:
            raise ValueError("set_logging: before the first iteration, 0 <= num_logged <= num_games")
        return n

    def set_stagger(self, on):
        lib().co_trainer_set_stagger(self._t, int(on))

    def num_requests(self, to_play=-1):
        return lib().co_trainer_num_requests(self._t, to_play)

    def num_samples(self):
        return lib().co_trainer_num_samples(self._t)

    def score(self):
        return float(lib().co_trainer_score(self._t))

    def avg_mate_length(self):
        return float(lib().co_trainer_avg_mate_length(self._t))

    def writeRequests(self, game_states, to_play=-1):
        lib().co_trainer_write_requests(self._t, _f32(game_states), to_play)

    def writeScores(self, filename):
        if not lib().co_trainer_write_scores(self._t, filename.encode() if isinstance(filename, str) else filename):
            raise OSError("cannot open %r" % filename)

    def writeSamples(self, game_states, eval_samples, prob_samples):
        lib().co_trainer_write_samples(self._t, _f32(game_states), _f32(eval_samples), _f32(prob_samples))

    def doIteration(self, evaluations, probabilities, to_play=-1):
        return bool(lib().co_trainer_do_iteration(self._t, _f32(evaluations), _f32(probabilities), to_play))

    # introspection
    def game_result(self, g):
        return lib().co_trainer_game_result(self._t, g)

    def game_to_play(self, g):
        return lib().co_trainer_game_to_play(self._t, g)

    def game_num_requests(self, g):
        return lib().co_trainer_game_num_requests(self._t, g)

    def game_num_samples(self, g):
        return lib().co_trainer_game_num_samples(self._t, g)

    def game_done(self, g):
        return bool(lib().co_trainer_game_done(self._t, g))

    def enable_trace(self, on=True):
        lib().co_trainer_enable_trace(self._t, int(on))

    def trace(self, g):
        n = lib().co_trainer_trace(self._t, g, None, 0)
        out = np.zeros(max(n, 1), dtype=np.int32)
        lib().co_trainer_trace(self._t, g, out.ctypes.data_as(C.POINTER(C.c_int32)), n)
        return out[:n]

    def counters(self):
        out = (C.c_int64 * 4)()
        lib().co_trainer_counters(self._t, out)
        return {"searches": out[0], "leaf_evals": out[1], "nodes_created": out[2], "plies": out[3]}


class DockerMC:
    """Same surface as the reference DockerMC (dockermc.h:13-51, docker/choose_move.pyx:21-42)."""

    def __init__(self, seed, max_searches, searches_per_eval, c_puct, epsilon, board, to_play, pieces):
        b = np.ascontiguousarray(board, dtype=np.int32)
        p = np.ascontiguousarray(pieces, dtype=np.int32)
        assert b.size == 64 and p.size == 6
        i32p = C.POINTER(C.c_int32)
        self.searches_per_eval = searches_per_eval
        self._d = lib().co_dockermc_create(seed, max_searches, searches_per_eval, c_puct, epsilon, b.ctypes.data_as(i32p),
                                           to_play, p.ctypes.data_as(i32p))
        if not self._d:
            raise ValueError("bad DockerMC arguments")

    def __del__(self):
        try:
            if self._d:
                lib().co_dockermc_destroy(self._d)
                self._d = None
        except Exception:
            pass

    def eval(self):
        return float(lib().co_dockermc_eval(self._d))

    def num_requests(self):
        return lib().co_dockermc_num_requests(self._d)

    def num_nodes(self):
        return lib().co_dockermc_num_nodes(self._d)

    def done(self):
        return bool(lib().co_dockermc_done(self._d))

    def drawn(self):
        return bool(lib().co_dockermc_drawn(self._d))

    def writeRequests(self, game_states):
        lib().co_dockermc_write_requests(self._d, _f32(game_states))

    def getLegalMoves(self):
        out = np.zeros(96, np.int32)
        lib().co_dockermc_get_legal_moves(self._d, out.ctypes.data_as(C.POINTER(C.c_int32)))
        return out

    def chooseMove(self):
        return lib().co_dockermc_choose_move(self._d)

    def doIteration(self, evaluations=None, probabilities=None):
        e = _f32(evaluations) if evaluations is not None else None
        p = _f32(probabilities) if probabilities is not None else None
        return bool(lib().co_dockermc_do_iteration(self._d, e, p))


class Tourney:
    """The reference's Tourney surface (tourney.h:13-46 / rating/tourney.pyx:15-31) on the oracle."""

    def __init__(self, num_threads=1, log_folder="", trace=False):
        self._log_folder = str(log_folder)
        self._t = lib().co_tourney_create(num_threads)
        if trace:
            lib().co_tourney_enable_trace(self._t, 1)

    def __del__(self):
        if getattr(self, "_t", None):
            lib().co_tourney_destroy(self._t)
            self._t = None

    def addPlayer(self, player_id, model_id, max_searches=1600, searches_per_eval=16, c_puct=1.0, epsilon=0.25,
                  random=False):
        if lib().co_tourney_add_player(self._t, player_id, model_id, max_searches, searches_per_eval, c_puct, epsilon,
                                       int(bool(random))) != 0:
            raise ValueError("addPlayer")

    def addMatch(self, player1, player2, logging=False):
        # logging: the match writes <log_folder>/match_<player1>_<player2>_<index>.txt (tourney.cpp:83-96)
        if lib().co_tourney_add_match_logged(self._t, player1, player2, self._log_folder.encode() if logging else None) < 0:
            raise ValueError("addMatch: unknown player")

    def all_done(self):
        return bool(lib().co_tourney_all_done(self._t))

    def num_requests(self, id):
        return lib().co_tourney_num_requests(self._t, id)

    def writeRequests(self, game_states, id):
        lib().co_tourney_write_requests(self._t, _f32(game_states), id)

    def doIteration(self, evaluations, probabilities, id):
        lib().co_tourney_do_iteration(self._t, _f32(evaluations), _f32(probabilities), id)

    def set_exact_offsets(self, on=True):
        """diagnostic, not in the reference: matches read their own rows (see corintho_oracle.c)"""
        lib().co_tourney_set_exact_offsets(self._t, int(bool(on)))

    def writeScores(self, filename):
        if lib().co_tourney_write_scores(self._t, str(filename).encode()) != 0:
            raise OSError("cannot write " + str(filename))

    def num_matches(self):
        return lib().co_tourney_num_matches(self._t)

    def match_info(self, i):
        L = lib()
        return {"done": L.co_tourney_match_done(self._t, i), "result": L.co_tourney_match_result(self._t, i),
                "to_play": L.co_tourney_match_to_play(self._t, i), "n_pending": L.co_tourney_match_num_requests(self._t, i)}

    def match_score(self, i):
        return lib().co_tourney_match_score(self._t, i)

    def trace(self, i):
        n = lib().co_tourney_trace(self._t, i, None, 0)
        out = np.zeros(max(n, 1), dtype=np.int32)
        lib().co_tourney_trace(self._t, i, out.ctypes.data_as(C.POINTER(C.c_int32)), n)
        return out[:n]

    def counters(self):
        out = (C.c_int64 * 4)()
        lib().co_tourney_counters(self._t, out)
        return {"searches": out[0], "leaf_evals": out[1], "nodes_created": out[2], "plies": out[3]}


def uniform_below(gen, n):
    """std::uniform_int_distribution<int32_t>(0, n - 1)(gen) as restated in the oracle"""
    return int(lib().co_uniform_below(gen._g, n))
