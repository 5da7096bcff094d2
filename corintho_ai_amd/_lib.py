"""Loader of libcorintho_hip.so (the HIP engine, C ABI in include/corintho_hip.h).

The product has no CPU fallback: if the library is missing or no gfx950 device
is visible, loading / creating a Trainer raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcorintho_hip.so")

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)


class CaConfig(C.Structure):
    """ca_config (include/corintho_hip.h)"""

    _fields_ = [
        ("num_games", C.c_int32),
        ("seed", C.c_int32),
        ("max_searches", C.c_int32),
        ("searches_per_eval", C.c_int32),
        ("c_puct", C.c_float),
        ("epsilon", C.c_float),
        ("num_logged", C.c_int32),
        ("num_threads", C.c_int32),
        ("testing", C.c_int32),
        ("device", C.c_int32),
        ("no_stagger", C.c_int32),
        ("arena_units", C.c_uint32),
        ("trace", C.c_int32),
        ("game_base", C.c_int32),
        ("total_games", C.c_int32),
        ("pools", C.c_int32),
        ("analyse", C.c_int32),
        ("resident", C.c_int32),
        ("eval_cache", C.c_int32),
        ("step_budget", C.c_int32),
    ]


class CaStats(C.Structure):
    _fields_ = [
        ("searches", C.c_int64),
        ("evals", C.c_int64),
        ("nodes", C.c_int64),
        ("plies", C.c_int64),
        ("iterations", C.c_int64),
        ("peak_arena_units", C.c_int64),
        ("mcts_ms", C.c_double),
        ("nn_ms", C.c_double),
        ("pack_ms", C.c_double),
        ("mcts_launches", C.c_int64),
        ("nn_launches", C.c_int64),
        ("nn_rows", C.c_int64),
        ("pools", C.c_int64),
        ("timed_launches", C.c_int64),
        ("nn_timed_rows", C.c_int64),
        ("mcts_timed_ms", C.c_double),
        ("nn_timed_ms", C.c_double),
        ("resident_slots", C.c_int64),
        ("nn_rows_evaluated", C.c_int64),
        ("steps_cut", C.c_int64),
        ("step_budget_last", C.c_int64),
    ]


# every symbol include/corintho_hip.h declares
EXPORTS = [
    "ca_last_error", "ca_device_check", "ca_trainer_create", "ca_trainer_destroy", "ca_trainer_num_requests",
    "ca_trainer_num_samples", "ca_trainer_score", "ca_trainer_avg_mate_length", "ca_trainer_write_requests",
    "ca_trainer_write_samples", "ca_trainer_write_scores", "ca_trainer_do_iteration", "ca_trainer_set_net",
    "ca_trainer_run", "ca_trainer_net_forward", "ca_trainer_net_bench", "ca_trainer_export_samples", "ca_trainer_pack_samples_device", "ca_trainer_pin_host", "ca_trainer_unpin_host", "ca_trainer_set_positions", "ca_trainer_analysis", "ca_trainer_finish", "ca_trainer_set_logging", "ca_trainer_reset", "ca_expand_samples", "ca_trainer_stats",
    "ca_trainer_game_info", "ca_trainer_trace", "ca_trainer_prof", "ca_rules_legal_moves", "ca_rules_do_move", "ca_rules_rows", "ca_rng_draw",
    "ca_fp_probe",
    "ca_tourney_create", "ca_tourney_destroy", "ca_tourney_set_log_folder", "ca_tourney_add_player", "ca_tourney_add_match", "ca_tourney_all_done",
    "ca_tourney_num_requests", "ca_tourney_write_requests", "ca_tourney_do_iteration", "ca_tourney_write_scores",
    "ca_tourney_set_net", "ca_tourney_run", "ca_tourney_set_exact_offsets", "ca_tourney_num_matches", "ca_tourney_match_info", "ca_tourney_match_score", "ca_tourney_trace", "ca_tourney_stats",
]


def declare(L):
    vp = C.c_void_p
    L.ca_last_error.restype = C.c_char_p
    L.ca_device_check.argtypes = [C.c_int]
    L.ca_trainer_create.argtypes = [C.POINTER(CaConfig), C.POINTER(vp)]
    L.ca_trainer_destroy.argtypes = [vp]
    L.ca_trainer_destroy.restype = None
    L.ca_trainer_num_requests.argtypes = [vp, C.c_int, i32p]
    L.ca_trainer_num_samples.argtypes = [vp, i32p]
    L.ca_trainer_score.argtypes = [vp, f32p]
    L.ca_trainer_avg_mate_length.argtypes = [vp, f32p]
    L.ca_trainer_write_requests.argtypes = [vp, f32p, C.c_int]
    L.ca_trainer_write_samples.argtypes = [vp, f32p, f32p, f32p]
    L.ca_trainer_write_scores.argtypes = [vp, C.c_char_p]
    L.ca_trainer_do_iteration.argtypes = [vp, f32p, f32p, C.c_int, i32p]
    L.ca_trainer_set_net.argtypes = [vp, C.c_int, C.c_int, f32p, C.c_size_t]
    L.ca_trainer_run.argtypes = [vp, C.c_int64, i32p]
    L.ca_trainer_net_forward.argtypes = [vp, C.c_int, f32p, C.c_int32, f32p, f32p]
    L.ca_trainer_net_bench.argtypes = [vp, C.c_int, f32p, C.c_int32, C.c_int32, f32p]
    L.ca_trainer_export_samples.argtypes = [vp, f32p, f32p]
    L.ca_trainer_pack_samples_device.argtypes = [vp, C.c_void_p, C.c_void_p, C.c_int32, i32p]
    L.ca_trainer_pin_host.argtypes = [vp, C.c_void_p, C.c_size_t, i32p]
    L.ca_trainer_unpin_host.argtypes = [vp, C.c_void_p]
    L.ca_trainer_set_positions.argtypes = [vp, i32p, i32p, i32p, i32p]
    L.ca_trainer_analysis.argtypes = [vp, i32p]
    L.ca_trainer_finish.argtypes = [vp]
    L.ca_trainer_set_logging.argtypes = [vp, C.c_char_p, C.c_int32]
    L.ca_trainer_reset.argtypes = [vp, C.c_int32]
    L.ca_expand_samples.argtypes = [C.c_int, f32p, f32p, C.c_int32, f32p, f32p, f32p]
    L.ca_trainer_stats.argtypes = [vp, C.POINTER(CaStats)]
    L.ca_trainer_game_info.argtypes = [vp, C.c_int, i32p]
    L.ca_trainer_trace.argtypes = [vp, C.c_int, i32p, C.c_int32, i32p]
    L.ca_trainer_prof.argtypes = [vp, C.POINTER(C.c_ulonglong)]
    L.ca_rules_legal_moves.argtypes = [C.c_int, u64p, u32p, C.c_int32, u32p, i32p]
    L.ca_rules_do_move.argtypes = [C.c_int, u64p, u32p, i32p, C.c_int32, f32p]
    L.ca_rules_rows.argtypes = [C.c_int, u64p, u32p, i32p, C.c_int32, u32p]
    L.ca_rng_draw.argtypes = [C.c_int, C.c_uint32, C.c_int32, C.c_int32, u32p]
    L.ca_fp_probe.argtypes = [C.c_int, f32p, C.c_int32, f32p]
    L.ca_tourney_create.argtypes = [C.c_int, C.c_uint32, C.c_int, C.POINTER(vp)]
    L.ca_tourney_set_log_folder.argtypes = [vp, C.c_char_p]
    L.ca_tourney_destroy.argtypes = [vp]
    L.ca_tourney_destroy.restype = None
    L.ca_tourney_add_player.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32]
    L.ca_tourney_add_match.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32]
    L.ca_tourney_all_done.argtypes = [vp, i32p]
    L.ca_tourney_num_requests.argtypes = [vp, C.c_int32, i32p]
    L.ca_tourney_write_requests.argtypes = [vp, f32p, C.c_int32]
    L.ca_tourney_do_iteration.argtypes = [vp, f32p, f32p, C.c_int32, C.c_int32]
    L.ca_tourney_write_scores.argtypes = [vp, C.c_char_p]
    L.ca_tourney_set_net.argtypes = [vp, C.c_int32, C.c_int32, f32p, C.c_size_t]
    L.ca_tourney_run.argtypes = [vp, C.c_int64, i32p]
    L.ca_tourney_set_exact_offsets.argtypes = [vp, C.c_int32]
    L.ca_tourney_num_matches.argtypes = [vp, i32p]
    L.ca_tourney_match_info.argtypes = [vp, C.c_int32, i32p]
    L.ca_tourney_match_score.argtypes = [vp, C.c_int32, f32p]
    L.ca_tourney_trace.argtypes = [vp, C.c_int32, i32p, C.c_int32, i32p]
    L.ca_tourney_stats.argtypes = [vp, C.POINTER(CaStats)]
    for name in EXPORTS:
        fn = getattr(L, name)
        if name not in ("ca_last_error", "ca_trainer_destroy", "ca_tourney_destroy"):
            fn.restype = C.c_int
    return L


_lib = None


def load():
    """The HIP engine.  Raises ImportError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "corintho_ai_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH
            )
        _lib = declare(C.CDLL(LIB_PATH))
    return _lib


class EngineError(RuntimeError):
    pass


def check(L, rc):
    if rc != 0:
        msg = L.ca_last_error()
        raise EngineError("corintho_hip error %d: %s" % (rc, msg.decode() if msg else "?"))
