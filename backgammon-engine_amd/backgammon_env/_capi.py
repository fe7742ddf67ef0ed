"""ctypes binding of libbgamd.so (include/bgamd.h).

There is NO CPU fallback: importing this module without the built HIP library, or creating an
env without a gfx950 device, raises.  The library is built in-tree by
`__graft_entry__.build()` (hipcc --offload-arch=gfx950).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BGAMD_LIB: a differently-flagged build of the SAME sources (tools/ab_build.sh, A/B measurements); the digest check applies to it too
LIB_PATH = os.environ.get("BGAMD_LIB") or os.path.join(os.path.dirname(_HERE), "libbgamd.so")

OK = 0
ROLL, AUTO_RESET, NO_FLIP, WANT_INDEX, ONLY_P1, ONLY_P2, WEIGHTS_SLOT1 = 1, 2, 4, 8, 16, 32, 64
F32, BF16, F16X2, F32_DENSE = 0, 1, 2, 3

# every symbol include/bgamd.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("bgamd_version", C.c_int, []),
    ("bgamd_source_hash", C.c_char_p, []),
    ("bgamd_error_string", C.c_char_p, [C.c_int]),
    ("bgamd_last_hip_error", C.c_char_p, []),
    ("bgamd_device_count", C.c_int, []),
    ("bgamd_env_create", C.c_int, [C.POINTER(_P), C.c_int64, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int64]),
    ("bgamd_env_destroy", C.c_int, [_P]),
    ("bgamd_env_num_games", C.c_int64, [_P]),
    ("bgamd_env_reset", C.c_int, [_P, _P]),
    ("bgamd_env_reset_episode", C.c_int, [_P, C.c_uint32, _P]),
    ("bgamd_env_reseed", C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_uint64, _P]),
    ("bgamd_env_reset_lanes", C.c_int, [_P, _P, _P]),
    ("bgamd_env_set_states", C.c_int, [_P, _P, _P, _P]),
    ("bgamd_env_get_states", C.c_int, [_P, _P, _P, _P]),
    ("bgamd_env_get_flags", C.c_int, [_P, _P, _P]),
    ("bgamd_env_snapshot", C.c_int, [_P, _P, _P]),
    ("bgamd_game_snapshot", C.c_int, [_P, C.POINTER(C.c_int32)]),
    ("bgamd_game_set_state", C.c_int, [_P, C.POINTER(C.c_int32), C.c_int]),
    ("bgamd_game_set_dice", C.c_int, [_P, C.c_int, C.c_int]),
    ("bgamd_game_roll", C.c_int, [_P, C.POINTER(C.c_int32)]),
    ("bgamd_game_legal_moves", C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int8)]),
    ("bgamd_game_try_move", C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("bgamd_game_enumerate", C.c_int64, [_P, C.c_int, C.c_int, C.c_int, _P, _P, _P, C.c_int64]),
    ("bgamd_env_set_dice", C.c_int, [_P, _P, _P]),
    ("bgamd_env_get_dice", C.c_int, [_P, _P, _P]),
    ("bgamd_env_roll", C.c_int, [_P, C.c_int, _P]),
    ("bgamd_env_enumerate", C.c_int, [_P, _P, _P, _P]),
    ("bgamd_env_candidates_info", C.c_int64, [_P, _P, _P, _P]),
    ("bgamd_env_candidates_read", C.c_int, [_P, C.c_int64, C.c_int64, _P, _P, _P, _P]),
    ("bgamd_env_step_random", C.c_int, [_P, C.c_int, _P, _P]),
    ("bgamd_env_step_random_walk", C.c_int, [_P, C.c_int, _P, _P]),
    ("bgamd_env_load_weights", C.c_int, [_P, _P]),
    ("bgamd_env_load_weights_slot", C.c_int, [_P, C.c_int, _P]),
    ("bgamd_weights_check", C.c_int, [_P]),
    ("bgamd_env_step_greedy", C.c_int, [_P, C.c_int, C.c_float, C.c_int, _P]),
    ("bgamd_env_run_greedy", C.c_int, [_P, C.c_int, C.c_float, C.c_int, C.c_int64, _P]),
    ("bgamd_env_last_choice", C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    ("bgamd_env_stats", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("bgamd_env_reset_stats", C.c_int, [_P, _P]),
    ("bgamd_env_try_move", C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    ("bgamd_env_legal_moves", C.c_int, [_P, _P, _P, _P, _P, _P]),
    ("bgamd_env_unique_rows_info", C.c_int64, [_P, _P, C.c_int64, _P]),
    ("bgamd_env_unique_rows_read", C.c_int, [_P, C.c_int64, C.c_int64, _P, _P, _P]),
    ("bgamd_env_set_trajectory", C.c_int, [_P, _P, C.c_int64]),
    ("bgamd_env_get_progress", C.c_int, [_P, _P, _P, _P]),
    ("bgamd_env_set_trajectory_ring", C.c_int, [_P, _P, C.c_int64, _P]),
    ("bgamd_env_trajectory_step", C.c_int64, [_P]),
    ("bgamd_encode_rows", C.c_int, [_P, C.c_int64, _P, _P]),
    ("bgamd_encode", C.c_int, [_P, _P, C.c_int64, _P, _P]),
    ("bgamd_evaluate", C.c_int, [_P, _P, _P, C.c_int64, C.c_int, _P, _P]),
    ("bgamd_evaluate_slot", C.c_int, [_P, C.c_int, _P, _P, C.c_int64, C.c_int, _P, _P]),
    ("bgamd_evaluate_incremental", C.c_int, [_P, C.c_int, _P, _P, C.c_int64, _P, _P, C.c_int64, _P, _P]),
    ("bgamd_build_flags", C.c_char_p, []),
    ("bgamd_env_kernel_choice", C.c_int, [_P, C.POINTER(C.c_int32)]),
    ("bgamd_env_time_kernels", C.c_int, [_P, C.c_int]),
    ("bgamd_env_kernel_times", C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    ("bgamd_pack_rows", C.c_int, [_P, _P, C.c_int64, _P, _P]),
    ("bgamd_td_create", C.c_int, [C.POINTER(_P), C.c_int64, C.c_int]),
    ("bgamd_td_destroy", C.c_int, [_P]),
    ("bgamd_td_set_weights", C.c_int, [_P, _P, _P]),
    ("bgamd_td_get_weights", C.c_int, [_P, _P, _P]),
    ("bgamd_td_begin", C.c_int, [_P, _P, C.c_int64, C.c_int64, _P, C.c_int64, _P, _P, _P]),
    ("bgamd_td_stream_schedule", C.c_int, [_P, C.c_int64, C.c_int64, _P, _P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("bgamd_td_begin_stream", C.c_int, [_P, _P, C.c_int64, C.c_int64, _P, _P, C.c_int64, _P, _P, _P]),
    ("bgamd_td_begin_stream_games", C.c_int, [_P, _P, C.c_int64, C.c_int64, _P, _P, C.c_int64, C.c_int64, _P, _P, _P, _P, _P]),
    ("bgamd_td_step", C.c_int, [_P, C.c_int64, C.c_int64, C.c_double, C.c_float, _P, _P]),
    ("bgamd_td_apply", C.c_int, [_P, _P, _P]),
    ("bgamd_td_replay", C.c_int, [_P, C.c_int64, C.POINTER(C.c_int64), C.c_double, C.c_float, _P]),
    ("bgamd_td_set_delay", C.c_int, [_P, C.c_int]),
    ("bgamd_td_comm_unique_id", C.c_int, [_P]),
    ("bgamd_td_comm_init", C.c_int, [_P, _P, C.c_int, C.c_int]),
    ("bgamd_td_comm_destroy", C.c_int, [_P]),
    ("bgamd_td_step_allreduce", C.c_int, [_P, C.c_int64, C.c_int64, C.c_double, C.c_float, _P]),
    ("bgamd_td_replay_allreduce", C.c_int, [_P, C.c_int64, C.POINTER(C.c_int64), C.c_int64, C.c_double, C.c_float, _P]),
    ("bgamd_td_stats", C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    ("bgamd_td_active_columns", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("bgamd_td_written_columns", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("bgamd_td_slots", C.c_int, [_P, _P]),
    ("bgamd_td_time", C.c_int, [_P, C.c_int]),
    ("bgamd_td_times", C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
]


class BgamdError(RuntimeError):
    pass


_lib = None


def load():
    """dlopen libbgamd.so and type every entry point.  Raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BgamdError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the ABI drifted
        fn.restype = res
        fn.argtypes = args
    # the library must be the one built from the sources next to it (a stale .so travels with gpurun snapshots)
    from . import _srchash
    have = lib.bgamd_source_hash().decode()
    if os.path.isdir(os.path.join(_srchash.PKG, "csrc")) and os.environ.get("BGAMD_ALLOW_STALE") != "1":
        want = _srchash.source_hash()
        if have != want:
            raise BgamdError(f"{LIB_PATH} was built from other sources (library {have}, sources {want}): "
                             "run `python -c 'import __graft_entry__ as g; g.build()'`")
    _lib = lib
    return lib


def source_hash() -> str:
    return load().bgamd_source_hash().decode()


def check(rc: int, what: str = ""):
    if rc >= 0:
        return rc
    lib = load()
    msg = lib.bgamd_error_string(int(rc)).decode()
    if rc == -2:
        msg += ": " + lib.bgamd_last_hip_error().decode()
    raise BgamdError(f"{what or 'bgamd'} failed ({rc}): {msg}")
