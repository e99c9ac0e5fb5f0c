"""ctypes binding of the C-ABI in include/mortal_amd.h (libmortal_amd.so, built by __graft_entry__.build()).

There is no fallback: if the shared library is missing or fails to load, importing this module raises.
"""
import ctypes as C
import os

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MORTAL_AMD_LIB") or os.path.join(_DIR, "libmortal_amd.so")  # env override: A/B builds

# Every symbol include/mortal_amd.h declares (tests/test_host.py checks the library exports all of them).
SYMBOLS = [
    "mj_last_error", "mj_abi_version", "mj_tables_upload", "mj_pool_create", "mj_pool_destroy", "mj_pool_reset",
    "mj_pool_configure", "mj_pool_set_refill", "mj_pool_set_start_stagger", "mj_table_apply_event", "mj_table_mark_row", "mj_table_query", "mj_replay_load", "mj_replay_step", "mj_replay_meta", "mj_pool_enable_log", "mj_log_lengths", "mj_log_read", "mj_step", "mj_step_q", "mj_step_ev", "mj_rows_count", "mj_rows_dev", "mj_encode",
    "mj_encode_oracle", "mj_oracle_obs_rows", "mj_encode_timing", "mj_sp_timing", "mj_sp_phase_ticks", "mj_pool_set_sp_schedule", "mj_sp_schedule_stats", "mj_random_policy", "mj_greedy_policy", "mj_counters", "mj_results", "mj_pool_first_error", "mj_debug_table",
    "mj_debug_table_size", "mj_debug_layout", "mj_obs_rows", "mj_algo_query",
]


class MortalAmdError(RuntimeError):
    pass


def _load(path=None):
    """dlopen the C-ABI library and declare its prototypes.  `path` is for the test suite only (the host emulation of the
    same sources, tests/host); the product always binds LIB_PATH."""
    path = path or LIB_PATH
    # torch ships its own libamdhip64; it must be in the process before libmortal_amd.so is opened, otherwise the
    # system copy gets bound first and the two HIP runtimes disagree about the visible devices
    import torch  # noqa: F401


    if not os.path.exists(path):
        raise MortalAmdError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). The HIP extension is mandatory; there is no CPU fallback."
        )
    L = C.CDLL(path)
    vp, i32, u64 = C.c_void_p, C.c_int, C.c_uint64
    L.mj_last_error.restype = C.c_char_p
    L.mj_tables_upload.argtypes = [vp, C.c_size_t]
    L.mj_pool_create.restype = vp
    L.mj_pool_create.argtypes = [i32, i32, i32, i32]
    L.mj_pool_destroy.argtypes = [vp]
    L.mj_pool_reset.argtypes = [vp, vp, vp, vp, vp, i32]
    L.mj_pool_configure.argtypes = [vp, i32, i32, i32, i32]
    L.mj_pool_set_refill.argtypes = [vp, u64]
    L.mj_pool_set_start_stagger.argtypes = [vp, C.c_uint32, vp]
    L.mj_step.argtypes = [vp, vp, vp, vp]
    L.mj_step_q.argtypes = [vp, vp, vp, vp, vp, vp]
    L.mj_step_ev.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.mj_rows_count.argtypes = [vp, vp, vp]
    L.mj_rows_dev.restype = vp
    L.mj_rows_dev.argtypes = [vp, i32]
    L.mj_encode.argtypes = [vp, i32, vp, vp, vp]
    L.mj_encode_timing.argtypes = [vp, i32, vp, vp]
    L.mj_sp_timing.argtypes = [vp, vp, vp]
    L.mj_sp_phase_ticks.argtypes = [vp, vp, vp]
    L.mj_pool_set_sp_schedule.argtypes = [vp, i32, i32, i32, i32, i32]
    L.mj_sp_schedule_stats.argtypes = [vp, vp, vp]
    L.mj_table_apply_event.argtypes = [vp, i32, vp, i32, vp]
    L.mj_table_mark_row.argtypes = [vp, i32, i32, i32, vp]
    L.mj_table_query.argtypes = [vp, i32, i32, i32, vp, vp, vp]
    L.mj_replay_load.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp]
    L.mj_replay_step.argtypes = [vp, vp]
    L.mj_replay_meta.argtypes = [vp, vp, vp]
    L.mj_pool_enable_log.argtypes = [vp, C.c_uint32]
    L.mj_log_lengths.argtypes = [vp, vp, vp]
    L.mj_log_read.argtypes = [vp, i32, i32, vp, vp]
    L.mj_encode_oracle.argtypes = [vp, i32, vp, vp]
    L.mj_oracle_obs_rows.argtypes = [i32]
    L.mj_random_policy.argtypes = [vp, i32, vp, u64, u64, vp, vp]
    L.mj_greedy_policy.argtypes = [vp, i32, vp, vp, u64, u64, vp, vp]
    L.mj_counters.argtypes = [vp, vp, vp]
    L.mj_results.argtypes = [vp, vp, vp, vp]
    L.mj_pool_first_error.argtypes = [vp, vp, vp]
    L.mj_debug_table.argtypes = [vp, i32, vp, C.c_size_t, vp]
    L.mj_debug_table_size.restype = C.c_size_t
    L.mj_debug_layout.restype = C.c_char_p
    L.mj_obs_rows.argtypes = [i32]
    L.mj_algo_query.argtypes = [vp, i32, vp, vp]
    return L


lib = _load()


def check(rc):
    if rc is None or (isinstance(rc, int) and rc < 0):
        raise MortalAmdError(lib.mj_last_error().decode())
    return rc
