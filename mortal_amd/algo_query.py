"""Debug / test access to the device code's pure rule functions through `mj_algo_query` (include/mortal_amd.h): shanten,
agari (search_yakus / has_yaku / agari + point), ankan-after-riichi, Point::calc — one device thread per query.  This is the
hook the reference's known-answer tests (algo/shanten.rs:158-201, algo/agari.rs:920-1379, algo/point.rs:121-153) reach the
HIP code through (tests/test_gpu_kats.py); the product path never calls it."""
import ctypes as C

import numpy as np

QUERY_DTYPE = np.dtype([
    ("tehai", np.uint8, 34), ("chis", np.uint8, 4), ("pons", np.uint8, 4), ("minkans", np.uint8, 4), ("ankans", np.uint8, 4),
    ("n_chis", np.uint8), ("n_pons", np.uint8), ("n_minkans", np.uint8), ("n_ankans", np.uint8),
    ("len_div3", np.uint8), ("is_menzen", np.uint8), ("bakaze", np.uint8), ("jikaze", np.uint8), ("winning_tile", np.uint8),
    ("is_ron", np.uint8), ("additional_hans", np.uint8), ("doras", np.uint8),
    ("op", np.uint8), ("arg0", np.uint8), ("arg1", np.uint8), ("arg2", np.uint8), ("pad", np.uint8, 6),
])
RESULT_DTYPE = np.dtype([(k, np.int32) for k in ("r0", "r1", "r2", "r3", "p0", "p1", "p2", "p3")])
assert QUERY_DTYPE.itemsize == 72 and RESULT_DTYPE.itemsize == 32

OP_SHANTEN, OP_SEARCH_YAKUS, OP_HAS_YAKU, OP_AGARI, OP_ANKAN_AFTER_RIICHI, OP_POINT, OP_DEAL_DIVMOD = range(7)


def queries(n):
    q = np.zeros(n, dtype=QUERY_DTYPE)
    q["bakaze"] = 27
    q["jikaze"] = 27
    q["len_div3"] = 4
    q["is_menzen"] = 1
    return q


def set_melds(q, i, chis=(), pons=(), minkans=(), ankans=()):
    for name, m in (("chis", chis), ("pons", pons), ("minkans", minkans), ("ankans", ankans)):
        q[name][i, :len(m)] = list(m)
        q["n_" + name][i] = len(m)
    q["is_menzen"][i] = int(not (chis or pons or minkans))


def run(q, lib=None):
    """Evaluate the queries on the device; returns the RESULT_DTYPE array."""
    if lib is None:
        from . import _lib, pool

        pool._ensure_tables()
        lib, check = _lib.lib, _lib.check
    else:
        def check(rc):
            if rc < 0:
                raise RuntimeError(lib.mj_last_error().decode())
    q = np.ascontiguousarray(q)
    out = np.zeros(len(q), dtype=RESULT_DTYPE)
    check(lib.mj_algo_query(q.ctypes.data_as(C.c_void_p), len(q), out.ctypes.data_as(C.c_void_p), None))
    return out
