"""Lookup-table payload loader.

The shanten tables (reference `algo/shanten.rs:11-44`) and the agari index
(`algo/agari.rs:24-51`) ship as one xz blob, `mortal_amd/data/mjtables.xz`,
re-laid-out by `tools/build_tables.py`:

    'MJT1' | u32 n_suhai | u32 n_jihai | u32 n_agari
    suhai  : n_suhai x 5 B   (10 nibbles per row, low nibble first)
    jihai  : n_jihai x 5 B
    agari  : n_agari x 24 B  (u32 key, u32 n_div, u32 div[4]), sorted by key
"""
import functools
import hashlib
import lzma
import os
import struct

import numpy as np

_PATH = os.path.join(os.path.dirname(__file__), "data", "mjtables.xz")
PAYLOAD_SHA256 = "4fa3d123ce5f06fe725d37150aafd3b7bb42e4983e2d5c1edb0737e8083f113a"


@functools.lru_cache(maxsize=1)
def payload() -> bytes:
    with open(_PATH, "rb") as f:
        data = lzma.decompress(f.read())
    if hashlib.sha256(data).hexdigest() != PAYLOAD_SHA256:
        raise RuntimeError("mjtables.xz is corrupt (sha256 mismatch)")
    return data


@functools.lru_cache(maxsize=1)
def arrays():
    """Returns dict(suhai=u64[n], jihai=u64[n], agari_keys=u32[n], agari_divs=u32[n,5]).

    Device layout: one u64 per shanten row (10 nibbles in the low 40 bits) so a
    lookup is a single 8-byte load; agari as a sorted key array + (n_div, div[4]).
    """
    p = payload()
    assert p[:4] == b"MJT1"
    ns, nj, na = struct.unpack_from("<III", p, 4)
    off = 16

    def rows(n):
        nonlocal off
        raw = np.frombuffer(p, dtype=np.uint8, count=n * 5, offset=off).reshape(n, 5).astype(np.uint64)
        off += n * 5
        out = np.zeros(n, dtype=np.uint64)
        for i in range(5):
            out |= raw[:, i] << np.uint64(8 * i)
        return out

    suhai = rows(ns)
    jihai = rows(nj)
    rec = np.frombuffer(p, dtype=np.uint32, count=na * 6, offset=off).reshape(na, 6)
    return dict(
        suhai=suhai,
        jihai=jihai,
        agari_keys=np.ascontiguousarray(rec[:, 0]),
        agari_divs=np.ascontiguousarray(rec[:, 1:6]),
    )
