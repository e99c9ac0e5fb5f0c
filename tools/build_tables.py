#!/usr/bin/env python3
"""Build mortal_amd/data/mjtables.xz from the reference's lookup-table data.

Runs ONLY in the build container (needs /root/reference).  The three tables are
third-party *data* (tomohxx shanten tables, Yamaoka agari index) that the
reference embeds as `libriichi/src/algo/data/*.gz`
(shanten.rs:11-44, agari.rs:24-51).  We re-lay them out for our own loaders:

  header  : magic 'MJT1', u32 n_suhai, u32 n_jihai, u32 n_agari
  suhai   : n_suhai  x 5 bytes  (10 nibbles, low nibble first)
  jihai   : n_jihai  x 5 bytes
  agari   : n_agari  x 24 bytes, sorted by key:
            u32 key, u32 n_div, u32 div[4]   (unused div = 0)

and xz-compress the lot.  The file travels with the repo (the GPU box has no
/root/reference).  sha256 of the decompressed payload is printed; the payload is exercised by the reference's
shanten / agari KATs (tests/test_oracle_kats.py, through the oracle's copy of the tables) and by every lock-step test.
"""
import gzip, hashlib, lzma, struct, sys, os

REF = "/root/reference/libriichi/src/algo/data"
OUT = os.path.join(os.path.dirname(__file__), "..", "mortal_amd", "data", "mjtables.xz")

def main():
    suhai = gzip.open(f"{REF}/shanten_suhai.bin.gz").read()
    jihai = gzip.open(f"{REF}/shanten_jihai.bin.gz").read()
    agari = gzip.open(f"{REF}/agari.bin.gz").read()
    assert len(suhai) == 1_940_777 * 5 and len(jihai) == 78_032 * 5
    recs = []
    off = 0
    for _ in range(9_362):
        key, n = struct.unpack_from("<IB", agari, off); off += 5
        divs = list(struct.unpack_from(f"<{n}I", agari, off)); off += 4 * n
        assert 1 <= n <= 4
        recs.append((key, n, divs + [0] * (4 - n)))
    assert off == len(agari)
    recs.sort()
    assert len({r[0] for r in recs}) == len(recs)
    payload = bytearray(b"MJT1")
    payload += struct.pack("<III", len(suhai) // 5, len(jihai) // 5, len(recs))
    payload += suhai + jihai
    for key, n, divs in recs:
        payload += struct.pack("<II4I", key, n, *divs)
    print("sha256", hashlib.sha256(payload).hexdigest(), "bytes", len(payload))
    with open(OUT, "wb") as f:
        f.write(lzma.compress(bytes(payload), preset=9 | lzma.PRESET_EXTREME))
    print("wrote", OUT, os.path.getsize(OUT))

if __name__ == "__main__":
    main()
