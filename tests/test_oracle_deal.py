"""Deal-from-seed under BOTH rand generations (arena/board.rs:99-109; Cargo.lock:1042-1043 pins rand 0.9.1).

The repository holds a vector only for the rand-0.8 shuffle (the seeded example log, tests/test_oracle_golden_log.py).
For the rand-0.9.1 shuffle this file pins the oracle (oracle/deal.cc) against a SECOND, independently written
implementation of the published algorithm (rand 0.9.1 `seq/slice.rs` partial_shuffle -> `seq/increasing_uniform.rs`
IncreasingUniform::next_index -> `distr/uniform_int.rs` sample_single_inclusive, Canon's method, u32) in a different
shape (chunk plan computed up front, pure-Python ChaCha12 + hashlib SHA3), plus the structural facts the algorithm
implies.  Still "parity unpinned" against the Rust crate itself (no rustc here), but no longer a single restatement.
"""
import hashlib
import struct

import numpy as np

MASK = 0xFFFFFFFF


def chacha12_words(seed32, n_words):
    """rand_chacha 0.9 ChaCha12Rng: 32-byte key, 64-bit block counter (words 12-13) from 0, stream 0, words in order."""
    key = struct.unpack("<8I", seed32)
    out = []
    counter = 0

    def rotl(x, n):
        return ((x << n) | (x >> (32 - n))) & MASK

    while len(out) < n_words:
        init = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574, *key, counter & MASK, counter >> 32, 0, 0]
        x = list(init)

        def qr(a, b, c, d):
            x[a] = (x[a] + x[b]) & MASK; x[d] = rotl(x[d] ^ x[a], 16)
            x[c] = (x[c] + x[d]) & MASK; x[b] = rotl(x[b] ^ x[c], 12)
            x[a] = (x[a] + x[b]) & MASK; x[d] = rotl(x[d] ^ x[a], 8)
            x[c] = (x[c] + x[d]) & MASK; x[b] = rotl(x[b] ^ x[c], 7)

        for _ in range(6):  # 6 double rounds = 12 rounds
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
        out += [(a + b) & MASK for a, b in zip(x, init)]
        counter += 1
    return out[:n_words]


def chunk_plan(length):
    """IncreasingUniform's chunks for a full shuffle: [(first divisor m, factors k, bound)], index 0 needs no chunk."""
    plan, m = [], 2
    while m <= length:
        prod, k = m, 1
        while prod * (m + k) <= MASK:  # u32::checked_mul
            prod *= m + k
            k += 1
        plan.append((m, k, prod))
        m += k
    return plan


def shuffle_rand09(seq, words):
    """for i in 0..len: swap(i, index_i), index_i uniform in [0, i]; indices come least-significant-first out of one
    Canon-sampled chunk per plan entry; the last index of a chunk is the remaining quotient."""
    seq = list(seq)
    it = iter(words)

    def canon(bound):  # sample_single_inclusive(0, bound - 1), u32, biased variant
        m = next(it) * bound
        hi, lo = m >> 32, m & MASK
        if lo > ((-bound) & MASK):
            new_hi = (next(it) * bound) >> 32
            hi += (lo + new_hi) >> 32  # carry of the 32-bit addition
        return hi

    idx = [0]
    for m, k, bound in chunk_plan(len(seq)):
        x = canon(bound)
        for j in range(k):
            if len(idx) == len(seq):
                break  # the tail of the last chunk is never consumed
            if j == k - 1:
                idx.append(x)
            else:
                idx.append(x % (m + j))
                x //= m + j
    for i, j in enumerate(idx):
        assert 0 <= j <= i
        seq[i], seq[j] = seq[j], seq[i]
    return seq


def shuffle_rand08(seq, words):
    """rand 0.8: for i in (1..len).rev(): swap(i, gen_range(0..=i)) with widening-multiply + zone rejection."""
    seq = list(seq)
    it = iter(words)
    for i in range(len(seq) - 1, 0, -1):
        rng_ = i + 1
        zone = ((rng_ << (32 - rng_.bit_length())) - 1) & MASK
        while True:
            m = next(it) * rng_
            if (m & MASK) <= zone:
                j = m >> 32
                break
        seq[i], seq[j] = seq[j], seq[i]
    return seq


def unshuffled():
    seq = [t for t in range(34) for _ in range(4)]
    seq[4 * 4], seq[13 * 4], seq[22 * 4] = 34, 35, 36  # board.rs:786-824: the first copy of each five is red
    return seq


def second_deal(nonce, key, kyoku, honba, algo):
    seed = hashlib.sha3_256(struct.pack("<QQBB", nonce, key, kyoku, honba)).digest()
    words = chacha12_words(seed, 400)
    return (shuffle_rand09 if algo == 1 else shuffle_rand08)(unshuffled(), words)


def test_chacha12_known_answer(oracle):
    """ChaCha12, all-zero 256-bit key and IV (Strombergson's ChaCha test vectors, TC1, 12 rounds): first block."""
    kat = bytes.fromhex("9bf49a6a0755f953811fce125f2683d50429c3bb49e074147e0089a52eae155f"
                        "0564f879d27ae3c02ce82834acfa8c793a629f2ca0de6919610be82f411326be")
    mine = struct.pack("<16I", *chacha12_words(bytes(32), 16))
    assert mine == kat
    seed = np.zeros(32, dtype=np.uint8)
    out = np.zeros(16, dtype=np.uint32)
    oracle.lib().mjo_chacha12(oracle.ptr(seed), 16, oracle.ptr(out))
    assert out.tobytes() == kat


def test_chunk_plan_is_the_published_one():
    plan = chunk_plan(136)
    assert plan[0] == (2, 11, 479001600)  # RESULT2 = inner(2): 2*3*...*12 = 12!, count 11 (increasing_uniform.rs)
    assert all(b <= MASK and b * (m + k) > MASK for m, k, b in plan)
    assert sum(k for _, k, _ in plan) >= 135 and plan[-1][0] <= 136


def test_oracle_rand09_equals_second_implementation(oracle):
    rng = np.random.default_rng(9)
    cases = [(10000, 0xD5DFAA4CEF265CD7, 0, 0), (1009, 0, 3, 2), (2**64 - 1, 2**63 + 5, 11, 7)]
    cases += [(int(rng.integers(0, 2**63)), int(rng.integers(0, 2**63)), int(rng.integers(0, 12)), int(rng.integers(0, 9)))
              for _ in range(200)]
    for nonce, key, kyoku, honba in cases:
        for algo in (0, 1):
            got = oracle.deal(nonce, key, kyoku, honba, algo=algo).tolist()
            assert got == second_deal(nonce, key, kyoku, honba, algo), (nonce, key, kyoku, honba, algo)
            assert sorted(got) == sorted(unshuffled())
    # the two generations really differ (a test that passes with algo ignored would be worthless)
    assert oracle.deal(10000, 1, 0, 0, algo=0).tolist() != oracle.deal(10000, 1, 0, 0, algo=1).tolist()


def test_rand09_bias_word_path_is_exercised():
    """Canon's second word is drawn when lo > 2^32 - bound: make sure the comparison suite reaches that branch."""
    hits = 0
    for nonce in range(300):
        seed = hashlib.sha3_256(struct.pack("<QQBB", nonce, 7, 0, 0)).digest()
        words = chacha12_words(seed, 64)
        it = iter(words)
        for m, k, bound in chunk_plan(136):
            w = next(it)
            if (w * bound) & MASK > ((-bound) & MASK):
                next(it)
                hits += 1
    assert hits > 50
