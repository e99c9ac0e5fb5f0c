// ORACLE — TEST INFRASTRUCTURE ONLY (see mjo.h).
// Deal-from-seed: arena/board.rs:99-123 (SHA3-256 -> ChaCha12Rng -> SliceRandom::shuffle), UNSHUFFLED :786-824.
//
// Third-party arithmetic absent from /root/reference (Cargo.lock:1042-1043,1052-1053,1236-1237):
//   sha3 0.10.8        -> FIPS-202 SHA3-256, restated below (checked against hashlib in tests)
//   rand_chacha 0.9.0  -> ChaCha, 12 rounds, 64-bit block counter in words 12-13, stream id 0,
//                         output words consumed in order
//   rand 0.9.1 shuffle -> DEAL_RAND09: restated from the published algorithm
//                         (IncreasingUniform chunked sampler + Canon's method); PARITY UNPINNED —
//                         the repository holds no vector for it.
//   rand 0.8 shuffle   -> DEAL_RAND08: reverse Fisher-Yates with the widening-multiply + zone
//                         rejection sampler; PINNED by the seeded game log in
//                         log-viewer/index.example.html:10-264 (tests/test_oracle_golden_log.py).
#include <algorithm>

#include "mjo.h"

namespace mjo {

// ---------------------------------------------------------------- SHA3-256
static inline u64 rotl64(u64 x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }
static void keccak_f(u64 st[25]) {
    static const u64 RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
                               0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
                               0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
                               0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
                               0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                               0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const int ROTC[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
    static const int PILN[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    for (int round = 0; round < 24; round++) {
        u64 bc[5];
        for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
        for (int i = 0; i < 5; i++) {
            u64 t = bc[(i + 4) % 5] ^ rotl64(bc[(i + 1) % 5], 1);
            for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
        }
        u64 t = st[1];
        for (int i = 0; i < 24; i++) {
            int j = PILN[i];
            u64 b = st[j];
            st[j] = rotl64(t, ROTC[i]);
            t = b;
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; i++) bc[i] = st[j + i];
            for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        st[0] ^= RC[round];
    }
}
void sha3_256(const u8* data, size_t len, u8 out[32]) {
    const size_t rate = 136;
    u64 st[25] = {};
    u8 block[136];
    while (len >= rate) {
        for (size_t i = 0; i < rate / 8; i++) {
            u64 w;
            memcpy(&w, data + i * 8, 8);
            st[i] ^= w;
        }
        keccak_f(st);
        data += rate;
        len -= rate;
    }
    memset(block, 0, rate);
    memcpy(block, data, len);
    block[len] ^= 0x06;
    block[rate - 1] ^= 0x80;
    for (size_t i = 0; i < rate / 8; i++) {
        u64 w;
        memcpy(&w, block + i * 8, 8);
        st[i] ^= w;
    }
    keccak_f(st);
    memcpy(out, st, 32);
}

// ---------------------------------------------------------------- ChaCha12
static inline u32 rotl32(u32 x, int n) { return (x << n) | (x >> (32 - n)); }
ChaCha12::ChaCha12(const u8 seed[32]) { memcpy(key, seed, 32); }
u32 ChaCha12::next_u32() {
    if (idx >= 16) {
        u32 s[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574};
        for (int i = 0; i < 8; i++) s[4 + i] = key[i];
        s[12] = (u32)counter;
        s[13] = (u32)(counter >> 32);
        s[14] = 0;
        s[15] = 0;
        u32 x[16];
        memcpy(x, s, sizeof x);
#define QR(a, b, c, d)                 \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16); \
    x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12); \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);  \
    x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7);
        for (int r = 0; r < 6; r++) {
            QR(0, 4, 8, 12) QR(1, 5, 9, 13) QR(2, 6, 10, 14) QR(3, 7, 11, 15)
            QR(0, 5, 10, 15) QR(1, 6, 11, 12) QR(2, 7, 8, 13) QR(3, 4, 9, 14)
        }
#undef QR
        for (int i = 0; i < 16; i++) buf[i] = x[i] + s[i];
        counter++;
        idx = 0;
    }
    return buf[idx++];
}

// ---------------------------------------------------------------- shuffles
static void shuffle_rand08(u8* seq, int len, ChaCha12& rng) {
    for (int i = len - 1; i >= 1; i--) {
        u32 range = (u32)(i + 1);
        u32 zone = (range << __builtin_clz(range)) - 1;
        u32 j;
        for (;;) {
            u32 v = rng.next_u32();
            u64 m = (u64)v * range;
            u32 hi = (u32)(m >> 32), lo = (u32)m;
            if (lo <= zone) {
                j = hi;
                break;
            }
        }
        std::swap(seq[i], seq[j]);
    }
}
static u32 random_range_rand09(ChaCha12& rng, u32 range) {  // random_range(..range), Canon's method
    u64 m = (u64)rng.next_u32() * range;
    u32 result = (u32)(m >> 32), lo = (u32)m;
    if (lo > (u32)(0u - range)) {
        u64 m2 = (u64)rng.next_u32() * range;
        u32 new_hi = (u32)(m2 >> 32);
        u32 sum = lo + new_hi;
        bool overflow = sum < lo;
        result += overflow;
    }
    return result;
}
static void calculate_bound_u32(u32 m, u32& bound, u8& count) {
    u32 product = m, current = m + 1;
    for (;;) {
        u64 p = (u64)product * current;
        if (p <= 0xffffffffull) {
            product = (u32)p;
            current += 1;
        } else {
            bound = product;
            count = (u8)(current - m);
            return;
        }
    }
}
static void shuffle_rand09(u8* seq, int len, ChaCha12& rng) {
    if (len <= 1) return;
    u32 n = 0, chunk = 0;
    u8 chunk_remaining = 1;  // n == 0
    for (int i = 0; i < len; i++) {
        u32 next_n = n + 1;
        u8 next_chunk_remaining;
        if (chunk_remaining >= 1) {
            next_chunk_remaining = chunk_remaining - 1;
        } else {
            u32 bound;
            u8 remaining;
            calculate_bound_u32(next_n, bound, remaining);
            chunk = random_range_rand09(rng, bound);
            next_chunk_remaining = remaining - 1;
        }
        u32 result;
        if (next_chunk_remaining == 0) {
            result = chunk;
        } else {
            result = chunk % next_n;
            chunk /= next_n;
        }
        chunk_remaining = next_chunk_remaining;
        n = next_n;
        std::swap(seq[i], seq[result]);
    }
}

void deal_from_seed(u64 nonce, u64 key, u8 kyoku, u8 honba, DealAlgo algo, u8 seq[136]) {  // board.rs:99-109
    u8 msg[18];
    memcpy(msg, &nonce, 8);  // to_le_bytes (host is little-endian)
    memcpy(msg + 8, &key, 8);
    msg[16] = kyoku;
    msg[17] = honba;
    u8 seed[32];
    sha3_256(msg, 18, seed);
    ChaCha12 rng(seed);
    // UNSHUFFLED (board.rs:786-824): 4 copies of each tile in id order; the first copy of each 5 is aka.
    for (int t = 0; t < 34; t++)
        for (int k = 0; k < 4; k++) seq[t * 4 + k] = (u8)t;
    seq[T_5M * 4] = T_5MR;
    seq[T_5P * 4] = T_5PR;
    seq[T_5S * 4] = T_5SR;
    if (algo == DEAL_RAND08) shuffle_rand08(seq, 136, rng);
    else shuffle_rand09(seq, 136, rng);
}

}  // namespace mjo
