// Per-table deal on device: kyoku seed = SHA3-256(nonce_le64 || key_le64 || [kyoku, honba]) -> ChaCha12
// stream -> Fisher-Yates over the 136-tile sequence (reference arena/board.rs:99-123, UNSHUFFLED :786-824).
// The third-party pieces (sha3 0.10.8, rand_chacha 0.9.0, rand shuffle) are restated from their published
// algorithms; `algo` selects the rand-0.8 shuffle (pinned by the reference's seeded game log) or rand-0.9.1's
// (unpinned).
#pragma once
#include "mj_algo.h"

MJD u64 rotl64(u64 x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }
MJD u32 rotl32(u32 x, int n) { return (x << n) | (x >> (32 - n)); }

__device__ static const u64 KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

// Keccak-f[1600] with the lane array fully in registers (all indices compile-time).
MJDN void keccak_f(u64 a[25]) {
    for (int round = 0; round < 24; round++) {
        u64 c[5], d[5];
#pragma unroll
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
        for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rotl64(c[(x + 1) % 5], 1);
#pragma unroll
        for (int i = 0; i < 25; i++) a[i] ^= d[i % 5];
        // rho + pi:  b[y, 2x+3y] = rot(a[x,y], r[x,y]);  index = x + 5y
        u64 b[25];
        constexpr int R[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
#pragma unroll
        for (int y = 0; y < 5; y++)
#pragma unroll
            for (int x = 0; x < 5; x++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl64(a[x + 5 * y], R[x + 5 * y]);
#pragma unroll
        for (int y = 0; y < 5; y++)
#pragma unroll
            for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= KECCAK_RC[round];
    }
}

// LDS work area of one wavefront, lane-interleaved (lane l owns wall[.][l] and rng[.][l]).  The Fisher-Yates swaps are a chain
// of ~136 dependent load/store pairs at data-dependent addresses and the RNG buffer is indexed dynamically: in HBM / scratch
// every link of that chain costs a memory round trip (the deal was the critical path of mj_k_step: a wavefront with one
// dealing lane ran 3x as long as one without), in LDS it costs ~100 cycles.
#define DEAL_LANES 64
struct DealPre {  // a dealt kyoku's four hands with their shanten numbers, computed by four lanes of the deal service (deal_wall_coop)
    u32 hand[4][4];   // hand.mp / hand.sz as two dwords each
    u8 akas[4];
    signed char shanten[4];
};
struct DealScratch {
    u8 wall[136][DEAL_LANES];
    u32 rng[16][DEAL_LANES];
    // the wave-cooperative deal (round 6): one table at a time, all 64 lanes
    alignas(16) u8 cw[144];   // the wall being shuffled
    u8 cres[144];             // the 136 swap partners
    u32 crng[64];             // four ChaCha12 blocks
    u32 cchunk[32];           // the 28 chunk values of the IncreasingUniform sampler
    u64 cseed[4];             // SHA3-256 of (nonce, key, kyoku, honba)
    DealPre pre[DEAL_LANES];  // per pool lane: what the service hands to the lane's kyoku_init
};
#if defined(__HIP_DEVICE_COMPILE__)
#define MJ_ASSUME_LDS(p) __builtin_assume(__builtin_amdgcn_is_shared((const void*)(p)))
#else
#define MJ_ASSUME_LDS(p) ((void)0)  // host pass of the single-source compile: the builtin only exists on the device
#endif

// floor(2^32 / n) for the divisors of the rand-0.9.1 shuffle (n = 1 .. 136): chunk / n as one multiply-high plus at most two correction
// steps instead of the ~40-instruction u32 division -- the shuffle is a serial chain of ONE lane, and that lane is mj_k_step's critical path.
struct DealMagic {
    u32 m[137];
    constexpr DealMagic() : m() {
        m[0] = 0;
        m[1] = 0xFFFFFFFFu;  // (2^32 / 1 does not fit: with 2^32 - 1 the estimate of x / 1 is x - 1 for x > 0, one correction step)
        for (u32 n = 2; n <= 136; n++) m[n] = (u32)(0x100000000ull / n);
    }
};
__device__ static const DealMagic DEAL_MAGIC = DealMagic();
MJD void deal_divmod(u32 x, u32 n, u32& q, u32& r) {  // exact: q = x / n, r = x % n for 1 <= n <= 136
    q = (u32)(((u64)x * DEAL_MAGIC.m[n]) >> 32);
    r = x - q * n;
#pragma unroll
    for (int k = 0; k < 2; k++)  // the estimate is low by at most 2 (floor(2^32 / n) * n > 2^32 - n, x < 2^32): a bounded correction
        if (r >= n) {
            q += 1;
            r -= n;
        }
}

struct ChaCha12Dev {
    u32 key[8];
    u32* buf;  // -> DealScratch::rng[0][lane], stride DEAL_LANES
    u32 counter;
    int idx;
    MJD void init(const u64 seed[4], u32* lds_buf) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            key[2 * i] = (u32)seed[i];
            key[2 * i + 1] = (u32)(seed[i] >> 32);
        }
        buf = lds_buf;
        counter = 0;
        idx = 16;
    }
    MJD void refill() {
        u32 s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                     key[4], key[5], key[6], key[7], counter, 0u, 0u, 0u};
        u32 x[16];
#pragma unroll
        for (int i = 0; i < 16; i++) x[i] = s[i];
#define QR(a, b, c, d)                                \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16);     \
    x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12);     \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);      \
    x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7);
        for (int r = 0; r < 6; r++) {
            QR(0, 4, 8, 12) QR(1, 5, 9, 13) QR(2, 6, 10, 14) QR(3, 7, 11, 15)
            QR(0, 5, 10, 15) QR(1, 6, 11, 12) QR(2, 7, 8, 13) QR(3, 4, 9, 14)
        }
#undef QR
#pragma unroll
        for (int i = 0; i < 16; i++) buf[i * DEAL_LANES] = x[i] + s[i];
        counter++;
        idx = 0;
    }
    MJD u32 next() {
        if (idx >= 16) refill();
        return buf[(idx++) * DEAL_LANES];
    }
};

// Writes the shuffled 136-tile sequence into wall[i * stride]; S = the wavefront's LDS work area, lane = this thread's column.
MJDN void deal_wall(u8* wall, int stride, DealScratch* S, int lane, u64 nonce, u64 key, int kyoku, int honba, int algo) {
    MJ_ASSUME_LDS(S);
    // SHA3-256 of an 18-byte message: one rate block (136 B), pad 0x06 .. 0x80
    u64 st[25];
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = 0;
    st[0] = nonce;
    st[1] = key;
    st[2] = (u64)(kyoku & 0xFF) | ((u64)(honba & 0xFF) << 8) | (0x06ull << 16);
    st[16] = 0x80ull << 56;
    keccak_f(st);
    ChaCha12Dev rng;
    rng.init(st, &S->rng[0][lane]);
    u8* w = &S->wall[0][lane];
    constexpr int WS = DEAL_LANES;

    for (int t = 0; t < 34; t++)
        for (int k = 0; k < 4; k++) w[(t * 4 + k) * WS] = (u8)t;
    w[(T_5M * 4) * WS] = T_5MR;
    w[(T_5P * 4) * WS] = T_5PR;
    w[(T_5S * 4) * WS] = T_5SR;

    if (algo == 0) {  // rand 0.8: for i in (1..n).rev(): swap(i, gen_range(0..=i))
        for (int i = 135; i >= 1; i--) {
            u32 range = (u32)(i + 1);
            u32 zone = (range << __clz(range)) - 1;
            u32 j;
            for (;;) {
                u64 m = (u64)rng.next() * range;
                if ((u32)m <= zone) {
                    j = (u32)(m >> 32);
                    break;
                }
            }
            u8 a = w[i * WS], b = w[j * WS];
            w[i * WS] = b;
            w[j * WS] = a;
        }
    } else {  // rand 0.9.1: forward loop with the chunked IncreasingUniform sampler
        u32 n = 0, chunk = 0;
        int chunk_remaining = 1;
        for (int i = 0; i < 136; i++) {
            u32 next_n = n + 1;
            int next_rem;
            if (chunk_remaining >= 1) {
                next_rem = chunk_remaining - 1;
            } else {
                u32 product = next_n, current = next_n + 1;
                for (;;) {
                    u64 p = (u64)product * current;
                    if (p > 0xffffffffull) break;
                    product = (u32)p;
                    current += 1;
                }
                u32 bound = product;
                int remaining = (int)(current - next_n);
                u64 m = (u64)rng.next() * bound;
                u32 res = (u32)(m >> 32), lo = (u32)m;
                if (lo > (u32)(0u - bound)) {
                    u32 new_hi = (u32)(((u64)rng.next() * bound) >> 32);
                    res += (lo + new_hi) < lo;
                }
                chunk = res;
                next_rem = remaining - 1;
            }
            u32 result;
            if (next_rem == 0) result = chunk;
            else {
                u32 q;
                deal_divmod(chunk, next_n, q, result);
                chunk = q;
            }
            chunk_remaining = next_rem;
            n = next_n;
            u8 a = w[i * WS], b = w[result * WS];
            w[i * WS] = b;
            w[result * WS] = a;
        }
    }
    for (int i = 0; i < 136; i++) wall[i * stride] = w[i * WS];
}

// ---- The deal as a service of the whole wavefront (round 6).  mj_k_step's duration was ONE lane: a wavefront with a dealing lane spent
// 141 us in start_kyoku -- SHA3-256, ChaCha12 and the shuffle as ~32 k instructions of a single lane, then four hands initialised one
// after the other (dependent table gathers) -- on top of the 127 us every wavefront needs.  Here the poll loop hands the deal out: the 64
// lanes build ONE table's wall together -- the four ChaCha12 blocks side by side on 16 lanes (a block = four lanes, one per column; the
// diagonal rounds rotate the rows across them), the 28 chunk values of rand 0.9.1's IncreasingUniform by one short serial scan (a chunk
// consumes one or two words depending on its own first word: Canon's method), the 136 swap partners chunk by chunk on 28 lanes, the swaps
// themselves by one lane in LDS (a data-dependent chain: 136 x two reads + two writes), the copy to the pool and to the owner's LDS column
// by all lanes, and the four hands with their shanten numbers on four lanes; SHA3-256's permutation runs on 25 lanes (keccak_f_lanes).  Same bytes as deal_wall: the lock-step tests deal
// every kyoku through this path, tests/test_oracle_deal.py pins the algorithm.  rand 0.8's rejection sampler is serial by nature: lane 0
// runs the reference-shaped loop of deal_wall for it.
// Keccak-f[1600] across 25 lanes of the wavefront (lane i holds word i = x + 5 y); every lane of the wavefront executes it (the cross-lane
// moves stay outside divergent control flow), lanes 25 .. 63 carry don't-care values.  Per round: column parities (four moves), theta's two
// neighbours, rho + pi as ONE move of the rotated word (lane (x', y') pulls from x = 3 (y' + 2 x') mod 5, y = x'), chi's two neighbours --
// nine 64-bit moves and ~40 instructions instead of the ~300 instructions of keccak_f's single-lane round.
__device__ static const u8 KECCAK_RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
MJD u64 deal_shfl64(u64 v, int src) {
    const u32 lo = (u32)__shfl((int)(u32)v, src), hi = (u32)__shfl((int)(u32)(v >> 32), src);
    return (u64)lo | ((u64)hi << 32);
}
MJD u64 keccak_f_lanes(u64 a, int lane) {
    const bool in = lane < 25;
    const int i = in ? lane : 0, x = i % 5, y = i / 5;
    int col[4];
#pragma unroll
    for (int k = 0; k < 4; k++) col[k] = in ? (i + 5 * (k + 1)) % 25 : lane;
    const int xm1 = in ? y * 5 + (x + 4) % 5 : lane, xp1 = in ? y * 5 + (x + 1) % 5 : lane, xp2 = in ? y * 5 + (x + 2) % 5 : lane;
    // pi: b[x', y'] = rot(a[x, y]) with x' = y, y' = (2 x + 3 y) mod 5  =>  this lane (x', y') = (x, y) pulls from (3 (y + 2 x) mod 5, x)
    const int pi_src = in ? (3 * ((y + 2 * x) % 5)) % 5 + 5 * x : lane;
    const unsigned rot = KECCAK_RHO[i];
    for (int round = 0; round < 24; round++) {
        u64 c = a;
#pragma unroll
        for (int k = 0; k < 4; k++) c ^= deal_shfl64(a, col[k]);
        const u64 d = deal_shfl64(c, xm1) ^ rotl64(deal_shfl64(c, xp1), 1);
        a ^= d;
        const u64 t = (a << rot) | (a >> ((64u - rot) & 63u));
        const u64 b = deal_shfl64(t, pi_src);
        a = b ^ (~deal_shfl64(b, xp1) & deal_shfl64(b, xp2));
        if (lane == 0) a ^= KECCAK_RC[round];
    }
    return a;
}

struct DealPlan {
    u8 start[28], rem[28];
    u32 bound[28];
    constexpr DealPlan() : start(), rem(), bound() {
        int k = 0;
        u32 n = 0;
        int i = 1;  // (position 0 takes the initial chunk: no draw)
        n = 1;
        while (i < 136) {
            const u32 next_n = n + 1;
            unsigned long long product = next_n;
            u32 current = next_n + 1;
            while (product * current <= 0xffffffffull) {
                product *= current;
                current += 1;
            }
            start[k] = (u8)i;
            rem[k] = (u8)(current - next_n);
            bound[k] = (u32)product;
            k++;
            n += current - next_n;
            i += (int)(current - next_n);
        }
    }
};
__device__ static const DealPlan DEAL_PLAN = DealPlan();
constexpr bool deal_plan_ok() {
    constexpr DealPlan p = DealPlan();
    return p.start[0] == 1 && p.bound[0] == 479001600u && p.rem[0] == 11 && p.start[1] == 12 && p.bound[2] == 3315312000u && p.start[27] == 135 &&
           p.bound[27] == 357399024u && p.rem[27] == 4;
}
static_assert(deal_plan_ok(), "the chunk plan of IncreasingUniform for 136 elements (28 chunks)");

// All 64 lanes call it with the SAME arguments (the dealing table's); `owner` = that table's lane: its pool column `wall` (stride MJ_LANES)
// and its column of S->wall / S->pre receive the result.
template <class TablesT>
MJDN void deal_wall_coop(u8* wall, int stride, DealScratch* S, int owner, u64 nonce, u64 key, int kyoku, int honba, int algo, const TablesT& T) {
    MJ_ASSUME_LDS(S);
    const int lane = threadIdx.x & 63;
    if (algo != 1) {  // rand 0.8: the serial path, on lane 0, into the owner's column
        if (lane == 0) deal_wall(wall, stride, S, owner, nonce, key, kyoku, honba, algo);
        mj_team_sync<64>();
    } else {
        {   // SHA3-256 of the 18-byte message: one rate block (136 B), pad 0x06 .. 0x80; the permutation on 25 lanes
            u64 a = lane == 0 ? nonce : lane == 1 ? key : lane == 2 ? ((u64)(kyoku & 0xFF) | ((u64)(honba & 0xFF) << 8) | (0x06ull << 16))
                  : lane == 16 ? (0x80ull << 56) : 0ull;
            a = keccak_f_lanes(a, lane);
            if (lane < 4) S->cseed[lane] = a;
        }
        // the unshuffled wall, three tiles per lane
        for (int j = lane; j < 136; j += 64) {
            const int t = j >> 2;
            S->cw[j] = (u8)((j & 3) == 0 && (t == T_5M || t == T_5P || t == T_5S) ? (t == T_5M ? T_5MR : t == T_5P ? T_5PR : T_5SR) : t);
        }
        mj_team_sync<64>();
        {   // four ChaCha12 blocks (counters 0 .. 3), a block = four lanes = the four columns of its 4 x 4 state (lanes 16 .. 63 repeat
            // them: the cross-lane rotations stay outside divergent control flow)
            const int c = lane & 3, blk = (lane >> 2) & 3;
            const u32 k0 = (u32)(S->cseed[c >> 1] >> (32 * (c & 1))), k1 = (u32)(S->cseed[2 + (c >> 1)] >> (32 * (c & 1)));
            const u32 cst = c == 0 ? 0x61707865u : c == 1 ? 0x3320646eu : c == 2 ? 0x79622d32u : 0x6b206574u;
            const u32 i0 = cst, i1 = k0, i2 = k1, i3 = c == 0 ? (u32)blk : 0u;
            u32 a = i0, b = i1, cc = i2, d = i3;
#define DEAL_QR()                                   \
    a += b; d = rotl32(d ^ a, 16);                  \
    cc += d; b = rotl32(b ^ cc, 12);                \
    a += b; d = rotl32(d ^ a, 8);                   \
    cc += d; b = rotl32(b ^ cc, 7);
            for (int r = 0; r < 6; r++) {
                DEAL_QR()  // columns
                b = __shfl(b, (lane & ~3) | ((c + 1) & 3));
                cc = __shfl(cc, (lane & ~3) | ((c + 2) & 3));
                d = __shfl(d, (lane & ~3) | ((c + 3) & 3));
                DEAL_QR()  // diagonals
                b = __shfl(b, (lane & ~3) | ((c + 3) & 3));
                cc = __shfl(cc, (lane & ~3) | ((c + 2) & 3));
                d = __shfl(d, (lane & ~3) | ((c + 1) & 3));
            }
#undef DEAL_QR
            if (lane < 16) {
                S->crng[blk * 16 + c] = a + i0;
                S->crng[blk * 16 + 4 + c] = b + i1;
                S->crng[blk * 16 + 8 + c] = cc + i2;
                S->crng[blk * 16 + 12 + c] = d + i3;
            }
        }
        mj_team_sync<64>();
        if (lane == 0) {  // the chunk values: Canon's method, one or two words per chunk (28 chunks, at most 56 of the 64 words)
            int idx = 0;
            for (int k = 0; k < 28; k++) {
                const u32 bound = DEAL_PLAN.bound[k];
                const u64 m = (u64)S->crng[idx++] * bound;
                u32 res = (u32)(m >> 32);
                const u32 lo = (u32)m;
                if (lo > (u32)(0u - bound)) {
                    const u32 new_hi = (u32)(((u64)S->crng[idx++] * bound) >> 32);
                    res += (lo + new_hi) < lo;
                }
                S->cchunk[k] = res;
            }
            S->cres[0] = 0;
        }
        mj_team_sync<64>();
        if (lane < 28) {  // the swap partners of a chunk's positions: successive divisions, the last position takes what is left
            u32 chunk = S->cchunk[lane];
            const int i0 = DEAL_PLAN.start[lane], rem = DEAL_PLAN.rem[lane];
            for (int q = 0; q < rem; q++) {
                const int i = i0 + q;
                if (i >= 136) break;
                u32 result;
                if (q == rem - 1) result = chunk;
                else {
                    u32 qq;
                    deal_divmod(chunk, (u32)(i + 1), qq, result);
                    chunk = qq;
                }
                S->cres[i] = (u8)result;
            }
        }
        mj_team_sync<64>();
        if (lane == 0) {  // the shuffle itself: a chain of data-dependent swaps
            for (int i = 1; i < 136; i++) {
                const int r = S->cres[i];
                const u8 a = S->cw[i], b = S->cw[r];
                S->cw[i] = b;
                S->cw[r] = a;
            }
        }
        mj_team_sync<64>();
        for (int j = lane; j < 136; j += 64) {  // to the pool and to the owner's column of the lane-interleaved scratch
            const u8 t = S->cw[j];
            wall[j * stride] = t;
            S->wall[j][owner] = t;
        }
        mj_team_sync<64>();
    }
    if (lane < 4) {  // the four hands and their shanten numbers (kyoku_init takes them from S->pre[owner])
        Hand h = {0, 0};
        u8 akas = 0;
        for (int i = 0; i < 13; i++) {
            const int t = S->wall[lane * 13 + i][owner];
            h.inc(deaka(t));
            if (is_aka(t)) akas |= 1 << (t - T_5MR);
        }
        const int sv = calc_all(T, h, 4);
        DealPre& P = S->pre[owner];
        P.hand[lane][0] = (u32)h.mp; P.hand[lane][1] = (u32)(h.mp >> 32); P.hand[lane][2] = (u32)h.sz; P.hand[lane][3] = (u32)(h.sz >> 32);
        P.akas[lane] = akas;
        P.shanten[lane] = (signed char)sv;
    }
    mj_team_sync<64>();
}
