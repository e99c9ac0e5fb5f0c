// Device-side riichi primitives for gfx950: tile helpers, register-packed hands, shanten, agari, points.
//
// Hands are NOT arrays here: the 34 tile counts (0..4) are packed 3 bits per tile into two 64-bit
// registers (`mp` = man+pin, `sz` = sou+honours).  A modified copy of a hand is a register move, the
// base-5 shanten keys come from static shifts, and nothing is dynamically indexed in scratch or LDS.
//
// Functional specs (reference, /root/reference/libriichi/src): algo/shanten.rs:51-150,
// algo/agari.rs:126-157,203-912, algo/point.rs:13-112, tile.rs:68-150.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mj_state.h"

#ifndef MJD  // tests/host_algo_check.hip defines both as __host__ __device__ to run the pure functions on the CPU
#define MJD __device__ __forceinline__
#define MJDN __device__ __noinline__
#endif

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;

// Two constructs are spelled through a macro / helper so that the host SIMT emulator of the test suite
// (tests/host/emu, -DMJ_EMU) can give them their lane-group meaning; on the device they are the plain HIP forms.
#ifndef MJ_EMU
#define MJ_DYN_SHARED(T, name) extern __shared__ T name[]
// LDS hand-off between the W lanes of a team inside one wavefront.  The lanes run in lock-step and the LDS serves a
// wavefront's accesses in issue order, so a later ds_read sees an earlier ds_write of the same wavefront: all that is needed
// is that the compiler keeps the order (wavefront-scope fence + scheduling barrier, no instruction).  A workgroup-scope
// fence (__threadfence_block) would also drain every outstanding HBM load (s_waitcnt vmcnt(0)) at each hand-off.
template <int W> MJD void mj_team_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}
// the same for teams of `width` consecutive lanes, any width (team k = lanes [k * width, (k + 1) * width) of the wavefront)
MJD void mj_team_sync_n(int width) {
    (void)width;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}
#else
template <int W> inline void mj_team_sync() { emu::group_sync(W); }
inline void mj_team_sync_n(int width) { emu::group_sync(width); }
#endif

// ---- IEEE f32 division a / b with the divisor's refined reciprocal hoisted.  For operands that need no scaling (neither
// denormal, quotient not denormal, exponents less than 96 apart — true here: probabilities in [1e-25, 1]) the division
// sequence of the hardware / of LLVM's lowering is  r0 = rcp(b); e0 = fma(-b, r0, 1); r1 = fma(e0, r0, r0);
// q0 = a * r1; e1 = fma(-b, q0, a); q1 = fma(e1, r1, q0); e2 = fma(-b, q1, a); q = fma(e2, r1, q1)  — correctly rounded
// for any r0 within 1 ulp of 1/b.  A lane divides by the same not_tsumo value all the time, so r1 is computed once per
// state and a division costs 5 instructions instead of 10 (tests/host/algo_check.hip checks q == a / b on 10^8 operands
// of the domain for r0 = RN(1/b) and both neighbours).
MJD float sp_rcp_refined(float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float r0 = __builtin_amdgcn_rcpf(b);
#else
    const float r0 = 1.0f / b;
#endif
    const float e0 = __builtin_fmaf(-b, r0, 1.0f);
    return __builtin_fmaf(e0, r0, r0);
}
MJD float sp_div(float a, float b, float r1) {
    const float q0 = a * r1;
    const float e1 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(e1, r1, q0);
    const float e2 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(e2, r1, q1);
}
// The first correction step alone: q1 is already the correctly rounded quotient for EVERY operand pair the SP kernel divides
// (prob = tsumo_prob[c][j] * not_tsumo[s][j] / not_tsumo[s][i], calc.rs:135-167,486-548: 4.3 M pairs over all wall sizes,
// required-tile sums, counts and turn pairs, for r0 = RN(1/b) and both neighbours — enumerated in tests/host/algo_check.hip);
// it is NOT a general division (random operands need the second step about once in 10^3).
MJD float sp_div_domain(float a, float b, float r1) {
    const float q0 = a * r1;
    const float e1 = __builtin_fmaf(-b, q0, a);
    return __builtin_fmaf(e1, r1, q0);
}


enum : int { T_5M = 4, T_5P = 13, T_5S = 22, T_E = 27, T_S = 28, T_W = 29, T_N = 30, T_P = 31, T_F = 32, T_C = 33,
             T_5MR = 34, T_5PR = 35, T_5SR = 36, T_UNK = 37 };

MJD int deaka(int t) { return t >= 34 ? (t == 34 ? 4 : t == 35 ? 13 : t == 36 ? 22 : t) : t; }
MJD bool is_aka(int t) { return t >= 34 && t <= 36; }
MJD int akaize(int t) { return t == 4 ? 34 : t == 13 ? 35 : t == 22 ? 36 : t; }
MJD bool is_jihai(int t) { return t >= 27 && t <= 33; }
MJD bool is_yaokyuu(int t) { return t < 34 && (t >= 27 || t % 9 == 0 || t % 9 == 8); }
MJD int tile_next(int t) {  // tile.rs:117-132 (dora wrap)
    if (t >= 37) return t;
    t = deaka(t);
    int kind = t / 9, num = t % 9;
    if (kind < 3) return kind * 9 + (num + 1) % 9;
    if (num < 4) return 27 + (num + 1) % 4;
    return 31 + (num - 4 + 1) % 3;
}
MJD int tile_prev(int t) {  // tile.rs:134-150
    if (t >= 37) return t;
    t = deaka(t);
    int kind = t / 9, num = t % 9;
    if (kind < 3) return kind * 9 + (num + 8) % 9;
    if (num < 4) return 27 + (num + 3) % 4;
    return 31 + (num - 4 + 2) % 3;
}
constexpr u64 YAOKYUU_MASK = (1ull << 0) | (1ull << 8) | (1ull << 9) | (1ull << 17) | (1ull << 18) | (1ull << 26) |
                             (0x7Full << 27);

// ---------------------------------------------------------------- packed hand
struct Hand {
    u64 mp;  // tiles 0..17  at bit 3*t
    u64 sz;  // tiles 18..33 at bit 3*(t-18)
    MJD int get(int t) const { return t < 18 ? (int)((mp >> (3 * t)) & 7) : (int)((sz >> (3 * (t - 18))) & 7); }
    MJD void inc(int t) {
        if (t < 18) mp += 1ull << (3 * t);
        else sz += 1ull << (3 * (t - 18));
    }
    MJD void dec(int t) {
        if (t < 18) mp -= 1ull << (3 * t);
        else sz -= 1ull << (3 * (t - 18));
    }
    MJD void clear(int t) {
        if (t < 18) mp &= ~(7ull << (3 * t));
        else sz &= ~(7ull << (3 * (t - 18)));
    }
    MJD bool empty() const { return (mp | sz) == 0; }
    // ---- bit-parallel views of the 3-bit count fields (bit 0 of field f = bit 3f)
    static constexpr u64 LSB3 = 0x1249249249249249ull;
    static MJD u64 nz_fields(u64 x) { return (x | (x >> 1) | (x >> 2)) & LSB3; }   // count >= 1
    static MJD u64 ge2_fields(u64 x) { return ((x >> 1) | (x >> 2)) & LSB3; }      // count >= 2
    static MJD u64 compress3(u64 x) {  // bit 3f -> bit f (21 fields)
        x &= LSB3;
        x = (x ^ (x >> 2)) & 0x10c30c30c30c30c3ull;
        x = (x ^ (x >> 4)) & 0x100f00f00f00f00full;
        x = (x ^ (x >> 8)) & 0x001f0000ff0000ffull;
        x = (x ^ (x >> 16)) & 0x001f00000000ffffull;
        x = (x ^ (x >> 32)) & 0x00000000001fffffull;
        return x;
    }
    MJD u64 nonzero_mask() const {  // bit t set iff count(t) > 0
        return compress3(nz_fields(mp)) | (compress3(nz_fields(sz)) << 18);
    }
    // yaokyuu tiles as field masks: 1m 9m 1p 9p (mp fields 0 8 9 17), 1s 9s + honours (sz fields 0 8 9..15)
    static constexpr u64 YAO_MP = 0x0008000009000001ull, YAO_SZ = 0x0000249249000001ull;
    static MJD int sum_fields(u64 x) { return __popcll(x & LSB3) + 2 * __popcll((x >> 1) & LSB3) + 4 * __popcll((x >> 2) & LSB3); }
    MJD int total() const { return sum_fields(mp) + sum_fields(sz); }  // number of tiles
    MJD int n_kinds() const { return __popcll(nz_fields(mp)) + __popcll(nz_fields(sz)); }
    MJD bool has_quad() const { return (((mp | sz) >> 2) & LSB3) != 0; }  // some tile held four times (fields count 0..4)
    MJD int n_pairs() const { return __popcll(ge2_fields(mp)) + __popcll(ge2_fields(sz)); }
    MJD int n_yao_kinds() const { return __popcll(nz_fields(mp) & YAO_MP) + __popcll(nz_fields(sz) & YAO_SZ); }
    MJD int n_yao_pairs() const { return __popcll(ge2_fields(mp) & YAO_MP) + __popcll(ge2_fields(sz) & YAO_SZ); }
};

// ---------------------------------------------------------------- shanten (shanten.rs:51-150)
MJD u32 suit_key9(u64 bits) {  // 9 packed counts, first tile most significant (shanten.rs:82-84)
    u32 k = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) k = k * 5 + (u32)((bits >> (3 * i)) & 7);
    return k;
}
MJD u32 suit_key7(u64 bits) {
    u32 k = 0;
#pragma unroll
    for (int i = 0; i < 7; i++) k = k * 5 + (u32)((bits >> (3 * i)) & 7);
    return k;
}
// Row = 10 nibbles in a u64 (nibble j at bits 4j).  Index past the table -> zeros (`unwrap_or_default`).
MJD u64 sh_row(const u64* __restrict__ tab, u32 n, u32 idx) { return idx < n ? tab[idx] : 0ull; }
#define NIB(r, j) ((int)(((r) >> (4 * (j))) & 15))

MJD void sh_unpack(u64 r, int v[10]) {
#pragma unroll
    for (int j = 0; j < 10; j++) v[j] = NIB(r, j);
}
MJD void sh_add_suhai(int lhs[10], u64 tab, int m) {  // shanten.rs:51-69
#pragma unroll
    for (int j = 9; j >= 5; j--) {
        if (j > 5 + m) continue;
        int sht = min(lhs[j] + NIB(tab, 0), lhs[0] + NIB(tab, j));
#pragma unroll
        for (int k = 5; k < 9; k++) {
            if (k >= j) continue;
            sht = min(sht, min(lhs[k] + NIB(tab, j - k), lhs[j - k] + NIB(tab, k)));
        }
        lhs[j] = sht;
    }
#pragma unroll
    for (int j = 4; j >= 0; j--) {
        if (j > m) continue;
        int sht = lhs[j] + NIB(tab, 0);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (k >= j) continue;
            sht = min(sht, lhs[k] + NIB(tab, j - k));
        }
        lhs[j] = sht;
    }
}
MJD int sh_add_jihai_final(const int lhs[10], u64 tab, int m) {  // shanten.rs:71-80, returns lhs[5+m]
    int j = m + 5;
    int sht = min(lhs[j] + NIB(tab, 0), lhs[0] + NIB(tab, j));
#pragma unroll
    for (int k = 5; k < 9; k++) {
        if (k >= j) continue;
        sht = min(sht, min(lhs[k] + NIB(tab, j - k), lhs[j - k] + NIB(tab, k)));
    }
    return sht;
}
MJD int calc_normal(const MjTablesDev& T, Hand h, int len_div3) {  // shanten.rs:88-102
    u32 km = suit_key9(h.mp), kp = suit_key9(h.mp >> 27), ks = suit_key9(h.sz), kz = suit_key7(h.sz >> 27);
    u64 rm = sh_row(T.suhai, T.n_suhai, km);
    u64 rp = sh_row(T.suhai, T.n_suhai, kp);
    u64 rs = sh_row(T.suhai, T.n_suhai, ks);
    u64 rz = sh_row(T.jihai, T.n_jihai, kz);
    int v[10];
    sh_unpack(rm, v);
    sh_add_suhai(v, rp, len_div3);
    sh_add_suhai(v, rs, len_div3);
    return sh_add_jihai_final(v, rz, len_div3) - 1;
}
MJD int calc_chitoi(Hand h) {  // shanten.rs:104-118
    const int pairs = h.n_pairs(), kinds = h.n_kinds();
    int redunct = kinds >= 7 ? 0 : 7 - kinds;
    return 7 - pairs + redunct - 1;
}
MJD int calc_kokushi(Hand h) {  // shanten.rs:120-137
    const int pairs = h.n_yao_pairs(), kinds = h.n_yao_kinds();
    return 14 - kinds - (pairs > 0) - 1;
}
MJD int calc_all(const MjTablesDev& T, Hand h, int len_div3) {  // shanten.rs:139-150
    int s = calc_normal(T, h, len_div3);
    if (s <= 0 || len_div3 < 4) return s;
    s = min(s, calc_chitoi(h));
    if (s > 0) s = min(s, calc_kokushi(h));
    return s;
}


// ---------------------------------------------------------------- incremental shanten (for the SP kernel)
// A hand that differs from a base hand by +1/-1 tiles only changes the table row of the touched suit(s), so the
// base rows are loaded once and every probe costs one gather instead of four (and the gathers of a whole probe
// batch are independent, i.e. in flight together).
struct ShBase {
    u32 key[4];
    u64 row[4];
    int pairs, kinds, kpairs, kkinds;  // chitoi / kokushi counters (shanten.rs:104-137)
    // lane-dependent suit index: selects, so that key[] / row[] stay in registers (a dynamically indexed local array
    // lives in scratch memory, and every scratch access shares vmcnt with the table gathers)
    MJD u32 key_of(int s) const { return s == 0 ? key[0] : s == 1 ? key[1] : s == 2 ? key[2] : key[3]; }
    MJD u64 row_of(int s) const { return s == 0 ? row[0] : s == 1 ? row[1] : s == 2 ? row[2] : row[3]; }
};
MJD int sh_suit(int t) { return t < 9 ? 0 : t < 18 ? 1 : t < 27 ? 2 : 3; }
MJD u32 sh_pow(int t) {  // weight of tile t inside its suit key (first tile most significant): 5^e, e = 0..8
    const int e = t < 27 ? 8 - t % 9 : 6 - (t - 27);
    return ((e & 1) ? 5u : 1u) * ((e & 2) ? 25u : 1u) * ((e & 4) ? 625u : 1u) * ((e & 8) ? 390625u : 1u);
}
MJD u64 sh_load(const MjTablesDev& T, int suit, u32 key) {
    return suit < 3 ? sh_row(T.suhai, T.n_suhai, key) : sh_row(T.jihai, T.n_jihai, key);
}
// The two table base pointers as values (wave-uniform scalar loads from the constant block, once per function): a probe
// with a lane-dependent suit then selects between two registers, instead of first gathering the POINTER from
// &T.suhai / &T.jihai with a per-lane load and only then the row (two dependent memory round trips per probe).
#define MJ_HBM __attribute__((address_space(1)))
struct ShTab {
    const MJ_HBM u64* su;
    const MJ_HBM u64* ji;
    u32 nsu, nji;
};
MJD ShTab sh_tab(const MjTablesDev& T) {
    ShTab t;
    t.su = (const MJ_HBM u64*)T.suhai;
    t.ji = (const MJ_HBM u64*)T.jihai;
    t.nsu = T.n_suhai;
    t.nji = T.n_jihai;
    return t;
}
MJD u64 sh_load(const ShTab& T, int suit, u32 key) {
    const MJ_HBM u64* tab = suit < 3 ? T.su : T.ji;
    const u32 n = suit < 3 ? T.nsu : T.nji;
    const u64 r = tab[key < n ? key : 0u];  // unconditional load (always a valid address), so gathers can be batched
    return key < n ? r : 0ull;
}
MJD ShBase sh_base(const ShTab& T, Hand h) {
    ShBase b;
    b.key[0] = suit_key9(h.mp);
    b.key[1] = suit_key9(h.mp >> 27);
    b.key[2] = suit_key9(h.sz);
    b.key[3] = suit_key7(h.sz >> 27);
#pragma unroll
    for (int i = 0; i < 4; i++) b.row[i] = sh_load(T, i, b.key[i]);
    b.pairs = h.n_pairs();
    b.kinds = h.n_kinds();
    b.kpairs = h.n_yao_pairs();
    b.kkinds = h.n_yao_kinds();
    return b;
}
MJD ShBase sh_base(const MjTablesDev& T, Hand h) {
    ShBase b;
    b.key[0] = suit_key9(h.mp);
    b.key[1] = suit_key9(h.mp >> 27);
    b.key[2] = suit_key9(h.sz);
    b.key[3] = suit_key7(h.sz >> 27);
#pragma unroll
    for (int i = 0; i < 4; i++) b.row[i] = sh_load(T, i, b.key[i]);
    b.pairs = h.n_pairs();
    b.kinds = h.n_kinds();
    b.kpairs = h.n_yao_pairs();
    b.kkinds = h.n_yao_kinds();
    return b;
}
// Min-plus merges of the per-suit rows are commutative and associative (a row = cheapest tile distance for j mentsu
// without / with the pair; sh_add_suhai is the convolution over (mentsu count, pair flag)), and every merged entry is
// bounded by the corresponding entry of either operand (row[0] == 0), so merged vectors still fit 10 nibbles.  Hence
// the shanten of a hand that differs from a base hand in ONE suit is  final(merge of the three untouched rows, new row)
// and in TWO suits  final(merge(merge of the two untouched rows, new row a), new row b):  the expensive full merges are
// shared by all probes of a state and a probe costs one `sh_final` (~35 ops) instead of two full merges + final (~330).
MJD u64 sh_merge(u64 a, u64 b, int m) {
    int v[10];
    sh_unpack(a, v);
    sh_add_suhai(v, b, m);
    u64 r = 0;
#pragma unroll
    for (int j = 0; j < 10; j++) r |= (u64)(v[j] & 15) << (4 * j);
    return r;
}
// Entry 5+m of merge(a, b) (m mentsu + the pair) = min over x = 0..m of  a[5+x] + b[m-x]  and  a[m-x] + b[5+x]
// (sh_add_jihai_final with its loop written out).  To keep every nibble index static for a run-time m, the low halves
// are shifted up by 4-m nibbles and the vacated nibbles filled with 15: term x then reads nibble 4-x, and the terms with
// x > m read the filler, i.e. are >= 15 and never below the x = m term (a[5+m] + b[0], b[0] = 0 in every row).
MJD int sh_final(u64 a, u64 b, int m) {
    const u32 s = 4u * (u32)(4 - m);
    const u32 fill = (1u << s) - 1u;
    const u32 as = (((u32)a & 0xFFFFFu) << s) | fill, bs = (((u32)b & 0xFFFFFu) << s) | fill;
    const u32 ah = (u32)(a >> 20), bh = (u32)(b >> 20);
    int r = 255;
#pragma unroll
    for (int x = 0; x < 5; x++) {
        const int f1 = (int)((ah >> (4 * x)) & 15u) + (int)((bs >> (4 * (4 - x))) & 15u);
        const int f2 = (int)((as >> (4 * (4 - x))) & 15u) + (int)((bh >> (4 * x)) & 15u);
        r = min(r, min(f1, f2));
    }
    return r;
}
// the same through the reference-shaped loop (shanten.rs:71-80); kept for the host-side equivalence check
MJD int sh_final_ref(u64 a, u64 b, int m) {
    int v[10];
    sh_unpack(a, v);
    return sh_add_jihai_final(v, b, m);
}
MJD int sh_pair_idx(int a, int b) {  // unordered suit pair -> 0..5 : 01 02 03 12 13 23
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    return lo == 0 ? hi - 1 : lo == 1 ? hi + 1 : 5;
}
// For every suit s the merge of the three OTHER suits' rows: 6 merges, after which any hand one tile away from the base
// hand costs one gather + sh_final (the step kernel's discard / wait scans; the SP kernel spreads the same merges over lanes).
struct ShOthers {
    u64 r[4];
    MJD u64 of(int s) const { return s == 0 ? r[0] : s == 1 ? r[1] : s == 2 ? r[2] : r[3]; }
};
MJD ShOthers sh_others(const ShBase& B, int m) {
    const u64 m01 = sh_merge(B.row[0], B.row[1], m), m23 = sh_merge(B.row[2], B.row[3], m);
    ShOthers o;
    o.r[0] = sh_merge(m23, B.row[1], m);
    o.r[1] = sh_merge(m23, B.row[0], m);
    o.r[2] = sh_merge(m01, B.row[3], m);
    o.r[3] = sh_merge(m01, B.row[2], m);
    return o;
}
// Which draws can lower the shanten number L of a hand at all?  A superset, per suit group i (0..2 number suits, 3 honours),
// as 3-bit-field masks (bit 3j = j-th tile of the group) from the group's non-zero-count fields `hn` of the hand:
//   standard form: a drawn tile only helps inside a block with a tile the hand already holds, i.e. the same tile or, in a
//     number suit, a tile at most two ranks away (a block made of new tiles only is never cheaper than one seeded by a
//     left-over tile, and 3k+1 tiles cannot all sit in k blocks);
//   seven pairs (closed hands): pairs up a single (a held tile), or ANY new kind while the hand has fewer than 7 kinds --
//     which lowers min(standard, pairs, orphans) only if the hand's seven-pairs number is already <= L;
//   thirteen orphans (closed hands): any terminal / honour -- only if the hand's orphans number is already <= L.
// A hand that holds all four copies of some tile is not pruned at all: the tables know that its left-over fourth copy cannot
// seed a pair, so there an unrelated new tile can be the cheaper seed (333p + EEEE: any draw makes it tenpai).
// Checked against calc_all on 1.5 M random hands of every size and shape class in tests/host/algo_check.hip (19 M once).
struct ShDrawRule {
    bool all, yao;  // probe every tile / add the terminals and honours
};
MJD ShDrawRule sh_draw_rule(int L, int len_div3, int pairs, int kinds, int kpairs, int kkinds, bool has_quad) {
    ShDrawRule r;
    const bool closed = len_div3 == 4;
    r.yao = closed && 13 - kkinds - (kpairs > 0) <= L;
    r.all = has_quad || (closed && kinds < 7 && 6 - pairs + (7 - kinds) <= L);
    return r;
}
MJD u32 sh_draw_candidate_fields(u32 hn, int group, ShDrawRule r) {
    constexpr u32 ALL9 = 0x1249249u, ALL7 = 0x49249u;
    u32 c = group < 3 ? (hn | (hn << 3) | (hn << 6) | (hn >> 3) | (hn >> 6)) & ALL9 : hn;
    if (r.yao) c |= group < 3 ? 0x1000001u : ALL7;
    if (r.all) c = group < 3 ? ALL9 : ALL7;
    return c;
}
// normal-form shanten from the final entry + chitoi / kokushi (shanten.rs:139-150)
MJD int sh_finish(int fin, int len_div3, int pairs, int kinds, int kpairs, int kkinds) {
    int s = fin - 1;
    if (s <= 0 || len_div3 < 4) return s;
    s = min(s, 7 - pairs + (kinds >= 7 ? 0 : 7 - kinds) - 1);
    if (s > 0) s = min(s, 14 - kkinds - (kpairs > 0) - 1);
    return s;
}

// ---------------------------------------------------------------- points (point.rs:13-112)
// The reference's match table equals the textbook formula on its whole domain (its own test,
// point.rs:121-153, asserts exactly that), so the device uses the closed form.
struct Point {
    int ron, tsumo_ko, tsumo_oya;
};
MJD int ceil100(int x) { return (x + 99) / 100 * 100; }
MJD Point point_calc(bool is_oya, int fu, int han) {
    int base;
    if (han >= 13) base = 8000;
    else if (han >= 11) base = 6000;
    else if (han >= 8) base = 4000;
    else if (han >= 6) base = 3000;
    else {
        base = fu << (han + 2);
        if (han >= 5 || base >= 2000) base = 2000;
    }
    Point p;
    if (is_oya) {
        p.ron = ceil100(base * 6);
        p.tsumo_ko = ceil100(base * 2);
        p.tsumo_oya = 0;
    } else {
        p.ron = ceil100(base * 4);
        p.tsumo_ko = ceil100(base);
        p.tsumo_oya = ceil100(base * 2);
    }
    return p;
}
MJD Point point_yakuman(bool is_oya, int n) {
    Point p;
    if (is_oya) { p.ron = 48000 * n; p.tsumo_ko = 16000 * n; p.tsumo_oya = 0; }
    else { p.ron = 32000 * n; p.tsumo_ko = 8000 * n; p.tsumo_oya = 16000 * n; }
    return p;
}
MJD int tsumo_total(Point p, bool is_oya) { return is_oya ? p.tsumo_ko * 3 : p.tsumo_ko * 2 + p.tsumo_oya; }

// ---------------------------------------------------------------- agari (agari.rs)
// Everything below lives in registers: the up to four sets of a hand (and its up to 14 distinct tiles) are byte lists packed in
// 32 / 64-bit words, never arrays — a dynamically indexed local array is scratch memory, and round 2's array-based version
// spent most of the level-0 scoring pass of the SP kernel waiting for it.
#define PK8(p, i) ((int)(((p) >> (8 * (i))) & 0xFFu))
struct Melds {  // own open/closed sets, deaka'd tile ids (lowest tile for chi), one byte per set
    u32 chis, pons, minkans, ankans;
    u8 n_chis, n_pons, n_minkans, n_ankans;
};
MJD void melds_put(u32& list, int i, int tile) { list = (list & ~(0xFFu << (8 * i))) | ((u32)(tile & 0xFF) << (8 * i)); }
struct AgariIn {
    Hand tehai;  // 3n+2 incl. the winning tile
    Melds m;
    bool is_menzen;
    int bakaze, jikaze, winning_tile;
    bool is_ron;
};
struct Agari {
    int kind;  // 0 none, 1 normal, 2 yakuman
    int fu, han;  // yakuman: han = count
};
MJD bool agari_better(Agari a, Agari b) {  // a > b per agari.rs:180-195
    if (a.kind == 2 && b.kind == 2) return a.han > b.han;
    if (a.kind == 2) return true;
    if (b.kind == 2) return false;
    if (a.han != b.han) return a.han > b.han;
    return a.fu > b.fu;
}

// agari.rs:767-838: the distinct tiles in ascending id (byte i of lo | hi << 64) + the run-length key.  The reference walks all
// 34 kinds and closes a run at the first empty kind, at a suit boundary and after every honour; walking only the held kinds,
// a run is closed right before the next one opens and once at the end — the same bit positions.
struct Tile14 {
    u64 lo, hi;
    MJD int at(u32 i) const { return (int)(((i < 8 ? lo : hi) >> (8 * (i & 7))) & 0xFF); }
};
MJD u32 tile14_and_key(Hand h, Tile14& t14) {
    t14.lo = t14.hi = 0;
    u32 key = 0, n14 = 0;
    int bit = -1, prev = -2;
    for (u64 m = h.nonzero_mask(); m; m &= m - 1) {
        const int t = __ffsll((long long)m) - 1, c = h.get(t);
        const bool same_run = t < 27 && t == prev + 1 && (t % 9) != 0;
        if (prev >= 0 && !same_run) {  // close the previous run
            key |= 1u << bit;
            bit += 1;
        }
        if (n14 < 8) t14.lo |= (u64)t << (8 * n14);
        else t14.hi |= (u64)t << (8 * (n14 - 8));
        n14++;
        bit += 1;
        if (c == 2) { key |= 0b11u << bit; bit += 2; }
        else if (c == 3) { key |= 0b1111u << bit; bit += 4; }
        else if (c == 4) { key |= 0b111111u << bit; bit += 6; }
        prev = t;
    }
    if (prev >= 0) key |= 1u << bit;
    return key;
}
MJD int agari_find(const MjTablesDev& T, u32 key) {  // index or -1; hashed (1-2 gathers instead of a 14-step bisection)
    u32 pos = (key * 0x9E3779B1u) >> 17;
    for (int probe = 0; probe < 32768; probe++) {
        const u64 e = T.agari_hash[pos];
        if (e == 0ull) return -1;
        if ((u32)(e >> 32) == key) return (int)(u32)e - 1;
        pos = (pos + 1) & 32767;
    }
    return -1;
}

// One decomposition of the concealed part + the caller's melds (agari.rs:100-124, 287-761).
struct DivWork {
    u32 kotsu;      // byte list: menzen kotsu, then pons, minkans, ankans (at most four sets in all)
    u32 shuntsu;    // byte list: menzen shuntsu, then chis
    int n_mk, n_kotsu, n_ms, n_shuntsu;
    int pair_tile;
    bool has_chitoi, has_chuuren, has_ittsuu, has_ryanpeikou, has_ipeikou;
    bool wt_minkou;  // winning_tile_makes_minkou
};
MJD DivWork div_init(const AgariIn& in, const Tile14& t14, u32 v) {  // agari.rs:126-157, 288-338
    DivWork w;
    w.pair_tile = t14.at((v >> 6) & 15);
    const int nk = v & 7, ns = (v >> 3) & 7;
    w.n_mk = nk;
    w.n_ms = ns;
    w.kotsu = w.shuntsu = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (i < nk) melds_put(w.kotsu, i, t14.at((v >> (10 + i * 4)) & 15));
        if (i < ns) melds_put(w.shuntsu, i, t14.at((v >> (10 + (nk + i) * 4)) & 15));
    }
    int k = nk;
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (i < (int)in.m.n_pons && k < 4) melds_put(w.kotsu, k++, PK8(in.m.pons, i));
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (i < (int)in.m.n_minkans && k < 4) melds_put(w.kotsu, k++, PK8(in.m.minkans, i));
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (i < (int)in.m.n_ankans && k < 4) melds_put(w.kotsu, k++, PK8(in.m.ankans, i));
    w.n_kotsu = k;
    k = ns;
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (i < (int)in.m.n_chis && k < 4) melds_put(w.shuntsu, k++, PK8(in.m.chis, i));
    w.n_shuntsu = k;
    w.has_chitoi = (v >> 26) & 1;
    w.has_chuuren = (v >> 27) & 1;
    w.has_ittsuu = (v >> 28) & 1;
    w.has_ryanpeikou = (v >> 29) & 1;
    w.has_ipeikou = (v >> 30) & 1;
    // winning_tile_makes_minkou (agari.rs:314-338)
    bool r = false;
    if (in.is_ron) {
        bool in_kotsu = false;
#pragma unroll
        for (int i = 0; i < 4; i++) in_kotsu |= i < nk && PK8(w.kotsu, i) == in.winning_tile;
        if (in_kotsu) {
            if (in.winning_tile >= 27) r = true;
            else {
                const int kind = in.winning_tile / 9, num = in.winning_tile % 9;
                const int low = kind * 9 + (num >= 2 ? num - 2 : 0), high = kind * 9 + min(num, 6);
                bool covered = false;
#pragma unroll
                for (int i = 0; i < 4; i++) covered |= i < ns && PK8(w.shuntsu, i) >= low && PK8(w.shuntsu, i) <= high;
                r = !covered;
            }
        }
    }
    w.wt_minkou = r;
    return w;
}
MJD int div_calc_fu(const AgariIn& in, const DivWork& w, bool has_pinfu) {  // agari.rs:367-452
    if (w.has_chitoi) return 25;
    int fu = 20;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (i >= w.n_mk) continue;
        const int t = PK8(w.kotsu, i);
        const bool mink = w.wt_minkou && t == in.winning_tile, yao = is_yaokyuu(t);
        fu += (!mink && yao) ? 8 : (mink && !yao) ? 2 : 4;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (i < (int)in.m.n_pons) fu += is_yaokyuu(PK8(in.m.pons, i)) ? 4 : 2;
        if (i < (int)in.m.n_ankans) fu += is_yaokyuu(PK8(in.m.ankans, i)) ? 32 : 16;
        if (i < (int)in.m.n_minkans) fu += is_yaokyuu(PK8(in.m.minkans, i)) ? 16 : 8;
    }
    const int pt = w.pair_tile;
    if (pt >= T_P && pt <= T_C) fu += 2;
    else {
        if (pt == in.bakaze) fu += 2;
        if (pt == in.jikaze) fu += 2;
    }
    if (fu == 20) {
        if (!in.is_menzen) return 30;
        if (has_pinfu) return in.is_ron ? 30 : 20;
        return in.is_ron ? 40 : 30;
    }
    if (!in.is_ron) fu += 2;
    else if (in.is_menzen) fu += 10;
    if (!w.wt_minkou) {
        if (pt == in.winning_tile) fu += 2;
        else {
            bool kp = false;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (i >= w.n_ms) continue;
                const int s = PK8(w.shuntsu, i);
                kp |= s + 1 == in.winning_tile || (s % 9 == 0 && s + 2 == in.winning_tile) || (s % 9 == 6 && s == in.winning_tile);
            }
            if (kp) fu += 2;
        }
    }
    return ((fu - 1) / 10 + 1) * 10;
}

// Yaku search of one decomposition.  `any_only`: the caller only wants to know whether there is a yaku (has_yaku path) — every
// check only ever adds, so the reference's early returns change nothing but the time; fu is skipped then.
// `hand_nz`: the concealed hand's held kinds (the seven pairs of a chitoi decomposition).
MJD Agari div_search_yakus(const AgariIn& in, const DivWork& w, u64 hand_nz, bool any_only) {  // agari.rs:454-761
    int han = 0, yakuman = 0;
    const int pt = w.pair_tile;
    const bool pair_is_dragon = pt >= T_P && pt <= T_C;
    bool has_pinfu = false;
    if (w.n_ms == 4 && !pair_is_dragon && pt != in.bakaze && pt != in.jikaze) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int s = PK8(w.shuntsu, i), num = s % 9 + 1;
            has_pinfu |= (num <= 6 && s == in.winning_tile) || (num >= 2 && s + 2 == in.winning_tile);
        }
    }
    if (has_pinfu) han += 1;
    if (w.has_chitoi) han += 2;
    if (w.has_ryanpeikou) han += 3;
    if (w.has_chuuren) yakuman += 1;
    // tile sets as bit masks over the 34 kinds: kotsu (+ pair), shuntsu starts
    u64 kmask = 0;
    u32 s_bits = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (i < w.n_kotsu) kmask |= 1ull << PK8(w.kotsu, i);
        if (i < w.n_shuntsu) s_bits |= 1u << PK8(w.shuntsu, i);  // tile ids < 27
    }
    constexpr u64 SIMPLES = 0x7FFFFFFull & ~((1ull << 0) | (1ull << 8) | (1ull << 9) | (1ull << 17) | (1ull << 18) | (1ull << 26));  // 2..8 of each suit
    constexpr u64 HONOURS = 0x7Full << 27;
    constexpr u32 SEQ_SIMPLE = 0b000111110u | (0b000111110u << 9) | (0b000111110u << 18);  // sequences 234 .. 678 (start 2..6)
    {
        bool has_tanyao, all_yao, yao_jihai;
        u32 kinds_seen;  // bit kind (0..2) for suited, bit 3 for honours, over every set + pair
        const u64 solid = w.has_chitoi ? hand_nz : (kmask | (1ull << pt));  // the tiles of the pairs / triplets / the head
        const u32 seqs = w.has_chitoi ? 0u : s_bits;
        has_tanyao = (solid & ~SIMPLES) == 0 && (seqs & ~SEQ_SIMPLE) == 0;
        all_yao = (solid & ~YAOKYUU_MASK) == 0;
        yao_jihai = (solid & HONOURS) != 0;
        kinds_seen = ((solid & 0x1FFull) ? 1u : 0u) | ((solid & (0x1FFull << 9)) ? 2u : 0u) | ((solid & (0x1FFull << 18)) ? 4u : 0u) |
                     ((solid & HONOURS) ? 8u : 0u) | ((seqs & 0x1FFu) ? 1u : 0u) | ((seqs & (0x1FFu << 9)) ? 2u : 0u) | ((seqs & (0x1FFu << 18)) ? 4u : 0u);
        if (has_tanyao) han += 1;
        const bool has_toitoi = !w.has_chitoi && w.n_ms == 0 && in.m.n_chis == 0;
        if (has_toitoi) han += 2;
        // 字一色 / 混一色 / 清一色 (agari.rs:533-571)
        {
            const u32 suits = kinds_seen & 7;
            const bool has_jihai = (kinds_seen >> 3) & 1;
            if (suits == 0) yakuman += 1;
            else if ((suits & (suits - 1)) == 0) han += (has_jihai ? 2 : 5) + (in.is_menzen ? 1 : 0);
        }
        if (!w.has_chitoi) {
            // 一盃口 (agari.rs:573-596)
            if (w.has_ipeikou) han += 1;
            else if (in.m.n_ankans > 0 && in.is_menzen && w.n_ms >= 2) {
                u32 marks = 0;
                bool dup = false;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (i >= w.n_ms) continue;
                    const u32 b = 1u << PK8(w.shuntsu, i);
                    dup |= (marks & b) != 0;
                    marks |= b;
                }
                if (dup) han += 1;
            }
            // 一気通貫 (agari.rs:598-619)
            if (in.is_menzen && w.has_ittsuu) han += 2;
            else if (in.m.n_chis == 0 && w.has_ittsuu) han += 1;
            else if (w.n_shuntsu >= 3) {
                bool itt = false;
#pragma unroll
                for (int k = 0; k < 3; k++) itt |= ((s_bits >> (9 * k)) & 0b1001001u) == 0b1001001u;
                if (itt) han += 1;
            }
            // 三色同順 / 三色同刻 (agari.rs:621-647)
            {
                const u32 k_bits = (u32)(kmask & 0x7FFFFFFull);
                const u32 s3 = s_bits & (s_bits >> 9) & (s_bits >> 18) & 0x1FF;
                const u32 k3 = k_bits & (k_bits >> 9) & (k_bits >> 18) & 0x1FF;
                if (s3) han += in.is_menzen ? 2 : 1;
                else if (k3) han += 2;
            }
            // 暗刻 / 槓子 (agari.rs:649-666)
            const int ankous = in.m.n_ankans + w.n_mk - (w.wt_minkou ? 1 : 0);
            if (ankous == 4) yakuman += 1;
            else if (ankous == 3) han += 2;
            const int kans = in.m.n_ankans + in.m.n_minkans;
            if (kans == 4) yakuman += 1;
            else if (kans == 3) han += 2;
            // 緑一色 (agari.rs:668-676)
            {
                constexpr u64 GREEN = (1ull << 19) | (1ull << 20) | (1ull << 21) | (1ull << 23) | (1ull << 25) | (1ull << T_F);
                const bool g = ((kmask | (1ull << pt)) & ~GREEN) == 0 && (s_bits & ~(1u << 19)) == 0;
                if (g) yakuman += 1;
            }
            if (!has_tanyao) {  // 役牌, 三元, 四喜 (agari.rs:678-721)
                const u32 jz = (u32)(kmask >> 27) & 0x7F;
                if ((jz >> (in.bakaze - 27)) & 1) han += 1;
                if ((jz >> (in.jikaze - 27)) & 1) han += 1;
                const int saneins = __popc(jz & 0b1110000);
                if (saneins > 0) {
                    han += saneins;
                    if (saneins == 3) yakuman += 1;
                    else if (saneins == 2 && pair_is_dragon) han += 2;
                }
                const int winds = __popc(jz & 0b0001111);
                if (winds == 4) yakuman += 1;
                else if (winds == 3 && pt >= T_E && pt <= T_N) yakuman += 1;
            }
        }
        if (!has_tanyao && all_yao) {  // 老頭 / 全帯 (agari.rs:724-757)
            if (w.has_chitoi || has_toitoi) {
                if (yao_jihai) han += 2;
                else yakuman += 1;
            } else {
                constexpr u32 SEQ_TERMINAL = 0b001000001u | (0b001000001u << 9) | (0b001000001u << 18);  // 123 / 789
                if ((s_bits & ~SEQ_TERMINAL) == 0) han += (yao_jihai ? 1 : 2) + (in.is_menzen ? 1 : 0);
            }
        }
    }
    Agari a;
    if (yakuman > 0) { a.kind = 2; a.fu = 0; a.han = yakuman; }
    else if (han > 0) { a.kind = 1; a.han = han; a.fu = (any_only || han >= 5) ? 0 : div_calc_fu(in, w, has_pinfu); }
    else { a.kind = 0; a.fu = a.han = 0; }
    return a;
}

// agari.rs:260-288.  any_only => has_yaku()
MJDN Agari agari_search(const MjTablesDev& T, const AgariIn& in, bool any_only) {
    Agari best = {0, 0, 0};
    if (in.is_menzen && calc_kokushi(in.tehai) == -1) {
        best.kind = 2;
        best.han = 1;
        return best;
    }
    Tile14 t14;
    const u32 key = tile14_and_key(in.tehai, t14);
    const int idx = agari_find(T, key);
    if (idx < 0) return best;
    const u32* rec = T.agari_divs + (size_t)idx * 5;
    const int n = (int)rec[0];
    const u64 hand_nz = in.tehai.nonzero_mask();
    for (int i = 0; i < n; i++) {
        const DivWork w = div_init(in, t14, rec[1 + i]);
        const Agari a = div_search_yakus(in, w, hand_nz, any_only);
        if (a.kind == 0) continue;
        if (any_only) return a;
        if (best.kind == 0 || !agari_better(best, a)) best = a;
    }
    return best;
}
// agari.rs:228-258
MJDN Agari agari_full(const MjTablesDev& T, const AgariIn& in, int additional_hans, int doras) {
    Agari a = agari_search(T, in, false);
    if (a.kind == 1) { a.han += additional_hans + doras; return a; }
    if (a.kind == 2) return a;
    Agari none = {0, 0, 0};
    if (additional_hans == 0) return none;
    if (additional_hans + doras >= 5) { Agari r = {1, 0, additional_hans + doras}; return r; }
    Tile14 t14;
    const u32 key = tile14_and_key(in.tehai, t14);
    const int idx = agari_find(T, key);
    if (idx < 0) return none;
    const u32* rec = T.agari_divs + (size_t)idx * 5;
    int fu = -1;
    for (int i = 0; i < (int)rec[0]; i++) fu = max(fu, div_calc_fu(in, div_init(in, t14, rec[1 + i]), false));
    if (fu < 0) return none;
    Agari r = {1, fu, additional_hans + doras};
    return r;
}
MJD Point agari_point(Agari a, bool is_oya) { return a.kind == 2 ? point_yakuman(is_oya, a.han) : point_calc(is_oya, a.fu, a.han); }

// agari.rs:854-912 with strict = false (the only mode the state machine uses, update.rs:277-278)
MJDN bool check_ankan_after_riichi(const MjTablesDev& T, Hand tehai, int len_div3, int tile) {
    int tid = deaka(tile);
    if (tehai.get(tid) != 4) return false;
    if (tid >= 27) return true;
    Hand before = tehai;
    before.dec(tid);
    for (int t = 0; t < 34; t++) {
        if (before.get(t) == 4) continue;
        Hand tmp = before;
        tmp.inc(t);
        if (calc_all(T, tmp, len_div3) != -1) continue;
        if (t == tid) return false;
        Hand after = tehai;
        after.clear(tid);
        after.inc(t);
        Tile14 t14;
        if (agari_find(T, tile14_and_key(after, t14)) < 0) return false;
    }
    return true;
}
