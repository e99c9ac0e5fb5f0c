// Shanten by table id: the min-plus algebra of algo/shanten.rs:51-102 lives on a tiny closed set of vectors.
//
// The reference's tables (shanten.rs:11-44) map a suit key (5^9 / 5^7 count patterns) to a ROW of 10 distances
// (j mentsu without / with the pair).  Only 126 distinct rows occur among the 1.94 M number-suit keys and 55 among the honour
// keys (130 in total, the all-zero row of `unwrap_or_default` for keys past the table included), and the set of vectors
// reachable from them by the min-plus merge `add_suhai` (shanten.rs:51-69) is CLOSED at 180 elements.  So, instead of
// unpacking nibbles and taking ~45 minima per merge, a hand's normal-form shanten is a walk through three small tables:
//   id      : suit key -> row id                                   (1 byte per key)
//   mrg     : (vector id, row id) -> id of their merge              (180 x 130 bytes)
//   opt[m]  : (vector id, row id) -> the final value `add_jihai`-style (shanten.rs:71-80) for m = len_div3, together with
//             WHICH entries of the row attain that minimum (the optimal ways to split the m mentsu + pair between this suit
//             and the other three).
// With the optimal entries known, "which tiles of this suit lower the hand's shanten number when drawn" and "which can be
// discarded without raising it" are unions of per-key, per-entry 9-bit tile masks:
//   wait[key][e] = tiles t of the suit with row(key + t)[e] == row(key)[e] - 1
//   keep[key][e] = tiles d held in the suit with row(key - d)[e] == row(key)[e]
// (one tile changes a row entry by at most one: checked over the whole table at build time) — draw t lowers the normal-form
// number N of the hand iff t is in wait[e] for an entry e that is optimal against the merge of the OTHER three suits, and
// discard d keeps it iff d is in keep[e] for such an e.  The SP kernel (mj_sp.hip) gets a state's required draws and, after each
// of them, its shanten-keeping discards from 4 such unions each instead of ~40 incremental probes of ~165 instructions.
// Seven pairs / thirteen orphans enter through the closed forms of sp_req_set / sp_keep_set below (the exact case analysis of
// calc_all, shanten.rs:139-150).  tests/host/algo_check.hip checks both sets against brute-force calc_all loops.
//
// The only keys whose records cannot be trusted are those with a neighbour (one more tile) past the end of the table, where the
// reference reads an all-zero row: 8 reachable 13-tile patterns (3,4,4,1,1 ...).  Their records carry SPT_FALLBACK and the
// callers take the brute-force path.
#pragma once
#include "mj_algo.h"

#define SPT_NB 136  // row-id stride of mrg / opt (>= number of distinct rows, 130)
#define SPT_NV 192  // vector ids (>= closure size, 180)
#define SPT_FALLBACK 1u

struct alignas(16) SpRec {
    u32 x, y, z, w;
};
// wait record of a key: x / y / z = the 9-bit tile masks of entries 1..3 / 4..6 / 7..9 (entry e at bits 9 * ((e - 1) % 3)),
//                       w = flags (SPT_FALLBACK)
// keep record of a key: the same for keep[], w = keep[0] (every held tile: entry 0 is the empty split, distance 0)
// opt record          : x / y / z = 0x1FF in the field of every entry that attains the final minimum, x bits 27..30 = the final
//                       value (normal-form shanten + 1), w = 0x1FF if entry 0 attains it
struct SpTabDev {
    const u8* id;        // [n_su + n_ji]: number-suit keys, then honour keys (ONE array: a lane-dependent suit selects an offset,
                         //                not a pointer — hipcc turns a select between two pointers of a local struct into a scratch array)
    const u8* mrg;       // [SPT_NV][SPT_NB]
    const SpRec* opt;    // [5][SPT_NV][SPT_NB]
    const SpRec* wk;     // [n_su + n_ji][2]: wait, keep
    u32 n_su, n_ji;
    u32 zero_id;         // id of the all-zero row (keys past the table)
};

// the same table block with explicit global-address-space pointers (kernels: global_load instead of flat_load)
struct SpTabG {
    const MJ_HBM u8* id;
    const MJ_HBM u8* mrg;
    const MJ_HBM SpRec* opt;
    const MJ_HBM SpRec* wk;
    u32 n_su, n_ji, zero_id;
};
MJD SpTabG sp_tab_g(const SpTabDev& T) {
    SpTabG g;
    g.id = (const MJ_HBM u8*)T.id;
    g.mrg = (const MJ_HBM u8*)T.mrg;
    g.opt = (const MJ_HBM SpRec*)T.opt;
    g.wk = (const MJ_HBM SpRec*)T.wk;
    // the three scalars pass through readfirstlane: an opaque value in an SGPR.  Otherwise hipcc folds a lane-dependent
    // `suit < 3 ? n_su : n_ji` over two loads into ONE load from a selected address and keeps a copy of the block in scratch.
#if defined(__HIP_DEVICE_COMPILE__)
    g.n_su = (u32)__builtin_amdgcn_readfirstlane((int)T.n_su);
    g.n_ji = (u32)__builtin_amdgcn_readfirstlane((int)T.n_ji);
    g.zero_id = (u32)__builtin_amdgcn_readfirstlane((int)T.zero_id);
#else
    g.n_su = T.n_su;
    g.n_ji = T.n_ji;
    g.zero_id = T.zero_id;
#endif
    return g;
}

// ---- device / host-testable lookups (TT = SpTabDev or SpTabG)
template <class TT> MJD u32 spt_id(const TT& T, int suit, u32 key) {
    // the fields are read BEFORE the lane-dependent selects: with the loads inside the arms hipcc sinks them into one load
    // from a selected ADDRESS, which pins the whole table block in scratch memory
    const u32 n_su = T.n_su, n_ji = T.n_ji, zero = T.zero_id;
    const u32 n = suit < 3 ? n_su : n_ji;
    const u32 v = T.id[(key < n ? key : 0u) + (suit < 3 ? 0u : n_su)];  // unconditional load (always a valid address)
    return key < n ? v : zero;
}
template <class TT> MJD u32 spt_merge(const TT& T, u32 vec, u32 row) { return T.mrg[vec * SPT_NB + row]; }
template <class R> MJD SpRec spt_load(const R* p) {  // one 16-byte load
    SpRec r;
    r.x = p->x; r.y = p->y; r.z = p->z; r.w = p->w;
    return r;
}
template <class TT> MJD SpRec spt_opt(const TT& T, int m, u32 vec, u32 row) { return spt_load(&T.opt[((u32)m * SPT_NV + vec) * SPT_NB + row]); }
MJD int spt_fin(const SpRec& o) { return (int)((o.x >> 27) & 15u); }
// which = 0 wait, 1 keep; key must be inside the table
template <class TT> MJD SpRec spt_rec(const TT& T, int suit, u32 key, int which) {
    const u32 n_su = T.n_su;
    return spt_load(&T.wk[(size_t)(key + (suit < 3 ? 0u : n_su)) * 2 + which]);
}
template <class TT> MJD bool spt_in_table(const TT& T, int suit, u32 key) {
    const u32 n_su = T.n_su, n_ji = T.n_ji;
    return key < (suit < 3 ? n_su : n_ji);
}
MJD u32 spt_fold27(u32 v) { return (v | (v >> 9) | (v >> 18)) & 0x1FFu; }
MJD u32 spt_wait_tiles(const SpRec& wait, const SpRec& opt) { return spt_fold27((wait.x & opt.x) | (wait.y & opt.y) | (wait.z & opt.z)); }  // wait.x < 2^27
MJD u32 spt_keep_tiles(const SpRec& keep, const SpRec& opt) {
    return spt_fold27((keep.x & opt.x) | (keep.y & opt.y) | (keep.z & opt.z)) | (keep.w & opt.w);
}

// 34-bit tile sets of a packed hand by count class
struct SpCountSets {
    u64 c1, c2, nz;  // count == 1, count == 2, count >= 1
};
MJD SpCountSets sp_count_sets(Hand h) {
    const u64 L3 = Hand::LSB3;
    const u64 a0 = h.mp & L3, a1 = (h.mp >> 1) & L3, a2 = (h.mp >> 2) & L3;
    const u64 b0 = h.sz & L3, b1 = (h.sz >> 1) & L3, b2 = (h.sz >> 2) & L3;
    SpCountSets s;
    s.c1 = Hand::compress3(a0 & ~a1 & ~a2) | (Hand::compress3(b0 & ~b1 & ~b2) << 18);
    s.c2 = Hand::compress3(a1 & ~a0 & ~a2) | (Hand::compress3(b1 & ~b0 & ~b2) << 18);
    s.nz = Hand::compress3(a0 | a1 | a2) | (Hand::compress3(b0 | b1 | b2) << 18);
    return s;
}
constexpr u64 SP_ALL34 = (1ull << 34) - 1;
// The same per suit group, in 32-bit arithmetic: the group's nine 3-bit count fields (bit 3j = tile j) -> 9-bit tile masks.
constexpr __host__ __device__ u32 sp_compress9(u32 x) {  // bit 3j -> bit j, j = 0..8
    x &= 0x01249249u;
    x = (x ^ (x >> 2)) & 0x010C30C3u;
    x = (x ^ (x >> 4)) & 0x0100F00Fu;
    x = (x ^ (x >> 8)) & 0x010000FFu;
    return (x ^ (x >> 16)) & 0x1FFu;
}
constexpr bool sp_compress9_ok() {
    for (u32 m = 0; m < 512; m++) {
        u32 x = 0;
        for (int j = 0; j < 9; j++)
            if ((m >> j) & 1) x |= 1u << (3 * j);
        if (sp_compress9(x | 0xFE000000u) != m) return false;
    }
    return true;
}
static_assert(sp_compress9_ok(), "sp_compress9");
// fields = the group's 27 bits of a packed hand: count == 1 | count == 2 << 9 | count >= 1 << 18
MJD u32 sp_group_count_sets(u32 fields) {
    const u32 M = 0x01249249u, b0 = fields & M, b1 = (fields >> 1) & M, b2 = (fields >> 2) & M;
    return sp_compress9(b0 & ~b1 & ~b2) | (sp_compress9(b1 & ~b0 & ~b2) << 9) | (sp_compress9(b0 | b1 | b2) << 18);
}
MJD u32 sp_group_fields(Hand h, int group) {  // the 27 (honours: 21) field bits of suit group 0..3
    const u64 w = group < 2 ? h.mp : h.sz;
    return (u32)(w >> ((group & 1) * 27)) & 0x7FFFFFFu;
}
// count classes of h + t from those of h (hc = copies of t in h)
MJD SpCountSets sp_count_sets_add(SpCountSets cs, int t, int hc) {
    const u64 bit = 1ull << t;
    if (hc == 0) { cs.c1 |= bit; cs.nz |= bit; }
    else if (hc == 1) { cs.c1 &= ~bit; cs.c2 |= bit; }
    else if (hc == 2) cs.c2 &= ~bit;
    return cs;
}

// calc_all's combination of the three forms (shanten.rs:139-150), closed hands (len_div3 == 4)
MJD int sp_finish3(int sn, int c, int k) {
    if (sn <= 0) return sn;
    const int s = min(sn, c);
    return s > 0 ? min(s, k) : s;
}
// Generic (slow) path: the tiles whose form changes (a: normal, b: seven pairs, g: orphans; each by `delta`) give the target.
MJD u64 sp_form_sets_slow(int sn, int c, int k, int delta, int target, u64 A, u64 B, u64 G) {
    u64 r = 0;
    for (int q = 0; q < 8; q++) {
        const int fa = q & 1, fb = (q >> 1) & 1, fg = (q >> 2) & 1;
        if (sp_finish3(sn + fa * delta, c + fb * delta, k + fg * delta) != target) continue;
        r |= (fa ? A : ~A) & (fb ? B : ~B) & (fg ? G : ~G);
    }
    return r & SP_ALL34;
}
// Draws t with calc_all(h + t) == L - 1 for a 3n+1 hand h with calc_all(h) == L (state.rs:128-173 `shanten_diff == -1`).
//   fin     : normal-form final value of h (shanten + 1);  waitN : tiles that lower the normal-form number
//   pairs.. : the seven-pairs / orphans counters of h (shanten.rs:104-137);  cs : count classes of h
// The caller still has to intersect with the tiles left in the wall.
MJD u64 sp_req_set(int len_div3, int L, int fin, u64 waitN, int pairs, int kinds, int kpairs, int kkinds, const SpCountSets& cs) {
    const int sn = fin - 1;
    if (len_div3 < 4) return sn == L ? waitN : 0ull;
    const int c = 7 - pairs + (kinds >= 7 ? 0 : 7 - kinds) - 1, k = 14 - kkinds - (kpairs > 0) - 1;
    const u64 H0 = ~cs.nz & SP_ALL34;
    const u64 B = cs.c1 | (kinds < 7 ? H0 : 0ull);                                      // seven-pairs number drops by one
    const u64 G = (YAOKYUU_MASK & H0) | (kpairs == 0 ? (YAOKYUU_MASK & cs.c1) : 0ull);  // orphans number drops by one
    if (sn < L || c < L || k < L) return sp_form_sets_slow(sn, c, k, -1, L - 1, waitN, B, G);
    // x = sn - a(t) > 0  /  y = c - b(t) > 0
    const u64 Pn = sn >= 2 ? SP_ALL34 : sn == 1 ? ~waitN : 0ull;
    const u64 Pc = c >= 2 ? SP_ALL34 : c == 1 ? ~B : 0ull;
    u64 r = sn == L ? waitN : 0ull;
    if (c == L) r |= B & Pn;
    if (k == L) r |= G & Pn & Pc;
    return r & SP_ALL34;
}
// Discards d with calc_all(g - d) == T for a 3n+2 hand g with calc_all(g) == T (state.rs:100-126 `shanten_diff == 0`).
//   fin / keepN : normal-form final value of g, tiles whose removal keeps the normal-form number; counters / classes of g
MJD u64 sp_keep_set(int len_div3, int T, int fin, u64 keepN, int pairs, int kinds, int kpairs, int kkinds, const SpCountSets& cs) {
    const int sn = fin - 1;
    if (len_div3 < 4) return sn == T ? keepN & cs.nz : 0ull;
    const int c = 7 - pairs + (kinds >= 7 ? 0 : 7 - kinds) - 1, k = 14 - kkinds - (kpairs > 0) - 1;
    const u64 Bw = cs.c2 | (kinds <= 7 ? cs.c1 : 0ull);                                        // seven-pairs number rises by one
    const u64 Gw = (YAOKYUU_MASK & cs.c1) | (kpairs == 1 ? (YAOKYUU_MASK & cs.c2) : 0ull);     // orphans number rises by one
    const u64 Aw = ~keepN & SP_ALL34;                                                          // normal-form number rises by one
    if (sn < T || c < T || k < T) return sp_form_sets_slow(sn, c, k, +1, T, Aw, Bw, Gw) & cs.nz;
    const u64 Pn = sn >= 1 ? SP_ALL34 : sn == 0 ? Aw : 0ull;  // x = sn + a(d) > 0
    const u64 Pc = c >= 1 ? SP_ALL34 : c == 0 ? Bw : 0ull;    // y = c + b(d) > 0
    u64 r = sn == T ? keepN : 0ull;
    if (c == T) r |= ~Bw & Pn;
    if (k == T) r |= ~Gw & Pn & Pc;
    return r & cs.nz;
}

// ---- whole-hand forms of the two sets (one thread does all four suits): the row set-up of the SP kernel, the fallback of its
// chunk passes, and the subject of tests/host/algo_check.hip.
struct SpSuitView {
    u32 key[4], id[4];
    int pairs, kinds, kpairs, kkinds;
};
template <class TT> MJD SpSuitView sp_suit_view(const TT& T, Hand h) {
    SpSuitView v;
    v.key[0] = suit_key9(h.mp);
    v.key[1] = suit_key9(h.mp >> 27);
    v.key[2] = suit_key9(h.sz);
    v.key[3] = suit_key7(h.sz >> 27);
#pragma unroll
    for (int i = 0; i < 4; i++) v.id[i] = spt_id(T, i, v.key[i]);
    v.pairs = h.n_pairs();
    v.kinds = h.n_kinds();
    v.kpairs = h.n_yao_pairs();
    v.kkinds = h.n_yao_kinds();
    return v;
}
// id of the merge of the three suits other than s
template <class TT> MJD u32 sp_others_id(const TT& T, const u32 id[4], int s) {
    const int a = s == 0 ? 1 : 0, b = s <= 1 ? 2 : 1, c = s == 3 ? 2 : 3;
    return spt_merge(T, spt_merge(T, id[a], id[b]), id[c]);
}
// brute force (the reference's own loops, state.rs:100-173); used when a key's record carries SPT_FALLBACK
MJD u64 sp_req_brute(const MjTablesDev& MT, Hand h, int len_div3, int L) {
    u64 r = 0;
    for (int t = 0; t < 34; t++) {
        if (h.get(t) >= 4) continue;
        Hand x = h;
        x.inc(t);
        if (calc_all(MT, x, len_div3) == L - 1) r |= 1ull << t;
    }
    return r;
}
MJD u64 sp_keep_brute(const MjTablesDev& MT, Hand g, int len_div3, int Tg) {
    u64 r = 0;
    for (int d = 0; d < 34; d++) {
        if (g.get(d) == 0) continue;
        Hand x = g;
        x.dec(d);
        if (calc_all(MT, x, len_div3) == Tg) r |= 1ull << d;
    }
    return r;
}
// required draws of a 3n+1 hand with calc_all(h) == L (wall not applied); *fin_out = normal-form final value
template <class TT> MJD u64 sp_req_of_hand(const TT& T, const MjTablesDev& MT, Hand h, int len_div3, int L, int* fin_out = nullptr) {
    const SpSuitView v = sp_suit_view(T, h);
    u64 waitN = 0;
    int fin = 0;
    bool fallback = false;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        if (!spt_in_table(T, s, v.key[s])) { fallback = true; continue; }
        const SpRec o = spt_opt(T, len_div3, sp_others_id(T, v.id, s), v.id[s]);
        const SpRec w = spt_rec(T, s, v.key[s], 0);
        fallback |= (w.w & SPT_FALLBACK) != 0;
        waitN |= (u64)spt_wait_tiles(w, o) << (9 * s);
        fin = spt_fin(o);
    }
    if (fin_out) *fin_out = fin;
    if (fallback) return sp_req_brute(MT, h, len_div3, L);
    return sp_req_set(len_div3, L, fin, waitN, v.pairs, v.kinds, v.kpairs, v.kkinds, sp_count_sets(h));
}
// shanten-keeping discards of a 3n+2 hand with calc_all(g) == Tg
template <class TT> MJD u64 sp_keep_of_hand(const TT& T, const MjTablesDev& MT, Hand g, int len_div3, int Tg) {
    const SpSuitView v = sp_suit_view(T, g);
    u64 keepN = 0;
    int fin = 0;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        if (!spt_in_table(T, s, v.key[s])) return sp_keep_brute(MT, g, len_div3, Tg);
        const SpRec o = spt_opt(T, len_div3, sp_others_id(T, v.id, s), v.id[s]);
        keepN |= (u64)spt_keep_tiles(spt_rec(T, s, v.key[s], 1), o) << (9 * s);
        fin = spt_fin(o);
    }
    return sp_keep_set(len_div3, Tg, fin, keepN, v.pairs, v.kinds, v.kpairs, v.kkinds, sp_count_sets(g));
}

// ---------------------------------------------------------------------------------------------------------------------
// Host-side construction from the reference's two tables (called once per process by mj_tables_upload; the emulator build and
// tests/host/algo_check.hip use the same code).
#include <map>
#include <string>
#include <vector>

struct SpTabHost {
    std::vector<u8> id, mrg;          // id: [n_su + n_ji]
    std::vector<SpRec> opt, wk;       // wk: [n_su + n_ji][2]
    std::vector<u64> vec;  // id -> packed vector (ids < n_rows are the table rows)
    u32 n_rows = 0, zero_id = 0;
    std::string error;
};

static inline u64 spt_merge_full(u64 a, u64 b) {  // add_suhai with m = 4 on packed nibbles
    int v[10];
    for (int j = 0; j < 10; j++) v[j] = (int)((a >> (4 * j)) & 15);
    auto nb = [&](int j) { return (int)((b >> (4 * j)) & 15); };
    for (int j = 9; j >= 5; j--) {
        int sht = std::min(v[j] + nb(0), v[0] + nb(j));
        for (int k = 5; k < j; k++) sht = std::min(sht, std::min(v[k] + nb(j - k), v[j - k] + nb(k)));
        v[j] = sht;
    }
    for (int j = 4; j >= 0; j--) {
        int sht = v[j] + nb(0);
        for (int k = 0; k < j; k++) sht = std::min(sht, v[k] + nb(j - k));
        v[j] = sht;
    }
    u64 r = 0;
    for (int j = 0; j < 10; j++) {
        if (v[j] > 15) return ~0ull;
        r |= (u64)v[j] << (4 * j);
    }
    return r;
}

static inline bool sp_tab_build(const u64* suhai, u32 n_su, const u64* jihai, u32 n_ji, SpTabHost& H) {
    // 1. distinct rows -> ids (the all-zero row of keys past the table included)
    std::map<u64, u32> ids;
    ids[0ull] = 0;
    for (u32 i = 0; i < n_su; i++) ids.emplace(suhai[i] & 0xFFFFFFFFFFull, 0);
    for (u32 i = 0; i < n_ji; i++) ids.emplace(jihai[i] & 0xFFFFFFFFFFull, 0);
    u32 n = 0;
    for (auto& kv : ids) kv.second = n++;
    if (n > SPT_NB) { H.error = "more distinct shanten rows than SPT_NB"; return false; }
    H.n_rows = n;
    H.zero_id = ids[0ull];
    H.vec.assign(n, 0);
    for (auto& kv : ids) H.vec[kv.second] = kv.first;
    H.id.resize((size_t)n_su + n_ji);
    for (u32 i = 0; i < n_su; i++) H.id[i] = (u8)ids[suhai[i] & 0xFFFFFFFFFFull];
    for (u32 i = 0; i < n_ji; i++) H.id[(size_t)n_su + i] = (u8)ids[jihai[i] & 0xFFFFFFFFFFull];
    // 2. closure under merge(vector, row)
    std::map<u64, u32> vid(ids);
    H.mrg.assign((size_t)SPT_NV * SPT_NB, 0);
    for (u32 v = 0; v < (u32)H.vec.size(); v++)
        for (u32 b = 0; b < n; b++) {
            const u64 w = spt_merge_full(H.vec[v], H.vec[b]);
            if (w == ~0ull) { H.error = "merged entry does not fit a nibble"; return false; }
            auto it = vid.find(w);
            if (it == vid.end()) {
                if (H.vec.size() >= SPT_NV) { H.error = "merge closure larger than SPT_NV"; return false; }
                it = vid.emplace(w, (u32)H.vec.size()).first;
                H.vec.push_back(w);
            }
            H.mrg[(size_t)v * SPT_NB + b] = (u8)it->second;
        }
    // 3. final value + optimal row entries per (m, vector, row): min over x of vec[5+x] + row[m-x], vec[m-x] + row[5+x]
    H.opt.assign((size_t)5 * SPT_NV * SPT_NB, SpRec{0, 0, 0, 0});
    for (int m = 0; m <= 4; m++)
        for (u32 v = 0; v < (u32)H.vec.size(); v++)
            for (u32 b = 0; b < n; b++) {
                const u64 A = H.vec[v], B = H.vec[b];
                auto na = [&](int j) { return (int)((A >> (4 * j)) & 15); };
                auto nb = [&](int j) { return (int)((B >> (4 * j)) & 15); };
                int fin = 255;
                for (int x = 0; x <= m; x++) fin = std::min(fin, std::min(na(5 + x) + nb(m - x), na(m - x) + nb(5 + x)));
                u32 E = 0;
                for (int x = 0; x <= m; x++) {
                    if (na(5 + x) + nb(m - x) == fin) E |= 1u << (m - x);
                    if (na(m - x) + nb(5 + x) == fin) E |= 1u << (5 + x);
                }
                if (fin > 15) { H.error = "final value does not fit a nibble"; return false; }
                SpRec r{0, 0, 0, 0};
                for (int e = 1; e <= 9; e++)
                    if ((E >> e) & 1) (&r.x)[(e - 1) / 3] |= 0x1FFu << (9 * ((e - 1) % 3));
                if (E & 1) r.w = 0x1FFu;
                r.x |= (u32)fin << 27;
                H.opt[((size_t)m * SPT_NV + v) * SPT_NB + b] = r;
            }
    // 4. per-key wait / keep masks
    H.wk.assign(((size_t)n_su + n_ji) * 2, SpRec{0, 0, 0, 0});
    auto build_wk = [&](const u64* tab, u32 nk, int ntile, SpRec* out) -> bool {
        u32 pw[9];
        pw[ntile - 1] = 1;
        for (int i = ntile - 2; i >= 0; i--) pw[i] = pw[i + 1] * 5;
        for (u32 key = 0; key < nk; key++) {
            int dg[9], total = 0;
            u32 k = key;
            for (int i = ntile - 1; i >= 0; i--) { dg[i] = (int)(k % 5); k /= 5; total += dg[i]; }
            if (total > 14) continue;  // not a hand
            const u64 row = tab[key];
            SpRec w{0, 0, 0, 0}, kp{0, 0, 0, 0};
            for (int i = 0; i < ntile; i++) {
                if (dg[i] < 4) {
                    const u32 nkey = key + pw[i];
                    const bool inside = nkey < nk;
                    const u64 up = inside ? tab[nkey] : 0ull;
                    if (!inside) w.w |= SPT_FALLBACK;
                    for (int e = 0; e <= 9; e++) {
                        const int d = (int)((up >> (4 * e)) & 15) - (int)((row >> (4 * e)) & 15);
                        if (inside && (d > 0 || d < -1)) { H.error = "a drawn tile changes a row entry by more than one"; return false; }
                        if (d < 0 && e >= 1) (&w.x)[(e - 1) / 3] |= 1u << (9 * ((e - 1) % 3) + i);
                    }
                }
                if (dg[i] > 0) {
                    const u64 dn = tab[key - pw[i]];
                    for (int e = 0; e <= 9; e++) {
                        const int d = (int)((dn >> (4 * e)) & 15) - (int)((row >> (4 * e)) & 15);
                        if (d < 0 || d > 1) { H.error = "a removed tile changes a row entry by more than one"; return false; }
                        if (d == 0) {
                            if (e == 0) kp.w |= 1u << i;
                            else (&kp.x)[(e - 1) / 3] |= 1u << (9 * ((e - 1) % 3) + i);
                        }
                    }
                }
            }
            out[(size_t)key * 2] = w;
            out[(size_t)key * 2 + 1] = kp;
        }
        return true;
    };
    return build_wk(suhai, n_su, 9, H.wk.data()) && build_wk(jihai, n_ji, 7, H.wk.data() + (size_t)n_su * 2);
}
