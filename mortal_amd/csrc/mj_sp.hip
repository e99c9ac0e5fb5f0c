// Single-player tables of obs v4 (rows 889..1011) on device.
//
// Reference: PlayerState::single_player_tables (state/agent_helper.rs:509-593) -> SPCalculator::calc
// (algo/sp/calc.rs:84-133, production flags: no tegawari, no shanten-down, maximise EV, sorted) and the encoder block
// obs_repr.rs:564-692.  The reference is a memoised depth-first recursion (draw -> discard -> draw ...) over hand
// states with order-sensitive f32 sums.  On the GPU the same values are produced LEVEL-SYNCHRONOUSLY, one decision row
// per (persistent) workgroup at a time:
//   set-up   : candidates and their required tiles as workgroup-parallel incremental shanten probes;
//   expand   : for shanten level L = s .. 1, the 3n+1 states of level L in chunks of 16 (sp_expand_chunk: thread-per-task
//              passes over the whole workgroup) find their required draws t and the shanten-keeping discards d of h+t and
//              insert the children h+t-d into a per-workgroup hash set (64-bit tag claimed by atomicCAS);
//   evaluate : for L = 0 .. s: level 0 in three passes (probe / dense thread-per-item scoring / sum), levels > 0 by
//              teams of 32, 16 or 8 lanes (one lane per remaining draw, sp_eval_team) that reproduce
//              draw_without_tegawari (calc.rs:447-561) with the reference's exact loop order (draw tiles ascending, aka
//              after its plain tile; i, j ascending) and fold discards like discard_slow (calc.rs:563-637).
// Memoisation in the reference is a pure cache, so evaluating every reachable state exactly once gives bit-identical
// f32 results as long as each state's own accumulation order is kept — it is.  Compiled with -ffp-contract=off
// (Rust never fuses a*b+c).  DESIGN.md §6 has the cost model and the optimisation history.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "mj_rules.h"

#define SP_THREADS 256
#ifndef SP_VARIANT
#define SP_VARIANT 0  // experimental round-2 candidates (see "NEXT" below): bit 0 = 32-state expansion chunks, bit 1 = two turns per lane, bit 2 = per-wavefront 8-state chunks, bit 3 = cheaper state hash
#endif
#define SP_CAP 16384           // hash slots per workgroup (max observed states per decision ~4.3k)
#define SP_T 17                // MAX_TSUMOS_LEFT (sp/mod.rs:40)
#define SP_MAX_CAND 14

#define SP_POOL (SP_CAP * 32)   // child-slot pool entries per workgroup
#define SP_ITEMS (SP_CAP * 4)   // level-0 scoring items per workgroup
#define SP_L0_MAX 17            // winning draw entries per tenpai state (13 waits + 3 aka variants; 272-byte node area)
struct SpNode {                // one 3n+1 state
    u64 k0, k1, k2, k3;        // hand.mp | hand.sz + akas_in_hand<<48 | wall.mp | wall.sz + akas_in_wall<<48
    float tenpai[SP_T], win[SP_T], ev[SP_T];
    u32 child_off;             // expansion results, reused by the evaluation pass:
    u64 req;                   //   required draw tiles,
    u64 keep[34];              //   per required tile t the shanten-keeping discards of h + t,
};                             //   and the children's hash slots in pool[child_off ..] (order: t, variant, d ascending)
struct SpWork {                // per-workgroup scratch in HBM (persistent workgroups)
    u64 tag[SP_CAP];
    SpNode node[SP_CAP];
    u32 list[SP_CAP];          // slots grouped by level: level L occupies [lvl_begin[L], lvl_end[L])
    unsigned short pool[SP_POOL];
    u32 items[SP_ITEMS];       // level 0: (state, winning tile, variant) work items of the dense scoring pass
};

struct SpParams {
    const TableOne* snap;
    const uint32_t* rows;
    int n_rows;
    MjTablesDev tables;
    float* obs;                // [n_rows][1012][34]; rows 889.. are zero on entry (written by mj_k_encode<4>)
    SpWork* work;              // [gridDim.x]
    int* queue;                // dynamic row queue (zeroed before launch)
    unsigned long long* prof;  // NULL or [24] phase timers
    unsigned long long* err;   // [0] hash-capacity overflows, [1] rows; cycle sums: [2] setup [3] expand [4] eval L0 [5] eval L>0 [6] encode; [7] states
};

// algo/data/uradora_prob_table.txt (values restated; calc.rs:17)
__device__ static const float SP_URADORA[5][13] = {
    {0.639485f, 0.327801f, 0.0327134f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f},
    {0.406736f, 0.42281f, 0.147966f, 0.021674f, 0.0008142f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f},
    {0.257516f, 0.406819f, 0.246851f, 0.0757724f, 0.0122266f, 0.0008004f, 1.43e-5f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f},
    {0.162199f, 0.346513f, 0.301539f, 0.142396f, 0.0401276f, 0.0066491f, 0.0005575f, 1.85e-5f, 0.f, 0.f, 0.f, 0.f, 0.f},
    {0.101768f, 0.275319f, 0.313742f, 0.20189f, 0.081774f, 0.0215394f, 0.0035918f, 0.0003607f, 1.52e-5f, 3e-7f, 0.f, 0.f, 0.f}};
__device__ static const u8 SP_DISCARD_PRIO[38] = {  // tile.rs:21-28
    6, 5, 4, 3, 2, 3, 4, 5, 6, 6, 5, 4, 3, 2, 3, 4, 5, 6, 6, 5, 4, 3, 2, 3, 4, 5, 6, 7, 7, 7, 7, 7, 7, 7, 1, 1, 1, 0};
// The same table in closed form (no memory access): used inside the evaluation fold, where the order of two discards is
// the order of the keys  prio * 64 + (63 - tile)  (higher priority first, then the lower tile id; tile.rs:169-177).
constexpr __host__ __device__ int sp_discard_prio(int t) {
    return t < 27 ? 2 + (t % 9 > 4 ? t % 9 - 4 : 4 - t % 9) : t < 34 ? 7 : t < 37 ? 1 : 0;
}
constexpr bool sp_discard_prio_matches_table() {
    constexpr int tab[38] = {6, 5, 4, 3, 2, 3, 4, 5, 6, 6, 5, 4, 3, 2, 3, 4, 5, 6, 6, 5, 4, 3, 2, 3, 4, 5, 6, 7, 7, 7, 7, 7, 7, 7, 1, 1, 1, 0};
    for (int t = 0; t < 38; t++)
        if (sp_discard_prio(t) != tab[t]) return false;
    return true;
}
static_assert(sp_discard_prio_matches_table(), "sp_discard_prio != SP_DISCARD_PRIO");
MJD int sp_discard_key(int t) { return sp_discard_prio(t) * 64 + (63 - t); }
MJD int cmp_discard_priority(int l, int r) {  // tile.rs:169-177
    int pl = SP_DISCARD_PRIO[l], pr = SP_DISCARD_PRIO[r];
    if (pl != pr) return pl < pr ? -1 : 1;
    if (r != l) return r < l ? -1 : 1;
    return 0;
}

struct SpState {  // sp/state.rs:9-20 (n_extra_tsumo is always 0 with the production flags)
    Hand h, w;
    u32 akas;     // bits 0..2 akas_in_hand, 3..5 akas_in_wall
};
MJD void sp_discard(SpState& s, int tile) {  // state.rs:57-65
    s.h.dec(deaka(tile));
    if (is_aka(tile)) s.akas &= ~(1u << (tile - T_5MR));
}
MJD void sp_deal(SpState& s, int tile) {     // state.rs:77-86
    s.w.dec(deaka(tile));
    s.h.inc(deaka(tile));
    if (is_aka(tile)) {
        s.akas &= ~(8u << (tile - T_5MR));
        s.akas |= 1u << (tile - T_5MR);
    }
}
MJD void sp_key(const SpState& s, u64 k[4]) {
    k[0] = s.h.mp;
    k[1] = s.h.sz | ((u64)(s.akas & 7) << 48);
    k[2] = s.w.mp;
    k[3] = s.w.sz | ((u64)((s.akas >> 3) & 7) << 48);
}
MJD u64 sp_hash(const u64 k[4]) {
#if SP_VARIANT & 8
    // experimental: multilinear combination of the four key words (odd 64-bit multipliers) + one splitmix64 finaliser —
    // 6 instead of 8 64-bit multiplications and a quarter of the shift/xor steps
    u64 g = k[0] * 0x9E3779B97F4A7C15ull + k[1] * 0xC2B2AE3D27D4EB4Full + k[2] * 0x165667B19E3779F9ull + k[3] * 0xD6E8FEB86659FD93ull;
    g = (g ^ (g >> 30)) * 0xBF58476D1CE4E5B9ull;
    g = (g ^ (g >> 27)) * 0x94D049BB133111EBull;
    g ^= g >> 31;
    return g | 1ull;
#endif
    u64 h = 0x9E3779B97F4A7C15ull;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        h ^= k[i] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
        h = (h ^ (h >> 30)) * 0xBF58476D1CE4E5B9ull;
        h = (h ^ (h >> 27)) * 0x94D049BB133111EBull;
        h ^= h >> 31;
    }
    return h | 1ull;  // never 0 (0 = empty slot)
}

struct SpCtx {  // per-decision constants (LDS)
    Melds melds;
    int len_div3, bakaze, jikaze, is_menzen, num_doras_in_fuuro, n_dora, calc_double_riichi, calc_haitei,
        prefer_riichi, T, n_left;
    int dora_ind[5];
    float tsumo_prob[4][SP_T];
    float not_tsumo[124][SP_T];   // MAX_TILES_LEFT + 1 = 123 rows (calc.rs:14,148-167); row = sum of required tiles
    // level bookkeeping
    int lvl_begin[5], lvl_end[5];
    int n_list;
    int n_pool;
    int n_items;
    int overflow;
    unsigned long long* prof;  // optional phase timers (MJ_SP_PROF)
    // candidates
    int n_cand;
    int cand_tile[SP_MAX_CAND], cand_slot[SP_MAX_CAND], cand_down[SP_MAX_CAND], cand_nreq[SP_MAX_CAND];
    u64 cand_req[SP_MAX_CAND];
    int order[SP_MAX_CAND];
    float cand_tp0[SP_MAX_CAND], cand_wp0[SP_MAX_CAND], cand_ev0[SP_MAX_CAND];
};

// Out-of-line phase functions receive their LDS scratch / HBM work area as generic pointers; telling the compiler which
// address space they are in (an assumption for LDS, an explicit global-address-space pointer for HBM) turns every flat_load (which counts on BOTH vmcnt and lgkmcnt, so each LDS read waits for all
// HBM gathers in flight) into ds_read / global_load with independent counters.
#if defined(__HIP_DEVICE_COMPILE__)
#define SP_ASSUME_LDS(p) __builtin_assume(__builtin_amdgcn_is_shared((const void*)(p)))
#else
#define SP_ASSUME_LDS(p) ((void)0)  // host pass of the single-source compile: the builtin only exists on the device
#endif
#ifdef MJ_EMU
#define SP_HBM
#else
#define SP_HBM __attribute__((address_space(1)))
#endif

// hash-set insert; returns the slot or -1 on overflow.  `fresh` tells whether this call created the slot.
__device__ int sp_insert(SpWork* W, SpCtx* X, const SpState& s, bool& fresh) {
    u64 k[4];
    sp_key(s, k);
    const u64 h = sp_hash(k);
    u32 pos = (u32)(h >> 20) & (SP_CAP - 1);
    fresh = false;
    for (int probe = 0; probe < SP_CAP; probe++) {
        u64 old = atomicCAS((unsigned long long*)&W->tag[pos], 0ull, (unsigned long long)h);
        if (old == 0ull) {
            SpNode& n = W->node[pos];
            n.k0 = k[0]; n.k1 = k[1]; n.k2 = k[2]; n.k3 = k[3];
            fresh = true;
            return (int)pos;
        }
        if (old == h) return (int)pos;  // same 63-bit tag = same state (collision odds < 1e-11 per row)
        pos = (pos + 1) & (SP_CAP - 1);
    }
    X->overflow = 1;
    return -1;
}
// Two-step insert so that several first probes (one L2 atomic each) can be in flight per lane.  WP = SpWork* in any
// address space (the out-of-line phase functions pass their global-address-space pointer).
struct SpIns {
    u64 k[4], h, old;
    u32 pos;
};
template <class TagP>
MJD u64 sp_claim_tag(TagP tagp, u64 h, u64 expected = 0ull) {  // atomicCAS(tag, 0, h) -> previous value (relaxed, agent scope)
    __hip_atomic_compare_exchange_strong(tagp, &expected, h, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return expected;
}
// `real == false` (a lane that has no child in this slot of the batch) issues a compare-and-swap that cannot change the
// tag (expected = desired = all ones), so that both probes of a batch are unconditional and stay in flight together.
template <class WP>
__device__ __forceinline__ void sp_insert_begin(WP W, const SpState& s, SpIns& I, bool real = true) {
    sp_key(s, I.k);
    I.h = sp_hash(I.k);
    I.pos = (u32)(I.h >> 20) & (SP_CAP - 1);
    I.old = sp_claim_tag(&W->tag[I.pos], real ? I.h : ~0ull, real ? 0ull : ~0ull);
}
template <class WP>
__device__ __forceinline__ int sp_insert_finish(WP W, SpCtx* X, SpIns& I, bool& fresh) {
    fresh = false;
    u64 old = I.old;
    u32 pos = I.pos;
    for (int probe = 0; probe < SP_CAP; probe++) {
        if (old == 0ull) {
            auto& n = W->node[pos];
            n.k0 = I.k[0]; n.k1 = I.k[1]; n.k2 = I.k[2]; n.k3 = I.k[3];
            fresh = true;
            return (int)pos;
        }
        if (old == I.h) return (int)pos;
        pos = (pos + 1) & (SP_CAP - 1);
        old = sp_claim_tag(&W->tag[pos], I.h);
    }
    X->overflow = 1;
    return -1;
}
template <class NodeT>
MJD SpState sp_state_of(const NodeT& n) {
    SpState s;
    s.h.mp = n.k0;
    s.h.sz = n.k1 & 0xFFFFFFFFFFFFull;
    s.w.mp = n.k2;
    s.w.sz = n.k3 & 0xFFFFFFFFFFFFull;
    s.akas = (u32)((n.k1 >> 48) & 7) | ((u32)((n.k3 >> 48) & 7) << 3);
    return s;
}

// get_score (calc.rs:640-758).  `s` already contains the winning tile.
__device__ bool sp_get_score(const MjTablesDev& T, const SpCtx* X, const SpState& s, int win_tile, float scores[4]) {
    AgariIn in;
    in.tehai = s.h;
    in.m = X->melds;
    in.is_menzen = X->is_menzen != 0;
    in.bakaze = X->bakaze;
    in.jikaze = X->jikaze;
    in.winning_tile = deaka(win_tile);
    in.is_ron = false;
    const bool is_oya = X->jikaze == T_E;
    const int additional = X->is_menzen ? (X->prefer_riichi ? 2 : 1) : 0;
    int num_doras = 0;
    for (int i = 0; i < X->n_dora; i++) num_doras += s.h.get(tile_next(X->dora_ind[i]));
    num_doras += __popc(s.akas & 7) + X->num_doras_in_fuuro;
    num_doras &= 0xFF;
    Agari a = agari_full(T, in, additional, num_doras);
    if (a.kind == 0) return false;
    if (a.kind == 2) {
        float v = (float)tsumo_total(point_yakuman(is_oya, a.han), is_oya);
        scores[0] = scores[1] = scores[2] = scores[3] = v;
        return true;
    }
    const int fu = a.fu, han = a.han & 0xFF;
    const bool assume_riichi = X->is_menzen && X->prefer_riichi;
    if (assume_riichi && X->n_dora == 1) {
        int n_ind[5] = {0, 0, 0, 0, 0}, sum_ind = 0;
        for (int t = 0; t < 34; t++) {
            int c = s.h.get(t);
            if (c == 0) continue;
            int ic = s.w.get(tile_prev(t));
            n_ind[c] = (n_ind[c] + ic) & 0xFF;
            sum_ind = (sum_ind + ic) & 0xFF;
        }
        int n_left = 0;
        for (int t = 0; t < 34; t++) n_left += s.w.get(t);
        n_left &= 0xFF;
        float up[5];
        up[0] = (float)((n_left - sum_ind) & 0xFF) / (float)n_left;
        for (int i = 1; i < 5; i++) up[i] = (float)n_ind[i] / (float)n_left;
        for (int i = 0; i < 4; i++) {
            float sc = 0.f;
            for (int j = 0; j < 5; j++) {
                float p = up[j];
                if (p == 0.f) continue;
                float pt = (float)tsumo_total(point_calc(is_oya, fu, (han + i + j) & 0xFF), is_oya);
                sc += pt * p;
            }
            scores[i] = sc;
        }
    } else if (assume_riichi && X->n_dora > 1) {
        for (int i = 0; i < 4; i++) {
            float sc = 0.f;
            for (int j = 0; j < 13; j++) {
                float p = SP_URADORA[X->n_dora - 1][j];
                if (p == 0.f) continue;
                float pt = (float)tsumo_total(point_calc(is_oya, fu, (han + i + j) & 0xFF), is_oya);
                sc += pt * p;
            }
            scores[i] = sc;
        }
    } else {
        for (int i = 0; i < 4; i++) scores[i] = (float)tsumo_total(point_calc(is_oya, fu, (han + i) & 0xFF), is_oya);
    }
    return true;
}

// Per-team LDS scratch of sp_eval_team<32> (rows with more than 16 draws left).
#define SP_CH 8      // children gathered per batch in the evaluation pass
#define SP_CCAP 256  // children staged per super-chunk (a required tile has at most 2 x 14)
struct SpTeam {
    static constexpr int CH = SP_CH, CCAP = SP_CCAP;
    u64 keep[34];            // per required tile t: set of shanten-keeping discards of h + t
    int coff[34];            // per required tile t: offset of its first child inside the node's child list
    u8 tiles[36];            // required tiles in ascending order
    union {
        float sc[SP_L0_MAX][4];  // level 0: get_score() of every draw entry (filled by sp_l0_score through the node)
        struct {             // level > 0 evaluation
            float buf[SP_CH][3][SP_T];          // values of the current batch of children, one turn per lane
            unsigned short cs[SP_CCAP];         // child slots
            unsigned short meta[SP_CCAP];       // discard order key (9 bits) | last-of-group << 9 | draw count << 10
        } ev;
    } u;
};

// Half-width team scratch for sp_eval_team<16>: evaluation only, 16 of them alias the 8 full-width scratches.
struct SpHalf {
    static constexpr int CH = 4, CCAP = 96;
    u64 keep[34];
    int coff[34];
    u8 tiles[36];
    union {
        float sc[SP_L0_MAX][4];
        struct {
            float buf[CH][3][16];
            unsigned short cs[CCAP];
            unsigned short meta[CCAP];
        } ev;
    } u;
};

struct SpQuarter {  // sp_eval_team<8>: at most 8 draws left, four states per 32 lanes
    static constexpr int CH = 2, CCAP = 64;
    u64 keep[34];
    int coff[34];
    u8 tiles[36];
    union {
        float sc[SP_L0_MAX][4];
        struct {
            float buf[CH][3][8];
            unsigned short cs[CCAP];
            unsigned short meta[CCAP];
        } ev;
    } u;
};

// ---- Level 0 (tenpai states) is evaluated in three passes so that the expensive, divergent scoring of the winning
// draws (get_score: agari decomposition + yaku + fu) runs with every lane busy:
//   probe : sp_l0_probe_chunk — which draws win (34 shanten probes per state) -> node.req, one work item per draw entry
//   score : THREAD per item, dense across the workgroup                  -> 4 scores per item in the node (keep[] area)
//   sum   : team per state — sp_eval_team(L = 0) accumulates the scores in the reference's order
__device__ __noinline__ void sp_l0_score(const MjTablesDev& Tb, SpWork* W, const SpCtx* X, u32 item) {
    SP_ASSUME_LDS(X);
    const int slot = item & 0x3FFF, idx = (item >> 14) & 31, t = (item >> 19) & 63, variant = (item >> 25) & 1;
    SP_HBM SpNode& node = ((SP_HBM SpWork*)W)->node[slot];
    SpState S1 = sp_state_of(node);
    const int tile = variant ? akaize(t) : t;
    sp_deal(S1, tile);
    float scv[4];
    if (sp_get_score(Tb, X, S1, tile, scv)) {
        SP_HBM float* dst = reinterpret_cast<SP_HBM float*>(node.keep) + 4 * idx;
#pragma unroll
        for (int q = 0; q < 4; q++) dst[q] = scv[q];
        __hip_atomic_fetch_or(&node.child_off, 1u << idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Dense expansion / level-0 probe: SP_NS states of one level at a time, every phase a THREAD-PER-TASK pass over the whole
// workgroup (tasks = (state, suit), (state, merge), (state, tile), (state, draw tile, discard kind) ...), separated by
// workgroup barriers.  A 32-lane team per state (the first formulation) spends most of its instruction stream with few
// lanes busy (6 of 32 in the partial merges, 2 of 32 in the second probe round, ...); here every lane of every wave runs a
// task, so the instructions issued per state drop by about 4x.
// `safe` = the tile kinds d in the hand with shanten(h - d) <= L.  Only those can be shanten-keeping discards after a
// required draw t: shanten(h + t - d) == L - 1 needs shanten(h - d) <= L, since one more tile lowers a shanten number
// (normal, chiitoi and kokushi form alike) by at most one; the (t, d) probes run over these kinds only.  Arithmetic, table probes and the resulting req / keep sets / child
// order are exactly those of the team formulation (git history: sp_expand_team / sp_l0_probe).
#define SP_NS 16
struct SpChunk {
    u64 k[SP_NS][4];        // state keys (hand.mp, hand.sz | akas, wall.mp, wall.sz | akas)
    u64 row[SP_NS][4];      // base table rows of the four suits
    u64 r2[SP_NS][6];       // merges of two base rows (suit pairs 01 02 03 12 13 23)
    u64 r3[SP_NS][4];       // per suit: merge of the three OTHER base rows
    u64 rowt[SP_NS][34];    // row of h + t in suit(t) (0 when t is not in the wall)
    u64 rowd[SP_NS][34];    // row of h - d in suit(d) (0 when d is not in the hand)
    u64 V[SP_NS][13][3];    // per safe discard kind d and k-th other suit: merge(two untouched suits, row of h - d)
    u64 keep[SP_NS][34];    // per required tile t: shanten-keeping discards of h + t
    u64 req[SP_NS], safe[SP_NS];
    u32 bkey[SP_NS][4];     // base-5 suit keys
    u32 slot[SP_NS];
    int cnt[SP_NS][4];      // pairs, kinds, yaokyuu pairs, yaokyuu kinds (shanten.rs:104-137)
    int item_off[SP_NS + 1];  // prefix sums of n_tiles * n_kinds
    int child_base[SP_NS];
    unsigned short coff[SP_NS][34];  // per required tile: offset of its first child inside the state's child list
    u8 tiles[SP_NS][36], kinds[SP_NS][16];
    u8 n_tiles[SP_NS], n_kinds[SP_NS];
};

MJD SpState sp_chunk_state(const SpChunk* C, int s) {
    SpState S;
    S.h.mp = C->k[s][0];
    S.h.sz = C->k[s][1] & 0xFFFFFFFFFFFFull;
    S.w.mp = C->k[s][2];
    S.w.sz = C->k[s][3] & 0xFFFFFFFFFFFFull;
    S.akas = (u32)((C->k[s][1] >> 48) & 7) | ((u32)((C->k[s][3] >> 48) & 7) << 3);
    return S;
}

// Passes P0-P3, shared by the expansion (L >= 1) and the level-0 probe (L == 0): state keys, base rows, partial merges
// and the 34 "+t" (and for L >= 1 "-t") shanten probes of every state of the chunk -> req, safe, rowt, rowd.
__device__ __forceinline__ void sp_chunk_probe(SP_HBM SpWork* Wg, SpCtx* X, SpChunk* C, const ShTab& ST, int first, int n, int L) {
    const int tid = threadIdx.x;
    const int ld3 = X->len_div3;
    // P0a: keys
    for (int task = tid; task < n * 4; task += SP_THREADS) {
        const int s = task >> 2, j = task & 3;
        const u32 slot = Wg->list[first + s];
        C->k[s][j] = reinterpret_cast<const SP_HBM u64*>(&Wg->node[slot])[j];  // k0..k3 lead the node
        if (j == 0) {
            C->slot[s] = slot;
            C->req[s] = 0;
            C->safe[s] = 0;
        }
    }
    __syncthreads();
    // P0b: suit keys, base rows, chiitoi / kokushi counters
    for (int task = tid; task < n * 4; task += SP_THREADS) {
        const int s = task >> 2, i = task & 3;
        const SpState S = sp_chunk_state(C, s);
        const u32 key = i == 0 ? suit_key9(S.h.mp) : i == 1 ? suit_key9(S.h.mp >> 27) : i == 2 ? suit_key9(S.h.sz) : suit_key7(S.h.sz >> 27);
        C->bkey[s][i] = key;
        C->row[s][i] = sh_load(ST, i, key);
        C->cnt[s][i] = i == 0 ? S.h.n_pairs() : i == 1 ? S.h.n_kinds() : i == 2 ? S.h.n_yao_pairs() : S.h.n_yao_kinds();
    }
    __syncthreads();
    // P1: merges of two base rows
    for (int task = tid; task < n * 6; task += SP_THREADS) {
        const int s = task / 6, p = task % 6;
        const int a = p < 3 ? 0 : p < 5 ? 1 : 2, b = p < 3 ? p + 1 : p < 5 ? p - 1 : 3;
        C->r2[s][p] = sh_merge(C->row[s][a], C->row[s][b], ld3);
    }
    __syncthreads();
    // P2: per suit the merge of the three other base rows: 0: (1,2)+3, 1: (0,2)+3, 2: (0,1)+3, 3: (0,1)+2
    for (int task = tid; task < n * 4; task += SP_THREADS) {
        const int s = task >> 2, i = task & 3;
        const u64 pr = i == 0 ? C->r2[s][3] : i == 1 ? C->r2[s][1] : C->r2[s][0];
        C->r3[s][i] = sh_merge(pr, C->row[s][i == 3 ? 2 : 3], ld3);
    }
    __syncthreads();
    // P3: probes (state, tile)
    for (int task = tid; task < n * 34; task += SP_THREADS) {
        const int s = task / 34, t = task % 34;
        const SpState S = sp_chunk_state(C, s);
        const int st = sh_suit(t), hc = S.h.get(t), yao = (int)((YAOKYUU_MASK >> t) & 1);
        const bool in_wall = S.w.get(t) > 0, in_hand = L > 0 && hc > 0;
        const u32 kb = C->bkey[s][st], pw = sh_pow(t);
        const u64 rt = sh_load(ST, st, in_wall ? kb + pw : kb);
        const u64 rd = sh_load(ST, st, in_hand ? kb - pw : kb);
        const u64 r3 = C->r3[s][st];
        const int pairs = C->cnt[s][0], kinds = C->cnt[s][1], kpairs = C->cnt[s][2], kkinds = C->cnt[s][3];
        if (in_wall) {
            const int sh = sh_finish(sh_final(r3, rt, ld3), ld3, pairs + (hc == 1), kinds + (hc == 0), kpairs + (yao && hc == 1),
                                     kkinds + (yao && hc == 0));
            if (sh - L == -1) atomicOr((unsigned long long*)&C->req[s], 1ull << t);
        }
        if (in_hand) {  // `safe` (see below): only discards with shanten(h - d) <= L can keep shanten after a required draw
            const int sh = sh_finish(sh_final(r3, rd, ld3), ld3, pairs - (hc == 2), kinds - (hc == 1), kpairs - (yao && hc == 2),
                                     kkinds - (yao && hc == 1));
            if (sh <= L) atomicOr((unsigned long long*)&C->safe[s], 1ull << t);
        }
        C->rowt[s][t] = in_wall ? rt : 0ull;
        C->rowd[s][t] = in_hand ? rd : 0ull;
        C->keep[s][t] = 0;
    }
    __syncthreads();
}

// Level 0: which draws win (node.req) and one scoring work item per draw entry, in the reference's order (plain tile if a
// non-red copy is left, then the red five).
__device__ __noinline__ void sp_l0_probe_chunk(SpWork* W, SpCtx* X, SpChunk* C, int first, int n) {
    SP_ASSUME_LDS(X);
    SP_ASSUME_LDS(C);
    SP_HBM SpWork* const Wg = (SP_HBM SpWork*)W;
    const ShTab ST = sh_tab(c_mj_tables);
    sp_chunk_probe(Wg, X, C, ST, first, n, 0);
    const int s = threadIdx.x;
    if (s < n) {
        const SpState S = sp_chunk_state(C, s);
        const u64 req = C->req[s];
        const u32 slot = C->slot[s];
        int cnt = 0;
        for (u64 rest = req; rest; rest &= rest - 1) {
            const int t = __ffsll((long long)rest) - 1;
            const bool aka_in_wall = (t == T_5M && (S.akas & 8)) || (t == T_5P && (S.akas & 16)) || (t == T_5S && (S.akas & 32));
            cnt += (!aka_in_wall || S.w.get(t) >= 2) + aka_in_wall;
        }
        if (cnt > SP_L0_MAX) { X->overflow = 1; cnt = SP_L0_MAX; }
        const int base = atomicAdd(&X->n_items, cnt);
        int e = 0;
        for (u64 rest = req; rest; rest &= rest - 1) {
            const int t = __ffsll((long long)rest) - 1;
            const bool aka_in_wall = (t == T_5M && (S.akas & 8)) || (t == T_5P && (S.akas & 16)) || (t == T_5S && (S.akas & 32));
            for (int variant = 0; variant < 2; variant++) {
                if (variant == 0 ? (aka_in_wall && S.w.get(t) < 2) : !aka_in_wall) continue;
                if (e < cnt) {
                    if (base + e < SP_ITEMS) Wg->items[base + e] = slot | ((u32)e << 14) | ((u32)t << 19) | ((u32)variant << 25);
                    else X->overflow = 1;
                }
                e++;
            }
        }
        SP_HBM SpNode& node = Wg->node[slot];
        node.req = req;
        node.child_off = 0;  // bit i: draw entry i has a yaku (set by sp_l0_score)
    }
    __syncthreads();
}

// Levels >= 1: required draws, shanten-keeping discards and the children of SP_NS states.
__device__ __noinline__ void sp_expand_chunk(SpWork* W, SpCtx* X, SpChunk* C, int first, int n, int L) {
    SP_ASSUME_LDS(X);
    SP_ASSUME_LDS(C);
    SP_HBM SpWork* const Wg = (SP_HBM SpWork*)W;
    const ShTab ST = sh_tab(c_mj_tables);
    const int tid = threadIdx.x;
    const int ld3 = X->len_div3;
    sp_chunk_probe(Wg, X, C, ST, first, n, L);
    // P4a: ascending lists of the required tiles and of the safe discard kinds
    for (int task = tid; task < n * 34; task += SP_THREADS) {
        const int s = task / 34, t = task % 34;
        const u64 req = C->req[s], safe = C->safe[s], below = (1ull << t) - 1;
        if ((req >> t) & 1) C->tiles[s][__popcll(req & below)] = (u8)t;
        if ((safe >> t) & 1) C->kinds[s][__popcll(safe & below)] = (u8)t;
        if (t == 0) {
            C->n_tiles[s] = (u8)__popcll(req);
            C->n_kinds[s] = (u8)__popcll(safe);
        }
    }
    __syncthreads();
    if (tid == 0) {
        int off = 0;
        for (int s = 0; s < n; s++) {
            C->item_off[s] = off;
            off += (int)C->n_tiles[s] * (int)C->n_kinds[s];
        }
        C->item_off[n] = off;
    }
    // P4b: per safe discard kind d and each of the three other suits: V = merge(two untouched suits, row of h - d)
    for (int task = tid; task < n * 39; task += SP_THREADS) {
        const int s = task / 39, q = task % 39, ki = q / 3, k = q % 3;
        if (ki >= (int)C->n_kinds[s]) continue;
        const int d = C->kinds[s][ki], sd = sh_suit(d), st = k + (k >= sd);  // k-th suit != sd
        int x = -1, y = -1;  // the two suits other than st and sd
        for (int i = 0; i < 4; i++)
            if (i != st && i != sd) { if (x < 0) x = i; else y = i; }
        C->V[s][ki][k] = sh_merge(C->r2[s][sh_pair_idx(x, y)], C->rowd[s][d], ld3);
    }
    __syncthreads();
    // P5: (state, required t, safe d) probes of h + t - d (d == t never keeps: that is the state itself):
    // same suit -> final(r3[suit], gathered row of h+t-d); other suit -> final(V[d][suit t], row of h+t)
    const int n_items = C->item_off[n];
    auto item_decode = [&](int it, int& s, int& t, int& d, int& ki) {
        s = 0;
        while (s + 1 < n && C->item_off[s + 1] <= it) s++;
        const int local = it - C->item_off[s], nk = C->n_kinds[s];
        const int ti = local / nk;
        ki = local - ti * nk;
        t = C->tiles[s][ti];
        d = C->kinds[s][ki];
    };
    for (int it = tid; it < n_items; it += SP_THREADS) {
        int s, t, d, ki;
        item_decode(it, s, t, d, ki);
        if (d == t) continue;
        const SpState S = sp_chunk_state(C, s);
        const int st = sh_suit(t), sd = sh_suit(d);
        const int c = S.h.get(d), hct = S.h.get(t), yt = (int)((YAOKYUU_MASK >> t) & 1), yd = (int)((YAOKYUU_MASK >> d) & 1);
        int fin;
        if (sd == st) fin = sh_final(C->r3[s][st], sh_load(ST, st, C->bkey[s][st] + sh_pow(t) - sh_pow(d)), ld3);
        else fin = sh_final(C->V[s][ki][st - (st > sd)], C->rowt[s][t], ld3);
        const int pairs = C->cnt[s][0] + (hct == 1) - (c == 2), kinds = C->cnt[s][1] + (hct == 0) - (c == 1);
        const int kpairs = C->cnt[s][2] + (yt && hct == 1) - (yd && c == 2), kkinds = C->cnt[s][3] + (yt && hct == 0) - (yd && c == 1);
        if (sh_finish(fin, ld3, pairs, kinds, kpairs, kkinds) - (L - 1) == 0) atomicOr((unsigned long long*)&C->keep[s][t], 1ull << d);
    }
    __syncthreads();
    // P6: child list layout per state (for each required tile `variants(t) * popcount(keep[t])` slots) + node header
    if (tid < n) {
        const int s = tid;
        const SpState S = sp_chunk_state(C, s);
        int total = 0;
        const int nt = C->n_tiles[s];
        for (int ti = 0; ti < nt; ti++) {
            const int t = C->tiles[s][ti];
            const bool aka_in_wall = (t == T_5M && (S.akas & 8)) || (t == T_5P && (S.akas & 16)) || (t == T_5S && (S.akas & 32));
            const int nvar = aka_in_wall ? (S.w.get(t) >= 2 ? 2 : 1) : 1;
            C->coff[s][t] = (unsigned short)total;
            total += nvar * __popcll(C->keep[s][t]);
        }
        int child_base = atomicAdd(&X->n_pool, total);
        if (child_base + total > SP_POOL) { X->overflow = 1; child_base = 0; }
        C->child_base[s] = child_base;
        SP_HBM SpNode& node = Wg->node[C->slot[s]];
        node.child_off = (u32)child_base;
        node.req = C->req[s];
    }
    for (int task = tid; task < n * 34; task += SP_THREADS) {
        const int s = task / 34, t = task % 34;
        Wg->node[C->slot[s]].keep[t] = C->keep[s][t];
    }
    __syncthreads();
    // P7: children — the kept (t, d) of the same item space; each inserts its child state(s) (one per existing draw variant
    // of t) into the hash set and leaves the slot at its place of the reference's order (t, variant, d ascending)
    for (int it = tid; it < n_items; it += SP_THREADS) {
        int s, t, d, ki;
        item_decode(it, s, t, d, ki);
        const u64 keep = C->keep[s][t];
        if (d == t || !((keep >> d) & 1)) continue;
        const SpState S = sp_chunk_state(C, s);
        const int nk = __popcll(keep), rank = __popcll(keep & ((1ull << d) - 1));
        const int cnt = S.w.get(t);
        const bool aka_in_wall = (t == T_5M && (S.akas & 8)) || (t == T_5P && (S.akas & 16)) || (t == T_5S && (S.akas & 32));
        for (int variant = 0; variant < 2; variant++) {
            int vidx;  // index of this variant among the tile's existing draw entries
            if (!aka_in_wall) { if (variant == 1) continue; vidx = 0; }
            else if (variant == 0) { if (cnt < 2) continue; vidx = 0; }
            else vidx = cnt >= 2 ? 1 : 0;
            const int tile = (aka_in_wall && variant == 1) ? akaize(t) : t;
            SpState S2 = S;
            sp_deal(S2, tile);
            const int c = S2.h.get(d);
            int dt = d;  // aka variant rule (state.rs:116-121): the red five goes last
            if (d == T_5M && (S2.akas & 1) && c == 1) dt = T_5MR;
            else if (d == T_5P && (S2.akas & 2) && c == 1) dt = T_5PR;
            else if (d == T_5S && (S2.akas & 4) && c == 1) dt = T_5SR;
            sp_discard(S2, dt);
            SpIns I;
            sp_insert_begin(Wg, S2, I);
            bool fresh;
            const int cs = sp_insert_finish(Wg, X, I, fresh);
            if (fresh && cs >= 0) {
                const int idx = atomicAdd(&X->n_list, 1);
                if (idx < SP_CAP) Wg->list[idx] = (u32)cs;
                else X->overflow = 1;
            }
            const int pos = C->child_base[s] + (int)C->coff[s][t] + vidx * nk + rank;
            if (pos < SP_POOL) Wg->pool[pos] = (unsigned short)(cs < 0 ? 0xFFFF : cs);
        }
    }
    __syncthreads();
}

#if SP_VARIANT & 5
// ---------------------------------------------------------------------------------------------------------------------
// NEXT (experimental, compiled only with -DSP_VARIANT bit 0 or bit 2; NOT the default and not yet validated on a GPU):
// the same passes with a slimmer per-state scratch (the rows of h + t / h - d are gathered again where they are needed — L2
// hits — instead of being kept; the keep sets are 13-bit masks over the state's safe-discard ordinals until they are
// written out), as a template over the chunk size NS and the number of co-operating threads NT:
//   bit 0: NS = 32, NT = 256 — nine workgroup barriers and the gather latencies of a chunk are shared by twice as many
//          states, and the first passes fill 128-192 of the 256 lanes instead of 64-96;
//   bit 2: NS = 8,  NT = 64  — every wavefront runs its own chunks: no workgroup barrier inside a level at all (the passes
//          are separated by wave-level fences), four chunks in flight per workgroup.
template <int NS>
struct SpChunkT {
    u64 k[NS][4];
    u64 row[NS][4];
    u64 r2[NS][6];
    u64 r3[NS][4];
    u64 V[NS][13][3];    // per safe discard kind d and k-th other suit: merge(two untouched suits, row of h - d)
    u64 req[NS], safe[NS];
    u32 bkey[NS][4];
    u32 slot[NS];
    u32 keepw[NS][17];   // per required-tile ORDINAL ti: 16-bit mask over the safe-kind ordinals, two per word
    int item_off[NS + 1];
    int child_base[NS];
    unsigned short coff[NS][34];  // per required-tile ordinal: offset of its first child inside the state's child list
    u8 cnt[NS][4];       // pairs, kinds, yaokyuu pairs, yaokyuu kinds
    u8 tiles[NS][36], kinds[NS][16];
    u8 n_tiles[NS], n_kinds[NS];
};
static_assert(sizeof(SpChunkT<32>) <= sizeof(SpHalf) * (SP_THREADS / 16) && 4 * sizeof(SpChunkT<8>) <= sizeof(SpHalf) * (SP_THREADS / 16),
              "the chunk scratch must not grow the kernel's LDS");
template <int NT>
MJD void sp_sync() {  // barrier between two passes of NT co-operating threads
    if constexpr (NT == SP_THREADS) {
        __syncthreads();
    } else {
        mj_team_sync<NT>();
    }
}

template <class CT>
MJD SpState sp_chunk2_state(const CT* C, int s) {
    SpState S;
    S.h.mp = C->k[s][0];
    S.h.sz = C->k[s][1] & 0xFFFFFFFFFFFFull;
    S.w.mp = C->k[s][2];
    S.w.sz = C->k[s][3] & 0xFFFFFFFFFFFFull;
    S.akas = (u32)((C->k[s][1] >> 48) & 7) | ((u32)((C->k[s][3] >> 48) & 7) << 3);
    return S;
}
template <class CT>
MJD u32 sp_chunk2_keep(const CT* C, int s, int ti) { return (C->keepw[s][ti >> 1] >> (16 * (ti & 1))) & 0xFFFFu; }

// passes P0-P3 (see sp_chunk_probe); req / safe only, no rows kept
template <int NS, int NT>
__device__ __forceinline__ void sp_chunk2_probe(SP_HBM SpWork* Wg, SpCtx* X, SpChunkT<NS>* C, const ShTab& ST, int first, int n, int L) {
    const int tid = threadIdx.x & (NT - 1);
    const int ld3 = X->len_div3;
    for (int task = tid; task < n * 4; task += NT) {
        const int s = task >> 2, j = task & 3;
        const u32 slot = Wg->list[first + s];
        C->k[s][j] = reinterpret_cast<const SP_HBM u64*>(&Wg->node[slot])[j];
        if (j == 0) {
            C->slot[s] = slot;
            C->req[s] = 0;
            C->safe[s] = 0;
        }
    }
    sp_sync<NT>();
    for (int task = tid; task < n * 4; task += NT) {
        const int s = task >> 2, i = task & 3;
        const SpState S = sp_chunk2_state(C, s);
        const u32 key = i == 0 ? suit_key9(S.h.mp) : i == 1 ? suit_key9(S.h.mp >> 27) : i == 2 ? suit_key9(S.h.sz) : suit_key7(S.h.sz >> 27);
        C->bkey[s][i] = key;
        C->row[s][i] = sh_load(ST, i, key);
        C->cnt[s][i] = (u8)(i == 0 ? S.h.n_pairs() : i == 1 ? S.h.n_kinds() : i == 2 ? S.h.n_yao_pairs() : S.h.n_yao_kinds());
    }
    sp_sync<NT>();
    for (int task = tid; task < n * 6; task += NT) {
        const int s = task / 6, p = task % 6;
        const int a = p < 3 ? 0 : p < 5 ? 1 : 2, b = p < 3 ? p + 1 : p < 5 ? p - 1 : 3;
        C->r2[s][p] = sh_merge(C->row[s][a], C->row[s][b], ld3);
    }
    sp_sync<NT>();
    for (int task = tid; task < n * 4; task += NT) {
        const int s = task >> 2, i = task & 3;
        const u64 pr = i == 0 ? C->r2[s][3] : i == 1 ? C->r2[s][1] : C->r2[s][0];
        C->r3[s][i] = sh_merge(pr, C->row[s][i == 3 ? 2 : 3], ld3);
    }
    sp_sync<NT>();
    for (int task = tid; task < n * 34; task += NT) {
        const int s = task / 34, t = task % 34;
        const SpState S = sp_chunk2_state(C, s);
        const int st = sh_suit(t), hc = S.h.get(t), yao = (int)((YAOKYUU_MASK >> t) & 1);
        const bool in_wall = S.w.get(t) > 0, in_hand = L > 0 && hc > 0;
        const u32 kb = C->bkey[s][st], pw = sh_pow(t);
        const u64 rt = sh_load(ST, st, in_wall ? kb + pw : kb);
        const u64 rd = sh_load(ST, st, in_hand ? kb - pw : kb);
        const u64 r3 = C->r3[s][st];
        const int pairs = C->cnt[s][0], kinds = C->cnt[s][1], kpairs = C->cnt[s][2], kkinds = C->cnt[s][3];
        if (in_wall) {
            const int sh = sh_finish(sh_final(r3, rt, ld3), ld3, pairs + (hc == 1), kinds + (hc == 0), kpairs + (yao && hc == 1),
                                     kkinds + (yao && hc == 0));
            if (sh - L == -1) atomicOr((unsigned long long*)&C->req[s], 1ull << t);
        }
        if (in_hand) {
            const int sh = sh_finish(sh_final(r3, rd, ld3), ld3, pairs - (hc == 2), kinds - (hc == 1), kpairs - (yao && hc == 2),
                                     kkinds - (yao && hc == 1));
            if (sh <= L) atomicOr((unsigned long long*)&C->safe[s], 1ull << t);
        }
        if (t < 17) C->keepw[s][t] = 0;
    }
    sp_sync<NT>();
}

template <int NS, int NT>
__device__ __noinline__ void sp_l0_probe_chunk2(SpWork* W, SpCtx* X, SpChunkT<NS>* C, int first, int n) {
    SP_ASSUME_LDS(X);
    SP_ASSUME_LDS(C);
    SP_HBM SpWork* const Wg = (SP_HBM SpWork*)W;
    const ShTab ST = sh_tab(c_mj_tables);
    sp_chunk2_probe<NS, NT>(Wg, X, C, ST, first, n, 0);
    const int s = threadIdx.x & (NT - 1);
    if (s < n) {
        const SpState S = sp_chunk2_state(C, s);
        const u64 req = C->req[s];
        const u32 slot = C->slot[s];
        int cnt = 0;
        for (u64 rest = req; rest; rest &= rest - 1) {
            const int t = __ffsll((long long)rest) - 1;
            const bool aka_in_wall = (t == T_5M && (S.akas & 8)) || (t == T_5P && (S.akas & 16)) || (t == T_5S && (S.akas & 32));
            cnt += (!aka_in_wall || S.w.get(t) >= 2) + aka_in_wall;
        }
        if (cnt > SP_L0_MAX) { X->overflow = 1; cnt = SP_L0_MAX; }
        const int base = atomicAdd(&X->n_items, cnt);
        int e = 0;
        for (u64 rest = req; rest; rest &= rest - 1) {
            const int t = __ffsll((long long)rest) - 1;
            const bool aka_in_wall = (t == T_5M && (S.akas & 8)) || (t == T_5P && (S.akas & 16)) || (t == T_5S && (S.akas & 32));
            for (int variant = 0; variant < 2; variant++) {
                if (variant == 0 ? (aka_in_wall && S.w.get(t) < 2) : !aka_in_wall) continue;
                if (e < cnt) {
                    if (base + e < SP_ITEMS) Wg->items[base + e] = slot | ((u32)e << 14) | ((u32)t << 19) | ((u32)variant << 25);
                    else X->overflow = 1;
                }
                e++;
            }
        }
        SP_HBM SpNode& node = Wg->node[slot];
        node.req = req;
        node.child_off = 0;
    }
    sp_sync<NT>();
}

template <int NS, int NT>
__device__ __noinline__ void sp_expand_chunk2(SpWork* W, SpCtx* X, SpChunkT<NS>* C, int first, int n, int L) {
    SP_ASSUME_LDS(X);
    SP_ASSUME_LDS(C);
    SP_HBM SpWork* const Wg = (SP_HBM SpWork*)W;
    const ShTab ST = sh_tab(c_mj_tables);
    const int tid = threadIdx.x & (NT - 1);
    const int ld3 = X->len_div3;
    sp_chunk2_probe<NS, NT>(Wg, X, C, ST, first, n, L);
    // P4a: ascending lists of the required tiles and of the safe discard kinds
    for (int task = tid; task < n * 34; task += NT) {
        const int s = task / 34, t = task % 34;
        const u64 req = C->req[s], safe = C->safe[s], below = (1ull << t) - 1;
        if ((req >> t) & 1) C->tiles[s][__popcll(req & below)] = (u8)t;
        if ((safe >> t) & 1) C->kinds[s][__popcll(safe & below)] = (u8)t;
        if (t == 0) {
            C->n_tiles[s] = (u8)__popcll(req);
            C->n_kinds[s] = (u8)__popcll(safe);
        }
    }
    sp_sync<NT>();
    if (tid == 0) {
        int off = 0;
        for (int s = 0; s < n; s++) {
            C->item_off[s] = off;
            off += (int)C->n_tiles[s] * (int)C->n_kinds[s];
        }
        C->item_off[n] = off;
    }
    // P4b: V = merge(two untouched suits, row of h - d), the row gathered again (it was probed in P3: an L2 hit)
    for (int task = tid; task < n * 39; task += NT) {
        const int s = task / 39, q = task % 39, ki = q / 3, k = q % 3;
        if (ki >= (int)C->n_kinds[s]) continue;
        const int d = C->kinds[s][ki], sd = sh_suit(d), st = k + (k >= sd);
        int x = -1, y = -1;
        for (int i = 0; i < 4; i++)
            if (i != st && i != sd) { if (x < 0) x = i; else y = i; }
        const u64 rowd = sh_load(ST, sd, C->bkey[s][sd] - sh_pow(d));
        C->V[s][ki][k] = sh_merge(C->r2[s][sh_pair_idx(x, y)], rowd, ld3);
    }
    sp_sync<NT>();
    // P5: (state, required t, safe d) probes of h + t - d
    const int n_items = C->item_off[n];
    auto item_decode = [&](int it, int& s, int& ti, int& ki) {
        int lo = 0, hi = n;  // largest s with item_off[s] <= it
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (C->item_off[mid] <= it) lo = mid; else hi = mid;
        }
        s = lo;
        const int local = it - C->item_off[s], nk = C->n_kinds[s];
        ti = local / nk;
        ki = local - ti * nk;
    };
    for (int it = tid; it < n_items; it += NT) {
        int s, ti, ki;
        item_decode(it, s, ti, ki);
        const int t = C->tiles[s][ti], d = C->kinds[s][ki];
        if (d == t) continue;
        const SpState S = sp_chunk2_state(C, s);
        const int st = sh_suit(t), sd = sh_suit(d);
        const int c = S.h.get(d), hct = S.h.get(t), yt = (int)((YAOKYUU_MASK >> t) & 1), yd = (int)((YAOKYUU_MASK >> d) & 1);
        // one gather either way: the row of h+t-d (same suit) or the row of h+t (other suit)
        const u64 r = sh_load(ST, st, C->bkey[s][st] + sh_pow(t) - (sd == st ? sh_pow(d) : 0u));
        const int fin = sh_final(sd == st ? C->r3[s][st] : C->V[s][ki][st - (st > sd)], r, ld3);
        const int pairs = (int)C->cnt[s][0] + (hct == 1) - (c == 2), kinds = (int)C->cnt[s][1] + (hct == 0) - (c == 1);
        const int kpairs = (int)C->cnt[s][2] + (yt && hct == 1) - (yd && c == 2), kkinds = (int)C->cnt[s][3] + (yt && hct == 0) - (yd && c == 1);
        if (sh_finish(fin, ld3, pairs, kinds, kpairs, kkinds) - (L - 1) == 0) atomicOr(&C->keepw[s][ti >> 1], 1u << (ki + 16 * (ti & 1)));
    }
    sp_sync<NT>();
    // P6: child list layout per state + node header and keep sets (expanded from ordinals to tile masks)
    if (tid < n) {
        const int s = tid;
        const SpState S = sp_chunk2_state(C, s);
        int total = 0;
        const int nt = C->n_tiles[s];
        for (int ti = 0; ti < nt; ti++) {
            const int t = C->tiles[s][ti];
            const bool aka_in_wall = (t == T_5M && (S.akas & 8)) || (t == T_5P && (S.akas & 16)) || (t == T_5S && (S.akas & 32));
            const int nvar = aka_in_wall ? (S.w.get(t) >= 2 ? 2 : 1) : 1;
            C->coff[s][ti] = (unsigned short)total;
            total += nvar * __popc(sp_chunk2_keep(C, s, ti));
        }
        int child_base = atomicAdd(&X->n_pool, total);
        if (child_base + total > SP_POOL) { X->overflow = 1; child_base = 0; }
        C->child_base[s] = child_base;
        SP_HBM SpNode& node = Wg->node[C->slot[s]];
        node.child_off = (u32)child_base;
        node.req = C->req[s];
    }
    for (int task = tid; task < n * 34; task += NT) {
        const int s = task / 34, t = task % 34;
        const u64 req = C->req[s];
        u64 mask = 0;
        if ((req >> t) & 1) {
            for (u32 bits = sp_chunk2_keep(C, s, __popcll(req & ((1ull << t) - 1))); bits; bits &= bits - 1)
                mask |= 1ull << C->kinds[s][__ffs((int)bits) - 1];
        }
        Wg->node[C->slot[s]].keep[t] = mask;
    }
    sp_sync<NT>();
    // P7: children of the kept (t, d)
    for (int it = tid; it < n_items; it += NT) {
        int s, ti, ki;
        item_decode(it, s, ti, ki);
        const u32 bits = sp_chunk2_keep(C, s, ti);
        const int t = C->tiles[s][ti], d = C->kinds[s][ki];
        if (d == t || !((bits >> ki) & 1)) continue;
        const SpState S = sp_chunk2_state(C, s);
        const int nk = __popc(bits), rank = __popc(bits & ((1u << ki) - 1));
        const int cnt = S.w.get(t);
        const bool aka_in_wall = (t == T_5M && (S.akas & 8)) || (t == T_5P && (S.akas & 16)) || (t == T_5S && (S.akas & 32));
        for (int variant = 0; variant < 2; variant++) {
            int vidx;
            if (!aka_in_wall) { if (variant == 1) continue; vidx = 0; }
            else if (variant == 0) { if (cnt < 2) continue; vidx = 0; }
            else vidx = cnt >= 2 ? 1 : 0;
            const int tile = (aka_in_wall && variant == 1) ? akaize(t) : t;
            SpState S2 = S;
            sp_deal(S2, tile);
            const int c = S2.h.get(d);
            int dt = d;  // aka variant rule (state.rs:116-121): the red five goes last
            if (d == T_5M && (S2.akas & 1) && c == 1) dt = T_5MR;
            else if (d == T_5P && (S2.akas & 2) && c == 1) dt = T_5PR;
            else if (d == T_5S && (S2.akas & 4) && c == 1) dt = T_5SR;
            sp_discard(S2, dt);
            SpIns I;
            sp_insert_begin(Wg, S2, I);
            bool fresh;
            const int cs = sp_insert_finish(Wg, X, I, fresh);
            if (fresh && cs >= 0) {
                const int idx = atomicAdd(&X->n_list, 1);
                if (idx < SP_CAP) Wg->list[idx] = (u32)cs;
                else X->overflow = 1;
            }
            const int pos = C->child_base[s] + (int)C->coff[s][ti] + vidx * nk + rank;
            if (pos < SP_POOL) Wg->pool[pos] = (unsigned short)(cs < 0 ? 0xFFFF : cs);
        }
    }
    sp_sync<NT>();
}
#endif  // SP_VARIANT & 5

template <int J, int N, class F>
MJD void sp_static_for(F&& f) {  // f(integral_constant<J>) ... f(integral_constant<N - 1>): the index is a compile-time constant
    if constexpr (J < N) {
        f(std::integral_constant<int, J>{});
        sp_static_for<J + 1, N>(f);
    }
}
// lane N of every 16-lane DPP row to all lanes of that row (v_*_dpp row_newbcast:N, folded into the consuming instruction)
template <int N>
MJD float sp_row_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x150 + N, 0xF, 0xF, true));
}

// Evaluate one state (tenpai/win/ev arrays of its node) with a TEAM of TW lanes, one turn per lane.  TW = 16 whenever the
// row has at most 16 draws left (always, except during the first go-around of a kyoku): two states then share the 32
// lanes that one used to occupy, halving the instructions issued per state in the accumulate-bound evaluation pass.
template <int TW, class TMT>
__device__ __noinline__ void sp_eval_team(SpWork* W, SpCtx* X, TMT* TM, int slot, int L) {
    SP_ASSUME_LDS(X);
    SP_ASSUME_LDS(TM);
    SP_HBM SpWork* const Wg = (SP_HBM SpWork*)W;
    const int ln = threadIdx.x & (TW - 1);
    SP_HBM SpNode& node = Wg->node[slot];
    const SpState S = sp_state_of(node);
    const int T = X->T;
    float acc_t = 0.f, acc_w = 0.f, acc_e = 0.f;  // lane i: tenpai[i], win[i], ev[i]
    u64 req = 0;
    int child_base = 0;
    u32 l0_yaku = 0;  // level 0: bit i = draw entry i has a yaku
    {
        // One round trip for everything the node holds: key (above), req, child_off and the 272-byte keep[] area — the
        // keep sets left by the expansion pass (L > 0) or, in the same bytes, the 17 x 4 scores of sp_l0_score (L == 0).
        constexpr int NR = (34 + TW - 1) / TW;
        u64 kv[NR];
#pragma unroll
        for (int rnd = 0; rnd < NR; rnd++) {
            const int t = ln + TW * rnd;
            kv[rnd] = node.keep[min(t, 33)];
        }
        req = node.req;
        if (L > 0) child_base = (int)node.child_off;
        else l0_yaku = __hip_atomic_load(&node.child_off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // set by L2 atomics
        u32* sc32 = reinterpret_cast<u32*>(&TM->u.sc[0][0]);  // float[17][4] == 34 x 8 bytes
#pragma unroll
        for (int rnd = 0; rnd < NR; rnd++) {
            const int t = ln + TW * rnd;
            if (t < 34) {
                if (L > 0) {
                    TM->keep[t] = kv[rnd];
                } else {
                    sc32[2 * t] = (u32)kv[rnd];
                    sc32[2 * t + 1] = (u32)(kv[rnd] >> 32);
                }
            }
        }
        mj_team_sync<TW>();
    }

    // ---- D
    int sum_required = 0;
    for (u64 rest = req; rest; rest &= rest - 1) sum_required += S.w.get(__ffsll((long long)rest) - 1);
    sum_required &= 0xFF;
    const float* nt = X->not_tsumo[min(sum_required, 123)];
    const float my_m = ln < T ? nt[ln] : 0.f;  // not_tsumo_probs[i] of this lane's turn
    const bool assume_riichi = X->is_menzen && X->prefer_riichi;

    // accumulate one draw entry (calc.rs:486-548): lane i runs j = i .. T-1; `break`s become predicates (not_tsumo is
    // monotone).  nx_* = lane i's folded child values at turn i (L > 0) or scores (L == 0).
    // one (i = lane, j) term of calc.rs:486-548; vt/vw/ve = next[j + 1] of the folded child values, n = not_tsumo[j]
    auto term = [&](int j, float tpj, float n, float vt, float vw, float ve, bool is_scores, const float* scores) {
        const float prob = tpj * n / my_m;
        if (is_scores) {
            int han_plus = (int)(assume_riichi && X->calc_double_riichi && ln == 0) + (int)(assume_riichi && j == ln) +
                           (int)(X->calc_haitei && j == T - 1);
            acc_w += prob;
            acc_e += prob * (han_plus == 0 ? scores[0] : han_plus == 1 ? scores[1] : han_plus == 2 ? scores[2] : scores[3]);
        } else {
            if (L == 1) acc_t += prob;
            if (j < T - 1) {
                if (L > 1) acc_t += prob * vt;
                acc_w += prob * vw;
                acc_e += prob * ve;
            }
        }
    };
    // accumulate one draw entry (calc.rs:486-548): lane i runs j = i .. T-1; `break`s become predicates (not_tsumo is
    // monotone).  nx_* = lane i's folded child values at turn i (L > 0) or scores (L == 0).
    auto accumulate = [&](int count, float nx_t, float nx_w, float nx_e, bool is_scores, const float* scores) {
        const float* tp = X->tsumo_prob[count - 1];
        const bool lane_on = ln < T && my_m != 0.f;
        if constexpr (TW == 16) {
            // a 16-lane team is exactly one DPP row: lane j's registers (next[j], not_tsumo[j], tsumo_prob[j]) reach the
            // whole team through row_newbcast:j operands — no LDS traffic in the loop (unrolled: the lane is an immediate)
            const float my_tp = ln < T ? tp[ln] : 0.f;
            sp_static_for<0, 16>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if (j >= T) return;  // uniform
                const float vt = sp_row_bcast<(j + 1) & 15>(nx_t), vw = sp_row_bcast<(j + 1) & 15>(nx_w), ve = sp_row_bcast<(j + 1) & 15>(nx_e);
                const float n = sp_row_bcast<j>(my_m), tpj = sp_row_bcast<j>(my_tp);
                if (!(lane_on && j >= ln && n != 0.f)) return;
                term(j, tpj, n, vt, vw, ve, is_scores, scores);
            });
        } else {
            // rolled on purpose: the kernel is occupancy-limited by registers, not by code size
#pragma unroll 1
            for (int j = 0; j < T; j++) {
                // next[j + 1] comes from lane j + 1 (only used when j < T - 1)
                const float vt = __shfl(nx_t, (j + 1) & (TW - 1), TW);
                const float vw = __shfl(nx_w, (j + 1) & (TW - 1), TW);
                const float ve = __shfl(nx_e, (j + 1) & (TW - 1), TW);
                const float n = nt[j];
                if (!(lane_on && j >= ln && n != 0.f)) continue;
                term(j, tp[j], n, vt, vw, ve, is_scores, scores);
            }
        }
    };

    if (L == 0) {
        int idx = 0;  // draw entry index (same enumeration as sp_l0_probe)
        for (u64 rest = req; rest; rest &= rest - 1) {
            const int t = __ffsll((long long)rest) - 1;
            const int cnt = S.w.get(t);
            const bool aka_in_wall = (t == T_5M && (S.akas & 8)) || (t == T_5P && (S.akas & 16)) || (t == T_5S && (S.akas & 32));
            for (int variant = 0; variant < 2; variant++) {
                int count;
                if (!aka_in_wall) {
                    if (variant == 1) break;
                    count = cnt;
                } else if (variant == 0) {
                    if (cnt < 2) continue;
                    count = cnt - 1;
                } else {
                    count = 1;
                }
                const int e = idx++;
                if (e >= SP_L0_MAX || !((l0_yaku >> e) & 1)) continue;  // no yaku with this tile
                float scores[4];
#pragma unroll
                for (int q = 0; q < 4; q++) scores[q] = TM->u.sc[e][q];
                accumulate(count, 0.f, 0.f, 0.f, true, scores);
            }
        }
    } else {
        // Children are consumed in the reference's order (t ascending, plain before aka draw, discard ascending) but
        // FETCHED in batches: slots of a whole super-chunk in one coalesced read, then TMT::CH children's value arrays
        // per round trip — instead of two dependent gathers per child.
        const int n_tiles = __popcll(req);
#pragma unroll
        for (int rnd = 0; rnd < (34 + TW - 1) / TW; rnd++) {
            const int t = ln + TW * rnd;
            if (t < 34 && ((req >> t) & 1)) TM->tiles[__popcll(req & ((1ull << t) - 1))] = (u8)t;
        }
        mj_team_sync<TW>();
        int ti_next = 0, cpos = child_base;
        while (ti_next < n_tiles) {
            // tiles [ti_next, ti_end) whose children fit the staging area
            int n_ch = 0, ti_end = ti_next;
            while (ti_end < n_tiles) {
                const int t = TM->tiles[ti_end];
                const bool aka_in_wall = (t == T_5M && (S.akas & 8)) || (t == T_5P && (S.akas & 16)) || (t == T_5S && (S.akas & 32));
                const int nvar = aka_in_wall ? (S.w.get(t) >= 2 ? 2 : 1) : 1;
                const int c = nvar * __popcll(TM->keep[t]);
                if (n_ch + c > TMT::CCAP) break;
                if (ln == 0) TM->coff[t] = n_ch;
                n_ch += c;
                ti_end++;
            }
            mj_team_sync<TW>();
            for (int i = ln; i < n_ch; i += TW) TM->u.ev.cs[i] = Wg->pool[min(cpos + i, SP_POOL - 1)];
            // per-child metadata, one lane per draw entry (tile, variant)
            for (int g = ln; g < 2 * (ti_end - ti_next); g += TW) {
                const int t = TM->tiles[ti_next + (g >> 1)], variant = g & 1;
                const int cnt = S.w.get(t);
                const bool aka_in_wall = (t == T_5M && (S.akas & 8)) || (t == T_5P && (S.akas & 16)) || (t == T_5S && (S.akas & 32));
                int tile, count, vidx;
                if (!aka_in_wall) { if (variant == 1) continue; tile = t; count = cnt; vidx = 0; }
                else if (variant == 0) { if (cnt < 2) continue; tile = t; count = cnt - 1; vidx = 0; }
                else { tile = akaize(t); count = 1; vidx = cnt >= 2 ? 1 : 0; }
                const u32 akas1 = is_aka(tile) ? (S.akas | (1u << (tile - T_5MR))) : S.akas;  // akas_in_hand after the draw
                u64 rest = TM->keep[t];
                const int nk = __popcll(rest);
                int pos = TM->coff[t] + vidx * nk;
                for (int k = 0; k < nk; k++, pos++) {
                    const int d = __ffsll((long long)rest) - 1;
                    rest &= rest - 1;
                    const int c = S.h.get(d) + (d == t);
                    int dt = d;  // aka variant rule (state.rs:116-121): the red five goes last
                    if (d == T_5M && (akas1 & 1) && c == 1) dt = T_5MR;
                    else if (d == T_5P && (akas1 & 2) && c == 1) dt = T_5PR;
                    else if (d == T_5S && (akas1 & 4) && c == 1) dt = T_5SR;
                    TM->u.ev.meta[pos] = (unsigned short)(sp_discard_key(dt) | ((k == nk - 1) ? 512 : 0) | (count << 10));
                }
            }
            mj_team_sync<TW>();
            // discard_slow (calc.rs:570-637) fold state of the current draw entry
            float nx_t = -3.40282347e+38f, nx_w = -3.40282347e+38f, nx_e = -3.40282347e+38f;
            int max_value = INT_MIN, max_key = sp_discard_key(T_UNK);
            for (int c0 = 0; c0 < n_ch; c0 += TMT::CH) {
                float v[TMT::CH][3];
#pragma unroll
                for (int q = 0; q < TMT::CH; q++) {
                    v[q][0] = v[q][1] = v[q][2] = 0.f;
                    if (c0 + q < n_ch && ln < T) {
                        const int cs = TM->u.ev.cs[c0 + q];
                        if (cs != 0xFFFF) {
                            const SP_HBM SpNode& ch = Wg->node[cs];
                            v[q][0] = ch.tenpai[ln];
                            v[q][1] = ch.win[ln];
                            v[q][2] = ch.ev[ln];
                        }
                    }
                }
                if (ln < T) {
#pragma unroll
                    for (int q = 0; q < TMT::CH; q++) {
                        TM->u.ev.buf[q][0][ln] = v[q][0];
                        TM->u.ev.buf[q][1][ln] = v[q][1];
                        TM->u.ev.buf[q][2][ln] = v[q][2];
                    }
                }
                const int nq = min(TMT::CH, n_ch - c0);
                for (int q = 0; q < nq; q++) {
                    const int m = TM->u.ev.meta[c0 + q];
                    if (TM->u.ev.cs[c0 + q] == 0xFFFF) {
                        X->overflow = 1;
                    } else if (ln < T) {
                        const float ce = TM->u.ev.buf[q][2][ln];
                        const int value = (int)ce;  // `as i32` (maximize_win_prob = false)
                        const int key = m & 511;  // cmp_discard_priority(dt, max_tile) > 0  <=>  key > max_key
                        if (value > max_value || (value == max_value && key > max_key)) {
                            nx_t = TM->u.ev.buf[q][0][ln];
                            nx_w = TM->u.ev.buf[q][1][ln];
                            nx_e = ce;
                            max_value = value;
                            max_key = key;
                        }
                    }
                    if (m & 512) {  // last child of this draw entry
                        accumulate(m >> 10, nx_t, nx_w, nx_e, false, nullptr);
                        nx_t = nx_w = nx_e = -3.40282347e+38f;
                        max_value = INT_MIN;
                        max_key = sp_discard_key(T_UNK);
                    }
                }
            }
            cpos += n_ch;
            ti_next = ti_end;
        }
    }
    for (int k = ln; k < SP_T; k += TW) {  // entries past T (and past the team width) are zero
        const bool mine = k == ln && k < T;
        node.tenpai[k] = mine ? acc_t : 0.f;
        node.win[k] = mine ? acc_w : 0.f;
        node.ev[k] = mine ? acc_e : 0.f;
    }
}

#if SP_VARIANT & 2
// ---------------------------------------------------------------------------------------------------------------------
// NEXT (experimental, compiled only with -DSP_VARIANT=2 or 3; not yet validated on a GPU): evaluation with TWO turns per
// lane.  The accumulate of calc.rs:486-548 is triangular (turn i sums j = i .. T-1), so with one turn per lane half of the
// lane-iterations are masked off.  Here lane p of a TH-lane team owns the turns p and T-1-p: (T - p) + (p + 1) = T + 1
// terms for every lane, all lanes busy, and a state needs only ceil(T / 2) lanes — twice as many states per wavefront.
// Each turn still adds its terms in the reference's order (draw entries in order, j ascending), so the f32 results are
// the same bits.  The folded child values of a draw entry go through a small LDS array (the terms of a lane read next[j+1]
// for its own j).
struct SpPair {
    static constexpr int CH = 2, CCAP = 48;
    u64 keep[34];
    unsigned short coff[34];
    u8 tiles[36];
    union {
        float sc[SP_L0_MAX][4];
        struct {
            float nxs[3][SP_T + 1];      // next[0 .. T-1] of the current draw entry: tenpai / win / ev
            unsigned short cs[CCAP];
            unsigned short meta[CCAP];   // discard order key (9 bits) | last-of-group << 9 | draw count << 10
        } ev;
    } u;
};
static_assert(sizeof(SpPair) * (SP_THREADS / 8) <= sizeof(SpHalf) * (SP_THREADS / 16), "SpPair must not grow the kernel's LDS");

template <int TH>
__device__ __noinline__ void sp_eval_pair(SpWork* W, SpCtx* X, SpPair* TM, int slot, int L) {
    SP_ASSUME_LDS(X);
    SP_ASSUME_LDS(TM);
    SP_HBM SpWork* const Wg = (SP_HBM SpWork*)W;
    const int p = threadIdx.x & (TH - 1);
    SP_HBM SpNode& node = Wg->node[slot];
    const SpState S = sp_state_of(node);
    const int T = X->T;
    const int i1 = p, i2 = T - 1 - p;          // the two turns of this lane
    const bool on1 = i1 <= i2, on2 = i2 > i1;  // (the middle turn of an odd T is taken once; lanes past it idle)
    float a1t = 0.f, a1w = 0.f, a1e = 0.f, a2t = 0.f, a2w = 0.f, a2e = 0.f;
    u64 req = 0;
    int child_base = 0;
    u32 l0_yaku = 0;
    {
        constexpr int NR = (34 + TH - 1) / TH;
        u64 kv[NR];
#pragma unroll
        for (int rnd = 0; rnd < NR; rnd++) kv[rnd] = node.keep[min(p + TH * rnd, 33)];
        req = node.req;
        if (L > 0) child_base = (int)node.child_off;
        else l0_yaku = __hip_atomic_load(&node.child_off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u32* sc32 = reinterpret_cast<u32*>(&TM->u.sc[0][0]);
#pragma unroll
        for (int rnd = 0; rnd < NR; rnd++) {
            const int t = p + TH * rnd;
            if (t < 34) {
                if (L > 0) {
                    TM->keep[t] = kv[rnd];
                } else {
                    sc32[2 * t] = (u32)kv[rnd];
                    sc32[2 * t + 1] = (u32)(kv[rnd] >> 32);
                }
            }
        }
        mj_team_sync<TH>();
    }
    int sum_required = 0;
    for (u64 rest = req; rest; rest &= rest - 1) sum_required += S.w.get(__ffsll((long long)rest) - 1);
    sum_required &= 0xFF;
    const float* nt = X->not_tsumo[min(sum_required, 123)];
    const float m1 = on1 ? nt[i1] : 0.f, m2 = on2 ? nt[i2] : 0.f;
    const bool assume_riichi = X->is_menzen && X->prefer_riichi;
    const int len1 = on1 ? T - i1 : 0, len2 = on2 ? T - i2 : 0;

    // one draw entry: T + 1 steps, step q of a lane is the term (i1, i1 + q) while q < len1, then (i2, i2 + q - len1)
    auto accumulate = [&](int count, bool is_scores, const float* scores) {
        const float* tp = X->tsumo_prob[count - 1];
#pragma unroll 1
        for (int q = 0; q <= T; q++) {
            const bool first = q < len1;
            const bool act = first || (q - len1) < len2;
            const int i = first ? i1 : i2;
            const int j = first ? i1 + q : i2 + (q - len1);
            const float mm = first ? m1 : m2;
            if (!(act && mm != 0.f)) continue;
            const float n = nt[j];
            if (n == 0.f) continue;
            const float prob = tp[j] * n / mm;
            float at = first ? a1t : a2t, aw = first ? a1w : a2w, ae = first ? a1e : a2e;
            if (is_scores) {
                const int han_plus = (int)(assume_riichi && X->calc_double_riichi && i == 0) + (int)(assume_riichi && j == i) +
                                     (int)(X->calc_haitei && j == T - 1);
                aw += prob;
                ae += prob * (han_plus == 0 ? scores[0] : han_plus == 1 ? scores[1] : han_plus == 2 ? scores[2] : scores[3]);
            } else {
                if (L == 1) at += prob;
                if (j < T - 1) {
                    if (L > 1) at += prob * TM->u.ev.nxs[0][j + 1];
                    aw += prob * TM->u.ev.nxs[1][j + 1];
                    ae += prob * TM->u.ev.nxs[2][j + 1];
                }
            }
            if (first) { a1t = at; a1w = aw; a1e = ae; }
            else { a2t = at; a2w = aw; a2e = ae; }
        }
    };

    if (L == 0) {
        int idx = 0;
        for (u64 rest = req; rest; rest &= rest - 1) {
            const int t = __ffsll((long long)rest) - 1;
            const int cnt = S.w.get(t);
            const bool aka_in_wall = (t == T_5M && (S.akas & 8)) || (t == T_5P && (S.akas & 16)) || (t == T_5S && (S.akas & 32));
            for (int variant = 0; variant < 2; variant++) {
                int count;
                if (!aka_in_wall) {
                    if (variant == 1) break;
                    count = cnt;
                } else if (variant == 0) {
                    if (cnt < 2) continue;
                    count = cnt - 1;
                } else {
                    count = 1;
                }
                const int e = idx++;
                if (e >= SP_L0_MAX || !((l0_yaku >> e) & 1)) continue;
                float scores[4];
#pragma unroll
                for (int q = 0; q < 4; q++) scores[q] = TM->u.sc[e][q];
                accumulate(count, true, scores);
            }
        }
    } else {
        const int n_tiles = __popcll(req);
#pragma unroll
        for (int rnd = 0; rnd < (34 + TH - 1) / TH; rnd++) {
            const int t = p + TH * rnd;
            if (t < 34 && ((req >> t) & 1)) TM->tiles[__popcll(req & ((1ull << t) - 1))] = (u8)t;
        }
        mj_team_sync<TH>();
        int ti_next = 0, cpos = child_base;
        while (ti_next < n_tiles) {
            int n_ch = 0, ti_end = ti_next;
            while (ti_end < n_tiles) {
                const int t = TM->tiles[ti_end];
                const bool aka_in_wall = (t == T_5M && (S.akas & 8)) || (t == T_5P && (S.akas & 16)) || (t == T_5S && (S.akas & 32));
                const int nvar = aka_in_wall ? (S.w.get(t) >= 2 ? 2 : 1) : 1;
                const int c = nvar * __popcll(TM->keep[t]);
                if (n_ch + c > SpPair::CCAP) break;
                if (p == 0) TM->coff[t] = (unsigned short)n_ch;
                n_ch += c;
                ti_end++;
            }
            mj_team_sync<TH>();
            for (int i = p; i < n_ch; i += TH) TM->u.ev.cs[i] = Wg->pool[min(cpos + i, SP_POOL - 1)];
            for (int g = p; g < 2 * (ti_end - ti_next); g += TH) {
                const int t = TM->tiles[ti_next + (g >> 1)], variant = g & 1;
                const int cnt = S.w.get(t);
                const bool aka_in_wall = (t == T_5M && (S.akas & 8)) || (t == T_5P && (S.akas & 16)) || (t == T_5S && (S.akas & 32));
                int tile, count, vidx;
                if (!aka_in_wall) { if (variant == 1) continue; tile = t; count = cnt; vidx = 0; }
                else if (variant == 0) { if (cnt < 2) continue; tile = t; count = cnt - 1; vidx = 0; }
                else { tile = akaize(t); count = 1; vidx = cnt >= 2 ? 1 : 0; }
                const u32 akas1 = is_aka(tile) ? (S.akas | (1u << (tile - T_5MR))) : S.akas;
                u64 rest = TM->keep[t];
                const int nk = __popcll(rest);
                int pos = (int)TM->coff[t] + vidx * nk;
                for (int k = 0; k < nk; k++, pos++) {
                    const int d = __ffsll((long long)rest) - 1;
                    rest &= rest - 1;
                    const int c = S.h.get(d) + (d == t);
                    int dt = d;
                    if (d == T_5M && (akas1 & 1) && c == 1) dt = T_5MR;
                    else if (d == T_5P && (akas1 & 2) && c == 1) dt = T_5PR;
                    else if (d == T_5S && (akas1 & 4) && c == 1) dt = T_5SR;
                    TM->u.ev.meta[pos] = (unsigned short)(sp_discard_key(dt) | ((k == nk - 1) ? 512 : 0) | (count << 10));
                }
            }
            mj_team_sync<TH>();
            // discard_slow (calc.rs:570-637) fold state of the current draw entry, one per turn of the lane
            float n1t = -3.40282347e+38f, n1w = -3.40282347e+38f, n1e = -3.40282347e+38f;
            float n2t = -3.40282347e+38f, n2w = -3.40282347e+38f, n2e = -3.40282347e+38f;
            int mv1 = INT_MIN, mk1 = sp_discard_key(T_UNK), mv2 = INT_MIN, mk2 = sp_discard_key(T_UNK);
            for (int c0 = 0; c0 < n_ch; c0 += SpPair::CH) {
                float v1[SpPair::CH][3], v2[SpPair::CH][3];
#pragma unroll
                for (int q = 0; q < SpPair::CH; q++) {
                    v1[q][0] = v1[q][1] = v1[q][2] = 0.f;
                    v2[q][0] = v2[q][1] = v2[q][2] = 0.f;
                    if (c0 + q < n_ch) {
                        const int cs = TM->u.ev.cs[c0 + q];
                        if (cs != 0xFFFF) {
                            const SP_HBM SpNode& ch = Wg->node[cs];
                            if (on1) { v1[q][0] = ch.tenpai[i1]; v1[q][1] = ch.win[i1]; v1[q][2] = ch.ev[i1]; }
                            if (on2) { v2[q][0] = ch.tenpai[i2]; v2[q][1] = ch.win[i2]; v2[q][2] = ch.ev[i2]; }
                        }
                    }
                }
                const int nq = min(SpPair::CH, n_ch - c0);
#pragma unroll
                for (int q = 0; q < SpPair::CH; q++) {
                    if (q >= nq) break;
                    const int m = TM->u.ev.meta[c0 + q];
                    if (TM->u.ev.cs[c0 + q] == 0xFFFF) {
                        X->overflow = 1;
                    } else {
                        const int key = m & 511;
                        if (on1) {
                            const int value = (int)v1[q][2];  // `as i32` (maximize_win_prob = false)
                            if (value > mv1 || (value == mv1 && key > mk1)) { n1t = v1[q][0]; n1w = v1[q][1]; n1e = v1[q][2]; mv1 = value; mk1 = key; }
                        }
                        if (on2) {
                            const int value = (int)v2[q][2];
                            if (value > mv2 || (value == mv2 && key > mk2)) { n2t = v2[q][0]; n2w = v2[q][1]; n2e = v2[q][2]; mv2 = value; mk2 = key; }
                        }
                    }
                    if (m & 512) {  // last child of this draw entry: publish next[], add the entry's terms
                        if (on1) { TM->u.ev.nxs[0][i1] = n1t; TM->u.ev.nxs[1][i1] = n1w; TM->u.ev.nxs[2][i1] = n1e; }
                        if (on2) { TM->u.ev.nxs[0][i2] = n2t; TM->u.ev.nxs[1][i2] = n2w; TM->u.ev.nxs[2][i2] = n2e; }
                        mj_team_sync<TH>();
                        accumulate(m >> 10, false, nullptr);
                        mj_team_sync<TH>();
                        n1t = n1w = n1e = n2t = n2w = n2e = -3.40282347e+38f;
                        mv1 = mv2 = INT_MIN;
                        mk1 = mk2 = sp_discard_key(T_UNK);
                    }
                }
            }
            cpos += n_ch;
            ti_next = ti_end;
        }
    }
    if (on1) { node.tenpai[i1] = a1t; node.win[i1] = a1w; node.ev[i1] = a1e; }
    if (on2) { node.tenpai[i2] = a2t; node.win[i2] = a2w; node.ev[i2] = a2e; }
    for (int k = T + p; k < SP_T; k += TH) {  // entries past T are zero
        node.tenpai[k] = 0.f;
        node.win[k] = 0.f;
        node.ev[k] = 0.f;
    }
}
#endif  // SP_VARIANT & 2

MJD int f32_total_cmp(float a, float b) {
    int x = __float_as_int(a), y = __float_as_int(b);
    x ^= (int)((unsigned)(x >> 31) >> 1);
    y ^= (int)((unsigned)(y >> 31) >> 1);
    return (x > y) - (x < y);
}

__global__ __launch_bounds__(SP_THREADS, 4) void mj_k_sp(SpParams P) {
    __shared__ SpCtx X;
    __shared__ TableOne st;
    __shared__ int s_row;
    __shared__ union SpTeams {
        SpTeam full[SP_THREADS / 32];
        SpChunk chunk;
#if SP_VARIANT & 4
        SpChunkT<8> wchunk[SP_THREADS / 64];
#elif SP_VARIANT & 1
        SpChunkT<32> chunk2;
#endif
#if SP_VARIANT & 2
        SpPair pair[SP_THREADS / 8];
#endif
        SpHalf half[SP_THREADS / 16];
        SpQuarter quarter[SP_THREADS / 8];
        struct {                 // row set-up (candidates + their required tiles), before any team runs
            u64 r2[6], r3[4];    // partial merges of the root hand's rows (mj_algo.h sh_merge)
            u64 rowt[34];        // row of root + t in suit(t)
            u64 rowd[34];        // row of root - d in suit(d)
            u64 U[34][3];        // discard d, k-th other suit: merge(two untouched suits, rowd[d])
            int sh_d[34];        // shanten of root - d (only for tiles in hand)
            u64 req[SP_MAX_CAND];
            int nreq[SP_MAX_CAND];
        } setup;
    } s_tm;
    SpTeam* s_team = s_tm.full;
    SpWork* W = P.work + blockIdx.x;
    const int tid = threadIdx.x;
    constexpr int O_SP = 889;  // Lay<4>::sp

    // the hash tags must start empty
    for (int i = tid; i < SP_CAP; i += SP_THREADS) W->tag[i] = 0ull;
    __syncthreads();

    for (;;) {
        if (tid == 0) s_row = atomicAdd(P.queue, 1);
        __syncthreads();
        const int row = s_row;
        if (row >= P.n_rows) break;
        const uint32_t desc = P.rows[row];
        const int table = ROW_TABLE(desc), p = ROW_SEAT(desc);
        long long t_0 = wall_clock64(), t_1 = t_0, t_2 = t_0, t_3 = t_0, t_4 = t_0;
        {
            const float4* src = reinterpret_cast<const float4*>(P.snap + table);
            float4* d4 = reinterpret_cast<float4*>(&st);
            for (int i = tid; i < (int)(sizeof(TableOne) / 16); i += SP_THREADS) d4[i] = src[i];
        }
        __syncthreads();
        LaneT<TableOne> L;
        L.B = &st;
        L.l = 0;
        L.T = &c_mj_tables;
        float* out = P.obs + (size_t)row * (1012 * 34);
        const u32 cans = F1(cans, p);
        const bool can_discard0 = (cans & CAN_DISCARD) != 0;
        const Hand h0 = load_hand(L, p);
        const int ld3 = F1(len_div3, p);
        const int tiles_left = F(tiles_left);

        // ---- single_player_tables preconditions (agent_helper.rs:509-530) + real_time_shanten (:467-503)
        int cur_shanten;
        {
            const int sh = F1(shanten, p);
            if (!can_discard0) cur_shanten = sh;
            else if (sh > 0) cur_shanten = F1(has_next_shanten, p) ? sh - 1 : sh;
            else if (F1(last_self_tsumo, p) != MJ_NONE) cur_shanten = ((F1(waits, p) >> deaka(F1(last_self_tsumo, p))) & 1) ? -1 : 0;
            else cur_shanten = calc_all(c_mj_tables, h0, ld3);
        }
        int tsumos_left, calc_haitei;
        if (can_discard0) {
            tsumos_left = tiles_left / 4;
            calc_haitei = tiles_left % 4 == 0;
        } else {
            int target = (F1(cans_target, p) + 4 - p) & 3;
            int at_next = max(tiles_left - (4 - target), 0);
            tsumos_left = at_next / 4;
            calc_haitei = at_next % 4 == 0;
        }
        const bool ok = tiles_left >= 4 && cur_shanten >= 0 && tsumos_left >= 1;
        __syncthreads();
        if (!ok) {
            // Err path (obs_repr.rs:612-623): max EV = minimal tsumo agari points, everything else zero
            if (tid == 0) {
                float v = 0.f;
                if (cans & CAN_AGARI) {
                    const bool is_ron = (cans & CAN_RON_AGARI) != 0;
                    if ((is_ron && (cans & CAN_RON_AGARI)) || (cans & CAN_TSUMO_AGARI)) {
                        Point pt;
                        if (seat_agari_points(L, p, is_ron, 0, pt)) v = (float)tsumo_total(pt, p == (F(kyoku) & 3));
                    }
                }
                X.cand_ev0[0] = v;
            }
            __syncthreads();
            const float v = X.cand_ev0[0];
            if (tid < 34) {
                out[(O_SP + 0) * 34 + tid] = fminf(fmaxf(v, 0.f), 100000.f) / 100000.f;
                out[(O_SP + 1) * 34 + tid] = fminf(fmaxf(v, 0.f), 30000.f) / 30000.f;
            }
            __syncthreads();
            continue;
        }

        // ---- calculator set-up (agent_helper.rs:532-586, calc.rs:84-167)
        bool can_discard = can_discard0;
        SpState root;
        root.h = h0;
        int akas_hand = F1(akas_in_hand, p) & 7;
        const bool after_riichi = can_discard0 && accepted(L, p);
        const int last_tsumo = F1(last_self_tsumo, p);
        if (after_riichi) {
            root.h.dec(deaka(last_tsumo));
            if (is_aka(last_tsumo)) akas_hand &= ~(1 << (last_tsumo - T_5MR));
            can_discard = false;
        }
        {
            Hand w = {0, 0};
            for (int t = 0; t < 34; t++) {
                int seen = F1(pub_seen, t) + h0.get(t);  // tiles_seen = public + own hand (incl. the tile just drawn)
                int left = (4 - seen) & 7;
                for (int k = 0; k < left; k++) w.inc(t);
            }
            root.w = w;
            int akas_seen = (F(pub_aka_seen) | F1(akas_in_hand, p)) & 7;
            root.akas = (u32)akas_hand | ((u32)(~akas_seen & 7) << 3);
        }
        if (tid == 0) {
            X.melds = load_melds(L, p);
            X.len_div3 = ld3;
            X.bakaze = table_bakaze(L);
            X.jikaze = seat_jikaze(L, p);
            X.is_menzen = (F1(pflags, p) & PF_IS_MENZEN) != 0;
            X.n_dora = F(n_dora_ind);
            for (int i = 0; i < 5; i++) X.dora_ind[i] = i < X.n_dora ? F1(dora_ind, i) : T_UNK;
            // num_doras_in_fuuro (agent_helper.rs:533-545) = doras_owned[0] - doras in hand - akas in hand
            int nf = 0;
            if (!(X.is_menzen && F1(ankan_n, p) == 0)) {
                int fn = F1(fuuro_n, p);
                for (int k = 0; k < fn; k++)
                    for (int j = 0; j < 4; j++) {
                        int t = F3(fuuro, p, k, j);
                        if (t != MJ_NONE) nf += dora_factor(L, deaka(t)) + (is_aka(t) ? 1 : 0);
                    }
                int na = F1(ankan_n, p);
                for (int k = 0; k < na; k++) {
                    int t = F2(ankan, p, k);
                    nf += 4 * dora_factor(L, t) + ((t == T_5M || t == T_5P || t == T_5S) ? 1 : 0);
                }
            }
            X.num_doras_in_fuuro = nf & 0xFF;
            X.prefer_riichi = F1(scores, p) >= 1000;
            X.calc_double_riichi = can_discard0 && (F1(pflags, p) & PF_CAN_W_RIICHI) != 0;
            X.calc_haitei = calc_haitei;
            X.T = tsumos_left;
            int n_left = 0;
            for (int t = 0; t < 34; t++) n_left += root.w.get(t);
            X.n_left = n_left & 0xFF;
            X.n_list = 0;
            X.n_pool = 0;
            X.overflow = 0;
            X.prof = P.prof;
            X.n_cand = 0;
            for (int l = 0; l < 5; l++) X.lvl_begin[l] = X.lvl_end[l] = 0;
        }
        __syncthreads();
        const int T = X.T, n_left = X.n_left;
        // build_tsumo_prob_table / build_not_tsumo_prob_table (calc.rs:135-167)
        if (tid < 4 * SP_T) {
            int i = tid / SP_T, j = tid % SP_T;
            X.tsumo_prob[i][j] = j < T ? (float)(i + 1) / (float)(n_left - j) : 0.f;
        }
        if (tid < 124) {
            float* r = X.not_tsumo[tid];
            for (int j = 0; j < SP_T; j++) r[j] = 0.f;
            if (tid <= 122 && tid < n_left + 1) {
                r[0] = 1.f;
                int lim = min(T - 1, n_left - tid);
                for (int j = 0; j < lim; j++) r[j + 1] = r[j] * (float)(n_left - tid - j) / (float)(n_left - j);
            }
        }
        __syncthreads();

        // ---- candidates: analyze_discard / analyze_draw (+ *_simple for shanten > 3)  (calc.rs:205-312).
        // All shanten numbers here are of hands one or two tiles away from the root hand, so they are incremental probes
        // (one table gather + one final merge step) on partial merges shared by the whole row, spread over the workgroup.
        auto& SU = s_tm.setup;
        const ShTab ST = sh_tab(c_mj_tables);
        const ShBase RB = sh_base(ST, root.h);
        const u64 root_mask = root.h.nonzero_mask();
        if (tid < 6) {
            const int a = tid < 3 ? 0 : tid < 5 ? 1 : 2, b = tid < 3 ? tid + 1 : tid < 5 ? tid - 1 : 3;
            SU.r2[tid] = sh_merge(RB.row_of(a), RB.row_of(b), ld3);
        } else if (tid >= 64 && tid < 98) {
            const int t = tid - 64, st = sh_suit(t);
            const u32 kb = RB.key_of(st), pw = sh_pow(t);
            const bool in_wall = root.w.get(t) > 0, in_hand = can_discard && ((root_mask >> t) & 1);
            const u64 rt = sh_load(ST, st, in_wall ? kb + pw : kb), rd = sh_load(ST, st, in_hand ? kb - pw : kb);
            SU.rowt[t] = in_wall ? rt : 0ull;
            SU.rowd[t] = in_hand ? rd : 0ull;
        }
        if (tid < SP_MAX_CAND) { SU.req[tid] = 0; SU.nreq[tid] = 0; }
        __syncthreads();
        if (tid < 4) {
            const u64 pr = tid == 0 ? SU.r2[3] : tid == 1 ? SU.r2[1] : SU.r2[0];
            SU.r3[tid] = sh_merge(pr, tid == 3 ? RB.row[2] : RB.row[3], ld3);
        } else if (can_discard && tid >= 64 && tid < 64 + 34 * 3) {
            const int d = (tid - 64) / 3, k = (tid - 64) % 3;
            if ((root_mask >> d) & 1) {
                const int sd = sh_suit(d), st = k + (k >= sd);
                int x = -1, y = -1;
                for (int q = 0; q < 4; q++)
                    if (q != sd && q != st) { if (x < 0) x = q; else y = q; }
                SU.U[d][k] = sh_merge(SU.r2[sh_pair_idx(x, y)], SU.rowd[d], ld3);
            }
        }
        __syncthreads();
        if (can_discard && tid < 34 && ((root_mask >> tid) & 1)) {  // shanten of root - d
            const int d = tid, sd = sh_suit(d), hd = root.h.get(d), yd = (int)((YAOKYUU_MASK >> d) & 1);
            SU.sh_d[d] = sh_finish(sh_final(SU.r3[sd], SU.rowd[d], ld3), ld3, RB.pairs - (hd == 2), RB.kinds - (hd == 1),
                                   RB.kpairs - (yd && hd == 2), RB.kkinds - (yd && hd == 1));
        }
        __syncthreads();
        if (tid == 0) {
            int n = 0;
            if (can_discard) {
                for (int d = 0; d < 34; d++) {
                    int c = root.h.get(d);
                    if (c == 0) continue;
                    int diff = SU.sh_d[d] - cur_shanten;
                    int dt = d;
                    if (d == T_5M && (root.akas & 1) && c == 1) dt = T_5MR;
                    else if (d == T_5P && (root.akas & 2) && c == 1) dt = T_5PR;
                    else if (d == T_5S && (root.akas & 4) && c == 1) dt = T_5SR;
                    if (cur_shanten <= 3 && diff != 0) continue;
                    X.cand_tile[n] = dt;
                    X.cand_down[n] = cur_shanten > 3 && diff == 1;
                    n++;
                }
            } else {
                X.cand_tile[0] = T_UNK;
                X.cand_down[0] = 0;
                n = 1;
            }
            X.n_cand = n;
        }
        __syncthreads();
        const int n_cand = X.n_cand;
        // required tiles of every candidate (state.rs:176-200): draws t that lower the shanten number of root - d (+ t)
        for (int w = tid; w < n_cand * 34; w += SP_THREADS) {
            const int c = w / 34, t = w % 34;
            const int wc = root.w.get(t);  // the discard does not change the wall
            if (wc == 0) continue;
            const int st = sh_suit(t), ht = root.h.get(t), yt = (int)((YAOKYUU_MASK >> t) & 1);
            int sh_base_c, sh_new;
            if (can_discard) {
                const int d = deaka(X.cand_tile[c]), sd = sh_suit(d), hd = root.h.get(d), yd = (int)((YAOKYUU_MASK >> d) & 1);
                sh_base_c = SU.sh_d[d];
                if (t == d) {
                    sh_new = calc_all(c_mj_tables, root.h, ld3);  // root - d + d
                } else {
                    int fin;
                    if (sd == st) fin = sh_final(SU.r3[st], sh_load(ST, st, RB.key_of(st) + sh_pow(t) - sh_pow(d)), ld3);
                    else fin = sh_final(SU.U[d][st - (st > sd)], SU.rowt[t], ld3);
                    sh_new = sh_finish(fin, ld3, RB.pairs - (hd == 2) + (ht == 1), RB.kinds - (hd == 1) + (ht == 0),
                                       RB.kpairs - (yd && hd == 2) + (yt && ht == 1), RB.kkinds - (yd && hd == 1) + (yt && ht == 0));
                }
            } else {
                sh_base_c = calc_all(c_mj_tables, root.h, ld3);
                sh_new = sh_finish(sh_final(SU.r3[st], SU.rowt[t], ld3), ld3, RB.pairs + (ht == 1), RB.kinds + (ht == 0),
                                   RB.kpairs + (yt && ht == 1), RB.kkinds + (yt && ht == 0));
            }
            if (sh_new < sh_base_c) {
                atomicOr((unsigned long long*)&SU.req[c], 1ull << t);
                atomicAdd(&SU.nreq[c], wc);
            }
        }
        __syncthreads();
        if (tid < n_cand) {
            X.cand_req[tid] = SU.req[tid];
            X.cand_nreq[tid] = SU.nreq[tid] & 0xFF;
            X.cand_slot[tid] = -1;
            X.cand_tp0[tid] = X.cand_wp0[tid] = X.cand_ev0[tid] = 0.f;
        }
        __syncthreads();

        // fewer draws left than the shanten number: tenpai / win / EV are exactly zero for every candidate (reaching tenpai
        // takes cur_shanten draws), so the state graph need not be built at all
        const bool with_probs = cur_shanten <= 3 && X.T >= cur_shanten;
        t_1 = wall_clock64();
        t_2 = t_3 = t_4 = t_1;
        if (with_probs) {
            // root states = level cur_shanten
            if (tid == 0) {
                for (int c = 0; c < n_cand; c++) {
                    SpState s = root;
                    if (can_discard) sp_discard(s, X.cand_tile[c]);
                    bool fresh;
                    int slot = sp_insert(W, &X, s, fresh);
                    X.cand_slot[c] = slot;
                    if (fresh && slot >= 0) W->list[X.n_list++] = (u32)slot;
                }
                X.lvl_begin[cur_shanten] = 0;
                X.lvl_end[cur_shanten] = X.n_list;
            }
            __syncthreads();
            // expand top-down
            for (int lv = cur_shanten; lv >= 1; lv--) {
                const int b = X.lvl_begin[lv], e = X.lvl_end[lv];
#if SP_VARIANT & 4
                for (int c0 = b + 8 * (tid >> 6); c0 < e; c0 += 8 * (SP_THREADS / 64))  // every wavefront its own chunks
                    sp_expand_chunk2<8, 64>(W, &X, &s_tm.wchunk[tid >> 6], c0, min(8, e - c0), lv);
#elif SP_VARIANT & 1
                for (int c0 = b; c0 < e; c0 += 32) sp_expand_chunk2<32, SP_THREADS>(W, &X, &s_tm.chunk2, c0, min(32, e - c0), lv);
#else
                for (int c0 = b; c0 < e; c0 += SP_NS) sp_expand_chunk(W, &X, &s_tm.chunk, c0, min(SP_NS, e - c0), lv);
#endif
                __syncthreads();
                if (tid == 0) {
                    X.lvl_begin[lv - 1] = e;
                    X.lvl_end[lv - 1] = min(X.n_list, SP_CAP);
                }
                __syncthreads();
            }
            t_2 = wall_clock64();
            // evaluate bottom-up
            for (int lv = 0; lv <= cur_shanten; lv++) {
                const int b = X.lvl_begin[lv], e = X.lvl_end[lv];
                if (lv == 0) {
                    if (tid == 0) X.n_items = 0;
                    __syncthreads();
#if SP_VARIANT & 4
                    for (int c0 = b + 8 * (tid >> 6); c0 < e; c0 += 8 * (SP_THREADS / 64))
                        sp_l0_probe_chunk2<8, 64>(W, &X, &s_tm.wchunk[tid >> 6], c0, min(8, e - c0));
#elif SP_VARIANT & 1
                    for (int c0 = b; c0 < e; c0 += 32) sp_l0_probe_chunk2<32, SP_THREADS>(W, &X, &s_tm.chunk2, c0, min(32, e - c0));
#else
                    for (int c0 = b; c0 < e; c0 += SP_NS) sp_l0_probe_chunk(W, &X, &s_tm.chunk, c0, min(SP_NS, e - c0));
#endif
                    __syncthreads();
                    const int n_items = min(X.n_items, SP_ITEMS);
                    for (int i = tid; i < n_items; i += SP_THREADS) sp_l0_score(c_mj_tables, W, &X, W->items[i]);
                    __syncthreads();
                }
                if (T <= 8) {
                    for (int i = b + (tid >> 3); i < e; i += SP_THREADS / 8)
                        sp_eval_team<8, SpQuarter>(W, &X, &s_tm.quarter[tid >> 3], (int)W->list[i], lv);
#if SP_VARIANT & 2
                } else if (T <= 16) {
                    for (int i = b + (tid >> 3); i < e; i += SP_THREADS / 8)
                        sp_eval_pair<8>(W, &X, &s_tm.pair[tid >> 3], (int)W->list[i], lv);
                } else {
                    for (int i = b + (tid >> 4); i < e; i += SP_THREADS / 16)
                        sp_eval_pair<16>(W, &X, &s_tm.pair[tid >> 4], (int)W->list[i], lv);
                }
#else
                } else if (T <= 16) {
                    for (int i = b + (tid >> 4); i < e; i += SP_THREADS / 16)
                        sp_eval_team<16, SpHalf>(W, &X, &s_tm.half[tid >> 4], (int)W->list[i], lv);
                } else {
                    for (int i = b + (tid >> 5); i < e; i += SP_THREADS / 32)
                        sp_eval_team<32, SpTeam>(W, &X, &s_team[tid >> 5], (int)W->list[i], lv);
                }
#endif
                __syncthreads();
                if (lv == 0) t_3 = wall_clock64();
            }
            t_4 = wall_clock64();
        }

        // ---- sort (calc.rs:181-188 / 196-199) + encode (obs_repr.rs:564-692)
        if (tid == 0) {
            for (int c = 0; c < n_cand; c++) {
                X.order[c] = c;
                int slot = X.cand_slot[c];
                if (with_probs && slot >= 0) {  // Candidate::from clamps (candidate.rs:46-70); shanten 0 => tenpai = 1
                    const SpNode& nd = W->node[slot];
                    float tp = cur_shanten == 0 ? 1.f : nd.tenpai[0];
                    X.cand_tp0[c] = fminf(fmaxf(tp, 0.f), 1.f);
                    X.cand_wp0[c] = fminf(fmaxf(nd.win[0], 0.f), 1.f);
                    X.cand_ev0[c] = fmaxf(nd.ev[0], 0.f);
                }
            }
            auto cmp = [&](int l, int r, int by) -> int {  // candidate.rs:73-106, by: 0 EV, 3 NotShantenDown
                if (X.cand_tile[l] == X.cand_tile[r]) return 0;
                int o;
                if (by <= 0 && (o = f32_total_cmp(X.cand_ev0[l], X.cand_ev0[r])) != 0) return o;
                if (by <= 1 && (o = f32_total_cmp(X.cand_wp0[l], X.cand_wp0[r])) != 0) return o;
                if (by <= 2 && (o = f32_total_cmp(X.cand_tp0[l], X.cand_tp0[r])) != 0) return o;
                if (!X.cand_down[l] && X.cand_down[r]) return 1;
                if (X.cand_down[l] && !X.cand_down[r]) return -1;
                if (X.cand_nreq[l] != X.cand_nreq[r]) return X.cand_nreq[l] < X.cand_nreq[r] ? -1 : 1;
                return cmp_discard_priority(X.cand_tile[l], X.cand_tile[r]);
            };
            const int by = with_probs ? 0 : 3;
            for (int i = 1; i < n_cand; i++) {  // stable insertion sort, descending: before(l, r) = cmp(r, l) < 0
                int v = X.order[i], j = i - 1;
                while (j >= 0 && cmp(X.order[j], v, by) < 0) {
                    X.order[j + 1] = X.order[j];
                    j--;
                }
                X.order[j + 1] = v;
            }
            // the candidate with the most required tiles: Iterator::max_by keeps the LAST maximum (obs_repr.rs:589-596)
            int best = -1;
            for (int k = 0; k < n_cand; k++) {
                int c = X.order[k];
                if (best < 0 || cmp(c, best, 3) >= 0) best = c;
            }
            X.lvl_begin[4] = best;
        }
        __syncthreads();
        {
            const int first = n_cand > 0 ? X.order[0] : -1;
            const float max_ev = (with_probs && first >= 0 && T > 0) ? X.cand_ev0[first] : 0.f;
            if (tid < 34) {
                out[(O_SP + 0) * 34 + tid] = fminf(fmaxf(max_ev, 0.f), 100000.f) / 100000.f;
                out[(O_SP + 1) * 34 + tid] = fminf(fmaxf(max_ev, 0.f), 30000.f) / 30000.f;
            }
            // required tiles
            if (can_discard0 && !after_riichi) {
                for (int c = tid / 34; c < n_cand; c += SP_THREADS / 34) {
                    int t = tid % 34;
                    if (tid >= (SP_THREADS / 34) * 34) break;
                    if ((X.cand_req[c] >> t) & 1) {
                        int dtid = deaka(X.cand_tile[c]);
                        out[(O_SP + 2 + (X.cand_down[c] ? 34 : 0) + dtid) * 34 + t] = 1.f;
                    }
                }
                if (tid == 0 && X.lvl_begin[4] >= 0) out[(O_SP + 70) * 34 + deaka(X.cand_tile[X.lvl_begin[4]])] = 1.f;
            } else if (can_discard0) {
                // discard after riichi: `cans.can_discard` is still true in the encoder (obs_repr.rs:580), the single
                // candidate's tile was patched to the drawn tile (agent_helper.rs:588-590)
                if (tid < 34 && first >= 0 && ((X.cand_req[first] >> tid) & 1)) out[(O_SP + 2 + deaka(last_tsumo)) * 34 + tid] = 1.f;
                if (tid == 0) out[(O_SP + 70) * 34 + deaka(last_tsumo)] = 1.f;
            } else {
                if (tid < 34 && first >= 0 && ((X.cand_req[first] >> tid) & 1)) out[(O_SP + 71) * 34 + tid] = 1.f;
            }
            // sp table (obs_repr.rs:644-692)
            const float ev_scale = max_ev < 1.f ? 0.f : 1.f / max_ev;
            bool table_ok = with_probs && first >= 0 && X.cand_tp0[first] > 0.f;
            if (table_ok) {
                if (can_discard0) {
                    for (int c = tid / SP_T; c < n_cand; c += SP_THREADS / SP_T) {
                        if (tid >= (SP_THREADS / SP_T) * SP_T) break;
                        const int turn = tid % SP_T;
                        if (turn >= T) continue;
                        const SpNode& nd = W->node[X.cand_slot[c]];
                        // take_while(p > 0) on the clamped tenpai probs
                        bool alive = true;
                        for (int q = 0; q <= turn && alive; q++) {
                            float tpq = cur_shanten == 0 ? 1.f : fminf(fmaxf(nd.tenpai[q], 0.f), 1.f);
                            alive = tpq > 0.f;
                        }
                        if (!alive) continue;
                        const int col = after_riichi ? deaka(last_tsumo) : deaka(X.cand_tile[c]);
                        float tpv = cur_shanten == 0 ? 1.f : fminf(fmaxf(nd.tenpai[turn], 0.f), 1.f);
                        float wpv = fminf(fmaxf(nd.win[turn], 0.f), 1.f);
                        float evv = fmaxf(nd.ev[turn], 0.f);
                        out[(O_SP + 72 + turn) * 34 + col] = tpv;
                        out[(O_SP + 72 + SP_T + turn) * 34 + col] = wpv;
                        out[(O_SP + 72 + 2 * SP_T + turn) * 34 + col] = fminf(evv * ev_scale, 1.f);
                    }
                } else {
                    const SpNode& nd = W->node[X.cand_slot[first]];
                    for (int w = tid; w < SP_T * 34; w += SP_THREADS) {
                        const int turn = w / 34, col = w % 34;
                        if (turn >= T) continue;
                        bool alive = true;
                        for (int q = 0; q <= turn && alive; q++) {
                            float tpq = cur_shanten == 0 ? 1.f : fminf(fmaxf(nd.tenpai[q], 0.f), 1.f);
                            alive = tpq > 0.f;
                        }
                        if (!alive) continue;
                        float tpv = cur_shanten == 0 ? 1.f : fminf(fmaxf(nd.tenpai[turn], 0.f), 1.f);
                        float wpv = fminf(fmaxf(nd.win[turn], 0.f), 1.f);
                        float evv = fmaxf(nd.ev[turn], 0.f);
                        out[(O_SP + 72 + turn) * 34 + col] = tpv;
                        out[(O_SP + 72 + SP_T + turn) * 34 + col] = wpv;
                        out[(O_SP + 72 + 2 * SP_T + turn) * 34 + col] = fminf(evv * ev_scale, 1.f);
                    }
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            long long t_5 = wall_clock64();
            atomicAdd(&P.err[1], 1ull);
            atomicAdd(&P.err[2], (unsigned long long)(t_1 - t_0));
            if (with_probs) {
                atomicAdd(&P.err[3], (unsigned long long)(t_2 - t_1));
                atomicAdd(&P.err[4], (unsigned long long)(t_3 - t_2));
                atomicAdd(&P.err[5], (unsigned long long)(t_4 - t_3));
                atomicAdd(&P.err[6], (unsigned long long)(t_5 - t_4));
                atomicAdd(&P.err[7], (unsigned long long)X.n_list);
            }
        }
        // ---- reset the hash set for the next row
        {
            const int n = min(X.n_list, SP_CAP);
            for (int i = tid; i < n; i += SP_THREADS) W->tag[W->list[i]] = 0ull;
            if (X.overflow && tid == 0) {
                atomicAdd(&P.err[0], 1ull);
                for (int i = 0; i < SP_CAP; i++) W->tag[i] = 0ull;
            }
        }
        __syncthreads();
    }
}
