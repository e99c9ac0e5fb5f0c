// Single-player tables of obs v4 (rows 889..1011) on device.
//
// Reference: PlayerState::single_player_tables (state/agent_helper.rs:509-593) -> SPCalculator::calc
// (algo/sp/calc.rs:84-133, production flags: no tegawari, no shanten-down, maximise EV, sorted) and the encoder block
// obs_repr.rs:564-692.  The reference is a memoised depth-first recursion (draw -> discard -> draw ...) over hand
// states with order-sensitive f32 sums.  On the GPU the same values are produced LEVEL-SYNCHRONOUSLY, one decision row
// per (persistent) workgroup at a time:
//   set-up   : candidates (the discards that keep the shanten number) and their required tiles from the table-id sets of
//              mj_sptab.h, three dependent chains side by side on three wavefronts (sp_row_front);
//   expand   : for shanten level L = s .. 1, the 3n+1 states of level L in chunks of 16, one chunk per WAVEFRONT
//              (sp_expand_chunk: thread-per-task passes over the 64 lanes, no workgroup barrier inside a level) find their
//              required draws t and the shanten-keeping discards d of h+t and insert the children h+t-d into a
//              per-workgroup hash set keyed by an EXACT 42-bit state id (see "state id" below; look, then atomicCAS on a
//              free slot; tags carry the row's epoch, nothing is cleared between rows);
//              the ordered child list of every state (slot, discard order key, draw count) goes to a pool in HBM;
//   evaluate : for L = 0 .. s: level 0 in three passes (probe / dense thread-per-item scoring / sum), then every level
//              by teams of T - depth lanes (one lane per remaining draw, sp_eval_wave0 / sp_eval_wave) that reproduce
//              draw_without_tegawari (calc.rs:447-561) with the reference's exact loop order (draw tiles ascending, aka
//              after its plain tile; i, j ascending) and fold discards like discard_slow (calc.rs:563-637);
//   write    : the whole SP block of the decision (123 rows; the encoder stops at row 890): sp_block_write.
// Work area per workgroup (SpWork): by hash slot only the tags and the 256-byte value nodes; keys, headers and level lists
// dense by list index (creation order), child lists / level-0 work items / their scores in append-only pools.
// Memoisation in the reference is a pure cache, so evaluating every reachable state exactly once gives bit-identical
// f32 results as long as each state's own accumulation order is kept — it is.  Compiled with -ffp-contract=off
// (Rust never fuses a*b+c); the only fused operations are the explicit fma steps of sp_div, which reproduce the IEEE
// division.  DESIGN.md §6 has the cost model and the optimisation history.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "mj_rules.h"
#include "mj_sptab.h"

// (c_sp_tab, the table-id shanten block set once by mj_tables_upload, is declared in mj_rules.h: the step kernel uses it too)

#ifndef SP_THREADS
#define SP_THREADS 256          // threads per workgroup = per decision row in flight.  Measured in round 3 (-DSP_THREADS=64: one row per
                                // wavefront, 16 single-wavefront workgroups per CU): the summed wavefront time drops by 14 % (nothing waits
                                // at workgroup barriers), but the heaviest rows (~6 k states) then take ~23 ms on their single wavefront
                                // and the launch lasts as long as they do: 39.5 vs 22.5 ms.  Four wavefronts per row it stays.
#endif
#define SP_CAP 16384           // hash slots per workgroup (max observed states per decision ~6k)
#define SP_T 17                // MAX_TSUMOS_LEFT (sp/mod.rs:40)
#define SP_MAX_CAND 14
#ifndef SP_NS
#define SP_NS 16               // states per expansion chunk (one chunk per wavefront)
#endif
#ifndef SP_SORT_MIN
#define SP_SORT_MIN 8            // levels up to this size are evaluated in list order (no cost sort)
#endif
#ifndef SP_ITEM_CAP
#define SP_ITEM_CAP 64         // draw items (state, required tile) per sub-batch of a chunk: one lane each
#endif
#ifndef SP_WPS
#define SP_WPS 4               // resident wavefronts per SIMD the kernel is compiled for (register budget 512 / SP_WPS per lane)
#endif
#define SP_WGS (4 * SP_WPS * 64 / SP_THREADS)  // resident workgroups per CU

#ifndef SP_CC_N
#define SP_CC_N 2048            // entries of the per-workgroup child cache in LDS (0 = off): row epoch << 56 | state id << 14 | slot.
                                // Measured (round 4, one box, mj_k_sp): none 20.01 ms, 512: 19.78, 1024: 19.73-19.77, 2048: 19.61 (16 KB: with the
                                // rest 40.8 KB per workgroup, the last size that keeps four workgroups on a CU)
#endif
#ifndef SP_TAIL_BATCH
#define SP_TAIL_BATCH 4         // rows per pop in the tail of the queue (rows without a state graph, one row per wavefront)
#endif
#define SP_Q_TAIL 32            // index of the tail's head word in SpParams::queue
// Small pools (round 6): a row whose state graph turns out large is PROMOTED by its 256-thread workgroup to a wide (SP_WIDE_THREADS)
// workgroup of mj_k_sp_wide, which runs beside mj_k_sp on another stream (see "promotion" at mj_k_sp).  Queue words, one 128-byte line each:
#define SP_Q_PROMO_DONE 96      // workgroups of mj_k_sp that have left their row loop (no promotion can follow)
#define SP_PROMO_LEVELS 2       // promotion queues: [0] the largest rows (taken first), [1] the rest
#define SP_Q_PROMO_ALLOC(p) (128 + 64 * (p))  // promotions so far = the next free entry / spare work area of queue p
#define SP_Q_PROMO_HEAD(p) (160 + 64 * (p))   // next entry of queue p a wide workgroup takes; + 1: the same for the sweep launch
#define SP_Q_PROMO_ENT 256      // entries [queue][SP_PROMO_CAP / 2]: work area + 1 (published after an agent-scope release), -1 = taken
#ifndef SP_PROMO_CAP
#define SP_PROMO_CAP 512        // promotions per launch (= spare work areas), half of them per queue
#endif
#define SP_Q_WORDS (SP_Q_PROMO_ENT + SP_PROMO_CAP)
#ifndef SP_WIDE_THREADS
#define SP_WIDE_THREADS 1024
#endif
#define SP_POOL (SP_CAP * 32)   // child-list pool entries per workgroup
#define SP_ITEMS (SP_CAP * 4)   // level-0 scoring items per workgroup
#define SP_L0_MAX 17            // winning draw entries per tenpai state (13 waits + 3 aka variants)
#define SP_EL_SLOT(v) ((v) & (SP_CAP - 1))  // an elist entry: hash slot | list index << 14
#define SP_EL_IDX(v) ((v) >> 14)
struct alignas(128) SpNode {    // one 3n+1 state's VALUES: 256 bytes = exactly two 128-byte lines, addressed by hash slot (round 5;
                                // rounds 1-4: 608 bytes with the state's key, header and level-0 scores inside, 16-byte value rows).
                                // A parent reads val[] of its children, ~6 parents per child: those reads are what the evaluation
                                // costs (a timing-only build without them: mj_k_sp 17.6 -> 15.8 ms), so a node holds nothing else
    float val[SP_T][3];         // per turn: tenpai prob, win prob, EV ((int)EV, the fold key of discard_slow, is recomputed by the reader)
    u8 pad_[52];
};
static_assert(sizeof(SpNode) == 256 && offsetof(SpNode, val) == 0, "SpNode layout");
struct alignas(16) SpHdr {      // a state's header, DENSE by list index: the level sort and the evaluation's state pipeline read a level's
                                // headers from one contiguous range instead of one node line each
    u32 child_off;              // level > 0: first pool entry of the child list; level 0: first work item (SpWork::items / l0sc)
    unsigned short n_ch;        // level > 0: number of pool entries; level 0: number of draw entries
    u8 sumreq, n_ent;           // sum over the required tiles of their wall counts (row of the not_tsumo table); draw entries
    u64 l0cnt2;                 // level 0: copies left in the wall of every draw entry, minus one, two bits each (its tsumo_prob row)
};
static_assert(sizeof(SpHdr) == 16 && offsetof(SpHdr, n_ch) == 4 && offsetof(SpHdr, sumreq) == 6 && offsetof(SpHdr, l0cnt2) == 8,
              "the evaluation reads the header as one or two u64");
struct SpKeys {                 // a state's key and exact id, DENSE by list index (the order of creation): the expansion / probe passes
                                // read the 16 states of a chunk as one contiguous block instead of 16 node lines + 16 tag lines
    u64 k0, k1, k2, k3;         // hand.mp | hand.sz + akas_in_hand<<48 | wall.mp | wall.sz + akas_in_wall<<48
    u64 dk;                     // state id (sp_dk_add)
};
// A child-list entry: hash slot of the child | discard order key << 14 | last-discard-of-its-draw-entry << 23 |
// draw count << 24 | invalid (hash set overflow) << 27.  Order: draw tile ascending, plain before red, discard ascending.
#define SP_ENT_SLOT(e) ((e) & 0x3FFFu)
#define SP_ENT_KEY(e) (((e) >> 14) & 511u)
#define SP_ENT_LAST (1u << 23)
#define SP_ENT_COUNT(e) (((e) >> 24) & 7u)
#define SP_ENT_INVALID (1u << 27)
struct alignas(16) SpF4 { float x, y, z, w; };
#define SP_HANDOFF_WORDS 1024   // >= sizeof(SpHandoff) / 4 (static_assert below)
struct alignas(128) SpWork {   // per-workgroup scratch in HBM (persistent workgroups), 8.1 MB
    u64 tag[SP_CAP];           // state id | row epoch << 42 | 1 << 63; a tag of another epoch (or 0) is an EMPTY slot
    SpNode node[SP_CAP];       // by hash slot
    SpKeys keys[SP_CAP];       // by list index
    SpHdr hdr[SP_CAP];         // by list index
    u32 list[SP_CAP];          // slots grouped by level: level L occupies [lvl_begin[L], lvl_end[L])
    u32 elist[SP_CAP];         // the same ranges ordered by child-list length for the evaluation (sp_sort_level): slot | list index << 14
    u32 pool[SP_POOL];         // child lists
    u32 items[SP_ITEMS];       // level 0: (list index, winning tile, variant) work items of the dense scoring pass
    SpF4 l0sc[SP_ITEMS];       // level 0: get_score() of every work item (sp_l0_score), all zero = no yaku
    u32 epoch;                 // the tag epoch of the last row with a state graph this workgroup processed (persists across launches)
    u32 pad_[31];
    alignas(16) u32 handoff[SP_HANDOFF_WORDS];  // a promoted row's context (SpHandoff: SpCtx + SpRowInfo + row + next level), written by its first workgroup
#ifdef MJ_EMU
    u32 idx_of[SP_CAP];        // emulator only: slot -> list index, for the id <-> key bijection check on every hit
#endif
};
static_assert(offsetof(SpWork, node) % 128 == 0 && sizeof(SpWork) % 128 == 0, "nodes on line boundaries");

// build_not_tsumo_prob_table (calc.rs:148-167) for EVERY wall size, built once on the host and shared by all workgroups:
// row [n_left][q] = P(no tile out of q useful ones on turns 0 .. j-1 | n_left tiles unseen), the running product
// cur * (n_left - q - j) / (n_left - j) in f32 exactly as the reference rounds it (x86 mulss / divss == the device's IEEE
// multiply / divide).  The reference's per-decision table only adds a cut-off at its number of draws T, and nobody reads a
// row past turn T - 1.  Round 3 built the 124 rows per decision into the workgroup's HBM area: 11.6 us per row of the queue.
#define SP_NT_ROWS 124           // MAX_TILES_LEFT + 1 = 123 rows (calc.rs:14) + one all-zero row for sums past that
#define SP_NT_STRIDE (SP_T + 3)  // 80-byte rows
__constant__ const float* c_sp_nt;  // [SP_NT_ROWS wall sizes][SP_NT_ROWS][SP_NT_STRIDE]
static inline void sp_not_tsumo_build(float* out /* [SP_NT_ROWS * SP_NT_ROWS * SP_NT_STRIDE] */) {
    for (int n_left = 0; n_left < SP_NT_ROWS; n_left++)
        for (int q = 0; q < SP_NT_ROWS; q++) {
            float* r = out + ((size_t)n_left * SP_NT_ROWS + q) * SP_NT_STRIDE;
            const bool row_on = q <= 122 && q < n_left + 1;
            const int lim = n_left - q;
            float cur = row_on ? 1.f : 0.f;
            for (int j = 0; j < SP_NT_STRIDE; j++) r[j] = 0.f;
            r[0] = cur;
            for (int j = 0; j < SP_T - 1; j++) {
                cur = (row_on && j < lim) ? cur * (float)(n_left - q - j) / (float)(n_left - j) : 0.f;
                r[j + 1] = cur;
            }
        }
}

struct SpParams {
    const TableOne* snap;
    const uint32_t* rows;
    int n_rows;
    MjTablesDev tables;
    float* obs;                // [n_rows][1012][34]; rows 0 .. 889 written by mj_k_encode<4>, rows 889 .. 1011 written here (sp_block_write)
    SpWork* work;              // [gridDim.x]
    int* queue;                // dynamic row queue (zeroed before launch); [0] head of the rows with a state graph, [1..8] class counts,
                               // [9..16] class cursors of the row sort, [SP_Q_TAIL] head of the queue's tail (rows without a graph)
    const uint32_t* order;     // [n_rows] queue position -> row index, heaviest cost class first (mj_k_order_classify / _scatter)
    unsigned long long* prof;  // NULL or [24] phase timers / counters (MJ_SP_PROF; mj_counters prints them)
    unsigned long long* err;   // [0] hash-capacity overflows, [1] rows; cycle sums: [2] setup [3] expand [4] eval L0 [5] eval L>0 [6] encode; [7] states
    uint32_t* rowdump;         // (builds with -DSP_ROWDUMP only: tools/build_variant.sh dump -DSP_ROWDUMP) NULL, or [n_rows][12] per-row records of the rows with a state graph (MJ_SP_ROWDUMP: cost-model data for the row order)
    // promotion of large rows to mj_k_sp_wide (small pools; see "promotion" at mj_k_sp)
    int promo_cap;             // spare work areas = promotions allowed in this launch (0: off); work[] holds grid + promo_cap areas
    int promo_min[4];          // [level]: park the row when the level about to be expanded has at least this many states
    int n_narrow;              // mj_k_sp_wide: workgroups of mj_k_sp in this launch (the DONE word's final value)
    int sweep;                 // mj_k_sp_wide: 1 = the sweep launch behind both kernels (never waits)
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

struct alignas(16) SpCtx {  // per-decision constants (LDS)
    Melds melds;
    int len_div3, bakaze, jikaze, is_menzen, num_doras_in_fuuro, n_dora, calc_double_riichi, calc_haitei,
        prefer_riichi, T, n_left;
    int dora_ind[5];
    float tsumo_prob[4][SP_T];
    // level bookkeeping
    int lvl_begin[5], lvl_end[5];
    int n_list;
    int n_pool;
    int n_items;
    int overflow;
    unsigned long long* prof;  // optional phase timers (MJ_SP_PROF)
    unsigned long long pt[8];  // per-row sums of the expansion pass timers / counters (flushed once per row)
    unsigned long long* cc;    // the workgroup's child cache in LDS (SP_CC_N entries; NULL: none) and this row's epoch (8 bits, never 0... see mj_k_sp)
    unsigned cc_epoch;
    unsigned tag_epoch;        // this row's epoch in the hash tags (sp_tag_free)
    // candidates
    int n_cand;
    int cand_tile[SP_MAX_CAND], cand_slot[SP_MAX_CAND], cand_down[SP_MAX_CAND], cand_nreq[SP_MAX_CAND];
    u64 cand_req[SP_MAX_CAND];
    u64 spec_req[2][34];  // row set-up: required draws of root - d for every held kind d, [1] = one shanten number up (4+ shanten roots);
                          // row output: the 34-bit masks of block rows 2 .. 69 (sp_block_write)
    u64 row7x[2];         // row output: masks of block rows 70 (the candidate with the most required tiles) and 71 (no discard)
    signed char colcand[36];  // row output: tile column -> candidate whose table it shows (-1: none)
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

// The out-of-line phase functions receive their pointers in VGPRs (the calling convention), and an array element addressed through
// such a pointer costs a 64-bit multiply-add plus 64-bit adds per access (v_mad_u64_u32, v_lshl_add_u64, v_add_co / v_addc: ~5 VALU,
// 12 % of sp_eval_wave's VALU instructions in round 4).  sp_uniform() tells the compiler the pointer is the same in every lane (SGPR
// pair); sp_ld / sp_st address the work area as base + a 32-bit BYTE offset (the area is 12.5 MB), which selects the
// global_load vdst, voffset, s[base:base+1] form: one v_mad_u32_u24 per element address.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MJ_EMU)
template <class Tp> __device__ __forceinline__ Tp* sp_uniform(Tp* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (Tp*)(((unsigned long long)hi << 32) | lo);
}
#else
template <class Tp> MJD Tp* sp_uniform(Tp* p) { return p; }  // (host pass of the single-source compile, emulator)
#endif
template <class Tv, class Bp> __device__ __forceinline__ Tv sp_ld(Bp base, u32 byte_off) {  // scalar types
#ifdef MJ_EMU
    return *reinterpret_cast<const Tv*>(reinterpret_cast<const char*>(base) + byte_off);
#else
    return *reinterpret_cast<const SP_HBM Tv*>(reinterpret_cast<const SP_HBM char*>(base) + (unsigned long long)byte_off);
#endif
}
template <class Bp> __device__ __forceinline__ SpF4 sp_ld4(Bp base, u32 byte_off) {  // one 16-byte load (member-wise: SP_HBM is an address space)
#ifdef MJ_EMU
    return *reinterpret_cast<const SpF4*>(reinterpret_cast<const char*>(base) + byte_off);
#else
    const SP_HBM SpF4* p = reinterpret_cast<const SP_HBM SpF4*>(reinterpret_cast<const SP_HBM char*>(base) + (unsigned long long)byte_off);
    SpF4 r;
    r.x = p->x; r.y = p->y; r.z = p->z; r.w = p->w;
    return r;
#endif
}
struct SpF3 { float x, y, z; };
template <class Bp> __device__ __forceinline__ SpF3 sp_ld3(Bp base, u32 byte_off) {  // one 12-byte load
#ifdef MJ_EMU
    return *reinterpret_cast<const SpF3*>(reinterpret_cast<const char*>(base) + byte_off);
#else
    const SP_HBM SpF3* p = reinterpret_cast<const SP_HBM SpF3*>(reinterpret_cast<const SP_HBM char*>(base) + (unsigned long long)byte_off);
    SpF3 r;
    r.x = p->x; r.y = p->y; r.z = p->z;
    return r;
#endif
}
template <class Bp> __device__ __forceinline__ void sp_st3(Bp base, u32 byte_off, float x, float y, float z) {
#ifdef MJ_EMU
    *reinterpret_cast<SpF3*>(reinterpret_cast<char*>(base) + byte_off) = SpF3{x, y, z};
#else
    SP_HBM SpF3* p = reinterpret_cast<SP_HBM SpF3*>(reinterpret_cast<SP_HBM char*>(base) + (unsigned long long)byte_off);
    p->x = x; p->y = y; p->z = z;
#endif
}
// `ev as i32` (calc.rs:600-615, the fold key of discard_slow): saturating, NaN -> 0 = v_cvt_i32_f32.  EVs are finite and below 2^19.
MJD int sp_f2i(float v) {
#if defined(MJ_EMU)
    return v != v ? 0 : v >= 2147483648.f ? 2147483647 : v <= -2147483648.f ? (-2147483647 - 1) : (int)v;
#else
    return (int)v;
#endif
}

// ---- state id.  Every state of a row is the row's root hand/wall after some draws (wall -> hand, at most 3: one per
// shanten level) and some discards (hand -> out, at most 4: the candidate's discard + one per level).  The wall of a state
// is the root wall minus the multiset of drawn tiles, and its hand is the root hand plus that multiset minus the multiset
// of discarded tiles (tiles as 37 kinds: a red five is its own kind, matching akas_in_hand / akas_in_wall) — so
// (hand, wall)  <->  (draw multiset, discard multiset) is a bijection, and the two sorted multisets, 6 bits per tile
// (0 = none), are an EXACT 42-bit identifier: two different states of a row never share one, equal states always do.
// The hash set compares this id, not a hash of the 256-bit key (round 1 accepted a 63-bit hash match).
MJD u64 sp_dk_add(u64 dk, int draw_tile, int discard_tile) {  // -1 = none
    u32 d0 = (u32)dk & 63, d1 = (u32)(dk >> 6) & 63, d2 = (u32)(dk >> 12) & 63;
    u32 x0 = (u32)(dk >> 18) & 63, x1 = (u32)(dk >> 24) & 63, x2 = (u32)(dk >> 30) & 63, x3 = (u32)(dk >> 36) & 63;
    if (draw_tile >= 0) {  // sorted (descending) insertion; the lowest field is free by construction
        u32 x = (u32)draw_tile + 1, t;
        t = max(d0, x); x = min(d0, x); d0 = t;
        t = max(d1, x); x = min(d1, x); d1 = t;
        d2 = max(d2, x);
    }
    if (discard_tile >= 0) {
        u32 x = (u32)discard_tile + 1, t;
        t = max(x0, x); x = min(x0, x); x0 = t;
        t = max(x1, x); x = min(x1, x); x1 = t;
        t = max(x2, x); x = min(x2, x); x2 = t;
        x3 = max(x3, x);
    }
    return (u64)(d0 | (d1 << 6) | (d2 << 12) | (x0 << 18) | (x1 << 24)) | ((u64)(x2 | (x3 << 6)) << 30);
}
MJD u32 sp_dk_pos(u64 dk) {  // first probe position
    u32 h = (u32)dk * 0x9E3779B1u ^ ((u32)(dk >> 32) + 0x7F4A7C15u) * 0x85EBCA77u;
    h ^= h >> 15;
    return (h * 0x2C1B3C6Du) >> 18;  // top 14 bits: SP_CAP slots
}
static_assert(SP_CAP == 1 << 14, "sp_dk_pos returns 14 bits");
// Tags carry the EPOCH of the row that set them (21 bits between the 42-bit id and the valid bit): a slot whose tag belongs to another
// row is empty, so nothing has to be cleared between rows (rounds 1-4: one 8-byte store per state, 0.7 GB per launch as 32-byte
// sectors, after re-reading the list).  The epoch counts this workgroup's graph rows (SpWork::epoch, kept across launches); when
// it wraps, once in 2 M rows, the table is wiped.
#define SP_EPOCH_MAX 0x1FFFFFu
#ifndef SP_EPOCH_WRAP
#define SP_EPOCH_WRAP SP_EPOCH_MAX  // the epoch after which the table is wiped (tests build with a small one: the wrap is 2 M rows away otherwise)
#endif
#define SP_TAG(dk, ep) ((dk) | ((u64)(ep) << 42) | (1ull << 63))
MJD bool sp_tag_free(u64 t, u32 ep) { return (u32)((t >> 42) & SP_EPOCH_MAX) != ep; }  // (0: epoch 0, never a row's epoch)

// ---- register pressure across the persistent row loop (round 4; every step measured inside one gpurun call)
// mj_k_sp keeps 128 VGPRs / ~100 SGPRs for a loop body of 19 k instructions.  Two things sent registers through scratch memory,
// and scratch traffic of 1,024 workgroups does not stay in L2 (4 MB per XCD):
//  * a __noinline__ callee saves and restores its callee-saved VGPRs on every call: 37 stores + 37 loads per expansion chunk =
//    19 KB per 16 states.  The expansion is therefore inlined into the kernel (one call site); the evaluation / level-0 functions
//    stay calls (a call there is a wavefront's whole share of a level; inlining them was measured slower: their loops then spill).
//  * lane-derived constants (lane -> suit / state index, LDS addresses) are loop-invariant, so the compiler computed them at the
//    top of the kernel and spilled them (50 stores up front, reloads inside the probe passes).  An opaque copy of the lane id per
//    row / per chunk stops the hoisting: the constants are recomputed (1-2 VALU) where they are used.
// Measured: 20.93 -> 19.76 ms (inline) -> 18.45 ms (+ opaque lane id) on one box; 19.99 -> 18.17 ms on another.
#ifndef SP_OPAQUE_TID
#define SP_OPAQUE_TID 2  // 0: off, 1: per expansion chunk, 2: + per row of the queue
#endif
#if defined(MJ_EMU)
#define SP_OPQ(level, v) (v)
#else
__device__ __forceinline__ int sp_opaque_v(int v) { asm volatile("" : "+v"(v)); return v; }
#define SP_OPQ(level, v) ((SP_OPAQUE_TID) >= (level) ? sp_opaque_v(v) : (v))
#endif
// ... and of a uniform pointer (the table directory in constant memory, the work area).  Measured neutral to slightly slower
// (18.82 vs 18.45 ms): off by default.
#if defined(MJ_EMU) || !defined(SP_OPAQUE_PTR)
#define SP_OPQ_PTR(level, p) (p)
#else
template <typename Tp> __device__ __forceinline__ Tp* sp_opaque_s(Tp* p) { asm volatile("" : "+s"(p)); return p; }
#define SP_OPQ_PTR(level, p) ((SP_OPAQUE_PTR) >= (level) ? sp_opaque_s(p) : (p))
#endif
#ifndef SP_ATTR_EXPAND
#define SP_ATTR_EXPAND __forceinline__
#endif
#ifndef SP_ATTR_L0P
#define SP_ATTR_L0P __noinline__
#endif
#ifndef SP_ATTR_L0S
#define SP_ATTR_L0S __noinline__
#endif
#ifndef SP_ATTR_EVAL
#define SP_ATTR_EVAL __noinline__
#endif
#ifndef SP_ATTR_EVAL0
#define SP_ATTR_EVAL0 __noinline__
#endif

template <class TagP>
MJD u64 sp_claim_tag(TagP tagp, u64 h, u64 seen) {  // claim a slot seen empty (holding `seen`): atomicCAS(tag, seen, h) (relaxed, agent scope)
    u64 expected = seen;                            // -> 0 = claimed, else the tag found there (of this row: only this workgroup writes its table)
    __hip_atomic_compare_exchange_strong(tagp, &expected, h, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return expected == seen ? 0ull : expected;
}
// hash-set insert of the state `base` + draw `tile` - discard `dt` (either may be -1) with id `dk`: returns the slot or -1 on
// overflow.  Only a call that creates the slot builds the state itself and writes its key record (sp_new_state; an edge finds its
// child already present five times in six).
MJD SpState sp_apply(SpState s, int tile, int dt) {
    if (tile >= 0) sp_deal(s, tile);
    if (dt >= 0) sp_discard(s, dt);
    return s;
}
// A fresh state gets the next list index; its key and id go to the dense array there.
template <class WP>
__device__ __forceinline__ void sp_new_state(WP W, SpCtx* X, u32 slot, u64 dk, const SpState& st) {
    const int idx = atomicAdd(&X->n_list, 1);
    if (idx < SP_CAP) {
        u64 k[4];
        sp_key(st, k);
        W->list[idx] = slot;
        auto& e = W->keys[idx];
        e.k0 = k[0]; e.k1 = k[1]; e.k2 = k[2]; e.k3 = k[3]; e.dk = dk;
#ifdef MJ_EMU
        W->idx_of[slot] = (u32)idx;
#endif
    } else {
        X->overflow = 1;
    }
}
#ifdef MJ_EMU  // the emulator never pre-empts between the claim and the key write: check the bijection on every hit
template <class WP>
inline void sp_emu_check_hit(WP W, SpCtx* X, u32 slot, const SpState& st) {
    u64 k[4];
    sp_key(st, k);
    const auto& e = W->keys[W->idx_of[slot]];
    if (e.k0 != k[0] || e.k1 != k[1] || e.k2 != k[2] || e.k3 != k[3]) X->overflow = 1;
}
#endif
template <class WP>
__device__ __forceinline__ int sp_insert(WP W, SpCtx* X, u64 dk, const SpState& base, int tile, int dt) {  // (the row's root states)
    const u32 ep = X->tag_epoch;
    const u64 tag = SP_TAG(dk, ep);
    u32 pos = sp_dk_pos(dk);
    for (int probe = 0; probe < SP_CAP; probe++) {
        u64 old = W->tag[pos];
        if (sp_tag_free(old, ep)) old = sp_claim_tag(&W->tag[pos], tag, old);
        if (old == 0ull) {
            sp_new_state(W, X, pos, dk, sp_apply(base, tile, dt));
            return (int)pos;
        }
        if (old == tag) {
#ifdef MJ_EMU
            sp_emu_check_hit(W, X, pos, sp_apply(base, tile, dt));
#endif
            return (int)pos;
        }
        pos = (pos + 1) & (SP_CAP - 1);
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
    {   // member-wise (a struct copy is a memcpy, which keeps the generic address space of X: flat loads, both wait counters; with one
        // kernel the compiler knew X's LDS address, with three kernels sharing this function it does not)
        const Melds& xm = X->melds;
        in.m.chis = xm.chis; in.m.pons = xm.pons; in.m.minkans = xm.minkans; in.m.ankans = xm.ankans;
        in.m.n_chis = xm.n_chis; in.m.n_pons = xm.n_pons; in.m.n_minkans = xm.n_minkans; in.m.n_ankans = xm.n_ankans;
    }
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
        // indicator tiles left in the wall per number of copies in the hand (scalars: a dynamically indexed local array
        // would live in scratch), over the hand's tile kinds only
        int n1 = 0, n2 = 0, n3 = 0, n4 = 0;
        for (u64 m = s.h.nonzero_mask(); m; m &= m - 1) {
            const int t = __ffsll((long long)m) - 1, c = s.h.get(t), ic = s.w.get(tile_prev(t));
            n1 += c == 1 ? ic : 0;
            n2 += c == 2 ? ic : 0;
            n3 += c == 3 ? ic : 0;
            n4 += c == 4 ? ic : 0;
        }
        n1 &= 0xFF; n2 &= 0xFF; n3 &= 0xFF; n4 &= 0xFF;
        const int sum_ind = (n1 + n2 + n3 + n4) & 0xFF;
        const int n_left = s.w.total() & 0xFF;
        float up[5];
        up[0] = (float)((n_left - sum_ind) & 0xFF) / (float)n_left;
        up[1] = (float)n1 / (float)n_left;
        up[2] = (float)n2 / (float)n_left;
        up[3] = (float)n3 / (float)n_left;
        up[4] = (float)n4 / (float)n_left;
        float pt[8];  // the points of han + k, k = i + j: 8 values instead of 20 evaluations (static indices: registers)
#pragma unroll
        for (int k = 0; k < 8; k++) pt[k] = (float)tsumo_total(point_calc(is_oya, fu, (han + k) & 0xFF), is_oya);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float sc = 0.f;
#pragma unroll
            for (int j = 0; j < 5; j++) {
                if (up[j] == 0.f) continue;
                sc += pt[i + j] * up[j];
            }
            scores[i] = sc;
        }
    } else if (assume_riichi && X->n_dora > 1) {
        float pt[16];
#pragma unroll
        for (int k = 0; k < 16; k++) pt[k] = (float)tsumo_total(point_calc(is_oya, fu, (han + k) & 0xFF), is_oya);
        const float* ur = SP_URADORA[X->n_dora - 1];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float sc = 0.f;
#pragma unroll
            for (int j = 0; j < 13; j++) {
                const float p = ur[j];
                if (p == 0.f) continue;
                sc += pt[i + j] * p;
            }
            scores[i] = sc;
        }
    } else {
        for (int i = 0; i < 4; i++) scores[i] = (float)tsumo_total(point_calc(is_oya, fu, (han + i) & 0xFF), is_oya);
    }
    return true;
}


// ---- Level 0 (tenpai states) is evaluated in three passes so that the expensive, divergent scoring of the winning
// draws (get_score: agari decomposition + yaku + fu) runs with every lane busy:
//   probe : sp_l0_probe_chunk — which draws win (34 shanten probes per state) -> draw entries, one work item per entry
//   score : THREAD per item, dense across the workgroup                  -> 4 scores per work item (SpWork::l0sc)
//   sum   : team per state — sp_eval_wave0 accumulates the scores in the reference's order
__device__ __forceinline__ void sp_l0_score(const MjTablesDev& Tb, SpWork* W, const SpCtx* X, u32 item, int item_idx) {
    SP_ASSUME_LDS(X);
    const int li = item & 0x3FFF, t = (item >> 19) & 63, variant = (item >> 25) & 1;  // list index of the state, winning tile
    SP_HBM SpWork* const Wg = (SP_HBM SpWork*)W;
    SpState S1 = sp_state_of(Wg->keys[li]);
    const int tile = variant ? akaize(t) : t;
    sp_deal(S1, tile);
    float scv[4];
    const bool yaku = sp_get_score(Tb, X, S1, tile, scv);  // a hand with a yaku scores > 0: all-zero scores mark "no yaku" for the summation
    SP_HBM SpF4& dst = Wg->l0sc[item_idx];
    dst.x = yaku ? scv[0] : 0.f; dst.y = yaku ? scv[1] : 0.f; dst.z = yaku ? scv[2] : 0.f; dst.w = yaku ? scv[3] : 0.f;
}

// The dense scoring pass of one wavefront: its share of the row's work items in ONE call (a call per item and lane saved and restored the
// function's 18 callee-saved VGPRs every 64 items).
__device__ SP_ATTR_L0S void sp_l0_score_all(const MjTablesDev& Tb, SpWork* W, const SpCtx* X, int n_items, int tid, int stride) {
    SP_ASSUME_LDS(X);
    const SP_HBM u32* const items = ((SP_HBM SpWork*)W)->items;
    for (int i = tid; i < n_items; i += stride) sp_l0_score(Tb, W, X, items[i], i);
}

template <int J, int N, class F>
MJD void sp_static_for(F&& f) {  // f(integral_constant<J>) ... f(integral_constant<N - 1>): the index is a compile-time constant
    if constexpr (J < N) {
        f(std::integral_constant<int, J>{});
        sp_static_for<J + 1, N>(f);
    }
}

// Inclusive prefix sum over the 64 lanes of a wavefront (every lane active).  On the device: six DPP steps inside the VALU
// (row_shr 1 / 2 / 4 / 8 within the rows of 16, then row_bcast 15 / 31 across them) instead of six LDS-crossbar shuffles.
MJD u32 sp_wave_scan_incl(u32 v) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MJ_EMU)
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);  // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);  // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);  // row_shr:4
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);  // row_shr:8
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
    return v;
#else
    const int lane = (int)(threadIdx.x & 63);
    for (int d = 1; d < 64; d <<= 1) {
        const u32 q = __shfl_up(v, d);
        if (lane >= d) v += q;
    }
    return v;
#endif
}

// The value lane `lane` holds (wave-uniform lane number): v_readlane on the device, no LDS crossbar.
MJD int sp_lane_get(int v, int lane) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MJ_EMU)
    return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(lane));
#else
    return __shfl(v, lane);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Dense expansion / level-0 probe: SP_NS states of one level at a time PER WAVEFRONT, every phase a THREAD-PER-TASK pass
// over the 64 lanes, separated by wave-level LDS hand-offs (mj_team_sync<64>): no workgroup barrier inside a level, four
// chunks in flight per workgroup.  Shanten numbers are never computed here: the required draws of a state and, after each
// of them, the shanten-keeping discards come from the table-id formulation of mj_sptab.h —
//   P0  (state, suit)  : suit key -> row id; seven-pairs / orphans counters
//   P1  (state, suit)  : id of the merge of the other three suits (two byte gathers), optimal-entry record, the key's wait /
//                        keep records -> the suit's part of "draws that lower the normal-form number"; pair merges for P2
//   P1c (state)        : required draws = closed form over normal form / seven pairs / orphans (sp_req_set), wall applied
//   P2  (state, draw t): g = h + t differs from h in ONE suit: its new row id, then per suit the optimal entries against the
//                        other three (table walks) and the keep masks -> shanten-keeping discards of g (sp_keep_set)
//   P3  (state)        : child-list layout (reference order: draw tile ascending, plain before red, discard ascending)
//   P4  (edge)         : hash-set insert of h + t - d, child-list entry
// A state with a suit key whose neighbour lies past the end of the reference's table (SPT_FALLBACK, 8 one-suit 13-tile
// patterns) takes the brute-force loops of the reference instead (sp_*_brute).
#ifndef SP_EINFO_CAP
#define SP_EINFO_CAP 192  // child entries per sub-batch whose (item, discard) the layout pass leaves in LDS (the rest: binary search, bit loop)
#endif
struct SpChunk {
    u64 k[SP_NS][4];        // state keys (hand.mp, hand.sz | akas, wall.mp, wall.sz | akas)
    u64 dk[SP_NS];          // state ids
    u64 req[SP_NS];         // required draws
    SpRec keep[SP_NS][4];   // keep records of the four suit keys
    u32 key[SP_NS][4];      // base-5 suit keys
    u32 slot[SP_NS];
    unsigned short wn[SP_NS][4];  // per suit: tiles that lower the normal-form number (9 bits) | fallback << 15
    u8 id[SP_NS][4];        // row ids
    u8 r2[SP_NS][8];        // ids of the pair merges 01 02 03 12 13 23
    u8 r3[SP_NS][4];        // per suit: id of the merge of the three OTHER suits
    u8 cnt[SP_NS][4];       // pairs, kinds, yaokyuu pairs, yaokyuu kinds (shanten.rs:104-137)
    u8 fin[SP_NS];          // normal-form final value (shanten + 1)
    u8 fb[SP_NS];           // brute-force path
    alignas(16) u8 n_tiles[SP_NS];  // required draws (read as one 16-byte word)
    u32 gcs[SP_NS][4];      // per suit group of the hand: tiles held once | twice << 9 | at all << 18 (sp_group_count_sets)
    unsigned short wnz[SP_NS][4];  // per suit group: tiles left in the wall
    u64 cs[SP_NS][3];       // the hand's tiles held once / twice / at all (34-bit sets)
    int child_base[SP_NS];
    // the draw items (state, required tile) of the current sub-batch of states: one lane each
    u64 kept[SP_ITEM_CAP];                  // shanten-keeping discards after the draw
    unsigned short item[SP_ITEM_CAP];       // state | tile << SP_SB
    unsigned short coff[SP_ITEM_CAP];       // offset of the item's first child inside the state's child list
    unsigned short eoff[SP_ITEM_CAP + 2];   // prefix sums of the child entries over the items
    u32 einfo[SP_EINFO_CAP];                // the sub-batch's first child entries, decoded by their item lane in the layout pass:
                                            // item | kept discard << 6 | draw variant << 12 | last discard of the draw entry << 13
};
#define SP_NT 64  // co-operating threads of a chunk
#define SP_SB (SP_NS > 16 ? 5 : 4)  // bits of the state index inside an item
static_assert(SP_NS == 16 || SP_NS == 32, "n_tiles is read in 16-byte words; item entries hold the state in SP_SB bits");
static_assert(SP_ITEM_CAP == SP_NT, "one lane per draw item: the layout pass is a wavefront scan");

MJD SpState sp_chunk_state(const SpChunk* C, int s) {
    SpState S;
    S.h.mp = C->k[s][0];
    S.h.sz = C->k[s][1] & 0xFFFFFFFFFFFFull;
    S.w.mp = C->k[s][2];
    S.w.sz = C->k[s][3] & 0xFFFFFFFFFFFFull;
    S.akas = (u32)((C->k[s][1] >> 48) & 7) | ((u32)((C->k[s][3] >> 48) & 7) << 3);
    return S;
}
MJD bool sp_aka_in_wall(const SpState& S, int t) {
    return (t == T_5M && (S.akas & 8)) || (t == T_5P && (S.akas & 16)) || (t == T_5S && (S.akas & 32));
}
__device__ __noinline__ u64 sp_req_brute_dev(Hand h, int ld3, int L) { return sp_req_brute(c_mj_tables, h, ld3, L); }
__device__ __noinline__ u64 sp_keep_brute_dev(Hand g, int ld3, int Tg) { return sp_keep_brute(c_mj_tables, g, ld3, Tg); }

// Passes P0-P1c, shared by the expansion (L >= 1) and the level-0 probe (L == 0): state keys, row ids, optimal entries ->
// req[s] = the draws left in the wall that lower the shanten number of state s.
__device__ __forceinline__ void sp_chunk_probe(SP_HBM SpWork* Wg, SpCtx* X, SpChunk* C, const SpTabG& TG, int first, int n, int L) {
    const int tid = SP_OPQ(1, (int)(threadIdx.x & (SP_NT - 1)));
    const int ld3 = X->len_div3;
    for (int task = tid; task < n * 4; task += SP_NT) {  // the chunk's states are neighbours in the dense key array: one contiguous read
        const int s = task >> 2, j = task & 3;
        C->k[s][j] = reinterpret_cast<const SP_HBM u64*>(&Wg->keys[first + s])[j];  // k0..k3 lead the record
        if (j == 0) {
            C->slot[s] = Wg->list[first + s];
            C->dk[s] = Wg->keys[first + s].dk;
        }
    }
    mj_team_sync<SP_NT>();
    for (int task = tid; task < n * 4; task += SP_NT) {
        const int s = task >> 2, i = task & 3;
        const SpState S = sp_chunk_state(C, s);
        const u32 key = i == 0 ? suit_key9(S.h.mp) : i == 1 ? suit_key9(S.h.mp >> 27) : i == 2 ? suit_key9(S.h.sz) : suit_key7(S.h.sz >> 27);
        C->key[s][i] = key;
        C->id[s][i] = (u8)spt_id(TG, i, key);
        C->cnt[s][i] = (u8)(i == 0 ? S.h.n_pairs() : i == 1 ? S.h.n_kinds() : i == 2 ? S.h.n_yao_pairs() : S.h.n_yao_kinds());
        C->gcs[s][i] = sp_group_count_sets(sp_group_fields(S.h, i));
        C->wnz[s][i] = (unsigned short)((sp_group_count_sets(sp_group_fields(S.w, i)) >> 18) & 0x1FF);
    }
    mj_team_sync<SP_NT>();
    for (int task = tid; task < n * 4; task += SP_NT) {
        const int s = task >> 2, i = task & 3;
        const u32 i0 = C->id[s][0], i1 = C->id[s][1], i2 = C->id[s][2], i3 = C->id[s][3];
        // first pair of the other three suits: (1,2) (0,2) (0,1) (0,1); third: 3 3 3 2
        const u32 pa = i == 0 ? i1 : i0, pb = i <= 1 ? i2 : i1, pc = i == 3 ? i2 : i3;
        const u32 r2 = spt_merge(TG, pa, pb);
        C->r2[s][i == 0 ? 3 : i == 1 ? 1 : 0] = (u8)r2;
        if (i < 3) C->r2[s][i == 0 ? 2 : i == 1 ? 4 : 5] = (u8)spt_merge(TG, i == 0 ? i0 : i == 1 ? i1 : i2, i3);  // (i, 3)
        const u32 r3 = spt_merge(TG, r2, pc);
        C->r3[s][i] = (u8)r3;
        const u32 key = C->key[s][i], myid = i == 0 ? i0 : i == 1 ? i1 : i == 2 ? i2 : i3;
        const bool inside = spt_in_table(TG, i, key);
        const SpRec o = spt_opt(TG, ld3, r3, myid);
        const SpRec w = spt_rec(TG, i, inside ? key : 0u, 0), kp = spt_rec(TG, i, inside ? key : 0u, 1);
        C->wn[s][i] = (unsigned short)(spt_wait_tiles(w, o) | ((!inside || (w.w & SPT_FALLBACK)) ? 0x8000u : 0u));
        C->keep[s][i] = kp;
        if (i == 3) C->fin[s] = (u8)spt_fin(o);
    }
    mj_team_sync<SP_NT>();
    if (tid < n) {
        const int s = tid;
        const SpState S = sp_chunk_state(C, s);
        const u32 w0 = C->wn[s][0], w1 = C->wn[s][1], w2 = C->wn[s][2], w3 = C->wn[s][3];
        const bool fb = ((w0 | w1 | w2 | w3) & 0x8000u) != 0;
        const u64 waitN = (u64)(w0 & 0x1FF) | ((u64)(w1 & 0x1FF) << 9) | ((u64)(w2 & 0x1FF) << 18) | ((u64)(w3 & 0x1FF) << 27);
        const u32 g0 = C->gcs[s][0], g1 = C->gcs[s][1], g2 = C->gcs[s][2], g3 = C->gcs[s][3];
        SpCountSets cs;
        cs.c1 = (u64)(g0 & 0x1FF) | ((u64)(g1 & 0x1FF) << 9) | ((u64)(g2 & 0x1FF) << 18) | ((u64)(g3 & 0x1FF) << 27);
        cs.c2 = (u64)((g0 >> 9) & 0x1FF) | ((u64)((g1 >> 9) & 0x1FF) << 9) | ((u64)((g2 >> 9) & 0x1FF) << 18) | ((u64)((g3 >> 9) & 0x1FF) << 27);
        cs.nz = (u64)(g0 >> 18) | ((u64)(g1 >> 18) << 9) | ((u64)(g2 >> 18) << 18) | ((u64)(g3 >> 18) << 27);
        C->cs[s][0] = cs.c1;
        C->cs[s][1] = cs.c2;
        C->cs[s][2] = cs.nz;
        u64 req;
        if (fb) req = sp_req_brute_dev(S.h, ld3, L);
        else req = sp_req_set(ld3, L, (int)C->fin[s], waitN, (int)C->cnt[s][0], (int)C->cnt[s][1], (int)C->cnt[s][2], (int)C->cnt[s][3], cs);
        req &= (u64)C->wnz[s][0] | ((u64)C->wnz[s][1] << 9) | ((u64)C->wnz[s][2] << 18) | ((u64)C->wnz[s][3] << 27);
        C->req[s] = req;
        C->fb[s] = (u8)fb;
        C->n_tiles[s] = (u8)__popcll(req);
    }
    mj_team_sync<SP_NT>();
}

// Level 0: which draws win and one scoring work item per draw entry, in the reference's order (plain tile if a non-red
// copy is left, then the red five); the state's header gets the entry counts, their number and the not_tsumo row.
__device__ SP_ATTR_L0P void sp_l0_probe_chunk(SpWork* W, SpCtx* X, SpChunk* C, int first, int n) {
    SP_ASSUME_LDS(X);
    SP_ASSUME_LDS(C);
    SP_HBM SpWork* const Wg = (SP_HBM SpWork*)W;
    const SpTabG TG = sp_tab_g(*SP_OPQ_PTR(1, &c_sp_tab));
    sp_chunk_probe(Wg, X, C, TG, first, n, 0);
    const int s = threadIdx.x & (SP_NT - 1);
    if (s < n) {
        const SpState S = sp_chunk_state(C, s);
        const u64 req = C->req[s];
        const u32 slot = C->slot[s];
        int cnt = 0, sumreq = 0;
        for (u64 rest = req; rest; rest &= rest - 1) {
            const int t = __ffsll((long long)rest) - 1;
            const bool aka = sp_aka_in_wall(S, t);
            cnt += (!aka || S.w.get(t) >= 2) + aka;
            sumreq += S.w.get(t);
        }
        u64 cnt2 = 0;
        if (cnt > SP_L0_MAX) { X->overflow = 1; cnt = SP_L0_MAX; }
        const int base = atomicAdd(&X->n_items, cnt);
        int e = 0;
        for (u64 rest = req; rest; rest &= rest - 1) {
            const int t = __ffsll((long long)rest) - 1;
            const bool aka = sp_aka_in_wall(S, t);
            const int wc = S.w.get(t);
            for (int variant = 0; variant < 2; variant++) {
                if (variant == 0 ? (aka && wc < 2) : !aka) continue;
                if (e < cnt) {
                    if (base + e < SP_ITEMS) Wg->items[base + e] = (u32)(first + s) | ((u32)e << 14) | ((u32)t << 19) | ((u32)variant << 25);
                    else X->overflow = 1;
                    cnt2 |= (u64)(((!aka ? wc : variant == 0 ? wc - 1 : 1) - 1) & 3) << (2 * e);  // draw_without_tegawari's `count` (1..4)
                }
                e++;
            }
        }
        (void)slot;
        SP_HBM SpHdr& hd = Wg->hdr[first + s];
        hd.child_off = (u32)min(base, SP_ITEMS - 1);  // the state's first work item: its scores are l0sc[child_off ...]
        hd.n_ch = (unsigned short)cnt;
        hd.sumreq = (u8)(sumreq & 0xFF);
        hd.n_ent = (u8)cnt;
        hd.l0cnt2 = cnt2;
    }
    mj_team_sync<SP_NT>();
}

// Levels >= 1: required draws, shanten-keeping discards and the children of SP_NS states.
__device__ SP_ATTR_EXPAND void sp_expand_chunk(SpWork* W, SpCtx* X, SpChunk* C, int first, int n, int L) {
    SP_ASSUME_LDS(X);
    SP_ASSUME_LDS(C);
    SP_HBM SpWork* const Wg = (SP_HBM SpWork*)SP_OPQ_PTR(2, W);
    const SpTabG TG = sp_tab_g(*SP_OPQ_PTR(1, &c_sp_tab));
    const int tid = SP_OPQ(1, (int)(threadIdx.x & (SP_NT - 1)));
    const int ld3 = X->len_div3;
    const u32 tag_ep = X->tag_epoch;
    // optional pass timers (MJ_SP_PROF): wave wall-clock per pass, summed into prof[8..13] by lane 0
    const bool prof = X->prof != nullptr;
    long long tq0 = prof ? wall_clock64() : 0, tq1 = 0;
    long long acc_list = 0, acc_keep = 0, acc_layout = 0, acc_ins = 0;
    int n_items_total = 0, n_edges_total = 0;
    sp_chunk_probe(Wg, X, C, TG, first, n, L);
    if (prof) tq1 = wall_clock64();
    const long long t_probe = tq1 - tq0;

    // The draw items (state, required tile) are processed in sub-batches of whole states with at most SP_ITEM_CAP = 64 items:
    // one lane per item.  A chunk with more items is split into sub-batches of about equal size.  Lane q < n holds the item count
    // of state q; ONE wavefront scan gives every prefix the sub-batch logic needs (round 4: the unrolled walk over the 16 states
    // with its running sums was 370 instructions per sub-batch, a fifth of the layout pass another 100).
    const int nt_lane = tid < n ? (int)C->n_tiles[tid] : 0;
    const int pre_in = (int)sp_wave_scan_incl((u32)nt_lane), pre_ex = pre_in - nt_lane;
    const int total_items = sp_lane_get(pre_in, SP_NT - 1);
    const int n_sub = (total_items + SP_ITEM_CAP - 1) / SP_ITEM_CAP, target = n_sub > 1 ? (total_items + n_sub - 1) / n_sub : SP_ITEM_CAP;
    for (int sb = 0; sb < n;) {
        long long tp0 = prof ? wall_clock64() : 0, tp1 = 0, tp2 = 0, tp3 = 0, tp4 = 0;
        // the sub-batch [sb, se): state q > sb ends it if its items do not fit any more or the target is reached (both conditions
        // are monotone in q, so the first lane that raises its hand is the end)
        const int base = sp_lane_get(pre_ex, sb);
        const bool stop = tid > sb && tid < n && (pre_in - base > SP_ITEM_CAP || pre_ex - base >= target);
        const u64 stop_m = __ballot(stop);
        const int se = stop_m ? __ffsll((long long)stop_m) - 1 : n;  // uniform over the wavefront
        const int n_items = sp_lane_get(pre_ex, se) - base;          // (lane n holds 0 items: its exclusive prefix is the total)
        const int my_first = pre_ex - base;                          // first item of state `tid`
        // item list: one lane per state walks its required-draw set (ascending); a state without draws left gets its header now
        if (tid >= sb && tid < se) {
            const int s = tid;
            int it = my_first;
            for (u64 rest = C->req[s]; rest; rest &= rest - 1) C->item[it++] = (unsigned short)(s | ((__ffsll((long long)rest) - 1) << SP_SB));
            if (it == my_first) {
                SP_HBM SpHdr& hd = Wg->hdr[first + s];
                hd.child_off = 0;
                hd.n_ch = 0;
                hd.sumreq = 0;
                hd.n_ent = 0;
            }
        }
        mj_team_sync<SP_NT>();
        if (prof) tp1 = wall_clock64();
        // P2: the shanten-keeping discards of g = h + t, lane = item
        const bool has_item = tid < n_items;
        const int my_e = has_item ? (int)C->item[tid] : 0, my_s = my_e & (SP_NS - 1), my_t = my_e >> SP_SB;
        const SpState S = sp_chunk_state(C, my_s);
        u64 kept = 0;
        if (has_item) {
            const int s = my_s, t = my_t;
            const int st = sh_suit(t);
            const u32 key1 = C->key[s][st] + sh_pow(t);
            const int hc = S.h.get(t);
            if (C->fb[s] || !spt_in_table(TG, st, key1)) {
                Hand g = S.h;
                g.inc(t);
                kept = sp_keep_brute_dev(g, ld3, L - 1);
            } else {
                const u32 id1 = spt_id(TG, st, key1);
                const SpRec o1 = spt_opt(TG, ld3, (u32)C->r3[s][st], id1);
                u64 keepN = (u64)spt_keep_tiles(spt_rec(TG, st, key1, 1), o1) << (9 * st);
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    const int u = q + (q >= st);                                                   // q-th suit != st
                    const u32 others = spt_merge(TG, (u32)C->r2[s][5 - sh_pair_idx(st, u)], id1);  // the two untouched suits + the new row
                    const SpRec o = spt_opt(TG, ld3, others, (u32)C->id[s][u]);
                    keepN |= (u64)spt_keep_tiles(C->keep[s][u], o) << (9 * u);
                }
                const int yao = (int)((YAOKYUU_MASK >> t) & 1);
                SpCountSets cs;
                cs.c1 = C->cs[s][0];
                cs.c2 = C->cs[s][1];
                cs.nz = C->cs[s][2];
                kept = sp_keep_set(ld3, L - 1, spt_fin(o1), keepN, (int)C->cnt[s][0] + (hc == 1), (int)C->cnt[s][1] + (hc == 0),
                                   (int)C->cnt[s][2] + (yao && hc == 1), (int)C->cnt[s][3] + (yao && hc == 0), sp_count_sets_add(cs, t, hc));
            }
            kept &= ~(1ull << t);  // d == t gives the state itself back
            C->kept[tid] = kept;
        }
        if (prof) tp2 = wall_clock64();
        // P3: child list layout (for each required tile `variants(t) * kept discards` entries) + node header: one inclusive
        // wavefront scan over the item lanes of (child entries | wall copies << 16 | draw entries << 24); a state's items are
        // contiguous lanes, so its totals are a difference of two prefix values
        int n_entries;
        {
            const int wc = has_item ? S.w.get(my_t) : 0;
            const int nvar = has_item && sp_aka_in_wall(S, my_t) ? (wc >= 2 ? 2 : 1) : 1;
            const int nkeep = __popcll(kept), my_ent = nvar * nkeep;
            const u32 pv = sp_wave_scan_incl(has_item ? ((u32)my_ent | ((u32)wc << 16) | ((u32)(nkeep ? nvar : 0) << 24)) : 0u);
            // this lane's state: first item lane, items (one shuffle of the state lane's prefix | count)
            const int sfn = __shfl((pre_ex - base) | (nt_lane << 16), has_item ? my_s : sb);  // (lanes before sb are never asked)
            const int s_first = sfn & 0xFFFF, s_n = sfn >> 16;
            const int s_last = s_first + s_n - 1;
            const u32 before = __shfl(pv, max(s_first - 1, 0)), upto = __shfl(pv, max(s_last, 0)), all = __shfl(pv, SP_NT - 1);
            const u32 base = s_first > 0 ? before : 0u;
            n_entries = (int)(all & 0xFFFFu);
            if (has_item) {
                C->coff[tid] = (unsigned short)(((pv - base) & 0xFFFFu) - (u32)my_ent);
                const int e_first = (int)((pv & 0xFFFFu) - (u32)my_ent);
                C->eoff[tid] = (unsigned short)e_first;
                // this item's entries (draw variant x kept discard, discards ascending), decoded once here: the insert pass found an
                // entry's item by a binary search over eoff (6 dependent LDS reads) and its discard by a loop over the kept bits, per entry
                {
                    int r = 0;
                    for (u64 rest = kept; rest; rest &= rest - 1, r++) {
                        const u32 d = (u32)(__ffsll((long long)rest) - 1);
                        const u32 w = (u32)tid | (d << 6) | ((rest & (rest - 1)) == 0ull ? 1u << 13 : 0u);
                        if (e_first + r < SP_EINFO_CAP) C->einfo[e_first + r] = w;
                        if (nvar == 2 && e_first + nkeep + r < SP_EINFO_CAP) C->einfo[e_first + nkeep + r] = w | (1u << 12);
                    }
                }
                if (tid == s_last) {  // one lane per state: the pool space and the node header
                    const u32 tot = upto - base;
                    int total = (int)(tot & 0xFFFFu);
                    const int sumreq = (int)((tot >> 16) & 0xFFu), n_ent = (int)(tot >> 24);
                    int child_base = atomicAdd(&X->n_pool, total);
                    if (child_base + total > SP_POOL) { X->overflow = 1; child_base = SP_POOL; total = 0; }
                    C->child_base[my_s] = child_base;
                    SP_HBM SpHdr& hd = Wg->hdr[first + my_s];
                    hd.child_off = (u32)min(child_base, SP_POOL - 1);
                    hd.n_ch = (unsigned short)total;
                    hd.sumreq = (u8)(sumreq & 0xFF);
                    hd.n_ent = (u8)min(n_ent, 255);
                }
            }
            if (tid == 0) C->eoff[n_items] = (unsigned short)n_entries;
        }
        mj_team_sync<SP_NT>();
        if (prof) tp3 = wall_clock64();
        // P4: children, one lane per CHILD ENTRY (draw variant x kept discard): it inserts its child state into the hash set and
        // leaves its child-list entry at its place of the reference's order (t, variant, d ascending).  A lane takes TWO entries
        // per round and claims both hash slots before it looks at either answer: an insert is one L2 atomic round trip of
        // latency and little else, so two in flight per lane halve the rounds.
        struct Ent {  // one child entry, decoded
            bool on;
            int s, it, local, rank, nk, tile, dt, count;
            u64 dk;
            u32 pos;
        };
        auto decode = [&](int e) -> Ent {
            Ent E;
            E.on = e < n_entries;
            const int ee = E.on ? e : 0;
            int lo, vidx, d;
            if (ee < SP_EINFO_CAP) {  // decoded by the layout pass
                const u32 w = C->einfo[ee];
                lo = (int)(w & 63u);
                d = (int)((w >> 6) & 63u);
                vidx = (int)((w >> 12) & 1u);
                E.local = ee - (int)C->eoff[lo];
                E.nk = 2;                                // only `rank == nk - 1` (the last discard of the draw entry) is asked of these two
                E.rank = (w >> 13) & 1u ? 1 : 0;
            } else {
                lo = 0;
                int hi = n_items;  // largest item with eoff[item] <= e
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if ((int)C->eoff[mid] <= ee) lo = mid; else hi = mid;
                }
                E.local = ee - (int)C->eoff[lo];
                const u64 bits = C->kept[lo];
                E.nk = __popcll(bits);
                vidx = E.local >= E.nk ? 1 : 0;
                E.rank = E.local - vidx * E.nk;
                u64 mrest = bits;
                for (int r = E.rank; r > 0; r--) mrest &= mrest - 1;
                d = __ffsll((long long)mrest) - 1;
            }
            E.it = lo;
            E.s = C->item[lo] & (SP_NS - 1);
            const int t = C->item[lo] >> SP_SB;
            const SpState Sx = sp_chunk_state(C, E.s);
            const int cnt = Sx.w.get(t);
            const bool aka = sp_aka_in_wall(Sx, t);
            // the tile's draw entries: plain (all copies but the red one) if any, then the red five
            const bool red = aka && (vidx == 1 || cnt < 2);
            E.count = !aka ? cnt : red ? 1 : cnt - 1;
            E.tile = red ? akaize(t) : t;
            const u32 akas1 = red ? (Sx.akas | (1u << (E.tile - T_5MR))) : Sx.akas;  // akas_in_hand after the draw
            const int c = Sx.h.get(d & 63);  // d != t: the draw does not change its count
            int dt = d;  // aka variant rule (state.rs:116-121): the red five goes last
            if (d == T_5M && (akas1 & 1) && c == 1) dt = T_5MR;
            else if (d == T_5P && (akas1 & 2) && c == 1) dt = T_5PR;
            else if (d == T_5S && (akas1 & 4) && c == 1) dt = T_5SR;
            E.dt = dt;
            E.dk = sp_dk_add(C->dk[E.s], E.tile, dt);
            E.pos = sp_dk_pos(E.dk);
            return E;
        };
        auto finish = [&](const Ent& E, u64 first_old, int known) -> int {  // the rest of sp_insert after the first look, the list and the child entry
            const u64 tag = SP_TAG(E.dk, tag_ep);
            const SpState Sx = sp_chunk_state(C, E.s);
            u32 pos = E.pos;
            u64 old = first_old;
            int cs = known;  // >= 0: the child cache knew the slot
            bool fresh = false;
#ifdef MJ_EMU
            if (known >= 0 && Wg->tag[known] != tag) X->overflow = 1;  // a cache hit must name the slot that holds this very state
#endif
            for (int probe = 0; cs < 0 && probe < SP_CAP; probe++) {
                if (old == 0ull) {
                    fresh = true;
                    cs = (int)pos;
                    break;
                }
                if (old == tag) {
#ifdef MJ_EMU
                    sp_emu_check_hit(Wg, X, pos, sp_apply(Sx, E.tile, E.dt));
#endif
                    cs = (int)pos;
                    break;
                }
                pos = (pos + 1) & (SP_CAP - 1);
                old = Wg->tag[pos];
                if (sp_tag_free(old, tag_ep)) old = sp_claim_tag(&Wg->tag[pos], tag, old);
            }
            if (cs < 0) X->overflow = 1;
            if (fresh) sp_new_state(Wg, X, (u32)cs, E.dk, sp_apply(Sx, E.tile, E.dt));  // next list index: slot, key and id
            const int pos_out = C->child_base[E.s] + (int)C->coff[E.it] + E.local;
            const u32 ent = (cs < 0 ? SP_ENT_INVALID : (u32)cs) | ((u32)sp_discard_key(E.dt) << 14) | (E.rank == E.nk - 1 ? SP_ENT_LAST : 0u) |
                            ((u32)E.count << 24);
            if (pos_out < SP_POOL) Wg->pool[pos_out] = ent;
            return cs;
        };
        // The child cache (round 4): a child is looked up by ~6 parents, most of them neighbours in the level list, and five look-ups
        // in six only need its slot.  A direct-mapped table in LDS remembers (state id -> slot) of the children this workgroup saw
        // last: a hit costs one ds_read and no tag line from L2 / HBM (a 128-byte line per 8-byte tag otherwise).
#if SP_CC_N > 0
        unsigned long long* const cc = X->cc;
        SP_ASSUME_LDS(cc);  // (a generic pointer would turn the look-up into a flat_load, which waits for every HBM gather in flight)
#else
        unsigned long long* const cc = nullptr;
#endif
        const u64 cc_ep = (u64)X->cc_epoch << 42;
        auto cc_look = [&](const Ent& E) -> int {
            if (!cc || !E.on) return -1;
            const u64 v = cc[E.pos & (SP_CC_N - 1)];
            return (v >> 14) == (cc_ep | E.dk) ? (int)(v & 0x3FFFu) : -1;
        };
        auto cc_put = [&](const Ent& E, int cs) {
            if (cc && cs >= 0) cc[E.pos & (SP_CC_N - 1)] = ((cc_ep | E.dk) << 14) | (u64)(u32)cs;
        };
        for (int e0 = 0; e0 < n_entries; e0 += 2 * SP_NT) {
            const bool two = e0 + SP_NT < n_entries;  // uniform: a round with at most 64 entries decodes one entry per lane (-0.9 %)
            const Ent A = decode(e0 + tid);
            Ent B = A;
            B.on = false;
            if (two) B = decode(e0 + SP_NT + tid);
            u64 oa = 1ull, ob = 1ull;
            const int ka = cc_look(A), kb = cc_look(B);
            // five edges in six find their child already there: LOOK before claiming (a plain load; within a row a tag only ever goes
            // from free -- zero or another row's epoch -- to its final value, so a stale value can only look free, which costs the
            // atomic that would have been issued anyway).  mj_k_sp -3.0 % (round 4, same box): 144 M L2 atomics per launch
            // become ~25 M.
            if (A.on && ka < 0) oa = Wg->tag[A.pos];
            if (B.on && kb < 0) ob = Wg->tag[B.pos];
            if (A.on && ka < 0 && sp_tag_free(oa, tag_ep)) oa = sp_claim_tag(&Wg->tag[A.pos], SP_TAG(A.dk, tag_ep), oa);
            if (B.on && kb < 0 && sp_tag_free(ob, tag_ep)) ob = sp_claim_tag(&Wg->tag[B.pos], SP_TAG(B.dk, tag_ep), ob);
            if (A.on) {
                const int cs = finish(A, oa, ka);
                if (ka < 0) cc_put(A, cs);
            }
            if (B.on) {
                const int cs = finish(B, ob, kb);
                if (kb < 0) cc_put(B, cs);
            }
        }
        mj_team_sync<SP_NT>();
        if (prof) {
            tp4 = wall_clock64();
            acc_list += tp1 - tp0;
            acc_keep += tp2 - tp1;
            acc_layout += tp3 - tp2;
            acc_ins += tp4 - tp3;
            n_items_total += n_items;
            n_edges_total += n_entries;
        }
        sb = se;
    }
    if (prof && tid == 0) {
        atomicAdd(&X->pt[0], (unsigned long long)t_probe);       // P0-P1c keys, ids, optimal entries, required draws
        atomicAdd(&X->pt[1], (unsigned long long)acc_list);      // item lists
        atomicAdd(&X->pt[2], (unsigned long long)acc_keep);      // P2 keeping discards
        atomicAdd(&X->pt[3], (unsigned long long)acc_layout);    // P3 layout
        atomicAdd(&X->pt[4], (unsigned long long)acc_ins);       // P4 inserts
        atomicAdd(&X->pt[5], (unsigned long long)n_items_total); // draw items
        atomicAdd(&X->pt[6], (unsigned long long)n);             // states expanded
    }
}

// LDS of the evaluation: SP_EVW_WAVE_FLOATS per wavefront, split between its teams (a team is exactly T - off lanes wide, T = draws
// left, a constant of the row: floor(64 / (T - off)) teams per wavefront, so rows with 9 draws left run 7 states per wavefront
// instead of 4): levels > 0 park SP_EV_ENT rows-of-(T + 4) x 4 floats per team, level 0 SP_EV_ENT numerator rows of SP_EV0_STRIDE floats.
#define SP_EVAL_LDS_FLOATS(NT) (((NT) / 64) * 1408)

// Evaluation (bottom-up), by TEAMS of T - off lanes, lane = turn: tenpai / win / EV of calc.rs:447-561 into node.val[turn].
// A state `off` levels below the row's roots is reached after at least `off` draws, so nobody ever reads its values of the turns
// before that (a parent's lane of turn i reads val[j + 1], j >= i): the team is T - off lanes wide.
//   level 0 : for every draw entry with a yaku, accumulate its scores (sp_eval_wave0);
//   level > 0: walk the state's child list (written by sp_expand_chunk in the reference's order); per turn fold the children of a
//              draw entry like discard_slow (max of (int)EV, then discard priority), then accumulate (sp_eval_wave).
// accumulate = calc.rs:486-548: lane i adds, for j = i .. T-1 in order, prob(i, j) = tsumo_prob[count][j] * not_tsumo[j] /
// not_tsumo[i] times next[j + 1] — terms the reference skips (`break` on a zero probability, j < i for a lane that runs all j)
// are added as +0.0 products instead of being branched over (x + 0.0 == x for the non-negative sums here).
// The wavefront runs in lock-step, its teams side by side (uniform control flow): rounds 1-3 let every team run its accumulate at
// its own draw-entry boundaries (sp_eval_team: a third of the lanes per pass, every turn its own basic block behind an exec-mask
// branch); round 4 parks what the teams produce in LDS and accumulates all teams together.
#define SP_EV_ENT 4                       // children folded per step = upper bound of the entries parked per step
#define SP_EVW_WAVE_FLOATS 1408           // LDS per wavefront: teams x SP_EV_ENT x (T + 4) rows x 4 floats (T = 17: 4 teams)
MJD int sp_evw_team_floats(int T) { return SP_EV_ENT * (T + 4) * 4; }
// The accumulate of sp_eval_wave reads one 16-byte row per lane and turn: ds_read_b128 serves a wavefront in four groups of 16 lanes
// ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32: MI355X_MICROARCH.md, LDS) and a group is conflict-free when its 16 slots (address / 16 mod 16)
// differ.  Lane (team, turn) reads slot team * stride + turn + const, a group spans two or three teams: with the dense stride of
// 4 x (T + 4) rows (84 = 4 mod 16 for T = 17) lanes 20-27 (team 1) met lanes 12-15 (team 0) -- 0.41 of the kernel's LDS cycles were
// conflict cycles in round 5.  SP_EVW_STRIDE[T][off] = the team stride in rows, >= the dense one, that minimises the conflicts of a
// (T, off) geometry within the wavefront's 1,408 floats without costing a team (tools/experiments/r06_lds_stride.py enumerates the
// groups: most geometries become conflict-free, T = 17 at off = 1 / 2 stays at 1.5 cycles per group).
__device__ static const u8 SP_EVW_STRIDE[SP_T + 1][4] = {
    {16, 16, 16, 16}, {20, 20, 20, 20}, {25, 25, 24, 24}, {29, 28, 29, 28}, {32, 32, 32, 32}, {37, 36, 38, 38}, {42, 41, 44, 41}, {44, 44, 44, 44},
    {48, 50, 50, 50}, {57, 56, 55, 54}, {58, 57, 56, 56}, {67, 66, 69, 60}, {68, 67, 66, 69}, {77, 68, 69, 70}, {78, 77, 76, 75}, {79, 78, 77, 76},
    {80, 86, 86, 85}, {97, 84, 86, 86}};
#define SP_EV0_STRIDE 28                  // level 0: floats per parked entry: the numerators A[turn] of the turns before the last (20 >= T + 3,
                                          // zero from T - 1 on), the entry's 4 scores, the last turn's numerator
#define SP_EV0_SC 20
#define SP_EV0_LAST 24

// Level 0 (tenpai states): a draw entry is a winning draw with its four scores (sp_l0_score) and its wall count; nothing to fold.
// Per step up to SP_EV_ENT entries of every team: lane j parks A[j] = tsumo_prob[count][j] * not_tsumo[j], then every lane i adds
// prob(i, j) and prob(i, j) * score for j = i .. T-1 (the score picked by riichi-ippatsu at j == i, haitei at j == T-1: calc.rs:510-527).
template <int TN>
__device__ SP_ATTR_EVAL0 void sp_eval_wave0(SpWork* W, SpCtx* X, float* WL, int first, int end, int stride, int lane_in_team, int off_,
                                           int team_in_wave, bool team_on) {
    SP_ASSUME_LDS(X);
    SP_ASSUME_LDS(WL);
    SP_HBM SpWork* const Wg = (SP_HBM SpWork*)sp_uniform(W);
    SP_HBM SpNode* const nodeB = sp_uniform(&Wg->node[0]);  // uniform array bases + 32-bit byte offsets: see sp_uniform
    SP_HBM u32* const elistB = sp_uniform(&Wg->elist[0]);
    const int T = __builtin_amdgcn_readfirstlane(X->T), off = __builtin_amdgcn_readfirstlane(off_);
    const int ln = min(lane_in_team + off, SP_T - 1);  // this lane's turn (lanes outside any team: clamped, never stored)
    float* const eb = WL + (team_on ? team_in_wave : 0) * (SP_EV_ENT * SP_EV0_STRIDE);  // [SP_EV_ENT][SP_EV0_STRIDE]
    if (team_on)
        for (int r = lane_in_team; r < SP_EV_ENT * SP_EV0_STRIDE / 4; r += T - off) *reinterpret_cast<SpF4*>(eb + 4 * r) = SpF4{0.f, 0.f, 0.f, 0.f};
    const float tp0 = X->tsumo_prob[0][ln], tp1 = X->tsumo_prob[1][ln], tp2 = X->tsumo_prob[2][ln], tp3 = X->tsumo_prob[3][ln];
    const bool assume_riichi = X->is_menzen && X->prefer_riichi, haitei = X->calc_haitei != 0;
    // which of the entry's four scores a term takes (hand points + 0..3 han): double riichi on turn 0, ippatsu on the lane's own
    // turn, haitei on the last turn
    const int hp_base = (int)(assume_riichi && X->calc_double_riichi && ln == 0);
    const int hp_own = hp_base + (int)assume_riichi + (int)(haitei && ln == T - 1), hp_last = hp_base + (int)haitei;
    const SP_HBM float* const nt_rows = (const SP_HBM float*)c_sp_nt + (size_t)__builtin_amdgcn_readfirstlane(min(X->n_left, SP_NT_ROWS - 1)) * (SP_NT_ROWS * SP_NT_STRIDE);
    const int last = max(end - 1, 0);
    auto ld_slot = [&](int i) -> u32 { return sp_ld<u32>(elistB, 4u * (u32)min(i, last)); };  // hash slot | list index << 14
    SP_HBM SpF4* const scB = sp_uniform(&Wg->l0sc[0]);
    SP_HBM SpHdr* const hdrB = sp_uniform(&Wg->hdr[0]);
    auto ld_hdr = [&](u32 el) -> u64 { return sp_ld<unsigned long long>(hdrB, 16u * SP_EL_IDX(el)); };
    auto ld_cnt = [&](u32 el) -> u64 { return sp_ld<unsigned long long>(hdrB, 16u * SP_EL_IDX(el) + 8u); };  // two bits per draw entry: copies - 1
    auto ld_m = [&](u64 hdr) -> float { return sp_ld<float>(nt_rows, 4u * ((u32)min((int)((hdr >> 48) & 0xFF), SP_NT_ROWS - 1) * SP_NT_STRIDE + (u32)ln)); };
    // an entry's four scores are kept as four scalars and picked by one-hot WEIGHTS (x * 1 + 0 + 0 + 0 is exact): a struct or array
    // whose element is picked by a run-time index ends up in scratch behind flat loads
    // (scores by work item: a state's draw entries are consecutive items, (u32)header = the first one; items past the state's own
    // belong to other states -- finite values, never used)
    auto ld_sc = [&](u64 hdr, int e) -> SpF4 { return sp_ld4(scB, 16u * min((u32)hdr + (u32)e, (u32)(SP_ITEMS - 1))); };
    const float wb0 = hp_base == 0 ? 1.f : 0.f, wb1 = hp_base == 1 ? 1.f : 0.f;
    const float wo0 = hp_own == 0 ? 1.f : 0.f, wo1 = hp_own == 1 ? 1.f : 0.f, wo2 = hp_own == 2 ? 1.f : 0.f, wo3 = hp_own == 3 ? 1.f : 0.f;
    // the term of the last turn j = T - 1: the lane of that turn takes its own score, every other lane the haitei one
    const int hp_fin = ln == T - 1 ? hp_own : hp_last;
    const float wf0 = hp_fin == 0 ? 1.f : 0.f, wf1 = hp_fin == 1 ? 1.f : 0.f, wf2 = hp_fin == 2 ? 1.f : 0.f, wf3 = hp_fin == 3 ? 1.f : 0.f;
    const int nl = T - ln;  // this lane's terms: turns ln .. T - 1
    mj_team_sync<64>();

    int i = first;
    bool has = team_on && i < end;
    u32 s0 = ld_slot(i), s1 = ld_slot(i + stride), s2 = ld_slot(i + 2 * stride);
    u64 h0 = ld_hdr(s0), h1 = ld_hdr(s1);
    float m_raw = ld_m(h0);
    float sx[SP_EV_ENT], sy[SP_EV_ENT], sz[SP_EV_ENT], sw[SP_EV_ENT];
#pragma unroll
    for (int q = 0; q < SP_EV_ENT; q++) {
        const SpF4 p = ld_sc(h0, q);
        sx[q] = p.x; sy[q] = p.y; sz[q] = p.z; sw[q] = p.w;
    }
    u64 cw = ld_cnt(s0);
    while (__ballot(has) != 0ull) {
        const int n_ent = (int)((h0 >> 32) & 0xFFFF);
        // in flight under this state: the next state's first entries, the header after it, the slot after that
        const float m_n = ld_m(h1);
        float nx_[SP_EV_ENT], ny_[SP_EV_ENT], nz_[SP_EV_ENT], nw_[SP_EV_ENT];
#pragma unroll
        for (int q = 0; q < SP_EV_ENT; q++) {
            const SpF4 p = ld_sc(h1, q);
            nx_[q] = p.x; ny_[q] = p.y; nz_[q] = p.z; nw_[q] = p.w;
        }
        const u64 cwn = ld_cnt(s1);
        const u64 h2 = ld_hdr(s2);
        const u32 s3 = ld_slot(i + 3 * stride);
        const float my_m = m_raw != 0.f ? m_raw : 1.f, my_r = sp_rcp_refined(my_m);
        float acc_w = 0.f, acc_e = 0.f;
        int nmax = has ? n_ent : 0;  // the longest entry list of the wavefront's current states (uniform loop bound)
        for (int d = 32; d > 0; d >>= 1) nmax = max(nmax, __shfl_xor(nmax, d));
        for (int e0 = 0; e0 < nmax; e0 += SP_EV_ENT) {
            if (e0 > 0) {  // more than SP_EV_ENT draw entries (rare)
#pragma unroll
                for (int q = 0; q < SP_EV_ENT; q++) {
                    const SpF4 p = ld_sc(h0, e0 + q);
                    sx[q] = p.x; sy[q] = p.y; sz[q] = p.z; sw[q] = p.w;
                }
            }
            u32 use = 0;  // bit q: this team has a draw entry with a yaku in slot q of the step
#pragma unroll
            for (int q = 0; q < SP_EV_ENT; q++) {
                const bool u = has && e0 + q < n_ent && sx[q] != 0.f;  // a draw entry with a yaku (sp_l0_score)
                const u32 cnt = ((u32)(cw >> (2 * (e0 + q))) & 3u) + 1u;
                const float tpc = cnt <= 1 ? tp0 : cnt == 2 ? tp1 : cnt == 3 ? tp2 : tp3;
                if (u) {
                    // A[ln]; the LAST turn's numerator is parked apart and its place in the row stays zero: its term is the last one
                    // of every lane, added after the row's loop with the haitei score (below)
                    const float a = tpc * m_raw;
                    eb[q * SP_EV0_STRIDE + ln] = ln == T - 1 ? 0.f : a;
                    if (ln == T - 1) eb[q * SP_EV0_STRIDE + SP_EV0_LAST] = a;
                    if (lane_in_team == 0) *reinterpret_cast<SpF4*>(eb + q * SP_EV0_STRIDE + SP_EV0_SC) = SpF4{sx[q], sy[q], sz[q], sw[q]};  // the entry's scores ride along
                }
                use |= u ? (1u << q) : 0u;
            }
            mj_team_sync<64>();
#pragma nounroll
            for (int q = 0; q < SP_EV_ENT; q++) {  // ONE copy of the accumulate (a dynamic loop over the parked entries)
                const bool u = (use >> q) & 1;
                if (__ballot(u) == 0ull) continue;
                const float* ar = eb + q * SP_EV0_STRIDE;
                const float* al = ar + ln;  // lane-relative turns (see sp_eval_wave): this lane's terms j = ln, ln + 1, ...
                if (u) {
                    const SpF4 sq = *reinterpret_cast<const SpF4*>(ar + SP_EV0_SC);
                    // picked by one-hot weights (x * 1 + 0 + 0 + 0 is exact): a run-time index into a register tuple would go through scratch
                    const float s_base = sq.x * wb0 + sq.y * wb1;  // hp_base is 0 or 1
                    const float s_own = sq.x * wo0 + sq.y * wo1 + sq.z * wo2 + sq.w * wo3;
                    const float s_fin = sq.x * wf0 + sq.y * wf1 + sq.z * wf2 + sq.w * wf3;  // the last turn's score: haitei, or the lane's own
                    sp_static_for<0, (TN + 3) / 4>([&](auto gc) {
                        constexpr int g = decltype(gc)::value;
                        if (4 * g >= T - off) return;  // scalar
                        if (4 * g < nl) {
                            float a4[4];
#pragma unroll
                            for (int jj = 0; jj < 4; jj++) a4[jj] = al[4 * g + jj];  // past T - 2: zero
                            const float so_ = s_own, sb_ = s_base;  // values, not the lambda's references (else: pointer selects + flat loads)
#pragma unroll
                            for (int jj = 0; jj < 4; jj++) {
                                const float prob_ = sp_div_domain(a4[jj], my_m, my_r);
                                acc_w += prob_;
                                acc_e += prob_ * ((g == 0 && jj == 0) ? so_ : sb_);  // the lane's own turn first (riichi-ippatsu), then the plain score
                            }
                        }
                    });
                    const float prob_l = sp_div_domain(ar[SP_EV0_LAST], my_m, my_r);  // j = T - 1
                    acc_w += prob_l;
                    acc_e += prob_l * s_fin;
                }
            }
            mj_team_sync<64>();
        }
        if (has) {
            sp_st3(nodeB, SP_EL_SLOT(s0) * (u32)sizeof(SpNode) + (u32)offsetof(SpNode, val) + 12u * (u32)ln, 0.f, acc_w, acc_e);
        }
        s0 = s1; s1 = s2; s2 = s3;
        h0 = h1; h1 = h2;
        m_raw = m_n;
        cw = cwn;
#pragma unroll
        for (int q = 0; q < SP_EV_ENT; q++) {
            sx[q] = nx_[q]; sy[q] = ny_[q]; sz[q] = nz_[q]; sw[q] = nw_[q];
        }
        i += stride;
        has = team_on && i < end;
    }
}

// ---- Levels > 0.  Rounds 1-3 (sp_eval_team) let every team run the accumulate — T iterations of {LDS read, division, 3
// multiply-adds} — at ITS OWN draw-entry boundaries: with 3-4 teams per wavefront the boundaries rarely coincide, so the
// wavefront executed the accumulate up to 3-4 times per entry position with a third of its lanes, and every turn of the
// unrolled loop was its own basic block behind an exec-mask branch (three LDS reads, a wait, eight VALU, three SALU).
// Here a STEP = {fold SP_EV_ENT children per team (selects, no branches), park every completed draw
// entry's folded row in the team's LDS buffer, issue the loads of the next step, accumulate ALL parked entries of ALL teams
// together}: one b128 LDS read per turn (the row {nx_t, nx_w, nx_e, A}: the numerator A[j] = tsumo_prob[count][j] *
// not_tsumo[j] rides in the spare word of row j + 1), turns in groups of four behind one scalar branch, rows past T are zero
// (+0.0 terms).  The loads of the next step (child values, next child-list entries,
// the next states' headers) are issued before the accumulate and consumed after it.  The f32 operation order per lane is the
// reference's: entries in list order, turns ascending.
template <int TN, int LK>
__device__ SP_ATTR_EVAL void sp_eval_wave(SpWork* W, SpCtx* X, float* WL, int first, int end, int stride, int lane_in_team, int off_,
                                          int team_in_wave, bool team_on) {
    static_assert(LK >= 1, "level 0 has no children: sp_eval_wave0");
    SP_ASSUME_LDS(X);
    SP_ASSUME_LDS(WL);
    SP_HBM SpWork* const Wg = (SP_HBM SpWork*)sp_uniform(W);
    const int T = __builtin_amdgcn_readfirstlane(X->T), off = __builtin_amdgcn_readfirstlane(off_);
    const int ln = min(lane_in_team + off, SP_T - 1);  // this lane's turn (lanes outside any team: clamped, never stored)
    const int rows = T + 4;
    const int team_rows = __builtin_amdgcn_readfirstlane((int)SP_EVW_STRIDE[min(max(T, 0), SP_T)][min(max(off, 0), 3)]);  // >= SP_EV_ENT * rows
    float* const eb = WL + (team_on ? team_in_wave : 0) * (team_rows * 4);  // [SP_EV_ENT][rows][4] (+ the stride's padding, never touched)
    if (team_on)
        for (int r = lane_in_team; r < SP_EV_ENT * rows; r += T - off) *reinterpret_cast<SpF4*>(eb + 4 * r) = SpF4{0.f, 0.f, 0.f, 0.f};
    const float tp0 = X->tsumo_prob[0][ln], tp1 = X->tsumo_prob[1][ln], tp2 = X->tsumo_prob[2][ln], tp3 = X->tsumo_prob[3][ln];
    const SP_HBM float* const nt_rows = (const SP_HBM float*)c_sp_nt + (size_t)__builtin_amdgcn_readfirstlane(min(X->n_left, SP_NT_ROWS - 1)) * (SP_NT_ROWS * SP_NT_STRIDE);
    const int last = max(end - 1, 0);
    const u32 val_ln = (u32)offsetof(SpNode, val) + (u32)ln * 12u;  // this lane's turn inside a node
    // one uniform base per array (an SGPR pair each): a constant array offset added to the 32-bit element offset would be folded into
    // a 64-bit address again
    SP_HBM SpNode* const nodeB = sp_uniform(&Wg->node[0]);
    SP_HBM u32* const elistB = sp_uniform(&Wg->elist[0]);
    SP_HBM u32* const poolB = sp_uniform(&Wg->pool[0]);
    SP_HBM SpHdr* const hdrB = sp_uniform(&Wg->hdr[0]);
    auto ld_slot = [&](int i) -> u32 { return sp_ld<u32>(elistB, 4u * (u32)min(i, last)); };  // hash slot | list index << 14
    auto ld_hdr = [&](u32 el) -> u64 { return sp_ld<unsigned long long>(hdrB, 16u * SP_EL_IDX(el)); };
    auto ld_m = [&](u64 hdr) -> float { return sp_ld<float>(nt_rows, 4u * ((u32)min((int)((hdr >> 48) & 0xFF), SP_NT_ROWS - 1) * SP_NT_STRIDE + (u32)ln)); };
    auto ld_ent = [&](u32 at) -> u32 { return sp_ld<u32>(poolB, 4u * min(at, (u32)(SP_POOL - 1))); };
    auto ld_val = [&](u32 ent) -> SpF3 { return sp_ld3(nodeB, SP_ENT_SLOT(ent) * (u32)sizeof(SpNode) + val_ln); };  // one 12-byte load
    mj_team_sync<64>();

    // the pipeline: state 0 = current, 1 = next (header, first entries and not_tsumo value loaded), 2 = header loaded, 3 = slot
    int i = first;
    bool has = team_on && i < end;
    u32 s0 = ld_slot(i), s1 = ld_slot(i + stride), s2 = ld_slot(i + 2 * stride), s3 = ld_slot(i + 3 * stride);
    u64 h0 = ld_hdr(s0), h1 = ld_hdr(s1), h2 = ld_hdr(s2);
    float m_raw = ld_m(h0), m_nxt = ld_m(h1);
    u32 ent[SP_EV_ENT], entn[SP_EV_ENT], nent[SP_EV_ENT];
#pragma unroll
    for (int q = 0; q < SP_EV_ENT; q++) {
        ent[q] = ld_ent((u32)h0 + q);
        entn[q] = ld_ent((u32)h0 + SP_EV_ENT + q);
        nent[q] = ld_ent((u32)h1 + q);
    }
    SpF3 v[SP_EV_ENT];
#pragma unroll
    for (int q = 0; q < SP_EV_ENT; q++) v[q] = ld_val(ent[q]);

    int c0 = 0;
    float my_m = m_raw != 0.f ? m_raw : 1.f, my_r = sp_rcp_refined(my_m);
    float acc_t = 0.f, acc_w = 0.f, acc_e = 0.f;
    // discard_slow (calc.rs:570-637) fold state of the draw entry in progress
    // The fold's order (calc.rs:600-615: larger (int)EV first, then the discard priority) as ONE integer: (int)EV << 9 | order key.
    // (int)EV is non-negative and below 2^19 (six-fold yakuman of the dealer: 288,000), the key is 9 bits wide; the initial
    // value loses against every child (INT_MIN there, the unknown tile's key 63 - 37 here: any real key is larger).
    float nx_t = -3.40282347e+38f, nx_w = -3.40282347e+38f, nx_e = -3.40282347e+38f;
    int max_pack = -1;
    float* const row0 = eb + ln * 4;  // this lane's row of the first parked entry
    const int nl = T - ln;            // this lane's terms: turns ln .. T - 1

    while (__ballot(has) != 0ull) {
        const int n_ch = (int)((h0 >> 32) & 0xFFFF);
        // ---- fold SP_EV_ENT children; a completed draw entry parks its row
        int k = 0, koff = 0;
#pragma unroll
        for (int q = 0; q < SP_EV_ENT; q++) {
            const u32 e = ent[q];
            const bool valid = has && c0 + q < n_ch;
            const bool bad = (e & SP_ENT_INVALID) != 0;
            if (valid && bad) X->overflow = 1;
            // `as i32` of the child's EV (maximize_win_prob = false) above the discard order key (cmp_discard_priority > 0 <=> larger key)
            const int pack = (sp_f2i(v[q].z) << 9) | (int)SP_ENT_KEY(e);
#ifdef MJ_EMU
            // the packing's range (the team's first lane reads a turn its children never wrote — nobody reads what it folds)
            if (valid && !bad && lane_in_team > 0 && ((unsigned)sp_f2i(v[q].z) >> 22)) X->overflow = 1;
#endif
            const bool better = valid && !bad && pack > max_pack;
            nx_t = better ? v[q].x : nx_t;
            nx_w = better ? v[q].y : nx_w;
            nx_e = better ? v[q].z : nx_e;
            max_pack = better ? pack : max_pack;
            if (valid && (e & SP_ENT_LAST)) {  // last child of this draw entry (uniform in the team)
                const u32 cnt = SP_ENT_COUNT(e);  // 1..4 copies of the drawn tile
                const float tpc = cnt <= 1 ? tp0 : cnt == 2 ? tp1 : cnt == 3 ? tp2 : tp3;
                float* row = row0 + koff;
                row[0] = nx_t;
                row[1] = nx_w;
                row[2] = nx_e;
                row[7] = tpc * m_raw;  // A[ln], read with row ln + 1
                k++;
                koff += rows * 4;
                nx_t = nx_w = nx_e = -3.40282347e+38f;
                max_pack = -1;
            }
        }
        c0 += SP_EV_ENT;
        const bool done = has && c0 >= n_ch;
        // ---- the loads of the next step: its children's values, the entries after them, and (used only when a state ends)
        // the pipeline's tail.  Issued by every lane whether or not its team advances: no divergent control flow.
        u32 up[SP_EV_ENT], upn[SP_EV_ENT], nent2[SP_EV_ENT];
        SpF3 vn[SP_EV_ENT];
        const u32 up_off = done ? (u32)h1 + SP_EV_ENT : (u32)h0 + (u32)c0 + SP_EV_ENT;
#pragma unroll
        for (int q = 0; q < SP_EV_ENT; q++) up[q] = done ? nent[q] : entn[q];
#pragma unroll
        for (int q = 0; q < SP_EV_ENT; q++) vn[q] = ld_val(up[q]);
#pragma unroll
        for (int q = 0; q < SP_EV_ENT; q++) upn[q] = ld_ent(up_off + q);
        // the pipeline's tail moves only when some team finishes its state in this step (a wavefront-uniform branch)
        float m_n2 = 0.f;
        u64 h3 = 0ull;
        u32 s4 = 0u;
#pragma unroll
        for (int q = 0; q < SP_EV_ENT; q++) nent2[q] = 0u;
        if (__ballot(done) != 0ull) {
#pragma unroll
            for (int q = 0; q < SP_EV_ENT; q++) nent2[q] = ld_ent((u32)h2 + q);
            m_n2 = ld_m(h2);
            h3 = ld_hdr(s3);
            s4 = ld_slot(i + 4 * stride);
        }

        // ---- accumulate (calc.rs:486-548) the parked entries of every team, entry by entry.  Lane-RELATIVE turns (round 5): lane ln
        // reads the rows of ITS terms j = ln, ln + 1, ... (row j + 1 = {nx_t, nx_w, nx_e of turn j + 1, A[j]}), four per group, and
        // drops out (exec mask) once j passes T - 1 -- no term is ever computed to be masked away.  Rounds 1-4 read row j as a
        // broadcast and multiplied the half of the (lane, turn) pairs with j < ln by a select: one v_cndmask per term, its 17
        // compare masks hoisted into SGPR pairs and, at level 0, spilled to VGPR lanes (2 v_readlane per use).
        mj_team_sync<64>();
        const int kmax = __ballot(k >= 4) ? 4 : __ballot(k >= 3) ? 3 : __ballot(k >= 2) ? 2 : __ballot(k >= 1) ? 1 : 0;
        for (int en = 0; en < kmax; en++) {
            const float* er = row0 + 4 + en * rows * 4;  // row ln + 1 of entry en
            const bool mine = en < k;
            sp_static_for<0, (TN + 3) / 4>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                if (4 * g >= T - off) return;  // scalar: the team's first lane has T - off terms, nobody has more
                if (mine && 4 * g < nl) {
                    SpF4 r[4];
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) r[jj] = *reinterpret_cast<const SpF4*>(er + (4 * g + jj) * 4);  // rows past T are zero
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) {
                        const float prob = sp_div_domain(r[jj].w, my_m, my_r);
                        if constexpr (LK == 1) acc_t += prob;
                        else acc_t += prob * r[jj].x;
                        acc_w += prob * r[jj].y;
                        acc_e += prob * r[jj].z;
                    }
                }
            });
        }
        mj_team_sync<64>();  // the parked rows are consumed: the next step may overwrite them

        // ---- end of a state: its values, then the pipeline moves up
        if (done) {
            sp_st3(nodeB, SP_EL_SLOT(s0) * (u32)sizeof(SpNode) + val_ln, acc_t, acc_w, acc_e);
            s0 = s1; s1 = s2; s2 = s3; s3 = s4;
            h0 = h1; h1 = h2; h2 = h3;
            m_raw = m_nxt; m_nxt = m_n2;
#pragma unroll
            for (int q = 0; q < SP_EV_ENT; q++) nent[q] = nent2[q];
            c0 = 0;
            my_m = m_raw != 0.f ? m_raw : 1.f;
            my_r = sp_rcp_refined(my_m);
            acc_t = acc_w = acc_e = 0.f;
            i += stride;
            has = i < end;
        }
#pragma unroll
        for (int q = 0; q < SP_EV_ENT; q++) {
            ent[q] = up[q];
            entn[q] = upn[q];
            v[q] = vn[q];
        }
    }
}

// The teams of a wavefront evaluate consecutive states of a level, and the wavefront runs as long as its slowest team: order
// the level by child-list length (counting sort over 64 buckets, workgroup-wide) so that neighbours cost about the same (key: children + 4 x draw entries).  The
// order of states inside a level does not touch the results (each state is evaluated on its own).
template <int NT>
__device__ __forceinline__ void sp_sort_level(SpWork* W, int* hist /* LDS [64] */, int b, int e) {
    SP_HBM SpWork* const Wg = (SP_HBM SpWork*)W;
    const int tid = threadIdx.x;
    if (e - b <= SP_SORT_MIN) {  // a handful of states (the root level): one round of teams whatever the order
        for (int i = b + tid; i < e; i += NT) Wg->elist[i] = Wg->list[i] | ((u32)i << 14);
        __syncthreads();
        return;
    }
    if (tid < 64) hist[tid] = 0;
    __syncthreads();
    auto cost_key = [&](int i) {  // ~ fold work (children) + accumulate work (draw entries), longest first (headers: dense by list index)
        const SP_HBM SpHdr& hd = Wg->hdr[i];
        return 63 - min(((int)hd.n_ch + 4 * (int)hd.n_ent) >> 1, 63);
    };
    // a thread's first four states keep their slot and key in registers between the two passes (levels up to 4 x 256 states)
    u32 my_slot[4];
    int my_key[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int i = b + tid + q * NT;
        my_slot[q] = i < e ? (Wg->list[i] | ((u32)i << 14)) : 0u;
        my_key[q] = i < e ? cost_key(i) : 0;
        if (i < e) atomicAdd(&hist[my_key[q]], 1);
    }
    for (int i = b + tid + 4 * NT; i < e; i += NT) atomicAdd(&hist[cost_key(i)], 1);
    __syncthreads();
    if (tid < 64) {  // exclusive prefix over the 64 buckets: one wavefront scan
        const int c = hist[tid];
        hist[tid] = (int)sp_wave_scan_incl((u32)c) - c;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int i = b + tid + q * NT;
        if (i < e) Wg->elist[b + atomicAdd(&hist[my_key[q]], 1)] = my_slot[q];
    }
    for (int i = b + tid + 4 * NT; i < e; i += NT) Wg->elist[b + atomicAdd(&hist[cost_key(i)], 1)] = Wg->list[i] | ((u32)i << 14);
    __syncthreads();
}

MJD int f32_total_cmp(float a, float b) {
    int x = __float_as_int(a), y = __float_as_int(b);
    x ^= (int)((unsigned)(x >> 31) >> 1);
    y ^= (int)((unsigned)(y >> 31) >> 1);
    return (x > y) - (x < y);
}

// ---------------------------------------------------------------- queue order: heaviest cost class first
// A row costs between nothing (no single-player tables for this decision) and ~6,000 states; the persistent workgroups pop rows
// from one queue, so a heavy row popped last leaves the other workgroups idle for its whole duration (15 % of the kernel,
// measured).  84 % of all states belong to the 3-shanten hands that may discard, 8 % to the 3-shanten hands before a call
// decision, 7 % to the 2-shanten hands: a counting sort of the rows by (shanten, may discard, discard candidates) -- the
// table's incremental shanten bookkeeping, no calculator work -- puts the long rows first and leaves the short ones to
// level the tail.
// The order of the rows in the queue does not touch the results (obs rows are addressed by row index).
#define SP_N_CLASS 8
MJD int sp_row_class(const TableOne* snap, uint32_t desc) {
    LaneT<TableOne> L;
    L.B = const_cast<TableOne*>(snap + ROW_TABLE(desc));
    L.l = 0;
    L.T = nullptr;
    const int p = ROW_SEAT(desc);
    const u32 cans = F1(cans, p);
    const bool cd = (cans & CAN_DISCARD) != 0;
    const int tiles_left = F(tiles_left);
    int sh = F1(shanten, p), tsumos_left;
    if (cd) {
        if (sh > 0 && F1(has_next_shanten, p)) sh -= 1;
        tsumos_left = tiles_left / 4;
    } else {
        const int target = (F1(cans_target, p) + 4 - p) & 3;
        tsumos_left = max(tiles_left - (4 - target), 0) / 4;
    }
    if (tiles_left < 4 || sh > 3 || tsumos_left < max(sh, 1)) return 7;  // no state graph at all
    if (sh <= 1) return 6;
    if (sh == 2) return cd ? 4 : 5;
    if (!cd) return 2;
    // 3-shanten, may discard: the number of root states = discards that keep the shanten number (the masks the table keeps
    // for its legal-action logic) splits the class further: >= 7 candidates average 1.5 k states, 6 -> 0.9 k, <= 5 -> 0.4 k
    const int n_cand = __popcll(F1(has_next_shanten, p) ? F1(next_shanten, p) : F1(keep_shanten, p));
    return n_cand >= 7 ? 0 : n_cand == 6 ? 1 : 3;
}

__global__ __launch_bounds__(256) void mj_k_order_classify(const TableOne* snap, const uint32_t* rows, int n, uint8_t* cls, int* cnt) {
    __shared__ int h[SP_N_CLASS];
    if (threadIdx.x < SP_N_CLASS) h[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const int c = sp_row_class(snap, rows[i]);
        cls[i] = (uint8_t)c;
        atomicAdd(&h[c], 1);
    }
    __syncthreads();
    if (threadIdx.x < SP_N_CLASS && h[threadIdx.x]) atomicAdd(&cnt[threadIdx.x], h[threadIdx.x]);
}

__global__ __launch_bounds__(256) void mj_k_order_scatter(const uint8_t* cls, int n, const int* cnt, int* cursor, uint32_t* order) {
    __shared__ int h[SP_N_CLASS], base[SP_N_CLASS];
    if (threadIdx.x < SP_N_CLASS) h[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    int c = 0, r = 0;
    if (i < n) {
        c = cls[i];
        r = atomicAdd(&h[c], 1);
    }
    __syncthreads();
    if (threadIdx.x < SP_N_CLASS) {
        int b = 0;
        for (int k = 0; k < (int)threadIdx.x; k++) b += cnt[k];
        base[threadIdx.x] = b + (h[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], h[threadIdx.x]) : 0);
    }
    __syncthreads();
    if (i < n) order[base[c] + r] = (uint32_t)i;
}

// The SP block of one decision row, rows 889 .. 1011 of obs v4 (obs_repr.rs:564-692), written WHOLE by this kernel since round 5 -- the
// encoder stops at row 890 (rounds 1-4: mj_k_encode<4> zero-filled the 123 rows, 16.7 KB per decision, and mj_k_sp rewrote the cells that
// are not zero: 1.1 GB per cycle written twice, the second time as scattered 4-byte stores into lines that had left the L2).
//   block row 0 / 1 : max EV / 100 k and / 30 k in every column          (ev100 / ev30)
//   block rows 2..69: required tiles per discard (34 keep + 34 shanten-down rows): bit column of rowm[row - 2]
//   block row 70/71 : the candidate with the most required tiles / required tiles without a discard: row7[0 / 1]
//   block rows 72.. : tenpai / win / EV tables, [3][SP_T] rows, column = the candidate's tile (tv: [candidate][turn][4], alive_n[candidate] =
//                     number of leading turns with a positive tenpai probability: take_while of obs_repr.rs:655-660)
// Row 889 (8-byte aligned only) goes out as 34 scalar stores (the encoder's tile holds zeros there), rows 890.. as one 8-byte store per lane.
template <bool WAVE, int NT, class OutP>
__device__ __forceinline__ void sp_block_write(OutP out, const int tid, const float ev100, const float ev30, const u64* rowm, const u64* row7,
                                               const bool table_ok, const bool one_for_all, const signed char* colcand, const float* tv,
                                               const int* alive_n, const int T) {
    constexpr int O_SP = 889;  // Lay<4>::sp
    if (tid < 34) out[O_SP * 34 + tid] = ev100;
    // Three block rows per 51 lanes: lane -> (row in the group, column pair) is loop-invariant, an iteration is one mask read, two bit tests
    // and one 8-byte store per lane.
    constexpr int G = NT / 51;                       // groups of 3 rows in flight (1 for a wavefront, 5 for the workgroup)
    const int grp = tid / 51, l51 = tid - 51 * grp;  // group, lane inside it
    const int lr = l51 / 17, c2 = l51 - 17 * lr;     // row inside the group, column pair
    const bool lane_on = grp < G;
    // block rows 1 .. 71: EV / 30 k, then the required-tile masks
    for (int r0 = 1 + 3 * grp; r0 < 72; r0 += 3 * G) {
        const int rb = r0 + lr;
        if (lane_on && rb < 72) {
            float v0, v1;
            if (rb == 1) {
                v0 = v1 = ev30;
            } else {
                const u64 m = !rowm ? 0ull : rb < 70 ? rowm[rb - 2] : row7[rb - 70];
                const u32 two = (u32)(m >> (2 * c2)) & 3u;
                v0 = (float)(two & 1u);
                v1 = (float)(two >> 1);
            }
            out[(O_SP + rb) * 34 + 2 * c2] = v0;
            out[(O_SP + rb) * 34 + 2 * c2 + 1] = v1;
        }
    }
    // block rows 72 .. 122: the tables, or zeros
    for (int r0 = 72 + 3 * grp; r0 < 123; r0 += 3 * G) {
        const int rb = r0 + lr;
        if (lane_on && rb < 123) {
            float v0 = 0.f, v1 = 0.f;
            if (table_ok) {
                const int t = rb - 72, kind = t >= 2 * SP_T ? 2 : t >= SP_T ? 1 : 0, turn = t - kind * SP_T;
                const int ca = one_for_all ? 0 : (int)colcand[2 * c2], cb = one_for_all ? 0 : (int)colcand[2 * c2 + 1];
                if (ca >= 0 && turn < T && turn < alive_n[ca]) v0 = tv[(ca * SP_T + turn) * 4 + kind];
                if (cb >= 0 && turn < T && turn < alive_n[cb]) v1 = tv[(cb * SP_T + turn) * 4 + kind];
            }
            out[(O_SP + rb) * 34 + 2 * c2] = v0;
            out[(O_SP + rb) * 34 + 2 * c2 + 1] = v1;
        }
    }
}

template <class P> struct SpF4Of { typedef const SpRec* type; };                    // 16-byte view of a table record pointer,
template <> struct SpF4Of<const SP_HBM TableOne*> { typedef const SP_HBM SpRec* type; };  // in the pointer's own address space
struct SpRowInfo {  // what the row set-up hands to the graph phases and to the encoder
    bool ok, can_discard0, can_discard, after_riichi, with_probs;
    int last_tsumo, cur_shanten, T, n_cand, ld3;
    u32 cans;
    SpState root;
};
template <bool WAVE> MJD void sp_sync() {  // the threads that process ONE row: a workgroup, or (rows without a state graph) a wavefront
    if constexpr (WAVE) mj_team_sync<64>();
    else __syncthreads();
}
// Row set-up: the table record, the preconditions of single_player_tables, the calculator's constants and the candidates with
// their required tiles.  NT threads (tid = 0 .. NT - 1) work on the row: the whole workgroup, or one wavefront in the tail of the
// queue where the rows have no state graph (sp_row_class == 7) and four of them are processed side by side.
template <bool WAVE, int NT, class RowsP, class SnapP, class OutP>
__device__ __forceinline__ SpRowInfo sp_row_front(RowsP rows, SnapP snap, SpWork* W, SpCtx& X, TableOne* st, const int tid, const int row, OutP out,
                                                  unsigned long long* prof) {  // (never a reference to the kernel's parameter block: hipcc would copy it to scratch)
    SpRowInfo R;
    R.ok = false;
    R.with_probs = false;
    R.n_cand = 0;
    const uint32_t desc = rows[row];
    const int table = ROW_TABLE(desc), p = ROW_SEAT(desc);
        {
            const auto src = reinterpret_cast<typename SpF4Of<SnapP>::type>(snap + table);
            SpRec* d4 = reinterpret_cast<SpRec*>(st);
            for (int i = tid; i < (int)(sizeof(TableOne) / 16); i += NT) d4[i] = spt_load(&src[i]);
        }
        sp_sync<WAVE>();
        LaneT<TableOne> L;
        L.B = st;
        L.l = 0;
        L.T = &c_mj_tables;
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
        R.ok = ok;
        sp_sync<WAVE>();
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
            sp_sync<WAVE>();
            const float v = X.cand_ev0[0];
            sp_block_write<WAVE, NT>(out, tid, fminf(fmaxf(v, 0.f), 100000.f) / 100000.f, fminf(fmaxf(v, 0.f), 30000.f) / 30000.f, (const u64*)nullptr,
                                     (const u64*)nullptr, false, false, (const signed char*)nullptr, (const float*)nullptr, (const int*)nullptr, 0);
            sp_sync<WAVE>();
            return R;
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
        int n_left_all = 0;
        {
            Hand w = {0, 0};
            for (int t = 0; t < 34; t++) {
                int seen = F1(pub_seen, t) + h0.get(t);  // tiles_seen = public + own hand (incl. the tile just drawn)
                int left = (4 - seen) & 7;
                for (int k = 0; k < left; k++) w.inc(t);
                n_left_all += left;
            }
            root.w = w;
            int akas_seen = (F(pub_aka_seen) | F1(akas_in_hand, p)) & 7;
            root.akas = (u32)akas_hand | ((u32)(~akas_seen & 7) << 3);
        }
        const int T = tsumos_left, n_left = n_left_all & 0xFF;
        // fewer draws left than the shanten number: tenpai / win / EV are exactly zero for every candidate (reaching tenpai
        // takes cur_shanten draws), so neither the probability table nor the state graph are needed at all
        const bool with_probs = cur_shanten <= 3 && T >= cur_shanten;
        // Three independent chains of dependent table walks / LDS reads run SIDE BY SIDE on three wavefronts of the workgroup
        // (one after the other they were 6 + 10 + 10 us of a row's set-up, with 255 threads waiting at the barriers in between):
        //   lane CTX   : the calculator's constants;
        //   lane 0     : the discards that keep the shanten number -> the candidate list;
        //   lanes SPEC : the required draws of root - d for EVERY held kind d, before it is known which d are candidates.
        constexpr int CTX = (!WAVE && NT > 128) ? 128 : 0, SPEC = (!WAVE && NT > 64) ? 64 : 0;
        if (tid == CTX) {
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
            X.T = T;
            X.n_left = n_left;
            X.n_list = 0;
            X.n_pool = 0;
            X.overflow = 0;
            X.prof = prof;
            for (int k = 0; k < 8; k++) X.pt[k] = 0;
            for (int l = 0; l < 5; l++) X.lvl_begin[l] = X.lvl_end[l] = 0;
        }
        if (with_probs) {  // build_tsumo_prob_table (calc.rs:135-146); the not_tsumo rows come from the shared table c_sp_nt
            for (int q = tid; q < 4 * SP_T; q += NT) {
                int i = q / SP_T, j = q % SP_T;
                X.tsumo_prob[i][j] = j < T ? (float)(i + 1) / (float)(n_left - j) : 0.f;
            }
        }

        // ---- candidates: analyze_discard / analyze_draw (+ *_simple for shanten > 3)  (calc.rs:205-312), from the table-id
        // sets of mj_sptab.h: the discards of the root hand that keep its shanten number, and the draws that lower the shanten
        // number of root - d (state.rs:176-200)
        const SpTabG TG = sp_tab_g(c_sp_tab);
        if (tid >= SPEC && tid < SPEC + 34) {
            const int d = tid - SPEC;
            if (can_discard) {
                if (root.h.get(d)) {
                    Hand hc = root.h;
                    hc.dec(d);
                    X.spec_req[0][d] = sp_req_of_hand(TG, c_mj_tables, hc, ld3, cur_shanten);  // d keeps the shanten number
                    if (cur_shanten > 3) X.spec_req[1][d] = sp_req_of_hand(TG, c_mj_tables, hc, ld3, cur_shanten + 1);  // d gives one up
                }
            } else if (d == 0) {
                X.spec_req[0][0] = sp_req_of_hand(TG, c_mj_tables, root.h, ld3, calc_all(c_mj_tables, root.h, ld3));
            }
        }
        if (tid == 0) {
            int n = 0;
            if (can_discard) {
                const u64 keepers = sp_keep_of_hand(TG, c_mj_tables, root.h, ld3, cur_shanten);  // calc_all(root - d) == cur_shanten
                for (int d = 0; d < 34; d++) {
                    int c = root.h.get(d);
                    if (c == 0) continue;
                    const bool keeps = (keepers >> d) & 1;
                    int dt = d;
                    if (d == T_5M && (root.akas & 1) && c == 1) dt = T_5MR;
                    else if (d == T_5P && (root.akas & 2) && c == 1) dt = T_5PR;
                    else if (d == T_5S && (root.akas & 4) && c == 1) dt = T_5SR;
                    if (cur_shanten <= 3 && !keeps) continue;
                    X.cand_tile[n] = dt;
                    X.cand_down[n] = cur_shanten > 3 && !keeps;
                    n++;
                }
            } else {
                X.cand_tile[0] = T_UNK;
                X.cand_down[0] = 0;
                n = 1;
            }
            X.n_cand = n;
        }
        sp_sync<WAVE>();
        const int n_cand = X.n_cand;
        if (tid < n_cand) {
            const u64 spec = can_discard ? X.spec_req[X.cand_down[tid]][deaka(X.cand_tile[tid])] : X.spec_req[0][0];
            const u64 req = spec & root.w.nonzero_mask();  // the discard does not change the wall
            int nreq = 0;
            for (u64 rest = req; rest; rest &= rest - 1) nreq += root.w.get(__ffsll((long long)rest) - 1);
            X.cand_req[tid] = req;
            X.cand_nreq[tid] = nreq & 0xFF;
            X.cand_slot[tid] = -1;
            X.cand_tp0[tid] = X.cand_wp0[tid] = X.cand_ev0[tid] = 0.f;
        }
        sp_sync<WAVE>();
        R.cans = cans;
        R.can_discard0 = can_discard0;
        R.can_discard = can_discard;
        R.after_riichi = after_riichi;
        R.last_tsumo = last_tsumo;
        R.cur_shanten = cur_shanten;
        R.T = T;
        R.n_cand = n_cand;
        R.ld3 = ld3;
        R.root = root;
        R.with_probs = with_probs;
        return R;
}

// Sorting of the candidates + the encoder block of obs v4 (rows 889..1011).
template <bool WAVE, int NT, class OutP>
__device__ __forceinline__ void sp_row_write(const SpNode* nodes, SpCtx& X, const SpRowInfo& R, const int tid, OutP out, float* tv_area) {
    const int n_cand = R.n_cand, cur_shanten = R.cur_shanten, T = R.T, last_tsumo = R.last_tsumo;
    const bool with_probs = R.with_probs && tv_area != nullptr;  // (the queue tail of mj_k_sp: rows without a state graph, no staging area)
    const bool can_discard0 = R.can_discard0, after_riichi = R.after_riichi;
        // ---- sort (calc.rs:181-188 / 196-199) + encode (obs_repr.rs:564-692)
        if (tid < n_cand) {  // one lane per candidate fetches its turn-0 values (side by side, not a chain of dependent loads)
            const int c = tid;
            X.order[c] = c;
            const int slot = X.cand_slot[c];
            if (with_probs && slot >= 0) {  // Candidate::from clamps (candidate.rs:46-70); shanten 0 => tenpai = 1
                const SpNode& nd = nodes[slot];
                const float tp = cur_shanten == 0 ? 1.f : nd.val[0][0];
                X.cand_tp0[c] = fminf(fmaxf(tp, 0.f), 1.f);
                X.cand_wp0[c] = fminf(fmaxf(nd.val[0][1], 0.f), 1.f);
                X.cand_ev0[c] = fmaxf(nd.val[0][2], 0.f);
            }
        }
        sp_sync<WAVE>();
        if (tid == 0) {
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
        sp_sync<WAVE>();
        {
            const int first = n_cand > 0 ? X.order[0] : -1;
            const float max_ev = (with_probs && first >= 0 && T > 0) ? X.cand_ev0[first] : 0.f;
            // ---- the masks of block rows 2 .. 71 (required tiles), one 34-bit word per output row
            u64* const rowm = &X.spec_req[0][0];  // (the set-up's speculative sets are consumed)
            static_assert(sizeof(X.spec_req) == 68 * sizeof(u64), "block rows 2 .. 69 = [keep / shanten-down][34 discards]");
            for (int i = tid; i < 68; i += NT) rowm[i] = 0ull;  // (NT = 64 for the rows a single wavefront processes)
            if (tid < 2) X.row7x[tid] = 0ull;
            if (tid < 34) X.colcand[tid] = -1;
            sp_sync<WAVE>();
            if (can_discard0 && !after_riichi) {
                if (tid < n_cand) {
                    const int dtid = deaka(X.cand_tile[tid]);
                    rowm[(X.cand_down[tid] ? 34 : 0) + dtid] = X.cand_req[tid];
                    X.colcand[dtid] = (signed char)tid;
                }
                if (tid == 0 && X.lvl_begin[4] >= 0) X.row7x[0] = 1ull << deaka(X.cand_tile[X.lvl_begin[4]]);
            } else if (can_discard0) {
                // discard after riichi: `cans.can_discard` is still true in the encoder (obs_repr.rs:580), the single
                // candidate's tile was patched to the drawn tile (agent_helper.rs:588-590)
                if (tid == 0) {
                    if (first >= 0) rowm[deaka(last_tsumo)] = X.cand_req[first];
                    X.row7x[0] = 1ull << deaka(last_tsumo);
                    if (n_cand > 0) X.colcand[deaka(last_tsumo)] = 0;
                }
            } else {
                if (tid == 0 && first >= 0) X.row7x[1] = X.cand_req[first];
            }
            // sp table (obs_repr.rs:644-692)
            const float ev_scale = max_ev < 1.f ? 0.f : 1.f / max_ev;
            bool table_ok = with_probs && first >= 0 && X.cand_tp0[first] > 0.f;
            float* tv = tv_area;  // [candidate slot * SP_T + turn][4]: tenpai, win, ev, alive
            if (table_ok) {  // uniform over the workgroup
                // one load per (candidate, turn): the clamped values go to the LDS (the evaluation scratch is free now), and
                // take_while(p > 0) on the tenpai probs becomes a count of the leading positive turns
                static_assert(SP_EVAL_LDS_FLOATS(64) >= SP_MAX_CAND * SP_T * 4, "table staging fits the evaluation scratch");
                const int n_src = can_discard0 ? n_cand : 1;
                for (int q = tid; q < n_src * SP_T; q += NT) {
                    const int c = q / SP_T, turn = q % SP_T;
                    {
                        float tpv = 0.f, wpv = 0.f, evv = 0.f;
                        if (turn < T) {
                            const SpNode& nd = nodes[X.cand_slot[can_discard0 ? c : first]];
                            tpv = cur_shanten == 0 ? 1.f : fminf(fmaxf(nd.val[turn][0], 0.f), 1.f);
                            wpv = fminf(fmaxf(nd.val[turn][1], 0.f), 1.f);
                            evv = fmaxf(nd.val[turn][2], 0.f);
                        }
                        float* dst = tv + (c * SP_T + turn) * 4;
                        dst[0] = tpv; dst[1] = wpv; dst[2] = fminf(evv * ev_scale, 1.f); dst[3] = tpv > 0.f ? 1.f : 0.f;
                    }
                }
                sp_sync<WAVE>();
                if (tid < n_src) {  // (the node slots are read: cand_slot now holds the candidates' alive counts)
                    int n_alive = 0;
                    while (n_alive < SP_T && tv[(tid * SP_T + n_alive) * 4 + 3] != 0.f) n_alive++;
                    X.cand_slot[tid] = n_alive;
                }
            }
            sp_sync<WAVE>();
            sp_block_write<WAVE, NT>(out, tid, fminf(fmaxf(max_ev, 0.f), 100000.f) / 100000.f, fminf(fmaxf(max_ev, 0.f), 30000.f) / 30000.f,
                                     (const u64*)rowm, (const u64*)X.row7x, table_ok, !can_discard0, (const signed char*)X.colcand, (const float*)tv,
                                     (const int*)X.cand_slot, T);
        }
}

// A row without a state graph (queue tail, sp_row_class == 7), processed by ONE wavefront with its own context: set-up and encoder only.
struct SpWaveArea {
    TableOne st;
    SpCtx X;
};
__device__ __noinline__ unsigned long long sp_light_row(const uint32_t* rows, const TableOne* snap, float* obs, SpWork* W, SpWaveArea* A, int row) {
    SP_ASSUME_LDS(A);
    const int lane = threadIdx.x & 63;
    SP_HBM float* out = (SP_HBM float*)obs + (size_t)row * (1012 * 34);
    const long long t0 = wall_clock64();
    const SpRowInfo R = sp_row_front<true, 64>((const SP_HBM uint32_t*)rows, (const SP_HBM TableOne*)snap, W, A->X, &A->st, lane, row, out, nullptr);
    if (R.ok) {
        if (R.with_probs) A->X.overflow = 1;  // cannot happen: the row order put a row WITH a graph into the tail of the queue
        sp_row_write<true, 64>((const SpNode*)nullptr, A->X, R, lane, out, nullptr);
    }
    // (statistics are summed by the caller and flushed once per wavefront: one global atomic per row and counter on ONE cache line --
    // ~250 k per launch at ~30 ns each, serialised in one L2 channel -- kept a kernel with every other cost removed at 7.6 ms)
    const unsigned long long ret = ((unsigned long long)(wall_clock64() - t0) & 0x7FFFFFFFFFFFFFFFull) | (R.ok && A->X.overflow ? 1ull << 63 : 0ull);
    mj_team_sync<64>();
    return ret;
}

// ---- promotion (round 6): small pools.  A decision row is pinned to ONE workgroup, and the heaviest rows (~6 k states) need ~3 ms on four
// wavefronts: with 65,536 tables a workgroup chains ~20 rows and nobody notices, with 4,096 tables the launch lasts as long as its heaviest
// row while 1,000 of the 1,024 workgroups idle.  What a row will cost is only known once its graph is growing (the best predictor from
// the table record alone, the candidates' required tiles, leaves 2 ms rows undetected), so the decision is taken THERE: when a workgroup of
// mj_k_sp finds the level it is about to expand at least SpParams::promo_min[level] states large, it parks the row -- its context goes to the
// work area's `handoff` block, the area itself (hash set, nodes, lists: everything is in HBM already) is handed over through an entry of
// the promotion queue after an agent-scope release, and the workgroup continues with a spare area and the next row.  mj_k_sp_wide
// (SP_WIDE_THREADS threads, one workgroup per CU, launched BEFORE mj_k_sp on the pool's stream while mj_k_sp runs on a second stream)
// takes the entries in order, acquires, and finishes the row -- remaining expansion levels, evaluation, row output -- with four times the
// wavefronts.  Nothing flows back.  The results do not depend on who expands which level (a state's values depend on its child list's
// order, which is the reference's, not on creation order).  Every spin is bounded: a wide workgroup that waits longer than SP_WIDE_TIMEOUT
// gives up, and a third launch on the pool's stream after both (sweep = the wide kernel again, now with every producer finished) takes
// whatever entry is left, so the result never depends on the two kernels having run side by side.
struct SpHandoff {
    SpCtx X;
    SpRowInfo R;
    int row, next_lv;
};
static_assert(sizeof(SpHandoff) <= SP_HANDOFF_WORDS * 4 && sizeof(SpHandoff) % 4 == 0, "SpWork::handoff holds an SpHandoff");
#ifndef SP_WIDE_TIMEOUT
#define SP_WIDE_TIMEOUT 5000000ll  // wall_clock64 ticks (100 MHz): 50 ms (a legitimate wait ends with mj_k_sp_promo's last row: milliseconds)
#endif
// The hand-off's two fences (MI355X_MICROARCH.md, inter-workgroup visibility: per-XCD L2s are not coherent with each other, a CU's L1 is never
// refreshed by another CU's stores).  Producer: every wavefront drains its stores, workgroup barrier, ONE lane releases at agent scope
// (buffer_wbl2 sc1) and drains again (the compiler may drop the wait behind the write-back), then the relaxed agent-scope flag store.
// Consumer: one relaxed poll, ONE agent-scope acquire (buffer_inv sc1), workgroup barrier, plain loads.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MJ_EMU)
#define SP_DRAIN_STORES() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define SP_RELEASE_AGENT() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } while (0)
#define SP_ACQUIRE_AGENT() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#define SP_SPIN_PAUSE() __builtin_amdgcn_s_sleep(32)
#else
#define SP_DRAIN_STORES() ((void)0)
#define SP_RELEASE_AGENT() ((void)0)
#define SP_ACQUIRE_AGENT() ((void)0)
#define SP_SPIN_PAUSE() ((void)0)
#endif

#ifdef SP_ROWDUMP
#define SP_DUMP(P) ((P).rowdump != nullptr)
#else
#define SP_DUMP(P) false  // (the debug records cost the kernel five VGPR spills)
#endif
template <int NT>
struct SpLds {
    union Teams {
        TableOne st;                                 // the decision's table record: read during the row set-up only
        SpChunk wchunk[NT / SP_NT];                  // expansion / level-0 probe: one chunk per wavefront
        float ev[SP_EVAL_LDS_FLOATS(NT)];            // evaluation teams (T lanes each)
        SpWaveArea wave[NT / 64];                    // queue tail: one row without a state graph per wavefront
    };
};

template <int NT, bool WIDE, bool PROMO>
__device__ __forceinline__ void sp_kernel_body(SpParams P) {
    __shared__ SpCtx X;
    __shared__ int s_row, s_k;
    __shared__ unsigned long long s_stat[25];  // this workgroup's share of SpParams::err, flushed once (see sp_light_row)
    if (threadIdx.x < 25) s_stat[threadIdx.x] = 0ull;
    __shared__ typename SpLds<NT>::Teams s_tm;
#if SP_CC_N > 0
    __shared__ unsigned long long s_cc[SP_CC_N];  // the child cache of the expansion (sp_expand_chunk)
    for (int i = threadIdx.x; i < SP_CC_N; i += NT) s_cc[i] = 0ull;
    unsigned n_graph_rows = 0;
#endif
    // mj_k_sp: workgroup b owns work area b (and moves to spare area grid + k after its k-th promotion); mj_k_sp_wide: the promoted row's area
    SpWork* W = WIDE ? P.work + P.n_narrow + P.promo_cap + blockIdx.x : P.work + blockIdx.x;
    SpWork* const W_own = W;
    const int tid_wg = threadIdx.x, tid = tid_wg;

    // the hash tags start empty (zeroed once when the work area is allocated); a row's tags carry its epoch, no row clears anything
    unsigned tag_epoch = W->epoch;  // (uniform; written back when the workgroup leaves the row loop / parks a row)
    unsigned own_epoch = tag_epoch;  // (mj_k_sp_wide: the epoch of its own area while it works in a promoted row's)
    bool main_done = false, drained = false;  // (mj_k_sp_wide, lane 0: the row queue is empty; the last look at the promotion queues)

    const long long t_wg_in = SP_DUMP(P) ? wall_clock64() : 0;
    const long long t_wg0 = P.prof ? wall_clock64() : 0;  // MJ_SP_PROF: workgroup lifetime / queue + reset time (err[19..22])
#ifndef MJ_EMU
    const long long c_wg0 = P.prof ? clock64() : 0;       // the same lifetime in shader-clock cycles (s_memtime): err[24] / err[19] x 100 MHz = the clock the kernel ran at
#endif
    long long t_pop = 0, t_reset = 0;
    // the queue is ordered by cost class (mj_k_order_*): the rows of class 7 (no state graph at all) form its tail
    const int n_heavy = P.n_rows - P.queue[1 + 7];
    bool wave_mode = false;
    for (;;) {
        const int tid = SP_OPQ(2, tid_wg);
        const long long t_a = P.prof ? wall_clock64() : 0;
        int row, lv_first = 0;
        SpRowInfo R;
        long long t_0, t_1, t_2, t_3, t_4;
        bool promoted = false;  // (mj_k_sp_wide only) this row was parked by a workgroup of mj_k_sp: continue it from its hand-off block
        if constexpr (WIDE) {
            // ---- the next row: a promoted one if there is any (the largest first), else one from the row queue like mj_k_sp -- a wide
            // workgroup never idles while rows are left; when the queue is empty it waits for promotions until every producer has gone
            if (tid == 0) {
                int got = 0;  // > 0: promoted (work area + 1); < 0: -(queue position + 1); 0: nothing left
                const int cap = P.promo_cap / SP_PROMO_LEVELS;
                long long t_idle = 0;
                for (;;) {
                    for (int q = 0; q < SP_PROMO_LEVELS && !got; q++) {
                        int* const ents = P.queue + SP_Q_PROMO_ENT + q * (SP_PROMO_CAP / SP_PROMO_LEVELS);
                        int* const head = P.queue + SP_Q_PROMO_HEAD(q) + (P.sweep ? 1 : 0);
                        for (;;) {
                            const int h = __hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const int a = min(__hip_atomic_load(P.queue + SP_Q_PROMO_ALLOC(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), cap);
                            if (h >= a) break;
                            if (atomicCAS(head, h, h + 1) != h) continue;
                            // entry h is ours; its producer publishes it right behind the allocation (a copy + one release)
                            const long long t_w = wall_clock64();
                            int e;
                            while ((e = __hip_atomic_load(ents + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 && !P.sweep &&
                                   wall_clock64() - t_w < SP_WIDE_TIMEOUT)
                                SP_SPIN_PAUSE();
                            if (e == 0 && !P.sweep) atomicAdd(&P.err[25], 1ull);  // (left to the sweep)
                            if (e > 0 && atomicCAS(ents + h, e, -1) == e) {       // (the sweep skips what the first launch took)
                                got = e;
                                break;
                            }
                        }
                    }
                    if (got || P.sweep) break;
                    if (!main_done) {
                        const int r = atomicAdd(P.queue, 1);
                        if (r < n_heavy) {
                            got = -(r + 1);
                            break;
                        }
                        main_done = true;
                    }
                    // nothing to do right now.  Every producer gone (their entries were released before they arrived at the DONE word) and
                    // still nothing => finished
                    if (__hip_atomic_load(P.queue + SP_Q_PROMO_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= P.n_narrow) {
                        if (drained) break;
                        drained = true;  // (one more look at the queues)
                        continue;
                    }
                    const long long now = wall_clock64();
                    if (t_idle == 0) t_idle = now;
                    else if (now - t_idle > SP_WIDE_TIMEOUT) {  // the two kernels do not overlap (or mj_k_sp has not started): give up, the sweep follows
                        atomicAdd(&P.err[25], 1ull);
                        break;
                    }
                    SP_SPIN_PAUSE();
                }
                s_row = got;
                if (got > 0) SP_ACQUIRE_AGENT();  // the producer's release: this CU reads the area for the first time
            }
            __syncthreads();
            if (s_row == 0) break;
            if (P.prof) t_pop += wall_clock64() - t_a;
            t_0 = wall_clock64();
            if (s_row > 0) {
                promoted = true;
                if (tid == 0) atomicAdd(&P.err[P.sweep ? 27 : 26], 1ull);  // rows finished here (by the sweep launch: the two kernels did not overlap)
                W = P.work + (s_row - 1);
                SP_HBM const u32* src = (SP_HBM const u32*)W->handoff;
                u32* dx = reinterpret_cast<u32*>(&X);
                for (int i = tid; i < (int)(sizeof(SpCtx) / 4); i += NT) dx[i] = src[i];
                const SpHandoff* H = reinterpret_cast<const SpHandoff*>(W->handoff);
                R = H->R;
                row = H->row;
                lv_first = H->next_lv;
                __syncthreads();
                tag_epoch = X.tag_epoch;
            } else {
                W = W_own;
                tag_epoch = own_epoch;
                row = (int)P.order[-s_row - 1];
            }
            t_1 = t_2 = t_3 = t_4 = t_0;
        } else {
            if (tid == 0) s_row = atomicAdd(P.queue, 1);
            __syncthreads();
            if (s_row >= n_heavy) { wave_mode = n_heavy < P.n_rows; break; }  // no graph rows left: on to the tail (its own head word)
            row = (int)P.order[s_row];
            if (P.prof) t_pop += wall_clock64() - t_a;
            t_0 = wall_clock64();
            t_1 = t_2 = t_3 = t_4 = t_0;
        }
        float* out = P.obs + (size_t)row * (1012 * 34);
        if (!promoted) {
            R = sp_row_front<false, NT>(P.rows, P.snap, W, X, &s_tm.st, tid, row, out, P.prof);
            if (!R.ok) continue;
        }
        const int cur_shanten = R.cur_shanten, n_cand = R.n_cand, T = R.T;
        const bool with_probs = R.with_probs, can_discard = R.can_discard;
        const SpState root = R.root;
        // fewer draws left than the shanten number: tenpai / win / EV are exactly zero for every candidate (reaching tenpai
        // takes cur_shanten draws), so the state graph need not be built at all
        if (!promoted) {
            t_1 = wall_clock64();
            t_2 = t_3 = t_4 = t_1;
        }
        bool parked = false;
        if (with_probs) {
#if SP_CC_N > 0
            // a new row = a new epoch of the child cache (8 bits, 1..255; when they have gone round the cache is wiped)
            n_graph_rows++;
            if ((n_graph_rows & 255u) == 0u) {
                n_graph_rows++;
                for (int i = tid; i < SP_CC_N; i += NT) s_cc[i] = 0ull;
            }
            if (tid == 0) {
                X.cc = s_cc;
                X.cc_epoch = n_graph_rows & 255u;
            }
#else
            if (tid == 0) X.cc = nullptr;
#endif
            if (promoted) {
                if (tid == 0) {
                    X.prof = P.prof;
                    for (int k = 0; k < 8; k++) X.pt[k] = 0ull;  // (the first workgroup flushed its share of the pass timers)
                }
                __syncthreads();
            } else {
                // a new row = a new epoch of the hash tags
                if (tag_epoch >= SP_EPOCH_WRAP) {  // wrapped (once in 2 M rows): wipe the table, start over
                    for (int i = tid; i < SP_CAP; i += NT) W->tag[i] = 0ull;
                    tag_epoch = 0;
                    __syncthreads();
                }
                tag_epoch++;
                if (tid == 0) X.tag_epoch = tag_epoch;
                __syncthreads();
                // root states = level cur_shanten
                if (tid < n_cand) {  // one lane per candidate: the claims (one L2 atomic round trip each) run side by side
                    const int c = tid;
                    SpState s = root;
                    if (can_discard) sp_discard(s, X.cand_tile[c]);
                    X.cand_slot[c] = sp_insert(W, &X, sp_dk_add(0ull, -1, can_discard ? X.cand_tile[c] : -1), s, -1, -1);
                }
                __syncthreads();
                if (tid == 0) {
                    X.lvl_begin[cur_shanten] = 0;
                    X.lvl_end[cur_shanten] = X.n_list;
                }
                __syncthreads();
            }
            // expand top-down
            for (int lv = promoted ? lv_first : cur_shanten; lv >= 1; lv--) {
                const int b = X.lvl_begin[lv], e = X.lvl_end[lv];
                if constexpr (PROMO) {  // (mj_k_sp_promo only: compiled into mj_k_sp the parking code costs the full-size pool 2.7 % -- register allocation)
                    // ---- promotion: this level is large (the ones below it will be larger) and a wide workgroup may take the row from here
                    if (P.promo_cap > 0 && lv < cur_shanten && e - b >= P.promo_min[lv] && !X.overflow) {
                        const int pq = e - b >= 2 * P.promo_min[lv] ? 0 : 1;  // the largest rows have their own queue: wide workgroups take them first
                        if (tid == 0) s_k = atomicAdd(P.queue + SP_Q_PROMO_ALLOC(pq), 1);
                        __syncthreads();
                        const int k = s_k;
                        if (k < P.promo_cap / SP_PROMO_LEVELS) {
                            {
                                const u32* sx = reinterpret_cast<const u32*>(&X);
                                SP_HBM u32* dst = (SP_HBM u32*)W->handoff;
                                for (int i = tid; i < (int)(sizeof(SpCtx) / 4); i += NT) dst[i] = sx[i];
                                if (tid == 0) {
                                    SpHandoff* H = reinterpret_cast<SpHandoff*>(W->handoff);
                                    H->R = R;
                                    H->row = row;
                                    H->next_lv = lv;
                                    W->epoch = tag_epoch;
                                    if (SP_DUMP(P)) P.rowdump[(size_t)row * 12 + 8] = (uint32_t)t_0, P.rowdump[(size_t)row * 12 + 9] = (uint32_t)wall_clock64();
                                }
                            }
                            // every wavefront's stores of this row (nodes, keys, lists, pools ...) are in L2 before lane 0 releases them
                            SP_DRAIN_STORES();
                            __syncthreads();
                            if (tid == 0) {
                                SP_RELEASE_AGENT();
                                __hip_atomic_store(P.queue + SP_Q_PROMO_ENT + pq * (SP_PROMO_CAP / SP_PROMO_LEVELS) + k, (int)(W - P.work) + 1, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
                            }
                            W = P.work + gridDim.x + pq * (P.promo_cap / SP_PROMO_LEVELS) + k;  // the spare area of this promotion
                            tag_epoch = W->epoch;
                            parked = true;
                            break;
                        }
                    }
                }
                // every wavefront its own chunks; a level too small for four full chunks is split four ways (a chunk pass costs
                // the same for 4 states as for 16, so idle wavefronts are the only thing to lose)
                const int ns = min(SP_NS, max(1, (e - b + NT / SP_NT - 1) / (NT / SP_NT)));
                for (int c0 = b + ns * (tid / SP_NT); c0 < e; c0 += ns * (NT / SP_NT))
                    sp_expand_chunk(W, &X, &s_tm.wchunk[tid / SP_NT], c0, min(ns, e - c0), lv);
                __syncthreads();
                if (tid == 0) {
                    X.lvl_begin[lv - 1] = e;
                    X.lvl_end[lv - 1] = min(X.n_list, SP_CAP);
                }
                __syncthreads();
            }
            t_2 = wall_clock64();
            // evaluate bottom-up
            if (!parked)
            for (int lv = 0; lv <= cur_shanten; lv++) {
                const int b = X.lvl_begin[lv], e = X.lvl_end[lv];
                if (lv == 0) {
                    if (tid == 0) X.n_items = 0;
                    __syncthreads();
                    const int ns = min(SP_NS, max(1, (e - b + NT / SP_NT - 1) / (NT / SP_NT)));
                    for (int c0 = b + ns * (tid / SP_NT); c0 < e; c0 += ns * (NT / SP_NT))
                        sp_l0_probe_chunk(W, &X, &s_tm.wchunk[tid / SP_NT], c0, min(ns, e - c0));
                    __syncthreads();
                    const long long t_2a = wall_clock64();
                    const int n_items = min(X.n_items, SP_ITEMS);
                    if ((tid & ~63) < n_items) sp_l0_score_all(c_mj_tables, W, &X, n_items, tid, NT);  // (a wavefront without items: no call)
                    __syncthreads();
                    if (P.prof && tid == 0) {  // level-0 sub-phases: probe, scoring (the sum is the rest of the level-0 timer)
                        X.pt[7] += (unsigned long long)(t_2a - t_2);
                        s_stat[18] += (unsigned long long)(wall_clock64() - t_2a);
                    }
                }
                sp_sort_level<NT>(W, reinterpret_cast<int*>(s_tm.ev), b, e);
                {
                    // teams of exactly T lanes, floor(64 / T) per wavefront (the leftover lanes of a wavefront idle)
                    // ... of the turns that can be reached at this level: the first `off` turns are dead
                    const int off = min(cur_shanten - lv, T - 1), TW = T - off;
                    const int wl = tid & 63, tw = wl / TW, ln = wl - tw * TW;
                    float* wl_lds = s_tm.ev + (tid >> 6) * SP_EVW_WAVE_FLOATS;
                    const long long t_ev0 = P.prof ? wall_clock64() : 0;
                    if (lv == 0) {
                        const int tpw0 = min(64 / TW, SP_EVW_WAVE_FLOATS / (SP_EV_ENT * SP_EV0_STRIDE));
                        const int team0 = (tid >> 6) * tpw0 + tw, n_teams0 = (NT / 64) * tpw0;
                        const bool on0 = tw < tpw0;
                        if (b + (tid >> 6) * tpw0 >= e) {}  // no state for any team of this wavefront: no call (14-20 callee-saved VGPRs saved / restored each)
                        else if (T <= 8) sp_eval_wave0<8>(W, &X, wl_lds, b + team0, e, n_teams0, ln, off, tw, on0);
                        else if (T <= 16) sp_eval_wave0<16>(W, &X, wl_lds, b + team0, e, n_teams0, ln, off, tw, on0);
                        else sp_eval_wave0<17>(W, &X, wl_lds, b + team0, e, n_teams0, ln, off, tw, on0);
                    } else {
                        // levels > 0: the whole wavefront in lock-step (sp_eval_wave), its own team geometry (LDS per team differs)
                        const int tpw2 = min(64 / TW, SP_EVW_WAVE_FLOATS / sp_evw_team_floats(T));
                        const int team2 = (tid >> 6) * tpw2 + tw, n_teams2 = (NT / 64) * tpw2;
                        const bool on = tw < tpw2;
                        if (b + (tid >> 6) * tpw2 >= e) {}  // (see level 0)
                        else if (T <= 8) {
                            if (lv == 1) sp_eval_wave<8, 1>(W, &X, wl_lds, b + team2, e, n_teams2, ln, off, tw, on);
                            else sp_eval_wave<8, 2>(W, &X, wl_lds, b + team2, e, n_teams2, ln, off, tw, on);
                        } else if (T <= 16) {
                            if (lv == 1) sp_eval_wave<16, 1>(W, &X, wl_lds, b + team2, e, n_teams2, ln, off, tw, on);
                            else sp_eval_wave<16, 2>(W, &X, wl_lds, b + team2, e, n_teams2, ln, off, tw, on);
                        } else {
                            if (lv == 1) sp_eval_wave<17, 1>(W, &X, wl_lds, b + team2, e, n_teams2, ln, off, tw, on);
                            else sp_eval_wave<17, 2>(W, &X, wl_lds, b + team2, e, n_teams2, ln, off, tw, on);
                        }
                    }
                    if (P.prof && (tid & 63) == 0) atomicAdd(&s_stat[23], (unsigned long long)(wall_clock64() - t_ev0));  // wavefront time inside the evaluation (all levels)
                }
                __syncthreads();
                if (lv == 0) t_3 = wall_clock64();
            }
            t_4 = wall_clock64();
        }

        if (!parked) sp_row_write<false, NT>(W->node, X, R, tid, out, s_tm.ev);
        __syncthreads();
        if (tid == 0) {
            long long t_5 = wall_clock64();
            if (!parked) s_stat[1] += 1ull;
            s_stat[2] += (unsigned long long)(t_1 - t_0);
            if (with_probs) {
                s_stat[3] += (unsigned long long)(t_2 - t_1);
                if (!parked) {
                    s_stat[4] += (unsigned long long)(t_3 - t_2);
                    s_stat[5] += (unsigned long long)(t_4 - t_3);
                    s_stat[6] += (unsigned long long)(t_5 - t_4);
                    s_stat[7] += (unsigned long long)X.n_list;
                    s_stat[15] += (unsigned long long)X.n_pool;   // child-list entries (edges of the state graph)
                    s_stat[16] += (unsigned long long)X.n_items;  // level-0 draw entries scored
                }
                if (P.prof)
                    for (int k = 0; k < 7; k++) s_stat[8 + k] += X.pt[k];  // expansion pass timers (MJ_SP_PROF)
                if (P.prof) s_stat[17] += X.pt[7];                          // level-0 probe
            }
            if (SP_DUMP(P) && !parked) {  // (debug) queue position | wide << 31, shanten | T << 8 | n_cand << 16 | can_discard << 24, sum of the candidates' required kinds, states, edges, l0 items, ticks, level sizes
                uint32_t* d = P.rowdump + (size_t)row * 12;
                int nreq = 0;
                for (int c = 0; c < n_cand; c++) nreq += X.cand_nreq[c];
                d[0] = (uint32_t)(WIDE ? 0x40000000 | (promoted ? 0x20000000 : 0) : s_row); d[1] = (uint32_t)(cur_shanten | T << 8 | n_cand << 16 | (int)can_discard << 24); d[2] = (uint32_t)nreq;
                d[3] = with_probs ? (uint32_t)X.n_list : 0u; d[4] = with_probs ? (uint32_t)X.n_pool : 0u; d[5] = with_probs ? (uint32_t)X.n_items : 0u;
                d[6] = (uint32_t)(t_5 - t_0);
                if (!promoted) d[8] = (uint32_t)t_0, d[9] = 0u;
                d[10] = (uint32_t)t_0; d[11] = (uint32_t)t_5;
                d[7] = 0x80000000u | (with_probs ? (uint32_t)min(X.lvl_end[1] - X.lvl_begin[1], 32767) | (uint32_t)min(X.lvl_end[2] - X.lvl_begin[2], 32767) << 15 : 0u);
            }
        }
        if constexpr (WIDE) {
            if (!promoted) own_epoch = tag_epoch;
        }
        // ---- the hash set needs no reset (tag epochs); a row that overflowed it is counted
        const long long t_r = P.prof ? wall_clock64() : 0;
        if (X.overflow && tid == 0 && !parked) s_stat[0] += 1ull;
        __syncthreads();
        if (P.prof) t_reset += wall_clock64() - t_r;
    }
    if constexpr (WIDE) {
        if (tid == 0) W_own->epoch = own_epoch;
    } else {
        if (tid == 0) {
            W->epoch = tag_epoch;
            if (PROMO && P.promo_cap > 0) {  // arrival at the DONE word, behind this workgroup's entry stores (agent-scope stores, complete once vmcnt says so:
                                    // the areas they point to were released when they were parked -- no second write-back of the XCD's L2 here)
                SP_DRAIN_STORES();
                __hip_atomic_fetch_add(P.queue + SP_Q_PROMO_DONE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (P.prof && tid == 0) {
        const unsigned long long life = (unsigned long long)(wall_clock64() - t_wg0);
        s_stat[19] += life;
#ifndef MJ_EMU
        s_stat[24] += (unsigned long long)(clock64() - c_wg0);
#endif
        atomicMax(&P.err[20], life);
        s_stat[21] += (unsigned long long)t_pop;
        s_stat[22] += (unsigned long long)t_reset;
    }
    __syncthreads();  // (every thread leaves the row loop at the same pop)
    if (SP_DUMP(P) && tid == 0) {  // (debug) first workgroup start / last end of the graph-row loops, both kernels
        atomicMin(&P.err[28], (unsigned long long)t_wg_in);
        atomicMax(&P.err[WIDE ? 30 : 29], (unsigned long long)wall_clock64());
    }
    if (tid < 25 && tid != 20 && s_stat[tid]) atomicAdd(&P.err[tid], s_stat[tid]);  // the workgroup's statistics, once
    // ---- the tail of the queue: every wavefront takes its own rows (set-up + encoder only, no workgroup barrier any more)
    // The tail has its own head word (another 128-byte line than the heavy rows' head) and is popped SP_TAIL_BATCH rows at a time:
    // ~46 k light rows per launch against 4,096 wavefronts that need ~10 us per row ask for ~400 pops per microsecond, and one word
    // serves ~88 (MI355X_MICROARCH.md, dequeue row) -- one row per atomic made the tail dequeue-bound.
    if constexpr (!WIDE) {
    if (wave_mode) {
        const int wv = tid >> 6, lane = tid & 63;
        unsigned long long w_rows = 0, w_ticks = 0, w_over = 0;
        for (;;) {
            int q = 0;
            if (lane == 0) q = n_heavy + atomicAdd(P.queue + SP_Q_TAIL, SP_TAIL_BATCH);
            q = __shfl(q, 0);
            if (q >= P.n_rows) break;
            const int qe = min(q + SP_TAIL_BATCH, P.n_rows);
            for (; q < qe; q++) {
                const unsigned long long r = sp_light_row(P.rows, P.snap, P.obs, W, &s_tm.wave[wv], (int)P.order[q]);
                w_rows++;
                w_ticks += r & 0x7FFFFFFFFFFFFFFFull;
                w_over += r >> 63;
            }
        }
        if (lane == 0 && w_rows) {  // the wavefront's statistics, once
            atomicAdd(&P.err[1], w_rows);
            atomicAdd(&P.err[2], w_ticks);
            if (w_over) atomicAdd(&P.err[0], w_over);
        }
        if (SP_DUMP(P) && lane == 0) atomicMax(&P.err[31], (unsigned long long)wall_clock64());  // (debug) end of the tail
    }
    }
}

__global__ __launch_bounds__(SP_THREADS, SP_WPS) void mj_k_sp(SpParams P) { sp_kernel_body<SP_THREADS, false, false>(P); }
// the small-pool pair: mj_k_sp with the parking code, and the wide kernel that finishes parked rows / takes rows of the queue itself
__global__ __launch_bounds__(SP_THREADS, SP_WPS) void mj_k_sp_promo(SpParams P) { sp_kernel_body<SP_THREADS, false, true>(P); }
__global__ __launch_bounds__(SP_WIDE_THREADS, SP_WPS) void mj_k_sp_wide(SpParams P) { sp_kernel_body<SP_WIDE_THREADS, true, false>(P); }
