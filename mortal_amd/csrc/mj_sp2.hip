// Single-player tables of obs v4 (rows 889..1011), round 4: the PER-PHASE pipeline.
//
// mj_sp.hip (rounds 1-3) ran one decision row per persistent workgroup, level by level, inside ONE kernel: every level ended in a
// workgroup barrier (19 % of the wavefront time), a level of 7 root states kept one wavefront of four busy, the heaviest phase set the
// register budget (128 VGPRs = 4 wavefronts per SIMD) of all of them, and every thread-per-task pass ran with the lanes its ROW
// happened to fill (VALU lane utilisation 0.60).  Here the state graphs of ALL rows of a cycle live in one hash set / node array in
// HBM and every phase is its own launch over all rows at once:
//
//   mj_k_spg_setup    one wavefront per decision row: table record, preconditions, calculator constants, candidates (sp_row_front of
//                     mj_sp.hip); rows without a state graph are finished here; the others leave their context in HBM and their root
//                     states in the level lists
//   mj_k_spg_expand   level L = 3, 2, 1: one wavefront per chunk of 16 states (ANY rows): required draws, keeping discards, children
//                     -> hash set (exact id: row | draw multiset | discard multiset), child lists, level L - 1 list
//   mj_k_spg_probe    level 0: winning draws -> scoring items
//   mj_k_spg_score    one THREAD per scoring item, dense over all rows (agari / yaku / fu / points)
//   mj_k_spg_eval     level 0 .. 3: blocks of 64 states with the same number of draws left T and depth `off`, teams of T - off lanes
//                     (sp_eval_wave / the level-0 variant), bottom-up
//   mj_k_spg_write    one wavefront per row with a graph: candidate order + the 123 obs rows (sp_row_write)
//
// A level's states are produced into a RAW list (a wavefront appends to its own 64-entry block: one atomic per block, any mix of
// rows) together with a histogram over (T, off); mj_k_spg_regroup then moves them into the level LIST, grouped by (T, off) in
// 64-entry BLOCKS that are homogeneous in (T, off) and full except the last of each group — the consumer takes blocks (evaluation)
// or quarter blocks (expansion chunks of 16) from a counter.  The arithmetic, the child-list order and every f32 operation order are
// those of mj_sp.hip (bit-identical results; the order of STATES inside a level never mattered).
#include <hip/hip_runtime.h>

#define SPG_BLK 64                         // states per level-list block
#define SPG_SLOT_BITS 26
#define SPG_SLOT_MASK ((1u << SPG_SLOT_BITS) - 1u)
#define SPG_ID_MASK ((1ull << 42) - 1ull)  // exact state id inside a row (mj_sp.hip: sp_dk_add)
#define SPG_POOL_BLK 4096u                 // child-list entries a wavefront reserves at a time (>= one sub-batch: 64 items x 26)
#define SPG_MAX_PROBE 4096                 // linear probes before a hash insert gives up (overflow)
// state descriptor (level lists): slot | row-with-graph index << 26 | wall size << 43 | len_div3 << 50 | flags << 53 | T << 56 | off << 61
// (never 0: T >= 1)
#define SPG_D_SLOT(d) ((u32)(d) & SPG_SLOT_MASK)
#define SPG_D_ROWG(d) ((u32)((d) >> 26) & 0x1FFFFu)
#define SPG_D_NLEFT(d) ((int)((d) >> 43) & 127)
#define SPG_D_LD3(d) ((int)((d) >> 50) & 7)
#define SPG_D_FLAGS(d) ((int)((d) >> 53) & 7)  // 1 assume riichi, 2 double riichi possible, 4 haitei
#define SPG_D_T(d) ((int)((d) >> 56) & 31)
#define SPG_D_OFF(d) ((int)((d) >> 61) & 3)
#define SPG_D_KEY(d) ((int)((d) >> 56) & 127)  // T | off << 5: the group a state belongs to
#define SPG_D_UPPER(d) ((d) & ~(u64)SPG_SLOT_MASK)
#define SPG_D_CHILD_UPPER(d) (SPG_D_UPPER(d) + (SPG_D_OFF(d) < 3 ? (1ull << 61) : 0ull))  // the same row and T, one level deeper
// block info: T | off << 5 | count << 8
#define SPG_B_T(b) ((int)((b) & 31u))
#define SPG_B_OFF(b) ((int)(((b) >> 5) & 3u))
#define SPG_B_COUNT(b) ((int)(((b) >> 8) & 127u))
// child-list entry: slot | discard order key << 26 | last-of-its-draw-entry << 35 | draw count << 36 | invalid << 39
#define SPG_E_SLOT(e) ((u32)(e) & SPG_SLOT_MASK)
#define SPG_E_KEY(e) ((int)((e) >> 26) & 511)
#define SPG_E_LAST (1ull << 35)
#define SPG_E_COUNT(e) ((int)((e) >> 36) & 7)
#define SPG_E_INVALID (1ull << 39)
// scoring item: slot | draw entry << 26 | tile << 31 | red variant << 37 | row-with-graph index << 38
#define SPG_I_SLOT(i) ((u32)(i) & SPG_SLOT_MASK)
#define SPG_I_IDX(i) ((int)((i) >> 26) & 31)
#define SPG_I_TILE(i) ((int)((i) >> 31) & 63)
#define SPG_I_VARIANT(i) ((int)((i) >> 37) & 1)
#define SPG_I_ROWG(i) ((u32)((i) >> 38) & 0x1FFFFu)

// control words (u32, zeroed before every launch sequence)
enum { SPG_C_NBLK = 0 /* [4] blocks per level list */, SPG_C_NPOOL = 4, SPG_C_NITEMS = 5, SPG_C_NROWG = 6, SPG_C_OVERFLOW = 7,
       SPG_C_CUR_SETUP = 8, SPG_C_CUR_EXPAND = 9 /* [4] */, SPG_C_CUR_PROBE = 13, SPG_C_CUR_EVAL = 14 /* [4] */, SPG_C_CUR_WRITE = 18,
       SPG_C_NRAW = 20 /* [4] blocks per raw list */, SPG_C_CUR_REGROUP = 24 /* [4] */,
       SPG_C_HIST = 64 /* [4][128] states per (level, T | off << 5) */, SPG_C_GCUR = 64 + 512 /* [4][128] regroup cursors */,
       SPG_C_N = 64 + 1024 };

struct SpGRow {  // what the row writer needs besides the calculator context
    SpRowInfo R;
    int row;
};
struct SpG {  // the global work area (device pointers), one per pool
    u64* tag;          // [cap] 0 = empty, else 1 << 63 | row-with-graph << 42 | state id
    SpNode* node;      // [cap] indexed by hash slot
    u64* raw[4];       // [lst_cap] per level: state descriptors as produced (blocks of SPG_BLK per producing wavefront, 0 = unused)
    u64* lst[4];       // [lst_cap] per level: the same descriptors grouped by (T, off) in blocks of SPG_BLK (mj_k_spg_regroup)
    u32* blk[4];       // [lst_cap / SPG_BLK] block info
    u64* pool;         // [pool_cap] child lists
    u64* items;        // [items_cap] level-0 scoring items
    SpCtx* ctx;        // [ctx_cap] calculator context of the rows with a state graph
    SpGRow* rinfo;     // [ctx_cap]
    u32* ctl;          // [SPG_C_N]
    u32 cap_mask, lst_cap, pool_cap, items_cap, ctx_cap;
};
struct SpGParams {
    SpG g;
    const TableOne* snap;
    const uint32_t* rows;
    int n_rows;
    float* obs;                // [n_rows][1012][34]; rows 889.. are zero on entry (written by mj_k_encode<4>)
    const uint32_t* order;     // [n_rows] setup order: rows grouped by (shanten, draws left), rows without a graph last
    unsigned long long* err;   // the counters of mj_sp.hip: [0] overflows, [1] rows, [2..6] phase wavefront ticks, [7] states
    int level;                 // expand / eval: the level of this launch
};

__constant__ const float* c_spg_tp;  // [124][4]: (c + 1) / x as f32 (build_tsumo_prob_table, calc.rs:135-146: x = n_left - turn)
static inline void spg_tp_build(float* out /* [124 * 4] */) {
    for (int x = 0; x < 124; x++)
        for (int c = 0; c < 4; c++) out[x * 4 + c] = x > 0 ? (float)(c + 1) / (float)x : 0.f;
}

MJD u32 spg_pos(u64 key /* rowg << 42 | id */, u32 mask) {
    u32 h = (u32)key * 0x9E3779B1u ^ ((u32)(key >> 32) + 0x7F4A7C15u) * 0x85EBCA77u;
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 13;
    return h & mask;
}
#define SPG_TAG(rowg, dk) (((u64)(rowg) << 42) | (dk) | (1ull << 63))
// Workgroup b runs on XCD b % 8 (observed dispatch order; each XCD has its own L2): the virtual workgroup index gives every XCD a
// CONTIGUOUS range, so that neighbouring chunks / blocks of a level list — states of the same rows, sharing their children — are
// worked on by the same XCD at about the same time and meet in its L2.  A pure speed choice: any mapping is correct.
MJD u32 spg_vwg() {
    const u32 n = gridDim.x, b = blockIdx.x;
    return (n % 8u) ? b : (b % 8u) * (n / 8u) + b / 8u;
}
MJD u32 spg_wave_first(u32 v) {  // lane 0's value in every lane
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MJ_EMU)
    return (u32)__builtin_amdgcn_readfirstlane((int)v);
#else
    return __shfl(v, 0);
#endif
}

// A wavefront's open block of one level's RAW list: appended to in lock-step by the whole wavefront (uniform control flow).
// (A same-address global atomic costs ~20 ns chip-wide — measured: 344 k chunk pops took 8 ms — so nothing on these paths may
// touch one counter per chunk: a wavefront reserves SPG_RAW_RES entries per atomic, takes its chunks / blocks by a static stride,
// and adds its state count to the statistics once, when it exits.)
#define SPG_RAW_RES 256  // raw-list entries a wavefront reserves per atomic (4 blocks)
struct SpGOut {
    int blk, fill;   // first entry of the reservation / SPG_BLK (-1 = none), entries used (0 .. SPG_RAW_RES)
    int hkey, hcnt;  // states appended since the last histogram flush and their group
    int total;       // states appended by this wavefront (statistics: err[7], added once at exit)
};
#define SPG_OUT_INIT SpGOut{-1, 0, 0, 0, 0}
__device__ __forceinline__ void spg_out_flush_hist(const SpG& G, int L, SpGOut& o, unsigned long long* err) {
    if (o.hcnt > 0 && (threadIdx.x & 63) == 0)
        __hip_atomic_fetch_add(&((SP_HBM u32*)G.ctl)[SPG_C_HIST + L * 128 + o.hkey], (u32)o.hcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    o.total += o.hcnt;
    o.hcnt = 0;
}
__device__ __forceinline__ void spg_out_close(const SpG& G, int L, SpGOut& o, unsigned long long* err) {
    if (o.blk >= 0) {  // the unused tail of the reservation reads as "no state"
        const int lane = threadIdx.x & 63;
        for (int i = o.fill + lane; i < SPG_RAW_RES; i += 64) ((SP_HBM u64*)G.raw[L])[(size_t)o.blk * SPG_BLK + i] = 0ull;
    }
    spg_out_flush_hist(G, L, o, err);
    o.blk = -1;
    o.fill = 0;
}
// once per wavefront, when it leaves its kernel: the states it produced (err[7])
__device__ __forceinline__ void spg_out_stats(SpGOut& o, unsigned long long* err) {
    if (o.total > 0 && (threadIdx.x & 63) == 0)
        __hip_atomic_fetch_add(&((SP_HBM unsigned long long*)err)[7], (unsigned long long)o.total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    o.total = 0;
}
__device__ __forceinline__ void spg_out_open(const SpG& G, int L, SpGOut& o) {
    u32 b = 0;
    if ((threadIdx.x & 63) == 0)
        b = __hip_atomic_fetch_add(&((SP_HBM u32*)G.ctl)[SPG_C_NRAW + L], (u32)(SPG_RAW_RES / SPG_BLK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    b = spg_wave_first(b);
    if (b + SPG_RAW_RES / SPG_BLK > G.lst_cap / SPG_BLK) {  // raw list full: the states are lost (counted as an overflow), keep writing into the last piece
        if ((threadIdx.x & 63) == 0) ((SP_HBM u32*)G.ctl)[SPG_C_OVERFLOW] = 1u;
        b = G.lst_cap / SPG_BLK - SPG_RAW_RES / SPG_BLK;
    }
    o.blk = (int)b;
    o.fill = 0;
}
// every lane with `pred` appends its descriptor to level L's raw list; `key` (T | off << 5) is uniform over the wavefront
__device__ __forceinline__ void spg_append(const SpG& G, int L, SpGOut& o, int key, bool pred, u64 desc, unsigned long long* err) {
    const unsigned long long m = __ballot(pred);
    const int nf = __popcll(m);
    if (nf == 0) return;
    const int lane = threadIdx.x & 63;
    const int rank = __popcll(m & ((1ull << lane) - 1ull));
    if (o.hcnt > 0 && o.hkey != key) spg_out_flush_hist(G, L, o, err);
    o.hkey = key;
    o.hcnt += nf;
    int done = 0;
    while (done < nf) {
        if (o.blk < 0 || o.fill == SPG_RAW_RES) {
            o.blk = -1;  // a full reservation needs no tail
            spg_out_open(G, L, o);
        }
        const int take = min(SPG_RAW_RES - o.fill, nf - done);
        if (pred && rank >= done && rank < done + take) ((SP_HBM u64*)G.raw[L])[(size_t)o.blk * SPG_BLK + o.fill + rank - done] = desc;
        o.fill += take;
        done += take;
    }
}

// ---------------------------------------------------------------------------------------------------------------- regroup
// raw list of level L -> level list grouped by (T, off): group g owns the entries [beg[g], beg[g] + hist[g]) with beg[g] a multiple
// of SPG_BLK, so every block is homogeneous and all but the last block of a group are full.  A workgroup takes tiles of 1,024 raw
// entries, ranks them per group in LDS and reserves each group's piece with ONE global atomic per (tile, group present): a tile is 16
// producer blocks, i.e. a handful of groups.
__global__ __launch_bounds__(256) void mj_k_spg_regroup(SpGParams P) {
    __shared__ u32 s_beg[128], s_cnt[128], s_base[128];
    const SpG G = P.g;
    const int L = P.level, tid = threadIdx.x;
    SP_HBM u32* const ctl = (SP_HBM u32*)G.ctl;
    if (tid < 128) s_cnt[tid] = ctl[SPG_C_HIST + L * 128 + tid];
    __syncthreads();
    if (tid == 0) {
        u32 at = 0;
        for (int g = 0; g < 128; g++) {
            s_beg[g] = at;
            at += (s_cnt[g] + SPG_BLK - 1) / SPG_BLK * SPG_BLK;
        }
        if (at > G.lst_cap) {  // cannot happen while the raw list itself fits (it has at least as many entries)
            ctl[SPG_C_OVERFLOW] = 1u;
        }
        if (blockIdx.x == 0) ctl[SPG_C_NBLK + L] = min(at, G.lst_cap) / SPG_BLK;
    }
    __syncthreads();
    // block info: T | off << 5 | count << 8
    for (int g = 0; g < 128; g++) {
        const u32 nb = (s_cnt[g] + SPG_BLK - 1) / SPG_BLK;
        for (u32 b = blockIdx.x * 256u + tid; b < nb; b += gridDim.x * 256u) {
            const u32 at = s_beg[g] / SPG_BLK + b;
            if (at < G.lst_cap / SPG_BLK) ((SP_HBM u32*)G.blk[L])[at] = (u32)g | (min(s_cnt[g] - b * SPG_BLK, (u32)SPG_BLK) << 8);
        }
    }
    __syncthreads();  // s_cnt is reused as the tile's histogram
    const u32 n_raw = min(ctl[SPG_C_NRAW + L], G.lst_cap / SPG_BLK) * SPG_BLK;
    for (u32 tile = blockIdx.x * 1024u; tile < n_raw; tile += gridDim.x * 1024u) {
        if (tid < 128) s_cnt[tid] = 0;
        __syncthreads();
        u64 d[4];
        u32 rk[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const u32 i = tile + q * 256u + tid;
            d[q] = i < n_raw ? ((SP_HBM u64*)G.raw[L])[i] : 0ull;
            rk[q] = d[q] ? atomicAdd(&s_cnt[SPG_D_KEY(d[q])], 1u) : 0u;
        }
        __syncthreads();
        if (tid < 128 && s_cnt[tid]) s_base[tid] = __hip_atomic_fetch_add(&ctl[SPG_C_GCUR + L * 128 + tid], s_cnt[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (!d[q]) continue;
            const int g = SPG_D_KEY(d[q]);
            const u32 at = s_beg[g] + s_base[g] + rk[q];
            if (at < G.lst_cap) ((SP_HBM u64*)G.lst[L])[at] = d[q];
        }
        __syncthreads();
    }
}

// hash-set insert of (row, id): the slot, `fresh` when this call created it (the caller then writes the node's key), -1 on overflow
__device__ __forceinline__ int spg_insert(const SpG& G, u32 rowg, u64 dk, bool& fresh) {
    const u64 tag = SPG_TAG(rowg, dk);
    u32 pos = spg_pos(tag & ~(1ull << 63), G.cap_mask);
    fresh = false;
    for (int probe = 0; probe < SPG_MAX_PROBE; probe++) {
        const u64 old = sp_claim_tag(&((SP_HBM u64*)G.tag)[pos], tag);
        if (old == 0ull) { fresh = true; return (int)pos; }
        if (old == tag) return (int)pos;
        pos = (pos + 1) & G.cap_mask;
    }
    ((SP_HBM u32*)G.ctl)[SPG_C_OVERFLOW] = 1u;
    return -1;
}

// ---------------------------------------------------------------------------------------------------------------- row order
// Rows with the same (shanten, draws left) are neighbours in the setup order, so that a wavefront's consecutive rows append their
// roots to the same open block; rows without a graph go last.  An estimate from the table's bookkeeping (like sp_row_class): the
// block key itself comes from the calculator's own numbers.
#define SPG_N_CLASS 80
MJD int spg_row_class(const TableOne* snap, uint32_t desc) {
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
    if (tiles_left < 4 || sh > 3 || sh < 0 || tsumos_left < max(sh, 1)) return SPG_N_CLASS - 1;  // no state graph
    return (3 - sh) * 18 + min(tsumos_left, 17);  // deepest graphs first
}
__global__ __launch_bounds__(256) void mj_k_spg_classify(const TableOne* snap, const uint32_t* rows, int n, uint8_t* cls, int* cnt) {
    __shared__ int h[SPG_N_CLASS];
    if (threadIdx.x < SPG_N_CLASS) h[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const int c = spg_row_class(snap, rows[i]);
        cls[i] = (uint8_t)c;
        atomicAdd(&h[c], 1);
    }
    __syncthreads();
    if (threadIdx.x < SPG_N_CLASS && h[threadIdx.x]) atomicAdd(&cnt[threadIdx.x], h[threadIdx.x]);
}
__global__ __launch_bounds__(256) void mj_k_spg_scatter(const uint8_t* cls, int n, const int* cnt, int* cursor, uint32_t* order) {
    __shared__ int h[SPG_N_CLASS], base[SPG_N_CLASS];
    if (threadIdx.x < SPG_N_CLASS) h[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    int c = 0, r = 0;
    if (i < n) {
        c = cls[i];
        r = atomicAdd(&h[c], 1);
    }
    __syncthreads();
    if (threadIdx.x < SPG_N_CLASS) {
        int b = 0;
        for (int k = 0; k < (int)threadIdx.x; k++) b += cnt[k];
        base[threadIdx.x] = b + (h[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], h[threadIdx.x]) : 0);
    }
    __syncthreads();
    if (i < n) order[base[c] + r] = (uint32_t)i;
}

// ---------------------------------------------------------------------------------------------------------------- setup
#define SPG_SETUP_RUN 8  // consecutive rows of the setup order a wavefront takes per queue pop
__global__ __launch_bounds__(256) void mj_k_spg_setup(SpGParams P) {
    __shared__ SpWaveArea s_area[4];
    const SpG G = P.g;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    SpWaveArea* A = &s_area[wv];
    SpGOut out = SPG_OUT_INIT;  // the wavefront's open raw block of root states and its level (consecutive rows mostly share it)
    int out_level = 0, n_rows_done = 0;
    const long long t0 = wall_clock64();
    for (;;) {
        u32 q0 = 0;
        if (lane == 0) q0 = __hip_atomic_fetch_add(&((SP_HBM u32*)G.ctl)[SPG_C_CUR_SETUP], (u32)SPG_SETUP_RUN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        q0 = spg_wave_first(q0);
        if ((int)q0 >= P.n_rows) break;
        const int q1 = min((int)q0 + SPG_SETUP_RUN, P.n_rows);
        for (int q = (int)q0; q < q1; q++) {
            const int row = (int)P.order[q];
            SP_HBM float* outp = (SP_HBM float*)P.obs + (size_t)row * (1012 * 34);
            const SpRowInfo R = sp_row_front<true, 64>((const SP_HBM uint32_t*)P.rows, (const SP_HBM TableOne*)P.snap, nullptr, A->X, &A->st, lane, row,
                                                       outp, nullptr);
            n_rows_done++;
            if (!R.ok) continue;
            if (!R.with_probs) {  // no state graph: the candidates' required tiles only
                sp_row_write<true, 64>((const SpNode*)nullptr, A->X, R, lane, outp, (float*)nullptr);
                mj_team_sync<64>();
                continue;
            }
            // ---- a row with a state graph: its index, its root states (level = shanten number, depth 0), its context
            u32 rowg = 0;
            if (lane == 0) rowg = __hip_atomic_fetch_add(&((SP_HBM u32*)G.ctl)[SPG_C_NROWG], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            rowg = spg_wave_first(rowg);
            if (rowg >= G.ctx_cap) {
                if (lane == 0) ((SP_HBM u32*)G.ctl)[SPG_C_OVERFLOW] = 1u;
                continue;
            }
            const int L = R.cur_shanten, T = R.T;
            bool fresh = false;
            int slot = -1;
            if (lane < R.n_cand) {
                SpState s = R.root;
                if (R.can_discard) sp_discard(s, A->X.cand_tile[lane]);
                slot = spg_insert(G, rowg, sp_dk_add(0ull, -1, R.can_discard ? A->X.cand_tile[lane] : -1), fresh);
                A->X.cand_slot[lane] = slot;
                if (fresh) {
                    u64 k[4];
                    sp_key(s, k);
                    SP_HBM SpNode& n = ((SP_HBM SpNode*)G.node)[slot];
                    n.k0 = k[0]; n.k1 = k[1]; n.k2 = k[2]; n.k3 = k[3];
                }
            }
            const bool assume_riichi = A->X.is_menzen && A->X.prefer_riichi;
            const int flags = (assume_riichi ? 1 : 0) | ((assume_riichi && A->X.calc_double_riichi) ? 2 : 0) | (A->X.calc_haitei ? 4 : 0);
            const u64 upper = ((u64)rowg << 26) | ((u64)(min(A->X.n_left, 123) & 127) << 43) | ((u64)(R.ld3 & 7) << 50) | ((u64)flags << 53) |
                              ((u64)(T & 31) << 56);  // depth 0
            if (out.blk >= 0 && out_level != L) spg_out_close(G, out_level, out, P.err);
            out_level = L;
            spg_append(G, L, out, T, fresh && slot >= 0, upper | (u64)(u32)max(slot, 0), P.err);
            mj_team_sync<64>();
            {  // the context (with the candidates' slots) and the row info go to HBM for the later phases
                const SpRec* src = reinterpret_cast<const SpRec*>(&A->X);
                SP_HBM SpRec* dst = reinterpret_cast<SP_HBM SpRec*>(&((SP_HBM SpCtx*)G.ctx)[rowg]);
                static_assert(sizeof(SpCtx) % 16 == 0, "SpCtx is copied in 16-byte words");
                for (int i = lane; i < (int)(sizeof(SpCtx) / 16); i += 64) {
                    const SpRec v = src[i];
                    dst[i].x = v.x; dst[i].y = v.y; dst[i].z = v.z; dst[i].w = v.w;
                }
                if (lane == 0) {
                    SP_HBM SpGRow& ri = ((SP_HBM SpGRow*)G.rinfo)[rowg];
                    ri.R.ok = R.ok; ri.R.can_discard0 = R.can_discard0; ri.R.can_discard = R.can_discard; ri.R.after_riichi = R.after_riichi;
                    ri.R.with_probs = R.with_probs; ri.R.last_tsumo = R.last_tsumo; ri.R.cur_shanten = R.cur_shanten; ri.R.T = R.T;
                    ri.R.n_cand = R.n_cand; ri.R.ld3 = R.ld3; ri.R.cans = R.cans;
                    ri.row = row;
                }
            }
            mj_team_sync<64>();
        }
    }
    spg_out_close(G, out_level, out, P.err);
    spg_out_stats(out, P.err);
    if (lane == 0 && n_rows_done) __hip_atomic_fetch_add(&((SP_HBM unsigned long long*)P.err)[1], (unsigned long long)n_rows_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 0) __hip_atomic_fetch_add(&((SP_HBM unsigned long long*)P.err)[2], (unsigned long long)(wall_clock64() - t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------------------- expansion
struct SpChunkG : SpChunk {
    u64 desc[SP_NS];  // the states' descriptors (row, wall size, len_div3, flags)
    u8 ld3[SP_NS];
};

// Passes P0-P1c of mj_sp.hip's sp_chunk_probe for SP_NS states of ANY rows: state keys, row ids, optimal entries -> req[s]
__device__ __forceinline__ void spg_chunk_probe(const SpG& G, SpChunkG* C, const SpTabG& TG, const SP_HBM u64* states, int n, int L) {
    const int tid = threadIdx.x & (SP_NT - 1);
    SP_HBM SpNode* const nodes = (SP_HBM SpNode*)G.node;
    for (int task = tid; task < n * 4; task += SP_NT) {
        const int s = task >> 2, j = task & 3;
        const u64 d = states[s];
        const u32 slot = SPG_D_SLOT(d);
        C->k[s][j] = reinterpret_cast<const SP_HBM u64*>(&nodes[slot])[j];  // k0..k3 lead the node
        if (j == 0) {
            C->slot[s] = slot;
            C->desc[s] = d;
            C->ld3[s] = (u8)SPG_D_LD3(d);
            C->dk[s] = ((SP_HBM u64*)G.tag)[slot] & SPG_ID_MASK;
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
        const int ld3 = C->ld3[s];
        const u32 i0 = C->id[s][0], i1 = C->id[s][1], i2 = C->id[s][2], i3 = C->id[s][3];
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
        const int ld3 = C->ld3[s];
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

struct SpGPoolBlk {
    u32 base, fill, size;
};
// Levels >= 1: required draws, shanten-keeping discards and the children of SP_NS states (mj_sp.hip: sp_expand_chunk).
__device__ __noinline__ void spg_expand_chunk(const SpG& G, SpChunkG* C, const u64* states_, int n, int L, int key_out, SpGOut& out, SpGPoolBlk& pb,
                                              unsigned long long* err) {
    SP_ASSUME_LDS(C);
    const SP_HBM u64* states = (const SP_HBM u64*)states_;
    const SpTabG TG = sp_tab_g(c_sp_tab);
    const int tid = threadIdx.x & (SP_NT - 1);
    SP_HBM SpNode* const nodes = (SP_HBM SpNode*)G.node;
    SP_HBM u64* const pool = (SP_HBM u64*)G.pool;
    SP_HBM u32* const ctl = (SP_HBM u32*)G.ctl;
    spg_chunk_probe(G, C, TG, states, n, L);

    u32 ntw[SP_NS / 4];
#pragma unroll
    for (int w = 0; w < SP_NS / 16; w++) {
        const SpRec raw = reinterpret_cast<const SpRec*>(C->n_tiles)[w];  // 16-byte LDS reads
        ntw[4 * w] = raw.x; ntw[4 * w + 1] = raw.y; ntw[4 * w + 2] = raw.z; ntw[4 * w + 3] = raw.w;
    }
    auto nt_of = [&](auto sc) -> int {  // static state index
        constexpr int q = decltype(sc)::value;
        return q < SP_NS ? (int)((ntw[q >> 2] >> (8 * (q & 3))) & 0xFFu) : 0;
    };
    int total_items = 0;
    sp_static_for<0, SP_NS>([&](auto sc) { if (decltype(sc)::value < n) total_items += nt_of(sc); });
    const int n_sub = (total_items + SP_ITEM_CAP - 1) / SP_ITEM_CAP, target = n_sub > 1 ? (total_items + n_sub - 1) / n_sub : SP_ITEM_CAP;
    for (int sb = 0; sb < n;) {
        int se = sb, n_items = 0;  // uniform over the wavefront
        int my_first = 0;          // first item of state `tid`
        {
            bool stopped = false;
            sp_static_for<0, SP_NS>([&](auto sc) {
                constexpr int q = decltype(sc)::value;
                if (q < sb || q >= n || stopped) return;
                const int nt = nt_of(sc);
                if (q > sb && (n_items + nt > SP_ITEM_CAP || n_items >= target)) { stopped = true; return; }
                if (q == tid) my_first = n_items;
                n_items += nt;
                se = q + 1;
            });
        }
        if (tid >= sb && tid < se) {
            const int s = tid;
            int it = my_first;
            for (u64 rest = C->req[s]; rest; rest &= rest - 1) C->item[it++] = (unsigned short)(s | ((__ffsll((long long)rest) - 1) << SP_SB));
            if (it == my_first) {  // no draws left: an empty child list
                SP_HBM SpNode& node = nodes[C->slot[s]];
                node.child_off = 0;
                node.n_ch = 0;
                node.sumreq = 0;
                node.n_ent = 0;
            }
        }
        mj_team_sync<SP_NT>();
        // P2: the shanten-keeping discards of g = h + t, lane = item
        const bool has_item = tid < n_items;
        const int my_e = has_item ? (int)C->item[tid] : 0, my_s = my_e & (SP_NS - 1), my_t = my_e >> SP_SB;
        const SpState S = sp_chunk_state(C, my_s);
        u64 kept = 0;
        if (has_item) {
            const int s = my_s, t = my_t;
            const int ld3 = C->ld3[s];
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
        // P3: child list layout + node header (one wavefront scan over the item lanes); the sub-batch's child lists are one
        // contiguous piece of the wavefront's pool block
        int n_entries;
        {
            const int wc = has_item ? S.w.get(my_t) : 0;
            const int nvar = has_item && sp_aka_in_wall(S, my_t) ? (wc >= 2 ? 2 : 1) : 1;
            const int nkeep = __popcll(kept), my_ent = nvar * nkeep;
            const u32 pv = sp_wave_scan_incl(has_item ? ((u32)my_ent | ((u32)wc << 16) | ((u32)(nkeep ? nvar : 0) << 24)) : 0u);
            int s_first = 0, s_n = 0;  // this lane's state: first item lane, items
            sp_static_for<0, SP_NS>([&](auto sc) {
                constexpr int q = decltype(sc)::value;
                if (q < sb || q >= se) return;
                const int nt = nt_of(sc);
                if (q < my_s) s_first += nt;
                if (q == my_s) s_n = nt;
            });
            const int s_last = s_first + s_n - 1;
            const u32 before = __shfl(pv, max(s_first - 1, 0)), upto = __shfl(pv, max(s_last, 0)), all = __shfl(pv, SP_NT - 1);
            const u32 base = s_first > 0 ? before : 0u;
            n_entries = (int)(all & 0xFFFFu);
            if (pb.fill + (u32)n_entries > pb.size) {  // a new pool block for this wavefront (uniform)
                u32 b = 0;
                if (tid == 0) b = __hip_atomic_fetch_add(&ctl[SPG_C_NPOOL], SPG_POOL_BLK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b = spg_wave_first(b);
                if (b + SPG_POOL_BLK > G.pool_cap) {  // pool full: counted as an overflow, the entries land in the last block
                    if (tid == 0) ctl[SPG_C_OVERFLOW] = 1u;
                    b = G.pool_cap - SPG_POOL_BLK;
                }
                pb.base = b;
                pb.fill = 0;
                pb.size = SPG_POOL_BLK;
            }
            const u32 sub_base = pb.base + pb.fill;
            pb.fill += (u32)n_entries;
            if (has_item) {
                C->coff[tid] = (unsigned short)(((pv - base) & 0xFFFFu) - (u32)my_ent);
                C->eoff[tid] = (unsigned short)((pv & 0xFFFFu) - (u32)my_ent);
                if (tid == s_last) {  // one lane per state: the node header
                    const u32 tot = upto - base;
                    const int total = (int)(tot & 0xFFFFu);
                    const int sumreq = (int)((tot >> 16) & 0xFFu), n_ent = (int)(tot >> 24);
                    const u32 child_base = sub_base + (base & 0xFFFFu);
                    C->child_base[my_s] = (int)child_base;
                    SP_HBM SpNode& node = nodes[C->slot[my_s]];
                    node.child_off = child_base;
                    node.n_ch = (unsigned short)total;
                    node.sumreq = (u8)(sumreq & 0xFF);
                    node.n_ent = (u8)min(n_ent, 255);
                }
            }
            if (tid == 0) C->eoff[n_items] = (unsigned short)n_entries;
        }
        mj_team_sync<SP_NT>();
        // P4: children, one lane per CHILD ENTRY (draw variant x kept discard), two entries per lane and round
        struct Ent {
            bool on;
            int s, it, local, rank, nk, tile, dt, count;
            u64 dk;
            u32 pos;
        };
        auto decode = [&](int e) -> Ent {
            Ent E;
            E.on = e < n_entries;
            const int ee = E.on ? e : 0;
            int lo = 0, hi = n_items;  // largest item with eoff[item] <= e
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if ((int)C->eoff[mid] <= ee) lo = mid; else hi = mid;
            }
            E.it = lo;
            E.local = ee - (int)C->eoff[lo];
            const u64 bits = C->kept[lo];
            E.nk = __popcll(bits);
            const int vidx = E.local >= E.nk ? 1 : 0;
            E.rank = E.local - vidx * E.nk;
            u64 mrest = bits;
            for (int r = E.rank; r > 0; r--) mrest &= mrest - 1;
            const int d = __ffsll((long long)mrest) - 1;
            E.s = C->item[lo] & (SP_NS - 1);
            const int t = C->item[lo] >> SP_SB;
            const SpState Sx = sp_chunk_state(C, E.s);
            const int cnt = Sx.w.get(t);
            const bool aka = sp_aka_in_wall(Sx, t);
            const bool red = aka && (vidx == 1 || cnt < 2);  // the tile's draw entries: plain (all copies but the red one) if any, then the red five
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
            E.pos = spg_pos(((u64)SPG_D_ROWG(C->desc[E.s]) << 42) | E.dk, G.cap_mask);
            return E;
        };
        auto tag_of = [&](const Ent& E) -> u64 { return SPG_TAG(SPG_D_ROWG(C->desc[E.s]), E.dk); };
        auto finish = [&](const Ent& E, u64 first_old, bool& fresh) -> int {  // the rest of the insert after the first claim + the child entry
            const u64 tag = tag_of(E);
            u32 pos = E.pos;
            u64 old = first_old;
            int cs = -1;
            fresh = false;
            for (int probe = 0; probe < SPG_MAX_PROBE; probe++) {
                if (old == 0ull) {
                    u64 k[4];
                    sp_key(sp_apply(sp_chunk_state(C, E.s), E.tile, E.dt), k);
                    SP_HBM SpNode& nd = nodes[pos];
                    nd.k0 = k[0]; nd.k1 = k[1]; nd.k2 = k[2]; nd.k3 = k[3];
                    fresh = true;
                    cs = (int)pos;
                    break;
                }
                if (old == tag) {
#ifdef MJ_EMU  // the emulator never pre-empts between the claim and the key write: check the bijection on every hit
                    u64 k[4];
                    sp_key(sp_apply(sp_chunk_state(C, E.s), E.tile, E.dt), k);
                    SP_HBM SpNode& nd = nodes[pos];
                    if (nd.k0 != k[0] || nd.k1 != k[1] || nd.k2 != k[2] || nd.k3 != k[3]) ctl[SPG_C_OVERFLOW] = 1u;
#endif
                    cs = (int)pos;
                    break;
                }
                pos = (pos + 1) & G.cap_mask;
                old = sp_claim_tag(&((SP_HBM u64*)G.tag)[pos], tag);
            }
            if (cs < 0) ctl[SPG_C_OVERFLOW] = 1u;
            const u32 pos_out = (u32)C->child_base[E.s] + (u32)C->coff[E.it] + (u32)E.local;
            const u64 ent = (cs < 0 ? SPG_E_INVALID : (u64)(u32)cs) | ((u64)sp_discard_key(E.dt) << 26) | (E.rank == E.nk - 1 ? SPG_E_LAST : 0ull) |
                            ((u64)E.count << 36);
            if (pos_out < G.pool_cap) pool[pos_out] = ent;
            return cs;
        };
        for (int e0 = 0; e0 < n_entries; e0 += 2 * SP_NT) {
            const Ent A = decode(e0 + tid), B = decode(e0 + SP_NT + tid);
            u64 oa = 1ull, ob = 1ull;
            if (A.on) oa = sp_claim_tag(&((SP_HBM u64*)G.tag)[A.pos], tag_of(A));
            if (B.on) ob = sp_claim_tag(&((SP_HBM u64*)G.tag)[B.pos], tag_of(B));
            bool fa = false, fb = false;
            int ca = -1, cb = -1;
            if (A.on) ca = finish(A, oa, fa);
            if (B.on) cb = finish(B, ob, fb);
            // the new states join level L - 1 (depth + 1, same T): the wavefront's open block, one atomic per 64 states
            spg_append(G, L - 1, out, key_out, A.on && fa && ca >= 0, SPG_D_CHILD_UPPER(C->desc[A.s]) | (u64)(u32)max(ca, 0), err);
            spg_append(G, L - 1, out, key_out, B.on && fb && cb >= 0, SPG_D_CHILD_UPPER(C->desc[B.s]) | (u64)(u32)max(cb, 0), err);
        }
        mj_team_sync<SP_NT>();
        sb = se;
    }
}

#ifndef SPG_RUN
#define SPG_RUN 8  // consecutive chunks (of 16 states) a wavefront takes at a time
#endif
#ifndef SPG_EXPAND_WPS
#define SPG_EXPAND_WPS 4  // resident wavefronts per SIMD the expansion is compiled for
#endif
__global__ __launch_bounds__(256, SPG_EXPAND_WPS) void mj_k_spg_expand(SpGParams P) {
    __shared__ SpChunkG s_chunk[4];
    const SpG G = P.g;
    const int L = P.level;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 n_blk = min(((SP_HBM u32*)G.ctl)[SPG_C_NBLK + L], G.lst_cap / SPG_BLK);
    const u32 n_chunks = n_blk * (SPG_BLK / SP_NS);
    SpGOut out = SPG_OUT_INIT;
    SpGPoolBlk pb = SpGPoolBlk{0u, 0u, 0u};
    const long long t0 = wall_clock64();
    // RUNS of SPG_RUN consecutive chunks by a static stride over all wavefronts of the launch (a shared cursor would serialise
    // them, see SPG_RAW_RES).  A run's children end up next to each other in the wavefront's raw-list reservation, so the child
    // level keeps the states of one row together — what gives the evaluation its L1 / L2 hits on the children sibling states share
    // (chunk-by-chunk striding scattered every row over the whole list: 24 % L2 hits, the evaluation bound by HBM).
    for (u32 cc = (spg_vwg() * 4u + (u32)wv) * SPG_RUN; cc < n_chunks; cc += gridDim.x * 4u * SPG_RUN)
    for (u32 c = cc; c < min(cc + SPG_RUN, n_chunks); c++) {
        const u32 b = c / (SPG_BLK / SP_NS), sub = c % (SPG_BLK / SP_NS);
        const u32 info = ((SP_HBM u32*)G.blk[L])[b];
        const int n = min(SPG_B_COUNT(info) - (int)sub * SP_NS, SP_NS);
        if (n <= 0) continue;
        const int key_out = SPG_B_T(info) | (min(SPG_B_OFF(info) + 1, 3) << 5);
        spg_expand_chunk(G, &s_chunk[wv], G.lst[L] + (size_t)b * SPG_BLK + sub * SP_NS, n, L, key_out, out, pb, P.err);
    }
    spg_out_close(G, L - 1, out, P.err);
    spg_out_stats(out, P.err);
    if (lane == 0) __hip_atomic_fetch_add(&((SP_HBM unsigned long long*)P.err)[3], (unsigned long long)(wall_clock64() - t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------------------- level 0
// which draws win + one scoring item per draw entry (mj_sp.hip: sp_l0_probe_chunk)
#define SPG_ITEM_RES 1024u  // scoring items a wavefront reserves per atomic (an item word is never 0: bit 63 = valid)
#define SPG_I_VALID (1ull << 63)
struct SpGItemBlk {
    u32 base, fill;  // the wavefront's reservation: first item, items used (SPG_ITEM_RES = exhausted)
};
__device__ __noinline__ void spg_l0_probe_chunk(const SpG& G, SpChunkG* C, const u64* states_, int n, SpGItemBlk& ib) {
    SP_ASSUME_LDS(C);
    const SP_HBM u64* states = (const SP_HBM u64*)states_;
    const SpTabG TG = sp_tab_g(c_sp_tab);
    SP_HBM u32* const ctl = (SP_HBM u32*)G.ctl;
    spg_chunk_probe(G, C, TG, states, n, 0);
    const int s = threadIdx.x & (SP_NT - 1);
    const bool on = s < n;
    const SpState S = sp_chunk_state(C, on ? s : 0);
    const u64 req = on ? C->req[s] : 0ull;
    int cnt = 0, sumreq = 0;
    for (u64 rest = req; rest; rest &= rest - 1) {
        const int t = __ffsll((long long)rest) - 1;
        const bool aka = sp_aka_in_wall(S, t);
        cnt += (!aka || S.w.get(t) >= 2) + aka;
        sumreq += S.w.get(t);
    }
    if (cnt > SP_L0_MAX) { ctl[SPG_C_OVERFLOW] = 1u; cnt = SP_L0_MAX; }
    // the chunk's items are one contiguous piece of the wavefront's item reservation (a wavefront scan; one atomic per SPG_ITEM_RES items)
    const u32 incl = sp_wave_scan_incl((u32)cnt);
    const u32 total = __shfl(incl, SP_NT - 1);  // <= 16 x 17
    if (ib.fill + total > SPG_ITEM_RES) {
        SP_HBM u64* const items = (SP_HBM u64*)G.items;
        if (ib.fill < SPG_ITEM_RES && ib.base + SPG_ITEM_RES <= G.items_cap)  // the unused tail reads as "no item"
            for (u32 i = ib.fill + (u32)s; i < SPG_ITEM_RES; i += SP_NT) items[ib.base + i] = 0ull;
        u32 b = 0;
        if (s == 0) b = __hip_atomic_fetch_add(&ctl[SPG_C_NITEMS], SPG_ITEM_RES, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ib.base = spg_wave_first(b);
        ib.fill = 0;
    }
    const u32 gbase = ib.base + ib.fill;
    ib.fill += total;
    if (on) {
        const u32 base = gbase + incl - (u32)cnt;
        const u32 slot = C->slot[s];
        const u64 rowg = SPG_D_ROWG(C->desc[s]);
        SP_HBM SpNode& node = ((SP_HBM SpNode*)G.node)[slot];
        int e = 0;
        for (u64 rest = req; rest; rest &= rest - 1) {
            const int t = __ffsll((long long)rest) - 1;
            const bool aka = sp_aka_in_wall(S, t);
            const int wc = S.w.get(t);
            for (int variant = 0; variant < 2; variant++) {
                if (variant == 0 ? (aka && wc < 2) : !aka) continue;
                if (e < cnt) {
                    if (base + e < G.items_cap)
                        ((SP_HBM u64*)G.items)[base + e] = (u64)slot | ((u64)e << 26) | ((u64)t << 31) | ((u64)variant << 37) | (rowg << 38) | SPG_I_VALID;
                    else ctl[SPG_C_OVERFLOW] = 1u;
                    node.l0cnt[e] = (u8)(!aka ? wc : variant == 0 ? wc - 1 : 1);  // draw_without_tegawari's `count`
                }
                e++;
            }
        }
        node.child_off = 0;  // bit i: draw entry i has a yaku (set by the scoring pass)
        node.n_ch = (unsigned short)cnt;
        node.sumreq = (u8)(sumreq & 0xFF);
        node.n_ent = (u8)cnt;
    }
    mj_team_sync<SP_NT>();
}
__global__ __launch_bounds__(256, SPG_EXPAND_WPS) void mj_k_spg_probe(SpGParams P) {
    __shared__ SpChunkG s_chunk[4];
    const SpG G = P.g;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 n_blk = min(((SP_HBM u32*)G.ctl)[SPG_C_NBLK + 0], G.lst_cap / SPG_BLK);
    const u32 n_chunks = n_blk * (SPG_BLK / SP_NS);
    const long long t0 = wall_clock64();
    SpGItemBlk ib = SpGItemBlk{0u, SPG_ITEM_RES};
    for (u32 cc = (spg_vwg() * 4u + (u32)wv) * SPG_RUN; cc < n_chunks; cc += gridDim.x * 4u * SPG_RUN)
    for (u32 c = cc; c < min(cc + SPG_RUN, n_chunks); c++) {
        const u32 b = c / (SPG_BLK / SP_NS), sub = c % (SPG_BLK / SP_NS);
        const u32 info = ((SP_HBM u32*)G.blk[0])[b];
        const int n = min(SPG_B_COUNT(info) - (int)sub * SP_NS, SP_NS);
        if (n <= 0) continue;
        spg_l0_probe_chunk(G, &s_chunk[wv], G.lst[0] + (size_t)b * SPG_BLK + sub * SP_NS, n, ib);
    }
    if (ib.fill < SPG_ITEM_RES && ib.base + SPG_ITEM_RES <= G.items_cap)  // the unused tail of the last reservation
        for (u32 i = ib.fill + (u32)lane; i < SPG_ITEM_RES; i += 64) ((SP_HBM u64*)G.items)[ib.base + i] = 0ull;
    if (lane == 0) __hip_atomic_fetch_add(&((SP_HBM unsigned long long*)P.err)[4], (unsigned long long)(wall_clock64() - t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one thread per scoring item (mj_sp.hip: sp_l0_score), dense over all rows; the row's context is read from HBM
__global__ __launch_bounds__(256) void mj_k_spg_score(SpGParams P) {
    const SpG G = P.g;
    const u32 n_items = min(((SP_HBM u32*)G.ctl)[SPG_C_NITEMS], G.items_cap);
    const long long t0 = wall_clock64();
    for (u32 i = spg_vwg() * 256u + threadIdx.x; i < n_items; i += gridDim.x * 256u) {
        const u64 item = ((SP_HBM u64*)G.items)[i];
        if (!(item & SPG_I_VALID)) continue;  // the unused tail of a wavefront's reservation
        const int idx = SPG_I_IDX(item), t = SPG_I_TILE(item), variant = SPG_I_VARIANT(item);
        SP_HBM SpNode& node = ((SP_HBM SpNode*)G.node)[SPG_I_SLOT(item)];
        const SpCtx* X = &G.ctx[SPG_I_ROWG(item)];
        SpState S1 = sp_state_of(node);
        const int tile = variant ? akaize(t) : t;
        sp_deal(S1, tile);
        float scv[4];
        if (sp_get_score(c_mj_tables, X, S1, tile, scv)) {
#pragma unroll
            for (int q = 0; q < 4; q++) node.sc[idx][q] = scv[q];
            __hip_atomic_fetch_or(&node.child_off, 1u << idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&((SP_HBM unsigned long long*)P.err)[4], (unsigned long long)(wall_clock64() - t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------------------- evaluation
// One block of a level list (<= 64 states, the same T and depth) per wavefront at a time: teams of T - off lanes in lock-step.
// The structure is sp_eval_wave of mj_sp.hip (fold SP_EV_ENT children, park completed draw entries, accumulate all parked entries of
// all teams together); what a row's workgroup kept in its LDS context comes from the state's descriptor (wall size -> not_tsumo
// row and tsumo_prob entries, flags) or the block (T, off).  LK = 0: the level-0 variant — a draw entry is a winning draw with
// its four scores (yaku bit set by mj_k_spg_score), nothing to fold.
template <int TN, int LK>
__device__ __noinline__ void spg_eval_block(const SpG& G, float* WL, const u64* states_, int count, int T_, int off_, int lane_in_team, int team_in_wave,
                                            int tpw, bool team_on) {
    SP_ASSUME_LDS(WL);
    const SP_HBM u64* states = (const SP_HBM u64*)states_;
    SP_HBM SpNode* const nodes = (SP_HBM SpNode*)G.node;
    SP_HBM u64* const pool = (SP_HBM u64*)G.pool;
    const int T = __builtin_amdgcn_readfirstlane(T_), off = __builtin_amdgcn_readfirstlane(off_);
    const int ln = min(lane_in_team + off, SP_T - 1);  // this lane's turn (lanes outside any team: clamped, never stored)
    const int rows = T + 4;
    float* const eb = WL + (team_on ? team_in_wave : 0) * (SP_EV_ENT * rows * 4);  // [SP_EV_ENT][rows][4]
    if (team_on)
        for (int r = lane_in_team; r < SP_EV_ENT * rows; r += T - off) *reinterpret_cast<SpF4*>(eb + 4 * r) = SpF4{0.f, 0.f, 0.f, 0.f};
    const SP_HBM float* const nt_all = (const SP_HBM float*)c_sp_nt;
    const SP_HBM SpF4* const tp_all = (const SP_HBM SpF4*)c_spg_tp;
    const int last = max(count - 1, 0);
    auto ld_desc = [&](int i) -> u64 { return states[min(i, last)]; };
    auto ld_hdr = [&](u64 d) -> u64 {
        SP_HBM unsigned long long* hp = reinterpret_cast<SP_HBM unsigned long long*>(&nodes[SPG_D_SLOT(d)].child_off);
        return LK > 0 ? *hp : __hip_atomic_load(hp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // level 0: the yaku bits were set by L2 atomics
    };
    auto ld_m = [&](u64 d, u64 hdr) -> float {
        return nt_all[((size_t)SPG_D_NLEFT(d) * SP_NT_ROWS + min((int)((hdr >> 48) & 0xFF), SP_NT_ROWS - 1)) * SP_NT_STRIDE + ln];
    };
    auto ld_tp = [&](u64 d) -> SpF4 {
        const SP_HBM SpF4* p = &tp_all[max(SPG_D_NLEFT(d) - ln, 0)];
        SpF4 r;
        r.x = p->x; r.y = p->y; r.z = p->z; r.w = p->w;
        return r;
    };
    mj_team_sync<64>();
    int i = team_in_wave;
    bool has = team_on && i < count;
    const int stride = tpw;

    if constexpr (LK == 0) {
        // ---- level 0: per state the draw entries with a yaku; entry = (count, 4 scores); accumulate in the reference's order
        u64 d0 = ld_desc(i), d1 = ld_desc(i + stride), d2 = ld_desc(i + 2 * stride);
        u64 h0 = ld_hdr(d0), h1 = ld_hdr(d1);
        while (__ballot(has) != 0ull) {
            SP_HBM SpNode& node = nodes[SPG_D_SLOT(d0)];
            const u32 yaku = (u32)h0;
            const int n_ent = (int)((h0 >> 32) & 0xFFFF);
            const float m_raw = ld_m(d0, h0);
            const SpF4 tp = ld_tp(d0);
            const int flags = SPG_D_FLAGS(d0);
            const bool assume_riichi = (flags & 1) != 0, haitei = (flags & 4) != 0;
            const int hp_base = (int)(((flags & 2) != 0) && ln == 0);
            const u64 d3 = ld_desc(i + 3 * stride);  // in flight under this state
            const u64 h2 = ld_hdr(d2);
            const float my_m = m_raw != 0.f ? m_raw : 1.f, my_r = sp_rcp_refined(my_m);
            float acc_w = 0.f, acc_e = 0.f;
            const int nmax = [&] {  // the largest entry count of the wavefront's current states (uniform loop bound)
                int v = has ? n_ent : 0;
                for (int dlt = 32; dlt > 0; dlt >>= 1) v = max(v, __shfl_xor(v, dlt));
                return v;
            }();
            for (int e0 = 0; e0 < nmax; e0 += SP_EV_ENT) {
                // park the numerators A[j] = tsumo_prob[count][j] * not_tsumo[j] of up to SP_EV_ENT entries, fetch their scores
                float sc[SP_EV_ENT][4];
                bool use[SP_EV_ENT];
#pragma unroll
                for (int q = 0; q < SP_EV_ENT; q++) {
                    const int e = min(e0 + q, SP_L0_MAX - 1);
                    use[q] = has && e0 + q < n_ent && ((yaku >> (e0 + q)) & 1);
                    const int cnt = min(max((int)node.l0cnt[e], 1), 4);
#pragma unroll
                    for (int k = 0; k < 4; k++) sc[q][k] = node.sc[e][k];
                    const float tpc = cnt == 1 ? tp.x : cnt == 2 ? tp.y : cnt == 3 ? tp.z : tp.w;
                    if (use[q]) eb[(q * rows + ln + 1) * 4 + 3] = tpc * m_raw;
                }
                mj_team_sync<64>();
#pragma unroll
                for (int q = 0; q < SP_EV_ENT; q++) {
                    if (__ballot(use[q]) == 0ull) continue;
                    if (use[q]) {
                        const float* er = eb + q * rows * 4;
                        sp_static_for<0, TN>([&](auto jc) {
                            constexpr int j = decltype(jc)::value;
                            if (j >= T || j < off) return;  // scalar
                            float prob = sp_div_domain(er[(j + 1) * 4 + 3], my_m, my_r);
                            prob = ln <= j ? prob : 0.f;
                            const int hp = hp_base + (int)(assume_riichi && j == ln) + (int)(haitei && j == T - 1);
                            acc_w += prob;
                            acc_e += prob * (hp == 0 ? sc[q][0] : hp == 1 ? sc[q][1] : hp == 2 ? sc[q][2] : sc[q][3]);
                        });
                    }
                }
                mj_team_sync<64>();
            }
            if (has) {
                SP_HBM SpF4* dst = reinterpret_cast<SP_HBM SpF4*>(node.val[ln]);
                dst->x = 0.f; dst->y = acc_w; dst->z = acc_e; dst->w = __int_as_float((int)acc_e);
            }
            d0 = d1; d1 = d2; d2 = d3;
            h0 = h1; h1 = h2;
            i += stride;
            has = team_on && i < count;
        }
    } else {
        auto ld_ent = [&](u32 at) -> u64 { return pool[min(at, G.pool_cap - 1u)]; };
        auto ld_val = [&](u64 ent) -> SpF4 {  // one 16-byte load
            // (an entry past a child list's end is whatever the pool held: masked into the node array, its value never used)
            const SP_HBM SpF4* p = reinterpret_cast<const SP_HBM SpF4*>(nodes[SPG_E_SLOT(ent) & G.cap_mask].val[ln]);
            SpF4 r;
            r.x = p->x; r.y = p->y; r.z = p->z; r.w = p->w;
            return r;
        };
        // the pipeline: state 0 = current, 1 = next (header, first entries, not_tsumo value, tsumo_prob loaded), 2 = header loaded, 3 = descriptor
        u64 d0 = ld_desc(i), d1 = ld_desc(i + stride), d2 = ld_desc(i + 2 * stride), d3 = ld_desc(i + 3 * stride);
        u64 h0 = ld_hdr(d0), h1 = ld_hdr(d1), h2 = ld_hdr(d2);
        float m_raw = ld_m(d0, h0), m_nxt = ld_m(d1, h1);
        SpF4 tp = ld_tp(d0), tp_nxt = ld_tp(d1);
        u64 ent[SP_EV_ENT], entn[SP_EV_ENT], nent[SP_EV_ENT];
#pragma unroll
        for (int q = 0; q < SP_EV_ENT; q++) {
            ent[q] = ld_ent((u32)h0 + q);
            entn[q] = ld_ent((u32)h0 + SP_EV_ENT + q);
            nent[q] = ld_ent((u32)h1 + q);
        }
        SpF4 v[SP_EV_ENT];
#pragma unroll
        for (int q = 0; q < SP_EV_ENT; q++) v[q] = ld_val(ent[q]);
        int c0 = 0;
        float my_m = m_raw != 0.f ? m_raw : 1.f, my_r = sp_rcp_refined(my_m);
        float acc_t = 0.f, acc_w = 0.f, acc_e = 0.f;
        float nx_t = -3.40282347e+38f, nx_w = -3.40282347e+38f, nx_e = -3.40282347e+38f;  // discard_slow (calc.rs:570-637) fold state
        int max_value = INT_MIN, max_key = sp_discard_key(T_UNK);
        while (__ballot(has) != 0ull) {
            const int n_ch = (int)((h0 >> 32) & 0xFFFF);
            int k = 0;
#pragma unroll
            for (int q = 0; q < SP_EV_ENT; q++) {
                const u64 e = ent[q];
                const bool valid = has && c0 + q < n_ch;
                const bool bad = (e & SPG_E_INVALID) != 0;
                if (valid && bad) ((SP_HBM u32*)G.ctl)[SPG_C_OVERFLOW] = 1u;
                const int value = __float_as_int(v[q].w);  // `as i32` of the child's EV (maximize_win_prob = false)
                const int key = SPG_E_KEY(e);              // cmp_discard_priority(dt, max_tile) > 0  <=>  key > max_key
                const bool better = valid && !bad && (value > max_value || (value == max_value && key > max_key));
                nx_t = better ? v[q].x : nx_t;
                nx_w = better ? v[q].y : nx_w;
                nx_e = better ? v[q].z : nx_e;
                max_value = better ? value : max_value;
                max_key = better ? key : max_key;
                if (valid && (e & SPG_E_LAST)) {  // last child of this draw entry (uniform in the team)
                    const int cnt = min(max(SPG_E_COUNT(e), 1), 4);
                    const float tpc = cnt == 1 ? tp.x : cnt == 2 ? tp.y : cnt == 3 ? tp.z : tp.w;
                    float* row = eb + (k * rows + ln) * 4;
                    row[0] = nx_t;
                    row[1] = nx_w;
                    row[2] = nx_e;
                    row[7] = tpc * m_raw;  // A[ln], read with row ln + 1
                    k++;
                    nx_t = nx_w = nx_e = -3.40282347e+38f;
                    max_value = INT_MIN;
                    max_key = sp_discard_key(T_UNK);
                }
            }
            c0 += SP_EV_ENT;
            const bool done = has && c0 >= n_ch;
            // the loads of the next step (issued by every lane whether or not its team advances: no divergent control flow)
            u64 up[SP_EV_ENT], upn[SP_EV_ENT], nent2[SP_EV_ENT];
            SpF4 vn[SP_EV_ENT];
            const u32 up_off = done ? (u32)h1 + SP_EV_ENT : (u32)h0 + (u32)c0 + SP_EV_ENT;
#pragma unroll
            for (int q = 0; q < SP_EV_ENT; q++) up[q] = done ? nent[q] : entn[q];
#pragma unroll
            for (int q = 0; q < SP_EV_ENT; q++) vn[q] = ld_val(up[q]);
#pragma unroll
            for (int q = 0; q < SP_EV_ENT; q++) upn[q] = ld_ent(up_off + q);
#pragma unroll
            for (int q = 0; q < SP_EV_ENT; q++) nent2[q] = ld_ent((u32)h2 + q);
            const float m_n2 = ld_m(d2, h2);
            const SpF4 tp_n2 = ld_tp(d2);
            const u64 h3 = ld_hdr(d3);
            const u64 d4 = ld_desc(i + 4 * stride);

            mj_team_sync<64>();
            const int kmax = __ballot(k >= 4) ? 4 : __ballot(k >= 3) ? 3 : __ballot(k >= 2) ? 2 : __ballot(k >= 1) ? 1 : 0;
            for (int en = 0; en < kmax; en++) {
                if (en < k) {
                    const float* er = eb + en * rows * 4;
                    sp_static_for<0, (TN + 3) / 4>([&](auto gc) {
                        constexpr int g = decltype(gc)::value;
                        if (4 * g + 3 < off || 4 * g >= T) return;  // scalar: turns before `off` have no lane, rows past T are zero
                        SpF4 r[4];
#pragma unroll
                        for (int jj = 0; jj < 4; jj++) r[jj] = *reinterpret_cast<const SpF4*>(er + (4 * g + jj + 1) * 4);
#pragma unroll
                        for (int jj = 0; jj < 4; jj++) {
                            const int j = 4 * g + jj;
                            float prob = sp_div_domain(r[jj].w, my_m, my_r);
                            prob = ln <= j ? prob : 0.f;
                            if constexpr (LK == 1) acc_t += prob;
                            else acc_t += prob * r[jj].x;
                            acc_w += prob * r[jj].y;
                            acc_e += prob * r[jj].z;
                        }
                    });
                }
            }
            mj_team_sync<64>();
            if (done) {
                SP_HBM SpF4* dst = reinterpret_cast<SP_HBM SpF4*>(nodes[SPG_D_SLOT(d0)].val[ln]);
                dst->x = acc_t; dst->y = acc_w; dst->z = acc_e; dst->w = __int_as_float((int)acc_e);
                d0 = d1; d1 = d2; d2 = d3; d3 = d4;
                h0 = h1; h1 = h2; h2 = h3;
                m_raw = m_nxt; m_nxt = m_n2;
                tp = tp_nxt; tp_nxt = tp_n2;
#pragma unroll
                for (int q = 0; q < SP_EV_ENT; q++) nent[q] = nent2[q];
                c0 = 0;
                my_m = m_raw != 0.f ? m_raw : 1.f;
                my_r = sp_rcp_refined(my_m);
                acc_t = acc_w = acc_e = 0.f;
                i += stride;
                has = i < count;
            }
#pragma unroll
            for (int q = 0; q < SP_EV_ENT; q++) {
                ent[q] = up[q];
                entn[q] = upn[q];
                v[q] = vn[q];
            }
        }
    }
}

#ifndef SPG_EVAL_WPS
#define SPG_EVAL_WPS 4  // resident wavefronts per SIMD the evaluation is compiled for
#endif
template <int LK>
__global__ __launch_bounds__(256, SPG_EVAL_WPS) void mj_k_spg_eval(SpGParams P) {
    __shared__ float s_ev[4 * SP_EVW_WAVE_FLOATS];
    const SpG G = P.g;
    const int L = P.level;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 n_blk = min(((SP_HBM u32*)G.ctl)[SPG_C_NBLK + L], G.lst_cap / SPG_BLK);
    const long long t0 = wall_clock64();
    for (u32 b = spg_vwg() * 4u + (u32)wv; b < n_blk; b += gridDim.x * 4u) {
        const u32 info = ((SP_HBM u32*)G.blk[L])[b];
        const int count = SPG_B_COUNT(info), T = SPG_B_T(info);
        if (count <= 0 || T <= 0) continue;
        const int off = min(SPG_B_OFF(info), T - 1), TW = T - off;
        const int tpw = min(64 / TW, SP_EVW_WAVE_FLOATS / sp_evw_team_floats(T));
        const int tw = lane / TW, ln = lane - tw * TW;
        const bool on = tw < tpw;
        const u64* states = G.lst[L] + (size_t)b * SPG_BLK;
        float* wl = s_ev + wv * SP_EVW_WAVE_FLOATS;
        if (T <= 8) spg_eval_block<8, LK>(G, wl, states, count, T, off, ln, tw, tpw, on);
        else if (T <= 16) spg_eval_block<16, LK>(G, wl, states, count, T, off, ln, tw, tpw, on);
        else spg_eval_block<17, LK>(G, wl, states, count, T, off, ln, tw, tpw, on);
    }
    if (lane == 0) __hip_atomic_fetch_add(&((SP_HBM unsigned long long*)P.err)[5], (unsigned long long)(wall_clock64() - t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------------------- row writer
struct SpWriteArea {
    SpCtx X;
    float tv[SP_MAX_CAND * SP_T * 4];
};
__global__ __launch_bounds__(256) void mj_k_spg_write(SpGParams P) {
    __shared__ SpWriteArea s_area[4];
    const SpG G = P.g;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    SpWriteArea* A = &s_area[wv];
    const u32 n_rowg = min(((SP_HBM u32*)G.ctl)[SPG_C_NROWG], G.ctx_cap);
    const long long t0 = wall_clock64();
    for (u32 r = spg_vwg() * 4u + (u32)wv; r < n_rowg; r += gridDim.x * 4u) {
        {
            const SP_HBM SpRec* src = reinterpret_cast<const SP_HBM SpRec*>(&((SP_HBM SpCtx*)G.ctx)[r]);
            SpRec* dst = reinterpret_cast<SpRec*>(&A->X);
            for (int i = lane; i < (int)(sizeof(SpCtx) / 16); i += 64) dst[i] = spt_load(&src[i]);
        }
        const SP_HBM SpGRow& ri = ((SP_HBM SpGRow*)G.rinfo)[r];
        SpRowInfo R;
        R.ok = ri.R.ok; R.can_discard0 = ri.R.can_discard0; R.can_discard = ri.R.can_discard; R.after_riichi = ri.R.after_riichi;
        R.with_probs = ri.R.with_probs; R.last_tsumo = ri.R.last_tsumo; R.cur_shanten = ri.R.cur_shanten; R.T = ri.R.T;
        R.n_cand = ri.R.n_cand; R.ld3 = ri.R.ld3; R.cans = ri.R.cans;
        const int row = ri.row;
        mj_team_sync<64>();
        SP_HBM float* outp = (SP_HBM float*)P.obs + (size_t)row * (1012 * 34);
        sp_row_write<true, 64>((const SpNode*)G.node, A->X, R, lane, outp, A->tv);
        mj_team_sync<64>();
    }
    if (lane == 0) __hip_atomic_fetch_add(&((SP_HBM unsigned long long*)P.err)[6], (unsigned long long)(wall_clock64() - t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the launch sequence's epilogue: a capacity overflow anywhere (hash set, level lists, child-list pool, items, contexts) is one
// "overflow" of the cycle in the counters mj_counters reports (sp_overflow must stay 0: tests and bench.py check it)
__global__ void mj_k_spg_finish(SpGParams P) {
    if (threadIdx.x == 0 && ((SP_HBM u32*)P.g.ctl)[SPG_C_OVERFLOW])
        __hip_atomic_fetch_add(&((SP_HBM unsigned long long*)P.err)[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
