// Observation + legal-action-mask encoder (reference: state/obs_repr.rs:126-630, consts.rs:20-28).
//
// One workgroup (256 threads) per decision row; the (C,34) f32 plane stack (v4: 137,632 B) goes to HBM exactly
// once, as 16-byte coalesced stores out of an LDS tile:
//   1. load    : the row's table record (TableOne, written contiguously by mj_k_snapshot) is read into LDS with
//                16-byte loads;
//   2. derive  : unconditional-tenpai discards (agent_helper.rs:100-197: up to 14x34 shanten probes + yaku checks)
//                spread over the workgroup; per-seat scalars (rank, dora counts, kawa lengths) by single lanes;
//   3. record  : the scatter tasks run ONCE and leave what they assign in an op list in LDS (cell, value) + (row, fill value).
//      The scatter is THREAD-PER-TASK, not row-serial: lane t < 34 owns tile id t, one lane per kawa entry, per meld,
//      per scalar block — each lane has a short dependency chain and only idempotent or provably unique stores, so
//      the whole plane stack is decoded in a few hundred cycles of wall time instead of a 1000-load serial chain.
//   4. PASSES x { zero a (C/PASSES)-row LDS tile; apply the ops that fall into it; stream the tile out }.
//      Small tiles let 5 workgroups share a CU, so that one group's HBM stores overlap another's record phase.
// The 46-byte mask is produced in the same pass.  The v4 SP block (rows 889..1011) is written by mj_sp.hip.
//
// Row offsets are compile-time per version (Appendix C of SURVEY.md; each total equals consts.rs:22-25).
// exp(-0.2 k) (obs_repr.rs:228,266) and the v2/v3 RBF rows (obs_repr.rs:79-90) come from host-built LUTs
// (glibc expf, the libm Rust links on Linux) so those rows are bit-exact as well.
#include <hip/hip_runtime.h>

#include "mj_rules.h"

struct EncParams {
    const TableBlock* blocks;
    const uint32_t* rows;  // row descriptors (ROW_PACK)
    int n_rows;
    MjTablesDev tables;
    float* obs;            // [n_rows][C][34]
    uint8_t* masks;        // [n_rows][46]
    int version;
    int C;
    const TableOne* snap;     // per-table contiguous records written by mj_k_snapshot
    const float* decay_lut;   // [64]  expf(-0.2f * k)
    const float* rbf_score;   // [4096][9]  cap 500, 10 intervals, n = score/100 saturated at 4095 (all-zero rows)
    const float* rbf_6;       // [256][2]   cap 6, 3 intervals   (honba, kyotaku)
    const float* rbf_12;      // [256][2]   cap 12, 3 intervals  (doras owned)
    const float* rbf_23;      // [256][3]   cap 23, 4 intervals  (doras unseen)
    int* err_flag;            // [1] set when a row's op list overflowed (mj_counters reports it with the SP overflows: must stay 0)
};

#define ENC_THREADS 256
// obs v4: rows 889 .. 1011 (the SP block) belong to mj_k_sp; row 889 is only 8-byte aligned, so the split is at row 890 (889 leaves here as zeros)
constexpr int enc_rows_written(int version) { return version == 1 ? 938 : version == 2 ? 942 : version == 3 ? 934 : 890; }
#ifndef ENC_PASSES
#define ENC_PASSES 8   // output passes per decision row = LDS tile of C / 8 rows.  With the op list a pass is {zero, apply, stream}: more,
                       // smaller tiles only cost barriers and buy resident workgroups (5 per CU at 8 passes).  Measured (round 4, one box,
                       // mj_k_encode<3> / <4> per launch): rounds 1-3 code 1.873 / 1.888 ms -> op list with 4 passes 1.756 / 1.790 ->
                       // 6 passes 1.571 / 1.776 -> 8 passes 1.573 / 1.709 ms (0.67 of the HBM spec peak for both versions)
#endif
#ifndef ENC_OPS
#define ENC_OPS 1024   // cell assignments of one decision row: < 1,000 by construction (96 displayed discards x <= 7 cells, 64 meld tiles,
                       // 14 hand tiles x 4, the 34-wide rows); measured maximum 395 (enc_ops_high_water); an overflow is reported, never silent
#endif
#define ENC_FILLS 512  // whole-row fills of one decision row: <= 4 flag rows per displayed discard + ~100 scalar rows (measured maximum 145)
#ifndef ENC_NT
#define ENC_NT 1   // non-temporal stores for the plane stack: it is written once and never re-read by this kernel
#endif

#ifdef MJ_EMU
// host emulation only: the largest op / fill lists seen (MJ_ENC_STATS=1 prints them at exit) — how ENC_OPS / ENC_FILLS were sized
static int g_enc_max_ops = 0, g_enc_max_fills = 0;
static inline void enc_ops_high_water(int n_ops, int n_fills) {
    static bool reg = false;
    if (!reg) {
        reg = true;
        if (getenv("MJ_ENC_STATS")) atexit([] { fprintf(stderr, "[enc] op list high water: %d ops, %d fills\n", g_enc_max_ops, g_enc_max_fills); });
    }
    if (n_ops > g_enc_max_ops) g_enc_max_ops = n_ops;
    if (n_fills > g_enc_max_fills) g_enc_max_fills = n_fills;
}
#endif

// ---------------------------------------------------------------- static row map
template <int V>
struct Lay {
    static constexpr int hand = 0;
    static constexpr int akas = 4;
    static constexpr int scores = 7;
    static constexpr int score_stride = V == 1 ? 1 : V == 4 ? 2 : 10;
    static constexpr int rank = scores + 4 * score_stride;
    static constexpr int kyoku = rank + 4;
    static constexpr int honba = kyoku + 4;
    static constexpr int hk_rows = V == 1 ? 10 : V == 4 ? 1 : 2;
    static constexpr int kyotaku = honba + hk_rows;
    static constexpr int kaze = kyotaku + hk_rows;
    static constexpr int kig = kaze + 2;                      // kyoku-in-game (v>=2)
    static constexpr int dora_ind = kig + (V >= 2 ? 1 : 0);
    static constexpr int self_kawa = dora_ind + 7;            // 6x4 + 18x4
    static constexpr int self_decay = self_kawa + 96;         // v>=3
    static constexpr int opp0 = self_decay + (V >= 3 ? 1 : 0);
    static constexpr int opp_extra = V == 2 ? 6 : V >= 3 ? 3 : 0;
    static constexpr int opp_stride = 48 + 144 + opp_extra;
    static constexpr int tiles_left = opp0 + 3 * opp_stride;
    static constexpr int doras_owned = tiles_left + 1;
    static constexpr int owned_stride = V == 1 ? 12 : V == 4 ? 1 : 3;
    static constexpr int doras_unseen = doras_owned + 4 * owned_stride;
    static constexpr int kawa_ov = doras_unseen + (V == 1 ? 23 : V == 4 ? 1 : 4);
    static constexpr int fuuro = kawa_ov + 28;
    static constexpr int ankan = fuuro + 80;
    static constexpr int seen = ankan + 4;                    // v>=2: tiles_seen 1 + tedashi 9 + riichi tile 9
    static constexpr int tedashi = seen + 1;
    static constexpr int riichi_tile = tedashi + 9;
    static constexpr int riichi_flags = ankan + 4 + (V >= 2 ? 19 : 0);
    static constexpr int waits = riichi_flags + 6;
    static constexpr int furiten = waits + 1;
    static constexpr int shanten = furiten + 1;
    static constexpr int self_riichi = shanten + (V == 1 ? 6 : 7);
    static constexpr int kan_select = self_riichi + 1;
    static constexpr int target = kan_select + 1;
    static constexpr int discard = target + 3;
    static constexpr int cans = discard + 5;                  // riichi, chi x3, pon, daiminkan, ankan, kakan, agari, ryukyoku
    static constexpr int sp = cans + 10;
    static constexpr int total = sp + (V == 4 ? 123 : 0);
};
static_assert(Lay<1>::total == 938 && Lay<2>::total == 942 && Lay<3>::total == 934 && Lay<4>::total == 1012, "row map");

struct EncDerived {  // per-row scalars computed once in phase 2
    unsigned long long uncond;   // 37-bit set (raw ids)
    unsigned long long furiten[34], yaku[34];
    int klen[4], kpad[4], max_kawa_len;
    int owned[4], doras_seen, rank;
    unsigned long long dora_set, dc;  // tiles with dora factor > 0; discard candidates (37-bit)
    // unconditional-tenpai scan (phase 2a): partial min-plus merges of the hand's four table rows (mj_algo.h sh_merge)
    unsigned long long r2[6], r3[4];   // two-suit merges, "three other suits" merges
    unsigned long long rowt[34];       // row of h + t in suit(t)
    unsigned long long rowd[34];       // row of h - d in suit(d) (candidate discards only)
    unsigned long long U[34][3];       // candidate d, k-th other suit: merge(two untouched suits, rowd[d])
    float rowfill[256];                // per pass: value a whole tile row is filled with, < 0 = not filled
    // The plane stack as an OP LIST (round 4): the scatter tasks run ONCE per decision row and record what they assign; every
    // output pass then only zeroes its tile, applies the ops that fall into it and streams it out.  (Rounds 1-3 re-ran the whole
    // task chain in every pass with the stores of the other passes masked off: four times the address arithmetic, kawa decoding
    // and LUT reads, during which the workgroup issued no HBM store.)
    int n_ops, n_fills;
    float op_val[ENC_OPS];
    float fill_val[ENC_FILLS];
    unsigned short op_cell[ENC_OPS];   // row * 34 + column
    unsigned short fill_row[ENC_FILLS];
};

template <class LN> MJD u64 enc_discard_candidates_aka(const LN& L, int s) {  // agent_helper.rs:35-79
    if (accepted(L, s)) return BIT(F1(last_self_tsumo, s));
    Hand h = load_hand(L, s);
    u64 have = h.nonzero_mask();
    u64 ret;
    if (declared(L, s)) ret = have & (F1(shanten, s) == 1 ? F1(next_shanten, s) : F1(keep_shanten, s));
    else ret = have & ~F1(forbidden, s);
    int akas = F1(akas_in_hand, s);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        int t5 = 4 + 9 * k;
        if (((ret >> t5) & 1) && ((akas >> k) & 1)) {
            ret |= BIT(34 + k);
            if (!(h.get(t5) > 1)) ret &= ~BIT(t5);
        }
    }
    return ret;
}
MJD u64 fold37(u64 m) {  // 37-bit raw-id set -> 34-bit deaka'd set
    return (m & 0x3FFFFFFFFull) | (((m >> 34) & 1) << 4) | (((m >> 35) & 1) << 13) | (((m >> 36) & 1) << 22);
}

template <int V>
__global__ __launch_bounds__(ENC_THREADS) void mj_k_encode(EncParams P) {
    typedef Lay<V> O;
    constexpr int C = O::total;
    constexpr int CW = enc_rows_written(V);  // the rows THIS kernel writes: all, or (v4) those below the SP block, which mj_k_sp writes whole
    static_assert(CW % 2 == 0 || CW == C, "the streamed rows end on a 16-byte boundary");
    constexpr int TILE_ROWS = ((CW + ENC_PASSES - 1) / ENC_PASSES + 1) & ~1;  // even -> tile bytes % 16 == 0
    MJ_DYN_SHARED(float4, smem4);
    float* tile = reinterpret_cast<float*>(smem4);
    TableOne* st = reinterpret_cast<TableOne*>(tile + TILE_ROWS * 34);
    EncDerived* D = reinterpret_cast<EncDerived*>(reinterpret_cast<char*>(st) + ((sizeof(TableOne) + 15) & ~(size_t)15));

    const int tid = threadIdx.x;
    const int row = blockIdx.x;
    const uint32_t desc = P.rows[row];
    const int table = ROW_TABLE(desc), p = ROW_SEAT(desc);
    const bool at_kan_select = ROW_KAN(desc);

    // ---- 1. load the table record (contiguous, 16-byte loads)
    {
        const float4* src = reinterpret_cast<const float4*>(P.snap + table);
        float4* dst4 = reinterpret_cast<float4*>(st);
        for (int i = tid; i < (int)(sizeof(TableOne) / 16); i += ENC_THREADS) dst4[i] = src[i];
        if (tid < 34) { D->furiten[tid] = 0; D->yaku[tid] = 0; }
        if (tid == 0) { D->uncond = 0; D->n_ops = 0; D->n_fills = 0; }
    }
    __syncthreads();

    LaneT<TableOne> L = {st, 0, &c_mj_tables};
    const u32 cans = F1(cans, p);
    const int shanten = F1(shanten, p);
    const int oya_abs = F(kyoku) & 3;
    const Hand h = load_hand(L, p);

    // ---- 2a. unconditional-tenpai discards (agent_helper.rs:100-197); only consulted when shanten <= 1
    bool uncond_scan = false;
    if ((cans & CAN_DISCARD) && shanten <= 1) {
        const int lst = F1(last_self_tsumo, p);
        const int ld3 = F1(len_div3, p);
        bool skip = F(tiles_left) == 0 || (shanten == 1 && !F1(has_next_shanten, p));
        if (!skip) {
            if (lst != MJ_NONE) {
                if ((F1(waits, p) >> deaka(lst)) & 1) skip = true;
                else if (accepted(L, p)) {
                    skip = true;
                    if (tid == 0 && !(F1(pflags, p) & PF_AT_FURITEN)) D->uncond = BIT(lst);  // raw id
                }
            } else if (calc_all(c_mj_tables, h, ld3) == -1) {
                skip = true;
            }
        }
        if (!skip) {
            uncond_scan = true;
            const u64 cand = (shanten == 1 ? F1(next_shanten, p) : F1(keep_shanten, p)) & ~F1(forbidden, p);
            const u64 disc = F1(discarded, p);
            // Is h - d + t a complete hand?  (h differs from the probed hand in at most two suits, so the probe is one
            // table gather + one final merge step on top of merges shared by the whole scan — see mj_algo.h sh_merge —
            // instead of a from-scratch calc_all with four gathers and two full merges.)
            const ShBase B = sh_base(c_mj_tables, h);
            if (tid < 6) {
                const int a = tid < 3 ? 0 : tid < 5 ? 1 : 2, b = tid < 3 ? tid + 1 : tid < 5 ? tid - 1 : 3;
                D->r2[tid] = sh_merge(B.row[a], B.row[b], ld3);
            } else if (tid >= 64 && tid < 98) {
                const int t = tid - 64, st = sh_suit(t);
                D->rowt[t] = h.get(t) < 4 ? sh_load(c_mj_tables, st, B.key[st] + sh_pow(t)) : 0ull;
                D->rowd[t] = ((cand >> t) & 1) ? sh_load(c_mj_tables, st, B.key[st] - sh_pow(t)) : 0ull;
            }
            __syncthreads();
            if (tid < 4) {
                const u64 pr = tid == 0 ? D->r2[3] : tid == 1 ? D->r2[1] : D->r2[0];
                D->r3[tid] = sh_merge(pr, B.row[tid == 3 ? 2 : 3], ld3);
            } else if (tid >= 64 && tid < 64 + 34 * 3) {
                const int d = (tid - 64) / 3, k = (tid - 64) % 3;
                if ((cand >> d) & 1) {
                    const int sd = sh_suit(d), st = k + (k >= sd);  // k-th suit != sd
                    int x = -1, y = -1;  // the two suits other than sd and st
                    for (int q = 0; q < 4; q++)
                        if (q != sd && q != st) { if (x < 0) x = q; else y = q; }
                    D->U[d][k] = sh_merge(D->r2[sh_pair_idx(x, y)], D->rowd[d], ld3);
                }
            }
            __syncthreads();
            for (int w = tid; w < 34 * 34; w += ENC_THREADS) {
                const int d = w / 34, t = w % 34;
                if (!((cand >> d) & 1)) continue;
                const int ht = h.get(t), hd = h.get(d);
                if (t == d || ht == 4) continue;
                const int st = sh_suit(t), sd = sh_suit(d);
                int fin;
                if (sd == st) fin = sh_final(D->r3[st], sh_load(c_mj_tables, st, B.key[st] + sh_pow(t) - sh_pow(d)), ld3);
                else fin = sh_final(D->U[d][st - (st > sd)], D->rowt[t], ld3);
                const int yt = (int)((YAOKYUU_MASK >> t) & 1), yd = (int)((YAOKYUU_MASK >> d) & 1);
                const int pairs = B.pairs - (hd == 2) + (ht == 1), kinds = B.kinds - (hd == 1) + (ht == 0);
                const int kpairs = B.kpairs - (yd && hd == 2) + (yt && ht == 1), kkinds = B.kkinds - (yd && hd == 1) + (yt && ht == 0);
                if (sh_finish(fin, ld3, pairs, kinds, kpairs, kkinds) > -1) continue;
                Hand g = h;
                g.dec(d);
                g.inc(t);
                if ((disc >> t) & 1) atomicOr(&D->furiten[d], 1ull);
                else if (F1(pub_seen, t) + ht < 4) {  // tiles_seen (hand before the discard) != 4
                    if (seat_has_yaku(L, p, g, t, true)) atomicOr(&D->yaku[d], 1ull);
                }
            }
        }
    }
    // ---- 2b. per-seat scalars, one lane each
    if (tid == 64) {
        int mx = 0;
        for (int r = 0; r < 4; r++) {
            int a = (p + r) & 3;
            int pad = r < ((oya_abs - p) & 3) ? 1 : 0;  // pad_kawa_at_start seen from seat p (update.rs:819-824)
            int len = pad + F1(kawa_len, a);
            D->kpad[r] = pad;
            D->klen[r] = len;
            mx = max(mx, len);
        }
        D->max_kawa_len = mx;
        int my = F1(scores, p), rank = 0;  // rankings.rs:8-21: stable sort by -score, ties -> lower absolute seat
        for (int a = 0; a < 4; a++) {
            int s = F1(scores, a);
            if (s > my || (s == my && a < p)) rank++;
        }
        D->rank = rank;
        D->dc = (cans & CAN_DISCARD) ? enc_discard_candidates_aka(L, p) : 0ull;
    }
    if (tid >= 65 && tid < 69) {  // doras_owned[r] (derived: melds of seat a; + own hand for r == 0)
        int r = tid - 65, a = (p + r) & 3, owned = 0;
        int nf = F1(fuuro_n, a);
        for (int k = 0; k < nf; k++)
            for (int j = 0; j < 4; j++) {
                int t = F3(fuuro, a, k, j);
                if (t != MJ_NONE) owned += dora_factor(L, deaka(t)) + (is_aka(t) ? 1 : 0);
            }
        int na = F1(ankan_n, a);
        for (int k = 0; k < na; k++) {
            int t = F2(ankan, a, k);
            owned += 4 * dora_factor(L, t) + ((t == T_5M || t == T_5P || t == T_5S) ? 1 : 0);
        }
        if (r == 0) {
            owned += __popc(F1(akas_in_hand, p) & 7);
            int n = F(n_dora_ind);
            for (int i = 0; i < n; i++) owned += h.get(tile_next(F1(dora_ind, i)));
        }
        D->owned[r] = owned & 0xFF;
    }
    if (tid == 69) {  // doras_seen = sum(tiles_seen x factor) + akas seen; dora_set
        int n = F(n_dora_ind), seen = __popc((F(pub_aka_seen) | F1(akas_in_hand, p)) & 7);
        u64 ds = 0;
        for (int i = 0; i < n; i++) {
            int d = tile_next(F1(dora_ind, i));
            ds |= BIT(d);
            seen += F1(pub_seen, d) + h.get(d);
        }
        D->doras_seen = seen;
        D->dora_set = ds;
    }
    __syncthreads();
    if (uncond_scan && tid == 0) {
        const u64 cand = (shanten == 1 ? F1(next_shanten, p) : F1(keep_shanten, p)) & ~F1(forbidden, p);
        u64 ret = 0;
        for (int d = 0; d < 34; d++)
            if (((cand >> d) & 1) && !D->furiten[d] && D->yaku[d]) ret |= BIT(d);
        const int akas = F1(akas_in_hand, p);
        for (int k = 0; k < 3; k++) {
            int t5 = 4 + 9 * k;
            if (((ret >> t5) & 1) && ((akas >> k) & 1)) {
                ret |= BIT(34 + k);
                if (!(h.get(t5) > 1)) ret &= ~BIT(t5);
            }
        }
        D->uncond = ret;
    }
    __syncthreads();

    const int klen_all[4] = {D->klen[0], D->klen[1], D->klen[2], D->klen[3]};
    const int max_kawa_len = D->max_kawa_len;
    const u64 dora_set = D->dora_set;

    // ---- 2c. kawa entry tasks (one lane per entry of the four perspective lists): everything that does not depend on
    // the pass is decoded once here — slots, the decay-row ownership scan, the kawa_overview occurrence index
    bool kt_on = false;
    int kt_r = 0, kt_i = 0, kt_len = 0, kt_t = 0, kt_td = 0, kt_nk = 0, kt_turn = 0, kt_ovk = 0;
    u64 kt_e = 0;
    bool kt_later_any = false, kt_later_ted = false, kt_later_rii = false;
    float kt_decay = 0.f;
    if (tid >= 80 && tid < 80 + 4 * MJ_KAWA_MAX + 4) {
        const int q = tid - 80;
        kt_r = q / (MJ_KAWA_MAX + 1);
        kt_i = q % (MJ_KAWA_MAX + 1);
        kt_len = klen_all[kt_r];
        if (kt_i < kt_len) {
            const int a = (p + kt_r) & 3, pad = D->kpad[kt_r];
            kt_e = kt_i < pad ? 0ull : F2(kawa, a, kt_i - pad);
            if (kt_e & KW_VALID) {
                kt_on = true;
                kt_t = KW_TILE(kt_e);
                kt_td = deaka(kt_t);
                kt_nk = KW_NKAN(kt_e);
                for (int j = pad; j < kt_i; j++) {  // earlier entries: overview row (obs_repr.rs:299-301), v2 turn index
                    const u64 f = F2(kawa, a, j - pad);
                    if (!(f & KW_VALID)) continue;
                    kt_turn++;
                    kt_ovk += deaka(KW_TILE(f)) == kt_td;
                }
                if (V >= 3) {
                    // decay rows (obs_repr.rs:223-233,259-276): sequential assigns in the reference, so the LAST entry
                    // holding a tile (resp. the last tedashi / riichi one) owns the cell
                    for (int j = kt_i + 1; j < kt_len; j++) {
                        const u64 f = F2(kawa, a, j - pad);
                        if ((f & KW_VALID) && deaka(KW_TILE(f)) == kt_td) {
                            kt_later_any = true;
                            kt_later_ted |= KW_TEDASHI(f) != 0;
                            kt_later_rii |= KW_RIICHI(f) != 0;
                        }
                    }
                    kt_decay = P.decay_lut[max_kawa_len - 1 - kt_i];
                }
            }
        }
    }

    // ---- 2d. RBF rows of the integer features (obs v2 / v3, obs_repr.rs:79-90): their LUT values do not depend on the pass
    // either — fetched once here instead of as a chain of dependent loads inside every pass (lanes 34..37: nine per score)
    float rbf_v[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (V == 2 || V == 3) {
        const float* src = nullptr;
        int cnt = 0;
        if (tid >= 34 && tid < 38) {
            const unsigned long long q = (unsigned long long)(long long)F1(scores, (p + tid - 34) & 3) / 100ull;  // `score as usize / 100`
            src = P.rbf_score + (size_t)(q > 4095ull ? 4095u : (u32)q) * 9;
            cnt = 9;
        } else if (tid >= 40 && tid < 44) {
            src = P.rbf_12 + (size_t)D->owned[tid - 40] * 2;
            cnt = 2;
        } else if (tid == 44) {
            src = P.rbf_23 + (size_t)((F(n_dora_ind) * 4 + 3 - D->doras_seen) & 0xFF) * 3;
            cnt = 3;
        }
#pragma unroll
        for (int k = 0; k < 9; k++)
            if (k < cnt) rbf_v[k] = src[k];
        if (tid == 38) {
            rbf_v[0] = P.rbf_6[F(honba) * 2];
            rbf_v[1] = P.rbf_6[F(honba) * 2 + 1];
            rbf_v[2] = P.rbf_6[F(kyotaku) * 2];
            rbf_v[3] = P.rbf_6[F(kyotaku) * 2 + 1];
        }
    }

    // ---- 3. record: every task assigns its cells / row fills ONCE, into the op list
    {
        auto put = [&](int r, int c, float v) {
            const int i = atomicAdd(&D->n_ops, 1);
            if (i < ENC_OPS) {
                D->op_cell[i] = (unsigned short)(r * 34 + c);
                D->op_val[i] = v;
            }
        };
        auto fillr = [&](int r, float v) {
            // `arr.fill(row, v)`: one entry, merged into the row while it is streamed out (no row is both filled and assigned
            // cell-wise in any obs version)
            const int i = atomicAdd(&D->n_fills, 1);
            if (i < ENC_FILLS) {
                D->fill_row[i] = (unsigned short)r;
                D->fill_val[i] = v;
            }
        };
        // obs_repr.rs:59-107 for one integer feature at row base `b`
        auto int_encode = [&](int b, u32 n_in, int cap, bool rescale, int rbf, int v0) {  // RBF values: rbf_v[v0 ..] (section 2d)
            int n = (int)min(n_in, (u32)cap);
            if (V == 1) {
                for (int k = 0; k < n; k++) fillr(b + k, 1.f);
                return;
            }
            if (rescale) {
                fillr(b, (float)n / (float)cap);
                b += 1;
            }
            if (V != 4 && rbf) {
#pragma unroll
                for (int i = 1; i < 10; i++)
                    if (i < rbf) fillr(b + i - 1, rbf_v[v0 + i - 1]);
            }
        };

        if (tid < 34) {
            // ---- A. lane == tile id
            const int t = tid;
            const int c = h.get(t);
            for (int k = 0; k < c; k++) put(O::hand + k, t, 1.f);
            if (V >= 2) put(O::seen, t, (float)(F1(pub_seen, t) + c) / 4.f);
            if ((F1(waits, p) >> t) & 1) put(O::waits, t, 1.f);
            if (cans & CAN_DISCARD) {
                if ((fold37(D->dc) >> t) & 1) put(O::discard, t, 1.f);
                if ((F1(keep_shanten, p) >> t) & 1) put(O::discard + 1, t, 1.f);
                if ((F1(next_shanten, p) >> t) & 1) put(O::discard + 2, t, 1.f);
                if (shanten <= 1 && ((fold37(D->uncond) >> t) & 1)) put(O::discard + 3, t, 1.f);
            }
            if ((cans & CAN_ANKAN) && ((F1(ankan_cand, p) >> t) & 1)) put(O::cans + 6, t, 1.f);
            if ((cans & CAN_KAKAN) && ((F1(kakan_cand, p) >> t) & 1)) put(O::cans + 7, t, 1.f);
        } else if (tid < 38) {
            // ---- B. scores (rotated)
            const int i = tid - 34;
            const int s = F1(scores, (p + i) & 3);
            const int b = O::scores + i * O::score_stride;
            fillr(b, (float)min(max(s, 0), 100000) / 100000.f);
            if (V == 2 || V == 3) {
                unsigned long long q = (unsigned long long)(long long)s / 100ull;  // `score as usize / 100`
                u32 n = q > 4095ull ? 4095u : (u32)q;
                int_encode(b + 1, n, 500, false, 10, 0);
            } else if (V == 4) {
                fillr(b + 1, (float)min(max(s, 0), 30000) / 30000.f);
            }
        } else if (tid == 38) {
            const int akas = F1(akas_in_hand, p);
            for (int i = 0; i < 3; i++)
                if ((akas >> i) & 1) fillr(O::akas + i, 1.f);
            fillr(O::rank + D->rank, 1.f);
            const int kw = F(kyoku) & 3;
            if (V == 1) {
                for (int k = 0; k < kw; k++) fillr(O::kyoku + k, 1.f);
            } else {
                fillr(O::kyoku + kw, 1.f);
            }
            const int cap = (V == 1 || V == 4) ? 10 : 6;
            int_encode(O::honba, F(honba), cap, V == 4, 3, 0);
            int_encode(O::kyotaku, F(kyotaku), cap, V == 4, 3, 2);
            const int bakaze = table_bakaze(L);
            put(O::kaze, bakaze, 1.f);
            put(O::kaze + 1, seat_jikaze(L, p), 1.f);
            if (V >= 2) int_encode(O::kig, min(bakaze - T_E, 1) * 4 + kw, 7, true, 0, 0);
        } else if (tid == 39) {
            // dora indicators tile set (obs_repr.rs:694-712) + tiles_left
            const int n = F(n_dora_ind);
            for (int i = 0; i < n; i++) {
                int t = F1(dora_ind, i), td = deaka(t), k = 0;
                for (int j = 0; j < i; j++) k += deaka(F1(dora_ind, j)) == td;
                put(O::dora_ind + k, td, 1.f);
                if (is_aka(t)) fillr(O::dora_ind + 4 + (t - T_5MR), 1.f);
            }
            fillr(O::tiles_left, (float)F(tiles_left) / 69.f);
        } else if (tid < 44) {
            const int r = tid - 40;
            const int ow = D->owned[r];
            int_encode(O::doras_owned + r * O::owned_stride, (u32)ow, 12, true, 3, 0);
        } else if (tid == 44) {
            const u32 unseen = (u32)((F(n_dora_ind) * 4 + 3 - D->doras_seen) & 0xFF);
            int_encode(O::doras_unseen, unseen, 23, true, 4, 0);
        } else if (tid == 45) {
            for (int r = 1; r < 4; r++) {
                if (declared(L, (p + r) & 3)) fillr(O::riichi_flags + r - 1, 1.f);
                if (accepted(L, (p + r) & 3)) fillr(O::riichi_flags + 3 + r - 1, 1.f);
            }
            if (F1(pflags, p) & PF_AT_FURITEN) fillr(O::furiten, 1.f);
            if (V == 1) {
                for (int k = 0; k < min(shanten, 6); k++) fillr(O::shanten + k, 1.f);
            } else {
                fillr(O::shanten + min(shanten, 6), 1.f);
            }
            if (accepted(L, p)) fillr(O::self_riichi, 1.f);
            if (at_kan_select) fillr(O::kan_select, 1.f);
        } else if (tid == 46) {
            if (cans & CAN_PASS) {  // target tile (obs_repr.rs:408-429)
                int t = F1(last_kawa_tile, p), td = deaka(t);
                put(O::target, td, 1.f);
                if (is_aka(t)) fillr(O::target + 1, 1.f);
                if ((dora_set >> td) & 1) fillr(O::target + 2, 1.f);
            }
            if ((cans & CAN_DISCARD) && declared(L, p)) fillr(O::discard + 4, 1.f);
        } else if (tid == 47) {
            if (cans & CAN_RIICHI) fillr(O::cans + 0, 1.f);
            if (cans & CAN_CHI_LOW) fillr(O::cans + 1, 1.f);
            if (cans & CAN_CHI_MID) fillr(O::cans + 2, 1.f);
            if (cans & CAN_CHI_HIGH) fillr(O::cans + 3, 1.f);
            if (cans & CAN_PON) fillr(O::cans + 4, 1.f);
            if (cans & CAN_DAIMINKAN) fillr(O::cans + 5, 1.f);
            if (cans & CAN_AGARI) fillr(O::cans + 8, 1.f);
            if (cans & CAN_RYUKYOKU) fillr(O::cans + 9, 1.f);
        } else if (tid == 48) {
            {  // mask (obs_repr.rs:422-562)
                u64 m = 0;
                if (!at_kan_select) {
                    if (cans & CAN_PASS) m |= BIT(45);
                    if (cans & CAN_DISCARD) m |= D->dc & 0x1FFFFFFFFFull;
                    if (cans & CAN_RIICHI) m |= BIT(37);
                    if (cans & CAN_CHI_LOW) m |= BIT(38);
                    if (cans & CAN_CHI_MID) m |= BIT(39);
                    if (cans & CAN_CHI_HIGH) m |= BIT(40);
                    if (cans & CAN_PON) m |= BIT(41);
                    if (cans & CAN_KAN) m |= BIT(42);
                    if (cans & CAN_AGARI) m |= BIT(43);
                    if (cans & CAN_RYUKYOKU) m |= BIT(44);
                } else {
                    if ((cans & CAN_PASS) && (cans & CAN_DAIMINKAN)) m |= BIT(deaka(F1(last_kawa_tile, p)));
                    if (cans & CAN_ANKAN) m |= F1(ankan_cand, p);
                    if (cans & CAN_KAKAN) m |= F1(kakan_cand, p);
                }
                u8* mask = P.masks + (size_t)row * 46;
                for (int i = 0; i < 46; i++) mask[i] = (u8)((m >> i) & 1);
            }
        } else if (tid < 55) {
            if (V >= 2) {  // last tedashi / riichi tile of the three opponents (obs_repr.rs:337-367)
                const int k = tid - 49, r = 1 + k % 3;
                const int a = (p + r) & 3;
                const int su = k < 3 ? F1(last_tedashi, a) : F1(riichi_sutehai, a);
                const int b = (k < 3 ? O::tedashi : O::riichi_tile) + (r - 1) * 3;
                if (su & SU_VALID) {
                    int t = su & 63;
                    put(b, deaka(t), 1.f);
                    if (is_aka(t)) fillr(b + 1, 1.f);
                    if (su & SU_DORA) fillr(b + 2, 1.f);
                }
            }
        } else if (tid >= 60 && tid < 64) {
            const int r = tid - 60, a = (p + r) & 3, na = F1(ankan_n, a);
            for (int k = 0; k < na; k++) put(O::ankan + r, F2(ankan, a, k), 1.f);
        } else if (tid >= 64 && tid < 80) {
            // ---- C. fuuro sets (obs_repr.rs:303-321): row = #same tile earlier in the set
            const int r = (tid - 64) >> 2, k = (tid - 64) & 3, a = (p + r) & 3;
            if (k < F1(fuuro_n, a)) {
                const int b = O::fuuro + (r * 4 + k) * 5;
                int tl[4];
                for (int j = 0; j < 4; j++) tl[j] = F3(fuuro, a, k, j);
                for (int j = 0; j < 4; j++) {
                    if (tl[j] == MJ_NONE) continue;
                    int td = deaka(tl[j]), i = 0;
                    for (int q = 0; q < j; q++) i += tl[q] != MJ_NONE && deaka(tl[q]) == td;
                    put(b + i, td, 1.f);
                    if (is_aka(tl[j])) fillr(b + 4, 1.f);
                }
            }
        } else if (kt_on) {
            // ---- D. one lane per kawa entry of the perspective lists (r = relative seat, i = index incl. start pad)
            const int r = kt_r, i = kt_i, len = kt_len, t = kt_t, td = kt_td, nk = kt_nk;
            const u64 e = kt_e;
            {   // kawa_overview = the Some() entries of a kawa as a tile set (obs_repr.rs:299-301)
                const int b = O::kawa_ov + r * 7;
                put(b + (kt_ovk & 3), td, 1.f);
                if (is_aka(t)) fillr(b + 4 + (t - T_5MR), 1.f);
            }
            // slots: first six, last eighteen
            int slots[2], ns = 0;
            if (r == 0) {
                if (i < 6) slots[ns++] = O::self_kawa + i * 4;
                if (len - 1 - i < 18) slots[ns++] = O::self_kawa + 24 + (len - 1 - i) * 4;
                for (int s = 0; s < ns; s++) {  // obs_repr.rs:714-734
                    const int b = slots[s];
                    for (int k = 0; k < nk; k++) put(b, deaka(KW_KAN(e, k)), 1.f);
                    put(b + 1, td, 1.f);
                    if (is_aka(t)) fillr(b + 2, 1.f);
                    if (KW_DORA(e)) fillr(b + 3, 1.f);
                }
            } else {
                const int ob = O::opp0 + (r - 1) * O::opp_stride;
                if (i < 6) slots[ns++] = ob + i * 8;
                if (len - 1 - i < 18) slots[ns++] = ob + 48 + (len - 1 - i) * 8;
                for (int s = 0; s < ns; s++) {  // obs_repr.rs:736-773
                    const int b = slots[s];
                    if (KW_HAS_CP(e)) {
                        put(b, KW_CP_MIN(e), 1.f);
                        put(b + 1, KW_CP_MAX(e), 1.f);
                    }
                    for (int k = 0; k < nk; k++) put(b + 2, deaka(KW_KAN(e, k)), 1.f);
                    put(b + 3, td, 1.f);
                    if (is_aka(t)) fillr(b + 4, 1.f);
                    if (KW_DORA(e)) fillr(b + 5, 1.f);
                    if (KW_TEDASHI(e)) fillr(b + 6, 1.f);
                    if (KW_RIICHI(e)) fillr(b + 7, 1.f);
                }
            }
            if (V >= 3) {
                if (r == 0) {
                    if (!kt_later_any) put(O::self_decay, td, kt_decay);
                } else {
                    const int b = O::opp0 + (r - 1) * O::opp_stride + 192;
                    if (!kt_later_any) put(b, td, kt_decay);
                    if (KW_TEDASHI(e) && !kt_later_ted) put(b + 1, td, kt_decay);
                    if (KW_RIICHI(e) && !kt_later_rii) put(b + 2, td, kt_decay);
                }
            } else if (V == 2 && r > 0) {
                // kt_turn = index among the Some() entries (obs_repr.rs:251-258)
                const int b = O::opp0 + (r - 1) * O::opp_stride + 192, rr = min(kt_turn / 6, 2);
                put(b + rr, td, 1.f);
                if (KW_TEDASHI(e)) put(b + 3 + rr, td, 1.f);
            }
        }
    }
    __syncthreads();
    const int n_ops = min(D->n_ops, ENC_OPS), n_fills = min(D->n_fills, ENC_FILLS);
    if (tid == 0 && (D->n_ops > ENC_OPS || D->n_fills > ENC_FILLS)) P.err_flag[0] = 1;  // capacity exceeded: the batch is reported as failed
#ifdef MJ_EMU
    enc_ops_high_water(D->n_ops, D->n_fills);
#endif

    // ---- 4. passes: zero a tile, apply the ops that fall into it, stream it out
    float4* dst = reinterpret_cast<float4*>(P.obs + (size_t)row * (C * 34));
    for (int pass = 0; pass < ENC_PASSES; pass++) {
        const int r0 = pass * TILE_ROWS, r1 = min(CW, r0 + TILE_ROWS);
        if (r0 >= CW) break;
        {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            constexpr int N4 = TILE_ROWS * 34 / 4;
#pragma unroll
            for (int k = 0; k < (N4 + ENC_THREADS - 1) / ENC_THREADS; k++) {
                const int i = tid + k * ENC_THREADS;
                if (i < N4) smem4[i] = z;
            }
            if (tid < TILE_ROWS) D->rowfill[tid] = -1.f;
        }
        __syncthreads();
        for (int i = tid; i < n_ops; i += ENC_THREADS) {
            const int cell = (int)D->op_cell[i] - r0 * 34;
            if (cell >= 0 && cell < (r1 - r0) * 34) tile[cell] = D->op_val[i];
        }
        for (int i = tid; i < n_fills; i += ENC_THREADS) {
            const int r = (int)D->fill_row[i] - r0;
            if (r >= 0 && r < r1 - r0) D->rowfill[r] = D->fill_val[i];
        }
        __syncthreads();
        {
            // Stream the tile out with 16-byte non-temporal stores (written once, never re-read by this kernel).
            typedef float vfloat4 __attribute__((ext_vector_type(4)));
            const int n4 = (r1 - r0) * 34 / 4;  // (r1-r0) is even
            vfloat4* d4 = reinterpret_cast<vfloat4*>(dst + (size_t)r0 * 34 / 4);
            const vfloat4* s4 = reinterpret_cast<const vfloat4*>(smem4);
            // Rows are only 32-byte aligned (C*136 B per row), so shift the lane->address map by the chunk's offset
            // inside its 128-byte line: every wave then stores a 1 KiB segment that starts on a 128-byte boundary and
            // no cache line is written in two halves by two different waves (PMC WRITE_SIZE showed 1.37x the
            // algorithmic bytes without this).
            const int shift = (int)((reinterpret_cast<uintptr_t>(d4) >> 4) & 7);
            constexpr int N4MAX = TILE_ROWS * 34 / 4;
#pragma unroll
            for (int k = 0; k < (N4MAX + 7 + ENC_THREADS - 1) / ENC_THREADS; k++) {
                const int i = tid - shift + k * ENC_THREADS;
                if (i < 0 || i >= n4) continue;
                vfloat4 v = s4[i];
                const int ra = (4 * i) / 34, off = 4 * i - 34 * ra;  // a 16-byte chunk touches at most two rows
                const float f0 = D->rowfill[ra], f1 = D->rowfill[min(ra + 1, TILE_ROWS - 1)];
                const float fa = f0, fb = off + 1 < 34 ? f0 : f1, fc = off + 2 < 34 ? f0 : f1, fd = off + 3 < 34 ? f0 : f1;
                if (fa >= 0.f) v.x = fa;
                if (fb >= 0.f) v.y = fb;
                if (fc >= 0.f) v.z = fc;
                if (fd >= 0.f) v.w = fd;
#if ENC_NT
                __builtin_nontemporal_store(v, d4 + i);
#else
                d4[i] = v;
#endif
            }
        }
        __syncthreads();
    }
}

// ================================================================ invisible ("oracle") observation, board.rs:679-782
// One workgroup per decision row; the whole 211|217 x 34 plane stack (<= 29.5 KB) is built in LDS in one pass and
// streamed out with 16-byte non-temporal stores.  Row map (v2..v4; v1 has a 6-row thermometer instead of 7+1):
//   per other seat (p+1, p+2, p+3): 4 hand thermometer, 3 aka, 7 shanten one-hot, 1 shanten/6, 1 waits, 1 furiten
//   then 69 x 2 yama (next draw first), 4 x 2 rinshan, 5 x 2 dora indicators (reveal order), 5 x 2 ura.
struct OracleEncParams {
    const uint32_t* rows;
    int n_rows;
    int version;
    const TableOne* snap;
    float* out;  // [n_rows][rows][34]
    int all_yama;  // dataset flavour (dataset/invisible.rs:217-224): every undrawn yama tile, not only the live ones
};

template <bool V1>
__global__ __launch_bounds__(ENC_THREADS) void mj_k_encode_oracle(OracleEncParams P) {
    constexpr int SEAT_ROWS = V1 ? 15 : 17;
    constexpr int ROWS = 3 * SEAT_ROWS + 138 + 8 + 10 + 10;
    constexpr int TILE_F = (ROWS * 34 + 3) & ~3;
    MJ_DYN_SHARED(float4, smem4);
    float* tile = reinterpret_cast<float*>(smem4);
    TableOne* st = reinterpret_cast<TableOne*>(tile + TILE_F);
    const int tid = threadIdx.x;
    const int row = blockIdx.x;
    const uint32_t desc = P.rows[row];
    const int table = ROW_TABLE(desc), p = ROW_SEAT(desc);
    {
        const float4* src = reinterpret_cast<const float4*>(P.snap + table);
        float4* dst4 = reinterpret_cast<float4*>(st);
        for (int i = tid; i < (int)(sizeof(TableOne) / 16); i += ENC_THREADS) dst4[i] = src[i];
        float4* t4 = reinterpret_cast<float4*>(tile);
        for (int i = tid; i < TILE_F / 4; i += ENC_THREADS) t4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    LaneT<TableOne> L = {st, 0, &c_mj_tables};
    auto fill = [&](int r, float v) {
        for (int c = 0; c < 34; c++) tile[r * 34 + c] = v;
    };
    auto encode_tile = [&](int r, int t) {
        tile[r * 34 + deaka(t)] = 1.f;
        if (is_aka(t)) fill(r + 1, 1.f);
    };
    if (tid < 34) {  // hand thermometer + waits, one tile kind per thread
        for (int k = 0; k < 3; k++) {
            const int s = (p + 1 + k) & 3, base = k * SEAT_ROWS;
            const int c = load_hand(L, s).get(tid);
            for (int i = 0; i < c; i++) tile[(base + i) * 34 + tid] = 1.f;
            if ((F1(waits, s) >> tid) & 1) tile[(base + SEAT_ROWS - 2) * 34 + tid] = 1.f;
        }
    } else if (tid < 64) {
        if (tid < 37) {  // per-seat scalars
            const int k = tid - 34, s = (p + 1 + k) & 3, base = k * SEAT_ROWS;
            const int akas = F1(akas_in_hand, s);
            for (int i = 0; i < 3; i++)
                if ((akas >> i) & 1) fill(base + 4 + i, 1.f);
            int n = F1(shanten, s);
            n = n < 0 ? 0 : n > 6 ? 6 : n;
            if (V1) {
                for (int i = 0; i < n; i++) fill(base + 7 + i, 1.f);
            } else {
                fill(base + 7 + n, 1.f);
                fill(base + 14, (float)n / 6.f);
            }
            if (F1(pflags, s) & PF_AT_FURITEN) fill(base + SEAT_ROWS - 1, 1.f);
        }
    } else {
        const int j = tid - 64;
        const int W0 = 3 * SEAT_ROWS, R0 = W0 + 138, D0 = R0 + 8, U0 = D0 + 10;
        if (j < 69) {
            if ((P.all_yama || j < (int)F(tiles_left)) && j < (int)F(yama_n)) encode_tile(W0 + 2 * j, F1(wall, 66 + F(yama_n) - 1 - j));
        } else if (j < 73) {
            const int i = j - 69, n = F(rinshan_n);
            if (i < n) encode_tile(R0 + 2 * i, F1(wall, 52 + n - 1 - i));
        } else if (j < 78) {
            const int i = j - 73;
            encode_tile(D0 + 2 * i, F1(wall, 60 - i));
        } else if (j < 83) {
            const int i = j - 78;
            encode_tile(U0 + 2 * i, F1(wall, 61 + i));
        }
    }
    __syncthreads();
    typedef float v4f __attribute__((ext_vector_type(4)));
    float* dst = P.out + (size_t)row * ROWS * 34;
    constexpr int N = ROWS * 34;  // 7174 / 7378 floats: row stride is 8-byte aligned, so split head / body / tail
    const int head = (int)(((16 - ((uintptr_t)dst & 15)) & 15) / 4);
    if (tid < head) dst[tid] = tile[tid];
    const int n4 = (N - head) / 4;
    for (int i = tid; i < n4; i += ENC_THREADS) {
        const float* s4 = tile + head + 4 * i;
        v4f v = {s4[0], s4[1], s4[2], s4[3]};
        __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(dst + head + 4 * i));
    }
    const int done = head + 4 * n4;
    if (tid < N - done) dst[done + tid] = tile[done + tid];
}

static size_t enc_oracle_lds_bytes(int version) {
    int rows = version == 1 ? 211 : 217;
    size_t tile = (size_t)((rows * 34 + 3) & ~3) * 4;
    return tile + ((sizeof(TableOne) + 15) & ~(size_t)15);
}
