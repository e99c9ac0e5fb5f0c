// Observation + legal-action-mask encoder (reference: state/obs_repr.rs:126-630, consts.rs:20-28).
//
// One workgroup per decision row.  The whole (C,34) f32 plane stack of the row (v4: 137,632 B) is staged in LDS:
//   1. gather  : the row's table is copied from the field-major pool into a single-table LDS struct with all loads
//                in flight at once (the pool layout is lane-major, so one table's fields are 64 elements apart);
//                at the same time the LDS plane stack is zero-filled with 16-byte stores;
//   2. derive  : unconditional-tenpai discards (agent_helper.rs:100-197) — up to 14x34 shanten probes, spread over
//                the workgroup;
//   3. scatter : the few hundred non-zero cells / rows are written into the LDS stack;
//   4. stream  : LDS -> HBM with 16-byte stores, fully coalesced: exactly the algorithmic bytes reach HBM, once.
// The 46-byte mask is produced in the same pass.
//
// exp(-0.2 k) (obs_repr.rs:228,266) comes from a host-built LUT (glibc expf == the libm Rust links on Linux) so the
// decay rows are bit-exact; the RBF rows of v2/v3 (obs_repr.rs:79-90) use a host LUT as well.
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
    int C;                 // rows per obs
    const MjGatherEnt* gather;
    int n_gather;
    const float* decay_lut;   // [64]  expf(-0.2f * k)
    // v2/v3 RBF rows (obs_repr.rs:79-90), host-built with the same libm: value[n][i-1] for i in 1..intervals
    const float* rbf_score;   // [4096][9]  cap 500, 10 intervals, n = score/100 (unclamped, saturated at 4095: all-zero rows)
    const float* rbf_6;       // [256][2]   cap 6, 3 intervals   (honba, kyotaku)
    const float* rbf_12;      // [256][2]   cap 12, 3 intervals  (doras owned)
    const float* rbf_23;      // [256][3]   cap 23, 4 intervals  (doras unseen)
    int with_sp;              // v4: 1 = SP rows come from sp_buf, 0 = left zero (caller must know!)
    const float* sp_buf;
};

#define ENC_THREADS 256

template <class LN> MJD u64 discard_candidates_aka_enc(const LN& L, int s) {  // agent_helper.rs:35-79
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

__global__ __launch_bounds__(ENC_THREADS) void mj_k_encode(EncParams P) {
    extern __shared__ float4 smem4[];
    float* obs = reinterpret_cast<float*>(smem4);
    const int C = P.C;
    const int n_cells = C * 34;
    // dynamic LDS carve (no static __shared__ in front: the base must stay 16-byte aligned)
    TableOne* st = reinterpret_cast<TableOne*>(obs + n_cells);
    unsigned long long* s_furiten = reinterpret_cast<unsigned long long*>(
        reinterpret_cast<char*>(st) + ((sizeof(TableOne) + 15) & ~(size_t)15));  // [34] any furiten wait per discard
    unsigned long long* s_yaku = s_furiten + 34;                                   // [34] any live wait with yaku
    unsigned long long* s_uncond_p = s_yaku + 34;
#define s_uncond (*s_uncond_p)

    const int tid = threadIdx.x;
    const int row = blockIdx.x;
    const uint32_t desc = P.rows[row];
    const int table = ROW_TABLE(desc), p = ROW_SEAT(desc);
    const bool at_kan_select = ROW_KAN(desc);
    const int version = P.version;

    // ---- 1. gather + zero
    {
        const char* src_base = reinterpret_cast<const char*>(P.blocks + (table >> 6));
        const int lane = table & 63;
        char* dst_base = reinterpret_cast<char*>(st);
        for (int i = tid; i < P.n_gather; i += ENC_THREADS) {
            MjGatherEnt e = P.gather[i];
            const char* s = src_base + e.src_off + lane * e.size;
            char* d = dst_base + e.dst_off;
            switch (e.size) {
                case 1: *reinterpret_cast<uint8_t*>(d) = *reinterpret_cast<const uint8_t*>(s); break;
                case 2: *reinterpret_cast<uint16_t*>(d) = *reinterpret_cast<const uint16_t*>(s); break;
                case 4: *reinterpret_cast<uint32_t*>(d) = *reinterpret_cast<const uint32_t*>(s); break;
                default: *reinterpret_cast<uint64_t*>(d) = *reinterpret_cast<const uint64_t*>(s); break;
            }
        }
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = tid; i < n_cells / 4; i += ENC_THREADS) smem4[i] = z;
        if (tid < 34) { s_furiten[tid] = 0; s_yaku[tid] = 0; }
        if (tid == 0) s_uncond = 0;
    }
    __syncthreads();

    LaneT<TableOne> L = {st, 0, &P.tables};
    const u32 cans = F1(cans, p);
    const int shanten = F1(shanten, p);
    const int oya_abs = F(kyoku) & 3;

    // ---- 2. unconditional-tenpai discards (agent_helper.rs:100-197), only consulted when shanten <= 1
    if ((cans & CAN_DISCARD) && shanten <= 1) {
        const int tiles_left = F(tiles_left);
        const int lst = F1(last_self_tsumo, p);
        const u8 pf = F1(pflags, p);
        Hand h = load_hand(L, p);
        const int ld3 = F1(len_div3, p);
        bool skip = tiles_left == 0 || (shanten == 1 && !F1(has_next_shanten, p));
        bool riichi_case = false;
        if (!skip) {
            if (lst != MJ_NONE) {
                if ((F1(waits, p) >> deaka(lst)) & 1) skip = true;
                else if (accepted(L, p)) { skip = true; riichi_case = true; }
            } else if (calc_all(P.tables, h, ld3) == -1) {
                skip = true;
            }
        }
        if (riichi_case) {
            if (tid == 0 && !(pf & PF_AT_FURITEN)) s_uncond = BIT(lst);  // 37-bit set, raw id
        } else if (!skip) {
            const u64 cand = (shanten == 1 ? F1(next_shanten, p) : F1(keep_shanten, p)) & ~F1(forbidden, p);
            const u64 disc = F1(discarded, p);
            for (int w = tid; w < 34 * 34; w += ENC_THREADS) {
                const int d = w / 34, t = w % 34;
                if (!((cand >> d) & 1)) continue;
                Hand g = h;
                g.dec(d);
                if (t == d || g.get(t) == 4) continue;
                g.inc(t);
                if (calc_all(P.tables, g, ld3) > -1) continue;
                if ((disc >> t) & 1) atomicOr(&s_furiten[d], 1ull);
                else if (F1(pub_seen, t) + h.get(t) < 4) {  // tiles_seen[t] != 4 (own hand before the discard)
                    if (seat_has_yaku(L, p, g, t, true)) atomicOr(&s_yaku[d], 1ull);
                }
            }
        }
        __syncthreads();
        if (!skip && tid == 0) {
            const u64 cand = (shanten == 1 ? F1(next_shanten, p) : F1(keep_shanten, p)) & ~F1(forbidden, p);
            u64 ret = 0;
            for (int d = 0; d < 34; d++)
                if (((cand >> d) & 1) && !s_furiten[d] && s_yaku[d]) ret |= BIT(d);
            const int akas = F1(akas_in_hand, p);
            for (int k = 0; k < 3; k++) {
                int t5 = 4 + 9 * k;
                if (((ret >> t5) & 1) && ((akas >> k) & 1)) {
                    ret |= BIT(34 + k);
                    if (!(h.get(t5) > 1)) ret &= ~BIT(t5);
                }
            }
            s_uncond = ret;
        }
        __syncthreads();
    }

    // ---- 3. scatter (wave 0; control flow is uniform across the wave)
    if (tid < 64) {
        const int lane = tid;
        auto fill = [&](int r, float v) {
            if (lane < 34) obs[r * 34 + lane] = v;
        };
        auto fill_rows = [&](int r, int n, float v) {
            for (int k = 0; k < n; k++) fill(r + k, v);
        };
        auto assign = [&](int r, int c, float v) {
            if (lane == 0) obs[r * 34 + c] = v;
        };
        auto int_encode = [&](int& idx, u32 n_in, int cap, bool one_hot, bool rescale, int rbf, const float* lut) {
            // obs_repr.rs:59-107
            int n = (int)min(n_in, (u32)cap);
            if (version == 1) {
                fill_rows(idx, n, 1.f);
                idx += cap;
                return;
            }
            if (one_hot) {
                fill(idx + n, 1.f);
                idx += cap + 1;
            }
            if (rescale) {
                fill(idx, (float)n / (float)cap);
                idx += 1;
            }
            if (version != 4 && rbf) {
                for (int i = 1; i < rbf; i++) fill(idx + i - 1, lut[i - 1]);
                idx += rbf - 1;
            }
        };
        u8* mask = P.masks + (size_t)row * 46;
        u64 mask_bits = 0;
        int idx = 0;
        Hand h = load_hand(L, p);
        // dora factor table for this kyoku (derived)
        int n_ind = F(n_dora_ind);
        u64 dora_set = 0;  // tiles with factor > 0
        for (int i = 0; i < n_ind; i++) dora_set |= BIT(tile_next(F1(dora_ind, i)));

        // hand thermometer (4) + akas (3)
        if (lane < 34) {
            int c = h.get(lane);
            for (int k = 0; k < c; k++) obs[(idx + k) * 34 + lane] = 1.f;
        }
        idx += 4;
        {
            int akas = F1(akas_in_hand, p);
            for (int i = 0; i < 3; i++)
                if ((akas >> i) & 1) fill(idx + i, 1.f);
        }
        idx += 3;
        // scores (rotated to the seat's perspective)
        int sc[4];
        for (int i = 0; i < 4; i++) sc[i] = F1(scores, (p + i) & 3);
        for (int i = 0; i < 4; i++) {
            int s = sc[i];
            fill(idx, (float)min(max(s, 0), 100000) / 100000.f);
            idx += 1;
            if (version == 2 || version == 3) {
                // IntegerEncoder(score as usize / 100, cap 500).rbf_intervals(10): 9 rows from the host LUT
                u32 n = (u32)(((unsigned long long)(long long)s) / 100ull > 4095ull ? 4095u : (u32)(((unsigned long long)(long long)s) / 100ull));
                int_encode(idx, n, 500, false, false, 10, P.rbf_score + (size_t)n * 9);
            } else if (version == 4) {
                fill(idx, (float)min(max(s, 0), 30000) / 30000.f);
                idx += 1;
            }
        }
        // rank (rankings.rs:8-21: stable sort by -score => ties favour the lower ABSOLUTE seat)
        {
            int my = F1(scores, p), rank = 0;
            for (int a = 0; a < 4; a++) {
                int s = F1(scores, a);
                if (s > my || (s == my && a < p)) rank++;
            }
            fill(idx + rank, 1.f);
        }
        idx += 4;
        const int kyoku_in_wind = F(kyoku) & 3;
        if (version == 1) fill_rows(idx, kyoku_in_wind, 1.f);
        else fill(idx + kyoku_in_wind, 1.f);
        idx += 4;
        {
            int cap = (version == 1 || version == 4) ? 10 : 6;
            int_encode(idx, F(honba), cap, false, version == 4, 3, P.rbf_6 + (size_t)F(honba) * 2);
            int_encode(idx, F(kyotaku), cap, false, version == 4, 3, P.rbf_6 + (size_t)F(kyotaku) * 2);
        }
        const int bakaze = table_bakaze(L), jikaze = seat_jikaze(L, p);
        assign(idx, bakaze, 1.f);
        assign(idx + 1, jikaze, 1.f);
        idx += 2;
        if (version >= 2) {
            int n = min(bakaze - T_E, 1) * 4 + kyoku_in_wind;
            int_encode(idx, n, 7, false, true, 0, nullptr);
        }
        // dora indicators (tile set, 7 rows)  obs_repr.rs:694-712
        {
            for (int i = 0; i < n_ind; i++) {
                int t = F1(dora_ind, i), td = deaka(t), k = 0;
                for (int j = 0; j < i; j++) k += deaka(F1(dora_ind, j)) == td;
                assign(idx + k, td, 1.f);
                if (is_aka(t)) fill(idx + 4 + (t - T_5MR), 1.f);
            }
            idx += 7;
        }
        // ---- kawa.  Perspective list of abs seat a = [None if (a-p)&3 < (oya-p)&3] ++ pool kawa[a]  (update.rs:819-824)
        int klen[4], kpad[4], max_kawa_len = 0;
        for (int r = 0; r < 4; r++) {
            int a = (p + r) & 3;
            kpad[r] = r < ((oya_abs - p) & 3) ? 1 : 0;
            klen[r] = kpad[r] + F1(kawa_len, a);
            max_kawa_len = max(max_kawa_len, klen[r]);
        }
        auto kawa_item = [&](int r, int i) -> u64 {  // i-th entry of relative seat r's list
            int j = i - kpad[r];
            return j < 0 ? 0ull : F2(kawa, (p + r) & 3, j);
        };
        auto enc_self_kawa = [&](u64 e, int base) {  // obs_repr.rs:714-734
            if (e & KW_VALID) {
                int nk = KW_NKAN(e);
                for (int k = 0; k < nk; k++) assign(base, deaka(KW_KAN(e, k)), 1.f);
                int t = KW_TILE(e);
                assign(base + 1, deaka(t), 1.f);
                if (is_aka(t)) fill(base + 2, 1.f);
                if (KW_DORA(e)) fill(base + 3, 1.f);
            }
        };
        auto enc_kawa = [&](u64 e, int base) {  // obs_repr.rs:736-773
            if (e & KW_VALID) {
                if (KW_HAS_CP(e)) {
                    assign(base, KW_CP_MIN(e), 1.f);
                    assign(base + 1, KW_CP_MAX(e), 1.f);
                }
                int nk = KW_NKAN(e);
                for (int k = 0; k < nk; k++) assign(base + 2, deaka(KW_KAN(e, k)), 1.f);
                int t = KW_TILE(e);
                assign(base + 3, deaka(t), 1.f);
                if (is_aka(t)) fill(base + 4, 1.f);
                if (KW_DORA(e)) fill(base + 5, 1.f);
                if (KW_TEDASHI(e)) fill(base + 6, 1.f);
                if (KW_RIICHI(e)) fill(base + 7, 1.f);
            }
        };
        {
            int n = min(klen[0], 6);
            for (int i = 0; i < n; i++) enc_self_kawa(kawa_item(0, i), idx + i * 4);
            idx += 24;
            n = min(klen[0], 18);
            for (int i = 0; i < n; i++) enc_self_kawa(kawa_item(0, klen[0] - 1 - i), idx + i * 4);
            idx += 72;
            if (version >= 3) {
                for (int turn = 0; turn < klen[0]; turn++) {
                    u64 e = kawa_item(0, turn);
                    if (e & KW_VALID) assign(idx, deaka(KW_TILE(e)), P.decay_lut[max_kawa_len - 1 - turn]);
                }
                idx += 1;
            }
        }
        for (int r = 1; r < 4; r++) {
            int n = min(klen[r], 6);
            for (int i = 0; i < n; i++) enc_kawa(kawa_item(r, i), idx + i * 8);
            idx += 48;
            n = min(klen[r], 18);
            for (int i = 0; i < n; i++) enc_kawa(kawa_item(r, klen[r] - 1 - i), idx + i * 8);
            idx += 144;
            if (version == 2) {
                int turn = 0;
                for (int i = 0; i < klen[r]; i++) {
                    u64 e = kawa_item(r, i);
                    if (!(e & KW_VALID)) continue;
                    int rr = min(turn / 6, 2), td = deaka(KW_TILE(e));
                    assign(idx + rr, td, 1.f);
                    if (KW_TEDASHI(e)) assign(idx + 3 + rr, td, 1.f);
                    turn++;
                }
                idx += 6;
            } else if (version >= 3) {
                for (int turn = 0; turn < klen[r]; turn++) {
                    u64 e = kawa_item(r, turn);
                    if (!(e & KW_VALID)) continue;
                    int td = deaka(KW_TILE(e));
                    float v = P.decay_lut[max_kawa_len - 1 - turn];
                    assign(idx, td, v);
                    if (KW_TEDASHI(e)) assign(idx + 1, td, v);
                    if (KW_RIICHI(e)) assign(idx + 2, td, v);
                }
                idx += 3;
            }
        }
        fill(idx, (float)F(tiles_left) / 69.f);
        idx += 1;

        // doras_owned[rel] / doras_seen (derived; update.rs:733-808,955-960)
        int akas_seen_n = __popc((int)(F(pub_aka_seen) | F1(akas_in_hand, p)) & 7);
        int doras_seen = akas_seen_n;
        for (int t = 0; t < 34; t++)
            if ((dora_set >> t) & 1) doras_seen += dora_factor(L, t) * (F1(pub_seen, t) + h.get(t));
        for (int r = 0; r < 4; r++) {
            int a = (p + r) & 3, owned = 0;
            int nf = F1(fuuro_n, a);
            for (int k = 0; k < nf; k++)
                for (int j = 0; j < 4; j++) {
                    int t = F3(fuuro, a, k, j);
                    if (t == MJ_NONE) continue;
                    owned += dora_factor(L, deaka(t)) + (is_aka(t) ? 1 : 0);
                }
            int na = F1(ankan_n, a);
            for (int k = 0; k < na; k++) {
                int t = F2(ankan, a, k);
                owned += 4 * dora_factor(L, t) + ((t == T_5M || t == T_5P || t == T_5S) ? 1 : 0);
            }
            if (r == 0) {
                owned += __popc(F1(akas_in_hand, p) & 7);
                for (int t = 0; t < 34; t++)
                    if ((dora_set >> t) & 1) owned += dora_factor(L, t) * h.get(t);
            }
            int_encode(idx, (u32)(owned & 0xFF), 12, false, true, 3, P.rbf_12 + (size_t)(owned & 0xFF) * 2);
        }
        {
            u32 unseen = (u32)((n_ind * 4 + 3 - doras_seen) & 0xFF);
            int_encode(idx, unseen, 23, false, true, 4, P.rbf_23 + (size_t)unseen * 3);
        }
        // kawa_overview: the Some() entries of each kawa (tile sets, 4 x 7 rows)
        for (int r = 0; r < 4; r++) {
            int a = (p + r) & 3, n = F1(kawa_len, a);
            u64 c0 = 0, c1 = 0;  // 2-bit occurrence counter per tile id, bit-sliced over the 34 ids
            for (int i = 0; i < n; i++) {
                u64 e = F2(kawa, a, i);
                if (!(e & KW_VALID)) continue;
                int t = KW_TILE(e), td = deaka(t);
                int k = (int)((c0 >> td) & 1) + 2 * (int)((c1 >> td) & 1);
                assign(idx + k, td, 1.f);
                u64 b = BIT(td), carry = c0 & b;
                c0 ^= b;
                c1 ^= carry;
                if (is_aka(t)) fill(idx + 4 + (t - T_5MR), 1.f);
            }
            idx += 7;
        }
        // fuuro_overview 4 x 4 x 5 (obs_repr.rs:303-321): row = #same tile earlier in the set
        for (int r = 0; r < 4; r++) {
            int a = (p + r) & 3, nf = F1(fuuro_n, a);
            for (int k = 0; k < nf; k++) {
                for (int j = 0; j < 4; j++) {
                    int t = F3(fuuro, a, k, j);
                    if (t == MJ_NONE) continue;
                    int td = deaka(t), i = 0;
                    for (int q = 0; q < j; q++) {
                        int u = F3(fuuro, a, k, q);
                        i += u != MJ_NONE && deaka(u) == td;
                    }
                    assign(idx + i, td, 1.f);
                    if (is_aka(t)) fill(idx + 4, 1.f);
                }
                idx += 5;
            }
            idx += (4 - nf) * 5;
        }
        for (int r = 0; r < 4; r++) {
            int a = (p + r) & 3, na = F1(ankan_n, a);
            for (int k = 0; k < na; k++) assign(idx, F2(ankan, a, k), 1.f);
            idx += 1;
        }
        if (version >= 2) {
            if (lane < 34) obs[idx * 34 + lane] = (float)(F1(pub_seen, lane) + h.get(lane)) / 4.f;
            idx += 1;
            for (int pass = 0; pass < 2; pass++)
                for (int r = 1; r < 4; r++) {
                    int a = (p + r) & 3;
                    int su = pass == 0 ? F1(last_tedashi, a) : F1(riichi_sutehai, a);
                    if (su & SU_VALID) {
                        int t = su & 63;
                        assign(idx, deaka(t), 1.f);
                        if (is_aka(t)) fill(idx + 1, 1.f);
                        if (su & SU_DORA) fill(idx + 2, 1.f);
                    }
                    idx += 3;
                }
        }
        for (int r = 1; r < 4; r++)
            if (declared(L, (p + r) & 3)) fill(idx + r - 1, 1.f);
        idx += 3;
        for (int r = 1; r < 4; r++)
            if (accepted(L, (p + r) & 3)) fill(idx + r - 1, 1.f);
        idx += 3;
        {
            u64 w = F1(waits, p);
            if (lane < 34 && ((w >> lane) & 1)) obs[idx * 34 + lane] = 1.f;
            idx += 1;
        }
        if (F1(pflags, p) & PF_AT_FURITEN) fill(idx, 1.f);
        idx += 1;
        if (version == 1) {
            fill_rows(idx, min(shanten, 6), 1.f);
            idx += 6;
        } else {
            fill(idx + min(shanten, 6), 1.f);
            idx += 7;
        }
        if (accepted(L, p)) fill(idx, 1.f);
        idx += 1;
        if (at_kan_select) fill(idx, 1.f);
        idx += 1;

        if (cans & CAN_PASS) {
            int t = F1(last_kawa_tile, p), td = deaka(t);
            assign(idx, td, 1.f);
            if (is_aka(t)) fill(idx + 1, 1.f);
            if ((dora_set >> td) & 1) fill(idx + 2, 1.f);
            if (!at_kan_select) mask_bits |= BIT(45);
            else if (cans & CAN_DAIMINKAN) mask_bits |= BIT(td);
        }
        idx += 3;
        if (cans & CAN_DISCARD) {
            u64 dc = discard_candidates_aka_enc(L, p);
            u64 dc34 = (dc & 0x3FFFFFFFFull) | (((dc >> 34) & 1) << 4) | (((dc >> 35) & 1) << 13) | (((dc >> 36) & 1) << 22);
            if (lane < 34 && ((dc34 >> lane) & 1)) obs[idx * 34 + lane] = 1.f;
            if (!at_kan_select) mask_bits |= dc & 0x1FFFFFFFFFull;
            u64 ks = F1(keep_shanten, p), ns = F1(next_shanten, p);
            if (lane < 34 && ((ks >> lane) & 1)) obs[(idx + 1) * 34 + lane] = 1.f;
            if (lane < 34 && ((ns >> lane) & 1)) obs[(idx + 2) * 34 + lane] = 1.f;
            if (shanten <= 1) {
                u64 u = s_uncond;
                u64 u34 = (u & 0x3FFFFFFFFull) | (((u >> 34) & 1) << 4) | (((u >> 35) & 1) << 13) | (((u >> 36) & 1) << 22);
                if (lane < 34 && ((u34 >> lane) & 1)) obs[(idx + 3) * 34 + lane] = 1.f;
            }
            if (declared(L, p)) fill(idx + 4, 1.f);
        }
        idx += 5;
        auto flag_row = [&](u32 bit, int action) {
            if (cans & bit) {
                fill(idx, 1.f);
                if (!at_kan_select) mask_bits |= BIT(action);
            }
            idx += 1;
        };
        flag_row(CAN_RIICHI, 37);
        flag_row(CAN_CHI_LOW, 38);
        flag_row(CAN_CHI_MID, 39);
        flag_row(CAN_CHI_HIGH, 40);
        flag_row(CAN_PON, 41);
        flag_row(CAN_DAIMINKAN, 42);
        if (cans & CAN_ANKAN) {
            u64 c = F1(ankan_cand, p);
            if (lane < 34 && ((c >> lane) & 1)) obs[idx * 34 + lane] = 1.f;
            if (at_kan_select) mask_bits |= c;
            else mask_bits |= BIT(42);
        }
        idx += 1;
        if (cans & CAN_KAKAN) {
            u64 c = F1(kakan_cand, p);
            if (lane < 34 && ((c >> lane) & 1)) obs[idx * 34 + lane] = 1.f;
            if (at_kan_select) mask_bits |= c;
            else mask_bits |= BIT(42);
        }
        idx += 1;
        if (cans & CAN_AGARI) {
            fill(idx, 1.f);
            if (!at_kan_select) mask_bits |= BIT(43);
        }
        idx += 1;
        flag_row(CAN_RYUKYOKU, 44);
        // v4 SP block (rows idx .. idx+122) is written by mj_k_sp_rows (mj_sp.hip) straight into HBM after this kernel.
        if (lane < 46) mask[lane] = (u8)((mask_bits >> lane) & 1);
    }
    __syncthreads();

    // ---- 4. stream out
    float4* dst = reinterpret_cast<float4*>(P.obs + (size_t)row * n_cells);
    for (int i = tid; i < n_cells / 4; i += ENC_THREADS) dst[i] = smem4[i];
}
