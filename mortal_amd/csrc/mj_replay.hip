// Log replay for the dataset loader (reference: dataset/gameplay.rs:239-443 Gameplay::load_events_by_player).
//
// A table replays one mjai game log: its events arrive as the same packed words the arena's event log uses
// (mj_state.h LG_*), the step kernel's PlayerState handlers (mj_rules.h ev_*) are applied one event at a time — all four
// seats at once, public state once per table — and after every event the tracked seats that can act get a training
// sample: a policy row (obs + mask through the normal snapshot / encode kernels) plus the label the log implies
// (gameplay.rs:296-409: the next event, with the reach_accepted / dora skip and the hora look-ahead).
#include <hip/hip_runtime.h>

#include "mj_rules.h"

struct ReplayParams {
    TableBlock* blocks;
    int n_tables;
    const uint64_t* script;      // all logs' event words, concatenated
    const uint32_t* script_off;  // [n_tables + 1] word offsets
    uint32_t* cursor;            // [n_tables] next word (relative)
    uint32_t* ev_index;          // [n_tables] events applied so far
    uint8_t* kyoku_idx;          // [n_tables] kyoku counter of the loader (gameplay.rs:279)
    const uint8_t* tracked;      // [n_tables] bit s: samples wanted for seat s
    int always_include_kan_select;
    int deal_algo;
    int* block_rows;             // [n_blocks][2] row counts for mj_k_scan / mj_k_assign
    int32_t* label;              // [n_tables][4] label of the pending main row (valid when main_row >= 0)
    int32_t* kan_label;          // [n_tables][4] label (tile id) of the pending kan-select row
    unsigned long long* counters;  // [0] events applied, [1] finished logs
};

struct RpEvent {
    int type, actor, target, pai, c[4], tsumogiri, len;
};
MJD int rp_len(uint64_t w) {
    const int t = (int)(w & 15);
    const int tag = (int)((w >> LG_TAG_BIT) & 1);  // arena logs fed back as scripts: the tag word is skipped
    return t == LG_START_KYOKU ? (((w >> LG_SK_WALL_BIT) & 1) ? 27 : 10) : (t == LG_HORA ? 4 : t == LG_RYUKYOKU ? 3 : 1) + tag;
}
MJD RpEvent rp_decode(uint64_t w) {
    RpEvent e;
    e.type = (int)(w & 15);
    e.actor = (int)((w >> 4) & 3);
    e.target = (int)((w >> 6) & 3);
    e.pai = (int)((w >> 8) & 63);
    for (int k = 0; k < 4; k++) e.c[k] = (int)((w >> (14 + 6 * k)) & 63);
    e.tsumogiri = (int)((w >> 38) & 1);
    e.len = rp_len(w);
    return e;
}

// PlayerState::update (state/update.rs:41-122) of all four seats for one event; `w` points at its header word.
template <class LN> MJDN void rp_apply(const LN& L, const RpEvent& ev, const uint64_t* w, int deal_algo = 0) {
    // wall cursors for the invisible obs (dataset/gameplay.rs:281-300): a draw after a kan comes from the rinshan
    if (ev.type == LG_TSUMO) {
        u32 fl = F(flags);
        if (fl & TF_DEAL_FROM_RINSHAN) {
            F(flags) = fl & ~TF_DEAL_FROM_RINSHAN;
            if (F(rinshan_n) > 0) F(rinshan_n) -= 1;
        } else if (F(yama_n) > 0) {
            F(yama_n) -= 1;
        }
    } else if (ev.type == LG_ANKAN || ev.type == LG_KAKAN || ev.type == LG_DAIMINKAN) {
        F(flags) |= TF_DEAL_FROM_RINSHAN;
    }
    switch (ev.type) {
        case LG_START_KYOKU: {
            F(kyoku) = (u8)ev.c[0];
            F(honba) = (u8)((w[0] >> LG_HONBA_SHIFT) & 0xFF);
            F(kyotaku) = (u8)((w[0] >> LG_KYOTAKU_SHIFT) & 0xFF);
            F1(scores, 0) = (int)(uint32_t)w[1];
            F1(scores, 1) = (int)(uint32_t)(w[1] >> 32);
            F1(scores, 2) = (int)(uint32_t)w[2];
            F1(scores, 3) = (int)(uint32_t)(w[2] >> 32);
            for (int i = 0; i < 136; i++) F1(wall, i) = T_UNK;
            for (int k = 0; k < 7; k++) {
                const uint64_t v = w[3 + k];
                for (int b = 0; b < 8; b++) {
                    const int i = k * 8 + b;
                    if (i < 52) F1(wall, i) = (u8)((v >> (8 * b)) & 0xFF);
                }
            }
            F1(wall, 60) = (u8)ev.pai;
            int aug_seed_dora = MJ_NONE;
            if ((w[0] >> LG_SK_DEAL_BIT) & 1) {
                // trust_seed (invisible.rs:36-71): the game came from this engine, rebuild the whole wall from its seed
                u8 logged[52];
                for (int i = 0; i < 52; i++) logged[i] = F1(wall, i);
                // the pool's shuffle first, then the other one: logs of either rand generation of the reference arena load
                // GameplayLoader(oracle, trust_seed, augmented) (dataset/gameplay.rs:126-164 + invisible.rs:36-71): the events are
                // augmented (Tile::augment, tile.rs:154-167: manzu <-> pinzu) but Invisible::new deals the wall from the seed as
                // it was — the reference mixes the two, and so does this: the logged haipai / dora marker are compared through
                // the swap and then kept, yama / rinshan / further indicators / ura stay as dealt
                const bool aug = (w[0] >> LG_SK_AUG_BIT) & 1;
                auto through = [&](int t) -> int {
                    if (!aug || t >= 37) return t;
                    const int d = deaka(t), sw = d < 9 ? d + 9 : d < 18 ? d - 9 : d;
                    return is_aka(t) ? akaize(sw) : sw;
                };
                bool same = false;
                for (int attempt = 0; attempt < 2 && !same; attempt++) {
                    deal_wall((uint8_t*)&L.B->wall[0][L.l], MJ_LANES, L.deal, L.l, F(seed_nonce), F(seed_key), F(kyoku), F(honba),
                              attempt == 0 ? deal_algo : 1 - deal_algo);
                    same = through(F1(wall, 60)) == ev.pai;
                    for (int i = 0; i < 52; i++) same = same && logged[i] == through(F1(wall, i));
                }
                if (!same) set_err(L, MJ_ERR_WALL);  // the seed does not reproduce the logged haipai under either shuffle
                else if (aug) {
                    for (int i = 0; i < 52; i++) F1(wall, i) = logged[i];
                    aug_seed_dora = F1(wall, 60);  // the invisible side keeps the seed's own first indicator
                    F1(wall, 60) = (u8)ev.pai;
                }
            } else if ((w[0] >> LG_SK_WALL_BIT) & 1) {
                for (int k = 0; k < 17; k++) {
                    const uint64_t v = w[10 + k];
                    for (int b = 0; b < 8; b++) F1(wall, k * 8 + b) = (u8)((v >> (8 * b)) & 0xFF);
                }
            }
            kyoku_init(L);
            if (aug_seed_dora != MJ_NONE) F1(wall, 60) = (u8)aug_seed_dora;
            F(flags) |= TF_HAIPAI_DONE;
            break;
        }
        case LG_TSUMO:
            F(tiles_left) -= 1;
            ev_tsumo(L, ev.actor, ev.pai);
            break;
        case LG_DAHAI: ev_dahai(L, ev.actor, ev.pai, ev.tsumogiri != 0); break;
        case LG_CHI: ev_chi_pon(L, false, ev.actor, ev.target, ev.pai, ev.c[0], ev.c[1]); break;
        case LG_PON: ev_chi_pon(L, true, ev.actor, ev.target, ev.pai, ev.c[0], ev.c[1]); break;
        case LG_DAIMINKAN: ev_daiminkan(L, ev.actor, ev.target, ev.pai, ev.c[0], ev.c[1], ev.c[2]); break;
        case LG_KAKAN: ev_kakan(L, ev.actor, ev.pai); break;
        case LG_ANKAN: ev_ankan(L, ev.actor, deaka(ev.c[0])); break;
        case LG_DORA: ev_dora(L, ev.pai); break;
        case LG_REACH: ev_reach(L, ev.actor); break;
        case LG_REACH_ACCEPTED: ev_reach_accepted(L, ev.actor); break;
        case LG_HORA: ev_prologue(L, ev.actor); break;
        case LG_RYUKYOKU:
        case LG_END_KYOKU: ev_prologue(L, -1); break;
        default: set_err(L, MJ_ERR_INTERNAL); break;
    }
}

// label of seat p after the current event (gameplay.rs:296-409); -1 = no sample.  nxt[0..2] = the three events after
// the current one (type 0 past the end of the log = end_game).
template <class LN> MJDN int rp_label(const LN& L, int p, const RpEvent nxt[3], int always_kan, int& kan_select) {
    kan_select = -1;
    const u32 cans = F1(cans, p);
    const RpEvent& next = (nxt[0].type == LG_REACH_ACCEPTED || nxt[0].type == LG_DORA) ? nxt[1] : nxt[0];
    switch (next.type) {
        case LG_DAHAI: return next.pai;
        case LG_REACH: return 37;
        case LG_CHI:
            if (next.actor == p) {  // ChiType::new (chi_type.rs): position of the called tile among the three
                const int a = deaka(next.c[0]), b = deaka(next.c[1]), t = deaka(next.pai);
                const int lo = a < b ? a : b, hi = a < b ? b : a;
                return t < lo ? 38 : t < hi ? 39 : 40;
            }
            break;
        case LG_PON:
            if (next.actor == p) return 41;
            break;
        case LG_DAIMINKAN:
            if (next.actor == p) {
                if (always_kan) kan_select = deaka(next.pai);
                return 42;
            }
            break;
        case LG_KAKAN:
            if (always_kan || __popcll(F1(kakan_cand, p)) > 1) kan_select = deaka(next.pai);
            return 42;
        case LG_ANKAN:
            if (always_kan || __popcll(F1(ankan_cand, p)) > 1) kan_select = deaka(next.c[0]);
            return 42;
        case LG_RYUKYOKU:
            if (cans & CAN_RYUKYOKU) return 44;
            break;
        default: break;
    }
    const bool has_any_ron = nxt[0].type == LG_HORA;
    if (has_any_ron) {
        for (int k = 0; k < 3; k++) {
            if (nxt[k].type == LG_END_KYOKU) break;
            if (nxt[k].type == LG_HORA && nxt[k].actor == p) return 43;
        }
    }
    const bool can_chi = (cans & (CAN_CHI_LOW | CAN_CHI_MID | CAN_CHI_HIGH)) != 0;
    if ((can_chi && next.type == LG_TSUMO) || ((cans & (CAN_PON | CAN_DAIMINKAN | CAN_RON_AGARI)) && !has_any_ron)) return 45;
    return -1;
}

// Apply events until at least one tracked seat has a sample (or the log ends).  One lane per table.
__global__ __launch_bounds__(64) void mj_k_replay(ReplayParams P) {
    __shared__ DealScratch s_deal;
    const int table = blockIdx.x * 64 + threadIdx.x;
    Lane L = {MJ_POOL_PTR(P.blocks + blockIdx.x), (int)threadIdx.x, &c_mj_tables};
    L.deal = &s_deal;
    int nr = 0;
    if (table < P.n_tables) {
        const uint64_t* sc = P.script + P.script_off[table];
        const uint32_t n_words = P.script_off[table + 1] - P.script_off[table];
        uint32_t cur = P.cursor[table];
        for (int s = 0; s < 4; s++) {
            F1(main_row, s) = -1;
            F1(kan_row, s) = -1;
        }
        const int tracked = P.tracked[table];
        // the loader's windows(4) never makes the last two log events (.., hora|ryukyoku, end_kyoku) "current"
        // (gameplay.rs:258-262): stop as soon as fewer than three events remain
        while (cur < n_words && F(err) == MJ_OK) {
            // look ahead: current + three following events
            uint32_t pos = cur;
            const RpEvent ev = rp_decode(sc[pos]);
            pos += ev.len;
            RpEvent nxt[3];
            int n_after = 0;
            for (int k = 0; k < 3; k++) {
                if (pos < n_words) {
                    nxt[k] = rp_decode(sc[pos]);
                    pos += nxt[k].len;
                    n_after++;
                } else {
                    nxt[k].type = 0;
                    nxt[k].actor = nxt[k].target = nxt[k].pai = 0;
                    nxt[k].len = 1;
                }
            }
            if (n_after < 2) {  // this event and what follows are never a window start
                cur = n_words;
                break;
            }
            rp_apply(L, ev, sc + cur, P.deal_algo);
            if (ev.type == LG_END_KYOKU) P.kyoku_idx[table] += 1;
            cur += ev.len;
            P.ev_index[table] += 1;
            // ---- samples
            for (int p = 0; p < 4; p++) {
                if (!((tracked >> p) & 1) || !(F1(cans, p) & CAN_ACT)) continue;
                int kan_select;
                const int label = rp_label(L, p, nxt, P.always_include_kan_select, kan_select);
                if (label < 0) continue;
                P.label[table * 4 + p] = label;
                F1(main_row, p) = nr++;
                if (kan_select >= 0) {
                    P.kan_label[table * 4 + p] = kan_select;
                    F1(kan_row, p) = nr++;
                }
            }
            if (nr) break;
        }
        P.cursor[table] = cur;
        if (cur >= n_words && !(F(flags) & TF_DONE)) {
            F(flags) |= TF_DONE;
            atomicAdd(&P.counters[1], 1ull);
        }
    }
    F1(n_rows, 0) = (u8)nr;
    F1(n_rows, 1) = 0;
    int rows0 = nr;
    for (int off = 32; off > 0; off >>= 1) rows0 += __shfl_down(rows0, off);
    if (threadIdx.x == 0) {
        P.block_rows[2 * blockIdx.x] = rows0;
        P.block_rows[2 * blockIdx.x + 1] = 0;
    }
}

// Per-row sample metadata after mj_k_assign: label, log id, seat, kyoku index, turn, shanten, kan flag, event index.
struct ReplayMetaParams {
    const TableBlock* blocks;
    const uint32_t* rows;
    int n_rows;
    const int32_t* label;
    const int32_t* kan_label;
    const uint8_t* kyoku_idx;
    const uint32_t* ev_index;
    int32_t* out;  // [n_rows][8]
};
__global__ void mj_k_replay_meta(ReplayMetaParams P) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= P.n_rows) return;
    const uint32_t d = P.rows[r];
    const int t = ROW_TABLE(d), s = ROW_SEAT(d), kan = ROW_KAN(d);
    const TableBlock* B = P.blocks + (t >> 6);
    const int l = t & 63;
    int32_t* o = P.out + (size_t)r * 8;
    o[0] = kan ? P.kan_label[t * 4 + s] : P.label[t * 4 + s];
    o[1] = t;
    o[2] = s;
    o[3] = P.kyoku_idx[t];
    o[4] = B->at_turn[s][l];
    o[5] = B->shanten[s][l];
    o[6] = kan;
    o[7] = (int32_t)P.ev_index[t];
}


// ================================================================ single-table access for libriichi.state.PlayerState
// (state/player_state.rs:142-167 pyo3 surface: update / encode_obs / getters; used by tests and debugging)
__global__ void mj_k_apply_event(TableBlock* blocks, int table, const uint64_t* words) {
    __shared__ DealScratch s_deal;
    Lane L = {MJ_POOL_PTR(blocks + (table >> 6)), table & 63, &c_mj_tables};
    L.deal = &s_deal;
    const RpEvent ev = rp_decode(words[0]);
    rp_apply(L, ev, words);
}
// mark one (table, seat) as the only policy row so that the normal snapshot / encode kernels produce its obs + mask
__global__ __launch_bounds__(64) void mj_k_mark_row(TableBlock* blocks, int n_tables, int table, int seat, int kan, int* block_rows) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    TableBlock* B = blocks + blockIdx.x;
    const int l = threadIdx.x;
    for (int s = 0; s < 4; s++) {
        B->main_row[s][l] = -1;
        B->kan_row[s][l] = -1;
    }
    const bool me = t == table && t < n_tables;
    if (me) {
        if (kan) B->kan_row[seat][l] = 0;
        else B->main_row[seat][l] = 0;
    }
    B->n_rows[0][l] = me ? 1 : 0;
    B->n_rows[1][l] = 0;
    if (l == 0) {
        block_rows[2 * blockIdx.x] = (table >> 6) == (int)blockIdx.x ? 1 : 0;
        block_rows[2 * blockIdx.x + 1] = 0;
    }
}
// queries / pokes: out int32[8]
enum { MJ_Q_AGARI_POINTS = 0, MJ_Q_RULE_BASED_AGARI = 1, MJ_Q_REAL_TIME_SHANTEN = 2, MJ_Q_DORAS_OWNED = 3,
       MJ_Q_ADD_DORA = 4, MJ_Q_SET_SCORES = 5, MJ_Q_SCENE = 6, MJ_Q_DECODE_ACTION = 7, MJ_Q_VALIDATE_REACTION = 8 };
__global__ void mj_k_query(TableBlock* blocks, int table, int seat, int what, const int32_t* args, int32_t* out) {
    Lane L = {MJ_POOL_PTR(blocks + (table >> 6)), table & 63, &c_mj_tables};
    const int p = seat;
    switch (what) {
        case MJ_Q_AGARI_POINTS: {  // args: is_ron, n_ura, ura[5]  (agent_helper.rs:377-462)
            u8 ura[5];
            for (int i = 0; i < 5; i++) ura[i] = (u8)args[2 + i];
            Point pt = {0, 0, 0};
            const u32 cans = F1(cans, p);
            const bool is_ron = args[0] != 0;
            bool ok = is_ron ? (cans & CAN_RON_AGARI) != 0 : (cans & CAN_TSUMO_AGARI) != 0;
            if (ok) ok = seat_agari_points(L, p, is_ron, args[1], pt, ura);
            out[0] = ok;
            out[1] = pt.ron;
            out[2] = pt.tsumo_ko;
            out[3] = pt.tsumo_oya;
            break;
        }
        case MJ_Q_RULE_BASED_AGARI: out[0] = rule_based_agari(L, p); break;
        case MJ_Q_REAL_TIME_SHANTEN: {  // agent_helper.rs:467-503
            const u32 cans = F1(cans, p);
            const int sh = F1(shanten, p);
            int r;
            if (!(cans & CAN_DISCARD)) r = sh;
            else if (sh > 0) r = F1(has_next_shanten, p) ? sh - 1 : sh;
            else if (F1(last_self_tsumo, p) != MJ_NONE) r = ((F1(waits, p) >> deaka(F1(last_self_tsumo, p))) & 1) ? -1 : 0;
            else r = calc_all(c_mj_tables, load_hand(L, p), F1(len_div3, p));
            out[0] = r;
            break;
        }
        case MJ_Q_DORAS_OWNED: {  // doras_owned[rel] (update.rs:780-808 recount): hand (own seat only) + melds + akas
            for (int rel = 0; rel < 4; rel++) {
                const int s = (p + rel) & 3;
                int n = 0;
                const int nd = F(n_dora_ind);
                const Hand h = load_hand(L, s);
                for (int i = 0; i < nd; i++) {
                    const int d = tile_next(F1(dora_ind, i));
                    if (rel == 0) n += h.get(d);
                    for (int k = 0; k < F1(fuuro_n, s); k++)
                        for (int j = 0; j < 4; j++) {
                            const int t = F3(fuuro, s, k, j);
                            if (t != MJ_NONE && deaka(t) == d) n++;
                        }
                    for (int k = 0; k < F1(ankan_n, s); k++)
                        if (F2(ankan, s, k) == d) n += 4;
                }
                if (rel == 0) n += __popc(F1(akas_in_hand, s));
                for (int k = 0; k < F1(fuuro_n, s); k++)
                    for (int j = 0; j < 4; j++) {
                        const int t = F3(fuuro, s, k, j);
                        if (t != MJ_NONE && is_aka(t)) n++;
                    }
                for (int k = 0; k < F1(ankan_n, s); k++) {
                    const int t = F2(ankan, s, k);
                    if (t == T_5M || t == T_5P || t == T_5S) n++;
                }
                out[rel] = n;
            }
            break;
        }
        case MJ_Q_ADD_DORA: {  // add_dora_indicator without the event prologue (update.rs:780-808)
            const int n = F(n_dora_ind);
            if (n < 5) {
                F1(dora_ind, n) = (u8)args[0];
                F(n_dora_ind) = (u8)(n + 1);
                pub_witness(L, args[0]);
            }
            break;
        }
        case MJ_Q_SET_SCORES:
            for (int i = 0; i < 4; i++) F1(scores, i) = args[i];
            break;
        case MJ_Q_SCENE: {  // agent/mortal.rs:200-250 set_scene: args[0] = enable_quick_eval -> quick discard (raw id) or -1, need_kan_select
            const u32 cans = F1(cans, p);
            out[0] = -1;
            out[1] = 0;
            if (args[0] && (cans & CAN_DISCARD) && !(cans & (CAN_RIICHI | CAN_TSUMO_AGARI | CAN_ANKAN | CAN_KAKAN | CAN_RYUKYOKU))) {
                const u64 dc = discard_candidates_aka(L, p);
                if (__popcll(dc) == 1) out[0] = __ffsll((long long)dc) - 1;
            }
            if (cans & (CAN_ANKAN | CAN_KAKAN))
                out[1] = !args[0] || __popcll(F1(ankan_cand, p)) + __popcll(F1(kakan_cand, p)) > 1;
            out[2] = F1(last_self_tsumo, p);
            break;
        }
        case MJ_Q_DECODE_ACTION: {  // agent/mortal.rs:338-573: args = action id, kan-select tile or -1 -> event word (LG_*), error
            const Reaction r = decode_action(L, p, args[0], args[1]);
            unsigned long long w = 0;
            switch (r.type) {
                case RX_DAHAI: w = LG_WORD(LG_DAHAI, r.actor, 0, r.pai, 0, 0, 0, 0, r.tsumogiri); break;
                case RX_CHI: w = LG_WORD(LG_CHI, r.actor, r.target, r.pai, r.c0, r.c1, 0, 0, 0); break;
                case RX_PON: w = LG_WORD(LG_PON, r.actor, r.target, r.pai, r.c0, r.c1, 0, 0, 0); break;
                case RX_DAIMINKAN: w = LG_WORD(LG_DAIMINKAN, r.actor, r.target, r.pai, r.c0, r.c1, r.c2, 0, 0); break;
                case RX_ANKAN: w = LG_WORD(LG_ANKAN, r.actor, 0, 0, akaize(r.pai), r.pai, r.pai, r.pai, 0); break;
                case RX_KAKAN: {
                    const int t = deaka(r.pai);
                    w = LG_WORD(LG_KAKAN, r.actor, 0, r.pai, is_aka(r.pai) ? t : akaize(t), t, t, 0, 0);
                    break;
                }
                case RX_REACH: w = LG_WORD(LG_REACH, r.actor, 0, 0, 0, 0, 0, 0, 0); break;
                case RX_HORA: w = LG_WORD(LG_HORA, r.actor, r.target, 0, 0, 0, 0, 0, 0); break;
                case RX_RYUKYOKU: w = LG_WORD(LG_RYUKYOKU, 0, 0, 0, 0, 0, 0, 0, 0); break;
                default: w = 0; break;  // pass
            }
            out[0] = (int32_t)(uint32_t)w;
            out[1] = (int32_t)(uint32_t)(w >> 32);
            out[2] = F(err);
            F(err) = MJ_OK;  // an illegal action is reported to the caller, the table stays usable
            break;
        }
        case MJ_Q_VALIDATE_REACTION: {  // the step kernel's own check of an explicit reaction word (mj_step_ev): args = word lo, hi -> error code
            const unsigned long long w = (unsigned long long)(uint32_t)args[0] | ((unsigned long long)(uint32_t)args[1] << 32);
            const u8 before = F(err);
            F(err) = MJ_OK;
            const Reaction r = reaction_from_word(L, p, w);
            out[0] = F(err);
            out[1] = r.type;
            F(err) = before;
            break;
        }
    }
}
