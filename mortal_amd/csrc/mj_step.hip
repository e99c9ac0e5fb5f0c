// Env-step kernels: one table per lane, 64 tables per wavefront.
//
//   mj_k_step   commit the policy's actions (agent/mortal.rs:292-573), advance every table to its next
//               decision point or game end (arena/game.rs:59-218, arena/board.rs:141-161,511-678), classify the
//               new decisions (quick-eval / kan-select, agent/mortal.rs:200-250) and count policy rows.
//   mj_k_rows   exclusive scan of the per-table row counts -> contiguous row ids per agent + row descriptors.
//   mj_k_random_policy   uniform-random legal action per row (BASELINE config 2), counter-based.
//   mj_k_greedy_policy   tenpai-seeking test / benchmark policy per row (always agari, mostly riichi, shanten-lowering discards).
#include <hip/hip_runtime.h>

#include "mj_rules.h"

struct Reaction {
    u8 type;  // RX_*
    u8 actor, target, pai;
    u8 c0, c1, c2;
    u8 tsumogiri;
    uint64_t tag;  // 0, or the log tag word of a policy decision (mj_state.h LG_TAG_BIT)
};
enum { RX_NONE = 0, RX_DAHAI, RX_REACH, RX_CHI, RX_PON, RX_DAIMINKAN, RX_KAKAN, RX_ANKAN, RX_HORA, RX_RYUKYOKU };

struct StepParams {
    TableBlock* blocks;
    int n_tables;
    MjTablesDev tables;
    const int* actions[2];     // per agent, indexed by row id of the previous cycle (NULL on the first cycle)
    const float* q_values[2];  // per agent [rows][46] or NULL (needed only by the agari guard)
    const uint64_t* reactions[2];  // per agent: explicit mjai reactions (one LG_* event word per row, 0 = none) instead of
                                   // action ids — mjai-log engines answer with events (agent/mjai_log.rs:65-87)
    uint64_t* log;             // [n_tables][log_cap] event words or NULL (logging off)
    uint32_t* log_len;         // [n_tables]
    uint32_t log_cap;
    uint32_t cycle;            // index of this mj_step call (log tags)
    int deal_algo;
    int enable_quick_eval[2];
    int enable_agari_guard[2];
    int game_length;           // 8 = hanchan
    int refill;                // steady-state mode: restart finished tables with fresh seeds
    uint64_t refill_stride;    // nonce increment on refill
    uint32_t start_stagger;    // steady-state mode: table t plays its first game from cycle hash(t) % start_stagger on (0 = all at once)
    unsigned long long* counters;  // [0] env steps, [1] games finished, [2] error count, [3] decisions, [4] quick-evals
    int* final_scores;         // [n_games_total][4] written when a game finishes
    uint8_t* final_done;       // [n_games_total]
    int n_games_total;
    int* block_rows;           // [n_blocks][2] rows wanted per 64-table block and agent (input of the row scan)
};

// ---------------------------------------------------------------- discard candidates (agent_helper.rs:35-79)
template <class LN> MJD u64 discard_candidates_aka(const LN& L, int s) {  // 37-bit set
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

// ---------------------------------------------------------------- action id -> reaction (mortal.rs:338-573)
// The consumed tiles (raw ids, red fives as 34..36) are all in the seat's hand, copies and red flags counted.
template <class LN> MJD bool hand_holds(const LN& L, int s, int a, int b, int c /* MJ_NONE = absent */) {
    const Hand h = load_hand(L, s);
    const int akas = F1(akas_in_hand, s);
    const int tiles[3] = {a, b, c};
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int x = tiles[i];
        if (x == MJ_NONE) continue;
        if (x >= 37) { ok = false; continue; }
        const int d = deaka(x);
        int need = 0, need_red = 0;
#pragma unroll
        for (int j = 0; j < 3; j++)
            if (tiles[j] != MJ_NONE && tiles[j] < 37 && deaka(tiles[j]) == d) { need++; need_red += is_aka(tiles[j]); }
        const int have = h.get(d), have_red = (d == T_5M || d == T_5P || d == T_5S) ? ((akas >> (d == T_5M ? 0 : d == T_5P ? 1 : 2)) & 1) : 0;
        ok = ok && need_red <= have_red && need - need_red <= have - have_red;
    }
    return ok;
}
// An explicit reaction of seat s (mjai-log engines): the LG_* header word of the event.  The Python host runs
// PlayerState::validate_reaction on it first (state/action.rs:91-228, as BoardState::step does, board.rs:524-533), but a C-ABI
// caller may not: the word is checked here against the seat's candidates AND its tiles (legal discard set, consumed tiles
// held with their red flags, the called tile, the shape of a chi / pon / kan, the chi type's own candidate flag — kuikae —, the
// seat the call is made on, tsumogiri only of the tile just drawn) before anything touches the packed hand.  A call or ron
// must name the seat that discarded (cans_target): stricter than validate_reaction's `(target + 1) % 4 == actor` /
// `target != actor`, which would let the wrong seat's river take the call mark.
template <class LN> MJDN Reaction reaction_from_word(const LN& L, int s, uint64_t w) {
    Reaction r = {RX_NONE, (u8)s, 0, MJ_NONE, MJ_NONE, MJ_NONE, MJ_NONE, 0, 0ull};
    const int t = (int)(w & 15);
    if (t == 0) return r;  // {"type":"none"}
    const u32 cans = F1(cans, s);
    const int actor = (int)((w >> 4) & 3), pai = (int)((w >> 8) & 63);
    const int c0 = (int)((w >> 14) & 63), c1 = (int)((w >> 20) & 63), c2 = (int)((w >> 26) & 63);
    r.target = (u8)((w >> 6) & 3);
    bool ok = actor == s || t == LG_RYUKYOKU;
    const int lkt = F1(last_kawa_tile, s);
    const bool called = pai < 37 && lkt != MJ_NONE && pai == lkt;  // the call names the tile just discarded
    switch (t) {
        case LG_DAHAI:
            r.type = RX_DAHAI; r.pai = (u8)pai; r.tsumogiri = (u8)((w >> 38) & 1);
            ok = ok && (cans & CAN_DISCARD) && pai < 37 && ((discard_candidates_aka(L, s) >> pai) & 1);
            ok = ok && (!r.tsumogiri || F1(last_self_tsumo, s) == pai);  // action.rs:124-130
            break;
        case LG_REACH: r.type = RX_REACH; ok = ok && (cans & CAN_RIICHI); break;
        case LG_CHI: {
            r.type = RX_CHI; r.pai = (u8)pai; r.c0 = (u8)c0; r.c1 = (u8)c1;
            ok = ok && (cans & (CAN_CHI_LOW | CAN_CHI_MID | CAN_CHI_HIGH)) && called && c0 < 37 && c1 < 37 && hand_holds(L, s, c0, c1, MJ_NONE);
            ok = ok && r.target == F1(cans_target, s) && ((r.target + 1) & 3) == s;  // action.rs:144 + the seat that discarded
            if (ok) {  // three consecutive tiles of one number suit
                const int x = deaka(pai), y = deaka(c0), z = deaka(c1);
                const int lo = min(x, min(y, z)), hi = max(x, max(y, z)), mid = x + y + z - lo - hi;
                ok = hi < 27 && lo / 9 == hi / 9 && mid == lo + 1 && hi == lo + 2;
                // ChiType::new (chi_type.rs:12-25): the called tile below / between / above the consumed pair picks the flag; a
                // legal shape whose own flag is off is a forbidden swap call (kuikae, update.rs:826-868)
                const u32 need = x == lo ? CAN_CHI_LOW : x == mid ? CAN_CHI_MID : CAN_CHI_HIGH;
                ok = ok && (cans & need);
            }
            break;
        }
        case LG_PON:
            r.type = RX_PON; r.pai = (u8)pai; r.c0 = (u8)c0; r.c1 = (u8)c1;
            ok = ok && (cans & CAN_PON) && called && c0 < 37 && c1 < 37 && deaka(c0) == deaka(pai) && deaka(c1) == deaka(pai) && hand_holds(L, s, c0, c1, MJ_NONE);
            ok = ok && r.target == F1(cans_target, s) && r.target != s;  // action.rs:164
            break;
        case LG_DAIMINKAN:
            r.type = RX_DAIMINKAN; r.pai = (u8)pai; r.c0 = (u8)c0; r.c1 = (u8)c1; r.c2 = (u8)c2;
            ok = ok && (cans & CAN_DAIMINKAN) && called && c0 < 37 && c1 < 37 && c2 < 37 && deaka(c0) == deaka(pai) && deaka(c1) == deaka(pai) &&
                 deaka(c2) == deaka(pai) && hand_holds(L, s, c0, c1, c2);
            ok = ok && r.target == F1(cans_target, s) && r.target != s;  // action.rs:179
            break;
        case LG_KAKAN:
            r.type = RX_KAKAN; r.pai = (u8)pai;
            ok = ok && (cans & CAN_KAKAN) && pai < 37 && ((F1(kakan_cand, s) >> deaka(pai)) & 1) && hand_holds(L, s, pai, MJ_NONE, MJ_NONE);
            break;
        case LG_ANKAN:
            r.type = RX_ANKAN; r.pai = (u8)deaka(c0);
            ok = ok && (cans & CAN_ANKAN) && c0 < 37 && ((F1(ankan_cand, s) >> deaka(c0)) & 1) && load_hand(L, s).get(deaka(c0)) == 4;
            break;
        case LG_HORA:
            r.type = RX_HORA;
            ok = ok && (r.target == s ? (cans & CAN_TSUMO_AGARI) : ((cans & CAN_RON_AGARI) && r.target == F1(cans_target, s)));
            break;
        case LG_RYUKYOKU: r.type = RX_RYUKYOKU; ok = ok && (cans & CAN_RYUKYOKU); break;
        default: ok = false; break;
    }
    if (!ok) {
        set_err(L, MJ_ERR_ILLEGAL_ACTION);
        r.type = RX_NONE;
    }
    return r;
}

template <class LN> MJDN Reaction decode_action(const LN& L, int s, int action, int kan_tile) {
    Reaction r = {RX_NONE, (u8)s, 0, MJ_NONE, MJ_NONE, MJ_NONE, MJ_NONE, 0, 0ull};
    const u32 cans = F1(cans, s);
    const int akas = F1(akas_in_hand, s);
    const int lkt = F1(last_kawa_tile, s);
    r.target = F1(cans_target, s);
    auto aka_of = [&](int pai, int a, int b) -> bool {  // match on the raw id of the called tile
        if (pai >= 27) return false;
        int num = pai % 9, kind = pai / 9;
        return (num == a || num == b) && ((akas >> kind) & 1);
    };
    if (action <= 36) {
        if (!(cans & CAN_DISCARD) || action < 0) { set_err(L, MJ_ERR_ILLEGAL_ACTION); return r; }
        // validate_reaction (action.rs:121-131): tile must be in hand (aka flag for red fives)
        Hand h = load_hand(L, s);
        bool ok = h.get(deaka(action)) > 0 && (!is_aka(action) || ((akas >> (action - T_5MR)) & 1));
        if (!ok) { set_err(L, MJ_ERR_ILLEGAL_ACTION); return r; }
        r.type = RX_DAHAI;
        r.pai = (u8)action;
        r.tsumogiri = F1(last_self_tsumo, s) == action;
        return r;
    }
    switch (action) {
        case 37:
            if (!(cans & CAN_RIICHI)) { set_err(L, MJ_ERR_ILLEGAL_ACTION); return r; }
            r.type = RX_REACH;
            return r;
        case 38: {
            if (!(cans & CAN_CHI_LOW) || lkt == MJ_NONE) { set_err(L, MJ_ERR_ILLEGAL_ACTION); return r; }
            int first = tile_next(lkt);
            bool aka = aka_of(lkt, 2, 3);
            r.type = RX_CHI;
            r.pai = (u8)lkt;
            r.c0 = (u8)(aka ? akaize(first) : first);
            r.c1 = (u8)(aka ? akaize(tile_next(first)) : tile_next(first));
            return r;
        }
        case 39: {
            if (!(cans & CAN_CHI_MID) || lkt == MJ_NONE) { set_err(L, MJ_ERR_ILLEGAL_ACTION); return r; }
            bool aka = aka_of(lkt, 3, 5);
            r.type = RX_CHI;
            r.pai = (u8)lkt;
            r.c0 = (u8)(aka ? akaize(tile_prev(lkt)) : tile_prev(lkt));
            r.c1 = (u8)(aka ? akaize(tile_next(lkt)) : tile_next(lkt));
            return r;
        }
        case 40: {
            if (!(cans & CAN_CHI_HIGH) || lkt == MJ_NONE) { set_err(L, MJ_ERR_ILLEGAL_ACTION); return r; }
            int last = tile_prev(lkt);
            bool aka = aka_of(lkt, 5, 6);
            r.type = RX_CHI;
            r.pai = (u8)lkt;
            r.c0 = (u8)(aka ? akaize(tile_prev(last)) : tile_prev(last));
            r.c1 = (u8)(aka ? akaize(last) : last);
            return r;
        }
        case 41: {
            if (!(cans & CAN_PON) || lkt == MJ_NONE) { set_err(L, MJ_ERR_ILLEGAL_ACTION); return r; }
            bool aka = (lkt == T_5M && (akas & 1)) || (lkt == T_5P && (akas & 2)) || (lkt == T_5S && (akas & 4));
            r.type = RX_PON;
            r.pai = (u8)lkt;
            r.c0 = (u8)(aka ? akaize(lkt) : deaka(lkt));
            r.c1 = (u8)deaka(lkt);
            return r;
        }
        case 42: {
            if (!(cans & CAN_KAN)) { set_err(L, MJ_ERR_ILLEGAL_ACTION); return r; }
            const u64 ac = F1(ankan_cand, s), kc = F1(kakan_cand, s);
            int tile;
            if (kan_tile >= 0) {
                tile = kan_tile;
                if (tile > 33 || !(((ac | kc) >> tile) & 1)) { set_err(L, MJ_ERR_ILLEGAL_ACTION); return r; }
            } else if (cans & CAN_DAIMINKAN) {
                tile = lkt;
            } else if (cans & CAN_ANKAN) {
                tile = __ffsll((long long)ac) - 1;
            } else {
                tile = __ffsll((long long)kc) - 1;
            }
            if (tile < 0 || tile == MJ_NONE) { set_err(L, MJ_ERR_ILLEGAL_ACTION); return r; }
            if (cans & CAN_DAIMINKAN) {
                r.type = RX_DAIMINKAN;
                r.pai = (u8)tile;
                if (is_aka(tile)) r.c0 = r.c1 = r.c2 = (u8)deaka(tile);
                else { r.c0 = (u8)akaize(tile); r.c1 = r.c2 = (u8)tile; }
            } else if ((cans & CAN_ANKAN) && ((ac >> deaka(tile)) & 1)) {
                r.type = RX_ANKAN;
                r.pai = (u8)deaka(tile);
            } else {
                bool aka = (tile == T_5M && (akas & 1)) || (tile == T_5P && (akas & 2)) || (tile == T_5S && (akas & 4));
                r.type = RX_KAKAN;
                r.pai = (u8)(aka ? akaize(tile) : deaka(tile));
            }
            return r;
        }
        case 43:
            if (!(cans & CAN_AGARI)) { set_err(L, MJ_ERR_ILLEGAL_ACTION); return r; }
            // validate_reaction: tsumo needs can_tsumo_agari, ron needs can_ron_agari (action.rs:198-204)
            if (r.target == s ? !(cans & CAN_TSUMO_AGARI) : !(cans & CAN_RON_AGARI)) { set_err(L, MJ_ERR_ILLEGAL_ACTION); return r; }
            r.type = RX_HORA;
            return r;
        case 44:
            if (!(cans & CAN_RYUKYOKU)) { set_err(L, MJ_ERR_ILLEGAL_ACTION); return r; }
            r.type = RX_RYUKYOKU;
            return r;
        default: return r;  // 45 = pass
    }
}

// ---------------------------------------------------------------- board (arena/board.rs)
template <class LN> MJD void abortive_ryukyoku(const LN& L, uint64_t tag = 0) {  // board.rs:502-509
    F(flags) |= TF_HAS_ABORTIVE;
    if (L.log) {
        const int z[4] = {0, 0, 0, 0};
        log_push_rx(L, LG_WORD(LG_RYUKYOKU, 0, 0, 0, 0, 0, 0, 0, 0), tag);
        log_push_i32x4(L, z);
    }
}

template <class LN> MJD void check_riichi_accepted(const LN& L) {  // board.rs:342-351
    int a = F(riichi_to_be_accepted);
    if (a != MJ_NONE) {
        F(riichi_to_be_accepted) = MJ_NONE;
        log_push(L, LG_WORD(LG_REACH_ACCEPTED, a, 0, 0, 0, 0, 0, 0, 0));
        ev_reach_accepted(L, a);
    }
}
template <class LN> MJD void add_new_dora(const LN& L) {  // board.rs:353-364
    int n = F(dora_n);
    if (n == 0) { set_err(L, MJ_ERR_WALL); return; }
    n -= 1;
    F(dora_n) = (u8)n;
    log_push(L, LG_WORD(LG_DORA, 0, 0, F1(wall, 56 + n), 0, 0, 0, 0, 0));
    ev_dora(L, F1(wall, 56 + n));
}

template <class LN> MJDN void exhaustive_ryukyoku(const LN& L) {  // board.rs:241-294
    const int oya = F(kyoku) & 3;
    int deltas[4] = {0, 0, 0, 0};
    u32 fl = F(flags);
    if (F1(shanten, oya) == 0) fl |= TF_CAN_RENCHAN;
    else fl &= ~TF_CAN_RENCHAN;
    bool nagashi = false;
    for (int i = 0; i < 4; i++) {
        if (!((fl >> (12 + i)) & 1)) continue;
        nagashi = true;
        for (int k = 0; k < 4; k++) {
            int d;
            if (i == oya) d = (k == i) ? 12000 : -4000;
            else d = (k == oya) ? -4000 : (k == i) ? 8000 : -2000;
            deltas[k] += d;
        }
    }
    if (!nagashi) {
        int n = 0;
        for (int i = 0; i < 4; i++) n += F1(shanten, i) == 0;
        int plus = n == 1 ? 3000 : n == 2 ? 1500 : n == 3 ? 1000 : 0;
        int minus = n == 1 ? -1000 : n == 2 ? -1500 : n == 3 ? -3000 : 0;
        if (plus > 0)
            for (int k = 0; k < 4; k++) deltas[k] += F1(shanten, k) == 0 ? plus : minus;
    }
    for (int k = 0; k < 4; k++) F1(kyoku_deltas, k) += deltas[k];
    F(flags) = fl;
    if (L.log) {
        log_push(L, LG_WORD(LG_RYUKYOKU, 0, 0, 0, 0, 0, 0, 0, 0));
        log_push_i32x4(L, deltas);
    }
}

// Log an agent's reaction: header (+ tag word when the reaction came from a policy row).
template <class LN> MJD void log_push_rx(const LN& L, uint64_t word, uint64_t tag) {
    if (!L.log) return;
    if (tag) {
        log_push(L, word | (1ull << LG_TAG_BIT));
        log_push(L, tag);
    } else {
        log_push(L, word);
    }
}

// Hora{actor, target, deltas, ura_markers}: the ura indicators are listed only for a winner in riichi (board.rs:418-426)
template <class LN> MJD void log_hora(const LN& L, int actor, int target, const int d[4], int n_ura, uint64_t tag) {
    if (!L.log) return;
    const int n = accepted(L, actor) ? n_ura : 0;
    log_push_rx(L, LG_WORD(LG_HORA, actor, target, 0, 0, 0, 0, 0, 0) | ((uint64_t)n << LG_NURA_SHIFT), tag);
    log_push_i32x4(L, d);
    uint64_t u = 0;
    for (int i = 0; i < n; i++) u |= (uint64_t)F1(wall, 61 + i) << (6 * i);
    log_push(L, u);
}

template <class LN> MJDN void handle_hora(const LN& L, int single_actor, int single_target, const Reaction rx[4]) {  // board.rs:366-471
    u32 fl = F(flags) | TF_HAS_HORA;
    const int oya = F(kyoku) & 3;
    const bool is_ron = single_actor != single_target;
    int honba_left = F(honba);
    int kyotaku_point = F(kyotaku) * 1000;
    F(kyotaku) = 0;
    const int n_ura = 5 - F(dora_n);
    Point pts[4];
    bool has[4] = {false, false, false, false};
    for (int i = 0; i < 4; i++) {
        if (rx[i].type != RX_HORA) continue;
        if (rx[i].actor == oya) fl |= TF_CAN_RENCHAN;
        has[i] = seat_agari_points(L, rx[i].actor, is_ron, n_ura, pts[i]);
        if (!has[i]) set_err(L, MJ_ERR_NOT_HORA);
    }
    F(flags) = fl;
    if (is_ron) {
        for (int k = 1; k <= 3; k++) {
            int actor = (single_target + k) & 3;
            if (!has[actor]) continue;
            Point p = pts[actor];
            int d[4] = {0, 0, 0, 0};
            int pao = F1(paos, actor);
            if (pao != MJ_NONE) {
                d[pao] = -p.ron / 2 - honba_left * 300;
                d[single_target] -= p.ron / 2;
            } else {
                d[single_target] = -p.ron - honba_left * 300;
            }
            d[actor] = p.ron + kyotaku_point + honba_left * 300;
            kyotaku_point = 0;
            honba_left = 0;
            for (int i = 0; i < 4; i++) F1(kyoku_deltas, i) += d[i];
            log_hora(L, actor, single_target, d, n_ura, rx[actor].tag);
        }
        return;
    }
    if (!has[single_actor]) return;
    Point p = pts[single_actor];
    int d[4];
    int pao = F1(paos, single_actor);
    if (pao != MJ_NONE) {
        d[0] = d[1] = d[2] = d[3] = 0;
        d[pao] = -p.ron - honba_left * 300;
    } else {
        for (int i = 0; i < 4; i++) d[i] = -p.tsumo_ko - honba_left * 100;
        if (single_actor != oya) d[oya] = -p.tsumo_oya - honba_left * 100;
    }
    d[single_actor] = tsumo_total(p, single_actor == oya) + kyotaku_point + honba_left * 300;
    for (int i = 0; i < 4; i++) F1(kyoku_deltas, i) += d[i];
    log_hora(L, single_actor, single_target, d, n_ura, rx[single_actor].tag);
}

// One BoardState::step (board.rs:511-678).  Returns true when the kyoku has ended.
template <class LN> MJDN bool board_step(const LN& L, const Reaction rx[4]) {
    if (F(accepted_riichis) == 4) {  // 四家立直
        abortive_ryukyoku(L);
        return true;
    }
    // winning reaction: Hora 0 < Daiminkan/Pon 1 < other 2 < None 3; ties -> lowest seat (min_by_key = first min)
    int best = 0, bestp = 4;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int t = rx[i].type;
        int p = t == RX_HORA ? 0 : (t == RX_DAIMINKAN || t == RX_PON) ? 1 : t == RX_NONE ? 3 : 2;
        if (p < bestp) { bestp = p; best = i; }
    }
    const Reaction ev = rx[best];
    u32 fl = F(flags);
    if ((fl & TF_CHECK_FOUR_KAN) && ev.type != RX_HORA) {  // 四槓散了
        abortive_ryukyoku(L);
        return true;
    }
    // update_nagashi_mangan_and_four_wind (board.rs:296-312)
    if (ev.type == RX_DAHAI) {
        if (!is_yaokyuu(ev.pai)) fl &= ~(TF_NAGASHI0 << ev.actor);
    } else if (ev.type == RX_CHI || ev.type == RX_PON || ev.type == RX_DAIMINKAN) {
        fl &= ~(TF_NAGASHI0 << ev.target);
        fl &= ~TF_CAN_FOUR_WIND;
    } else if (ev.type == RX_ANKAN) {
        fl &= ~TF_CAN_FOUR_WIND;
    }
    F(flags) = fl;

    switch (ev.type) {
        case RX_NONE: {
            if (F(tiles_left) == 0) {
                SPROF_T(t_e);
                exhaustive_ryukyoku(L);
                SPROF_ADD(17, t_e);
                return true;
            }
            check_riichi_accepted(L);
            int tile;
            fl = F(flags);
            if (fl & TF_DEAL_FROM_RINSHAN) {
                fl &= ~TF_DEAL_FROM_RINSHAN;
                int n = F(rinshan_n);
                if (n == 0) { set_err(L, MJ_ERR_WALL); return true; }
                n -= 1;
                F(rinshan_n) = (u8)n;
                tile = F1(wall, 52 + n);
            } else {
                int n = F(yama_n);
                if (n == 0) { set_err(L, MJ_ERR_WALL); return true; }
                n -= 1;
                F(yama_n) = (u8)n;
                tile = F1(wall, 66 + n);
            }
            F(tiles_left) -= 1;
            bool dora_now = (fl & TF_NEW_DORA_AT_TSUMO) != 0;
            fl &= ~TF_NEW_DORA_AT_TSUMO;
            F(flags) = fl;
            if (dora_now) add_new_dora(L);
            log_push(L, LG_WORD(LG_TSUMO, F(tsumo_actor), 0, tile, 0, 0, 0, 0, 0));
            SPROF_T(t_e);
            ev_tsumo(L, F(tsumo_actor), tile);
            SPROF_ADD(11, t_e);
            break;
        }
        case RX_DAHAI: {
            if (fl & TF_NEW_DORA_AT_DISCARD) {
                F(flags) = fl & ~TF_NEW_DORA_AT_DISCARD;
                add_new_dora(L);
            }
            log_push_rx(L, LG_WORD(LG_DAHAI, ev.actor, 0, ev.pai, 0, 0, 0, 0, ev.tsumogiri), ev.tag);
            SPROF_T(t_e);
            ev_dahai(L, ev.actor, ev.pai, ev.tsumogiri);
            SPROF_ADD(12, t_e);
            const int next_actor = (ev.actor + 1) & 3;
            F(tsumo_actor) = (u8)next_actor;
            fl = F(flags);
            if (fl & TF_CAN_FOUR_WIND) {  // check_four_wind (board.rs:314-340)
                bool abort = false;
                if (!(ev.pai >= T_E && ev.pai <= T_N)) {
                    fl &= ~TF_CAN_FOUR_WIND;
                } else if (F1(pflags, next_actor) & PF_CAN_W_RIICHI) {
                    int fw = F(four_wind_tile);
                    if (fw != MJ_NONE) {
                        if (fw != ev.pai) fl &= ~TF_CAN_FOUR_WIND;
                    } else {
                        F(four_wind_tile) = ev.pai;
                    }
                } else {
                    int fw = F(four_wind_tile);
                    if (fw != MJ_NONE) {
                        if (fw == ev.pai) abort = true;
                        else fl &= ~TF_CAN_FOUR_WIND;
                    } else {
                        set_err(L, MJ_ERR_FOUR_WIND);
                    }
                }
                F(flags) = fl;
                if (abort) {  // 四風連打
                    abortive_ryukyoku(L);
                    return true;
                }
            }
            if (F(kans) == 4) {
                bool all_lt4 = true;
                for (int s = 0; s < 4; s++) all_lt4 &= (F2(n_melds, s, 2) + F2(n_melds, s, 3)) < 4;
                if (all_lt4) F(flags) |= TF_CHECK_FOUR_KAN;
            }
            break;
        }
        case RX_CHI:
        case RX_PON:
            check_riichi_accepted(L);
            log_push_rx(L, LG_WORD(ev.type == RX_PON ? LG_PON : LG_CHI, ev.actor, ev.target, ev.pai, ev.c0, ev.c1, 0, 0, 0), ev.tag);
            {
                SPROF_T(t_e);
                ev_chi_pon(L, ev.type == RX_PON, ev.actor, ev.target, ev.pai, ev.c0, ev.c1);
                SPROF_ADD(13, t_e);
            }
            break;
        case RX_ANKAN:
            if (fl & TF_NEW_DORA_AT_DISCARD) {
                F(flags) = fl & ~TF_NEW_DORA_AT_DISCARD;
                add_new_dora(L);
            }
            // consumed = [akaize(t), t, t, t] (agent/mortal.rs:505-520)
            log_push_rx(L, LG_WORD(LG_ANKAN, ev.actor, 0, 0, akaize(ev.pai), ev.pai, ev.pai, ev.pai, 0), ev.tag);
            ev_ankan(L, ev.actor, ev.pai);
            add_new_dora(L);
            F(tsumo_actor) = ev.actor;
            F(flags) |= TF_DEAL_FROM_RINSHAN;
            F(kans) += 1;
            break;
        case RX_DAIMINKAN:
        case RX_KAKAN:
            if (fl & TF_NEW_DORA_AT_DISCARD) F(flags) = fl | TF_NEW_DORA_AT_TSUMO;
            check_riichi_accepted(L);
            if (ev.type == RX_DAIMINKAN) {
                log_push_rx(L, LG_WORD(LG_DAIMINKAN, ev.actor, ev.target, ev.pai, ev.c0, ev.c1, ev.c2, 0, 0), ev.tag);
                ev_daiminkan(L, ev.actor, ev.target, ev.pai, ev.c0, ev.c1, ev.c2);
            } else {
                // pai = the added tile; consumed = the pon it extends: red five first unless it is the added tile
                const int t = deaka(ev.pai);
                log_push_rx(L, LG_WORD(LG_KAKAN, ev.actor, 0, ev.pai, is_aka(ev.pai) ? t : akaize(t), t, t, 0, 0), ev.tag);
                ev_kakan(L, ev.actor, ev.pai);
            }
            F(flags) |= TF_NEW_DORA_AT_DISCARD | TF_DEAL_FROM_RINSHAN;
            F(tsumo_actor) = ev.actor;
            F(kans) += 1;
            break;
        case RX_REACH:
            log_push_rx(L, LG_WORD(LG_REACH, ev.actor, 0, 0, 0, 0, 0, 0, 0), ev.tag);
            ev_reach(L, ev.actor);
            F(riichi_to_be_accepted) = ev.actor;
            break;
        case RX_HORA: {
            SPROF_T(t_e);
            handle_hora(L, ev.actor, ev.target, rx);
            SPROF_ADD(16, t_e);
            return true;
        }
        case RX_RYUKYOKU:  // 九種九牌
            abortive_ryukyoku(L);  // (the reference logs this Ryukyoku without the agent's meta, board.rs:502-509)
            return true;
    }
    // update_paos (board.rs:473-499)
    if ((ev.type == RX_PON || ev.type == RX_DAIMINKAN) && is_jihai(ev.pai)) {
        const int s = ev.actor;
        u32 jz = 0;
        for (int i = 0; i < F2(n_melds, s, 1); i++) {
            int t = F2(pons, s, i);
            if (t >= T_E) jz |= 1u << (t - T_E);
        }
        for (int i = 0; i < F2(n_melds, s, 2); i++) {
            int t = F2(minkans, s, i);
            if (t >= T_E) jz |= 1u << (t - T_E);
        }
        bool sangen = (jz & 0b1110000) == 0b1110000, suushi = (jz & 0b0001111) == 0b0001111;
        if ((sangen && ev.pai >= T_P) || (suushi && ev.pai <= T_N)) F1(paos, s) = ev.target;
    }
    return false;
}

template <class LN> MJD bool any_can_act(const LN& L) {
    return ((F1(cans, 0) | F1(cans, 1) | F1(cans, 2) | F1(cans, 3)) & CAN_ACT) != 0;
}

// Game::poll (game.rs:59-178) with BoardState::poll (board.rs:141-161) inlined.
// Returns true when the table needs a kyoku dealt and `deal_service` is set (mj_k_step: the wavefront deals it together, deal_wall_coop,
// and calls again with `dealt`); false when the poll is over (a seat can act, or the game has ended).
template <class LN> MJDN bool game_poll(const LN& L, Reaction rx[4], const StepParams& P, bool deal_service = false, bool dealt = false) {
    for (;;) {
        u32 fl = F(flags);
        if (fl & TF_ENDED) return false;
        if (F(err) != MJ_OK) { F(flags) = fl | TF_ENDED; return false; }
        bool kyoku_end;
        if (!(fl & TF_KYOKU_STARTED)) {
            const int kyoku = F(kyoku), len = P.game_length;
            bool any30k = false;
            for (int i = 0; i < 4; i++) any30k |= F1(scores, i) >= 30000;
            if (kyoku >= len + 4 || (kyoku >= len && !(fl & TF_IN_RENCHAN) && any30k)) {
                F(flags) = fl | TF_ENDED;
                return false;
            }
            if (deal_service && !dealt) return true;  // (the caller deals, then calls again)
            SPROF_T(t_sk);
            start_kyoku(L, P.deal_algo, dealt);  // haipai + first tsumo == the first board step (board.rs:512-515)
            dealt = false;
            SPROF_ADD(3, t_sk);
            kyoku_end = false;
        } else {
            SPROF_T(t_bs);
            SPROF_CNT(9);
            kyoku_end = board_step(L, rx);
            SPROF_ADD(4, t_bs);
        }
        for (int i = 0; i < 4; i++) rx[i].type = RX_NONE;
        if (!kyoku_end) {
            SPROF_T(t_ca);
            const bool act = any_can_act(L);
            SPROF_ADD(5, t_ca);
            if (act) return false;
            continue;
        }
        SPROF_CNT(10);
        // ---- Poll::End (board.rs:149-157, game.rs:114-174)
        log_push(L, LG_WORD(LG_END_KYOKU, 0, 0, 0, 0, 0, 0, 0, 0));
        fl = F(flags);
        for (int i = 0; i < 4; i++) F1(scores, i) += F1(kyoku_deltas, i);
        if (fl & TF_HAS_ABORTIVE) fl |= TF_CAN_RENCHAN;
        fl &= ~(TF_KYOKU_STARTED | TF_IN_RENCHAN);
        for (int i = 0; i < 4; i++) F1(cans, i) = 0;
        bool tobi = false;
        for (int i = 0; i < 4; i++) tobi |= F1(scores, i) < 0;
        if (tobi) {
            F(flags) = fl | TF_ENDED;
            return false;
        }
        const int kyoku = F(kyoku);
        if (fl & TF_HAS_ABORTIVE) {
            F(honba) += 1;
        } else if (!(fl & TF_CAN_RENCHAN)) {
            F(kyoku) = (u8)(kyoku + 1);
            if (fl & TF_HAS_HORA) F(honba) = 0;
            else F(honba) += 1;
        } else {
            const int oya = kyoku & 3;
            if (kyoku >= P.game_length - 1 && F1(scores, oya) >= 30000) {
                int top = 0;
                for (int i = 1; i < 4; i++)
                    if (F1(scores, i) > F1(scores, top)) top = i;
                if (top == oya) {
                    F(flags) = fl | TF_ENDED;
                    return false;
                }
            }
            fl |= TF_IN_RENCHAN;
            F(honba) += 1;
        }
        F(flags) = fl;
    }
}

// One wavefront serves the 64 tables of a pool block.  A wavefront executes the union of its lanes' paths (deal, draw, call,
// win, ...); narrower wavefronts were measured in round 2 (32 / 16 tables per wavefront: 2.604 / 2.735 ms per v3 cycle against
// 2.596 — a wavefront of 16 tables still walks nearly every path, the split only multiplies the instructions issued) and the
// knob was removed.
__global__ __launch_bounds__(64) void mj_k_step(StepParams P) {
    __shared__ DealScratch s_deal;
    const int blk = blockIdx.x;
    const int table = blk * 64 + threadIdx.x;
    Lane L = {MJ_POOL_PTR(P.blocks + blk), (int)threadIdx.x, &c_mj_tables};
    L.deal = &s_deal;
    if (P.log && table < P.n_tables) {
        L.log = P.log + (size_t)table * P.log_cap;
        L.log_len = P.log_len + table;
        L.log_cap = P.log_cap;
    }
    SPROF_T(t_k0);
    u32 fl = F(flags);
    const bool active = table < P.n_tables && !(fl & TF_INACTIVE);
    bool live_after = false;
    int n_dec = 0, n_quick = 0;
    Reaction rx[4];
    const bool playing = active && !(fl & TF_DONE);
    if (playing) {
        // ---- commit (game.rs:180-218)
        const int pend = F(pending);
        for (int s = 0; s < 4; s++) {
            rx[s].type = RX_NONE;
            if (!((pend >> s) & 1)) continue;
            const int qp = F1(quick_pai, s);
            if (qp != MJ_NONE) {
                rx[s] = {RX_DAHAI, (u8)s, 0, (u8)qp, MJ_NONE, MJ_NONE, MJ_NONE, (u8)(F1(last_self_tsumo, s) == qp), 0ull};
                continue;
            }
            const int agent = (F(agent_of_seat) >> s) & 1;
            const int mr = F1(main_row, s), kr = F1(kan_row, s);
            if (P.reactions[agent]) {
                rx[s] = reaction_from_word(L, s, P.reactions[agent][mr]);
                if (L.log && rx[s].type != RX_NONE)
                    rx[s].tag = (uint64_t)(P.cycle & 0xFFFFFu) | ((uint64_t)(mr & 0x3FFFF) << 20) |
                                ((uint64_t)((F1(shanten, s) + 1) & 15) << 56) | ((uint64_t)((F1(pflags, s) & PF_AT_FURITEN) != 0) << 60) | (1ull << 63);
                continue;
            }
            int action = P.actions[agent] ? P.actions[agent][mr] : 45;
            const int kan_tile = (kr >= 0 && P.actions[agent]) ? P.actions[agent][kr] : -1;
            if (P.enable_agari_guard[agent] && action == 43 && !rule_based_agari(L, s)) {
                // mortal.rs:319-336: take the best alternative; q[43] := f32::MIN, then Iterator::max_by(total_cmp),
                // which keeps the LAST of equal maxima
                const float* q = P.q_values[agent] + (size_t)mr * 46;
                int best = 0;
                int best_key = INT_MIN;
                for (int a = 0; a < 46; a++) {
                    const float v = a == 43 ? -3.40282347e+38f : q[a];
                    int k = __float_as_int(v);
                    k ^= (int)((unsigned)(k >> 31) >> 1);  // f32::total_cmp key
                    if (k >= best_key) {
                        best_key = k;
                        best = a;
                    }
                }
                action = best;
            }
            rx[s] = decode_action(L, s, action, kan_tile);
            if (L.log)
                rx[s].tag = (uint64_t)(P.cycle & 0xFFFFFu) | ((uint64_t)(mr & 0x3FFFF) << 20) |
                            ((uint64_t)((kr >= 0 ? kr + 1 : 0) & 0x3FFFF) << 38) | ((uint64_t)((F1(shanten, s) + 1) & 15) << 56) |
                            ((uint64_t)((F1(pflags, s) & PF_AT_FURITEN) != 0) << 60) | (1ull << 63);
        }
        F(pending) = 0;
        SPROF_ADD(1, t_k0);
    }
    // ---- poll, with the deal as a service of the whole wavefront (mj_deal.h: deal_wall_coop): a lane whose table needs a kyoku leaves its
    // poll loop, the 64 lanes deal the tables that asked one after the other, the lanes go back in.  Wave-uniform control flow.
    {
        SPROF_T(t_gp);
        bool want = playing, dealt = false;
        for (;;) {
            bool need = false;
            if (want) need = game_poll(L, rx, P, true, dealt);
            unsigned long long m = __ballot(need);
            if (m == 0ull) break;
            const u64 my_nonce = need ? F(seed_nonce) : 0ull, my_key = need ? F(seed_key) : 0ull;
            const int my_kh = need ? ((int)F(kyoku) | ((int)F(honba) << 8)) : 0;
            for (; m; m &= m - 1) {
                const int d = __ffsll((long long)m) - 1;
                const u64 nonce = (u64)(u32)__shfl((int)(u32)my_nonce, d) | ((u64)(u32)__shfl((int)(u32)(my_nonce >> 32), d) << 32);
                const u64 key = (u64)(u32)__shfl((int)(u32)my_key, d) | ((u64)(u32)__shfl((int)(u32)(my_key >> 32), d) << 32);
                const int kh = __shfl(my_kh, d);
                deal_wall_coop((uint8_t*)&L.B->wall[0][d], MJ_LANES, &s_deal, d, nonce, key, kh & 0xFF, (kh >> 8) & 0xFF, P.deal_algo, c_mj_tables);
            }
            want = need;
            dealt = need;
        }
        SPROF_ADD(2, t_gp);
    }
    if (playing) {
        SPROF_T(t_cl);
        fl = F(flags);
        if (fl & TF_ENDED) {
            // game.rs:181-197: leftover kyotaku to the (first) top, emit the result
            int kt = F(kyotaku);
            if (kt > 0) {
                int top = 0;
                for (int i = 1; i < 4; i++)
                    if (F1(scores, i) > F1(scores, top)) top = i;
                F1(scores, top) += kt * 1000;
                F(kyotaku) = 0;
            }
            const u32 gid = F(game_id);
            if ((int)gid < P.n_games_total) {
                for (int i = 0; i < 4; i++) P.final_scores[gid * 4 + i] = F1(scores, i);
                P.final_done[gid] = F(err) == MJ_OK ? 1 : 2;
            }
            if (F(err) != MJ_OK) atomicAdd(&P.counters[2], 1ull);
            atomicAdd(&P.counters[1], 1ull);
            fl |= TF_DONE;
            F(flags) = fl;
            for (int a = 0; a < 2; a++) F1(n_rows, a) = 0;
        } else {
            live_after = true;
            // ---- classify the new decisions (mortal.rs:200-250)
            int pend_new = 0, nr[2] = {0, 0};
            for (int s = 0; s < 4; s++) {
                const u32 cans = F1(cans, s);
                F1(main_row, s) = -1;
                F1(kan_row, s) = -1;
                F1(quick_pai, s) = MJ_NONE;
                if (!(cans & CAN_ACT)) continue;
                pend_new |= 1 << s;
                n_dec++;
                const int agent = (F(agent_of_seat) >> s) & 1;
                if (P.enable_quick_eval[agent] && (cans & CAN_DISCARD) &&
                    !(cans & (CAN_RIICHI | CAN_TSUMO_AGARI | CAN_ANKAN | CAN_KAKAN | CAN_RYUKYOKU))) {
                    u64 dc = discard_candidates_aka(L, s);
                    if (__popcll(dc) == 1) {
                        F1(quick_pai, s) = (u8)(__ffsll((long long)dc) - 1);
                        n_quick++;
                        continue;
                    }
                }
                bool need_kan;
                if (!(cans & (CAN_ANKAN | CAN_KAKAN))) need_kan = false;
                else if (!P.enable_quick_eval[agent]) need_kan = true;
                else need_kan = __popcll(F1(ankan_cand, s)) + __popcll(F1(kakan_cand, s)) > 1;
                // local row offsets for now; mj_k_rows adds the table's base
                if (need_kan) F1(kan_row, s) = nr[agent]++;
                F1(main_row, s) = nr[agent]++;
            }
            F(pending) = (u8)pend_new;
            F1(n_rows, 0) = (u8)nr[0];
            F1(n_rows, 1) = (u8)nr[1];
        }
        SPROF_ADD(7, t_cl);
    } else if (active) {
        F1(n_rows, 0) = 0;
        F1(n_rows, 1) = 0;
    } else {
        F1(n_rows, 0) = 0;
        F1(n_rows, 1) = 0;
    }
    // env steps = tables still live after this cycle (reference: `actions += games.len()`, game.rs:303-304)
    unsigned long long live_mask = __ballot(live_after);
    int dec_sum = n_dec, quick_sum = n_quick;
    int rows0 = F1(n_rows, 0), rows1 = F1(n_rows, 1);
    for (int off = MJ_LANES / 2; off > 0; off >>= 1) {
        dec_sum += __shfl_down(dec_sum, off);
        quick_sum += __shfl_down(quick_sum, off);
        rows0 += __shfl_down(rows0, off);
        rows1 += __shfl_down(rows1, off);
    }
    if (threadIdx.x == 0) {
        P.block_rows[2 * blk] = rows0;
        P.block_rows[2 * blk + 1] = rows1;
        if (live_mask) atomicAdd(&P.counters[0], (unsigned long long)__popcll(live_mask));
        if (dec_sum) atomicAdd(&P.counters[3], (unsigned long long)dec_sum);
        if (quick_sum) atomicAdd(&P.counters[4], (unsigned long long)quick_sum);
    }
    SPROF_ADD(0, t_k0);
    SPROF_CNT(8);
}

// Park every table as finished (before the first cycle of a staggered steady-state run): mj_k_refill starts each at its own cycle.
__global__ __launch_bounds__(64) void mj_k_park(TableBlock* blocks, int n_tables) {
    const int table = blockIdx.x * 64 + threadIdx.x;
    Lane L = {MJ_POOL_PTR(blocks + blockIdx.x), (int)threadIdx.x, &c_mj_tables};
    if (table < n_tables && !(F(flags) & TF_INACTIVE)) F(flags) |= TF_DONE;
}

// Restart finished tables with fresh seeds (steady-state throughput mode; not used in parity runs).
__global__ __launch_bounds__(64) void mj_k_refill(StepParams P) {
    const int table = blockIdx.x * 64 + threadIdx.x;
    Lane L = {MJ_POOL_PTR(P.blocks + blockIdx.x), (int)threadIdx.x, &c_mj_tables};
    u32 fl = F(flags);
    if (table >= P.n_tables || (fl & TF_INACTIVE) || !(fl & TF_DONE)) return;
    // staggered first start (mj_pool_set_start_stagger): every table was parked as finished; it enters play at its own cycle,
    // so that a pool is spread over all phases of a hanchan from the beginning instead of marching in step
    if (P.start_stagger && P.cycle < (((uint32_t)table * 2654435761u) >> 8) % P.start_stagger) return;
    F(seed_nonce) += P.refill_stride;
    F(game_id) += (u32)P.n_tables;
    if (P.log_len) P.log_len[table] = 0;
    F(flags) = 0;
    F(kyoku) = 0;
    F(honba) = 0;
    F(kyotaku) = 0;
    F(err) = MJ_OK;
    F(pending) = 0;
    for (int i = 0; i < 4; i++) F1(scores, i) = 25000;
}

// ---------------------------------------------------------------- row assignment
// Row order: table, seat, kan-select row before the main row.  Three steps, all coalesced:
//   mj_k_step      leaves per-table counts (n_rows) and per-block sums (block_rows, wave reduction);
//   mj_k_scan      exclusive scan of the block sums (one workgroup; n_blocks <= a few thousand);
//   mj_k_assign    one lane per table: wave prefix (shfl) + block base -> global row ids + row descriptors.
struct RowsParams {
    TableBlock* blocks;
    int n_blocks;
    int* block_rows;     // [n_blocks][2] in: sums, out (after scan): exclusive bases
    uint32_t* rows[2];   // out: row descriptors per agent
    int* n_rows_out;     // out: [2] totals (device)
    int max_rows[2];
};
__global__ __launch_bounds__(1024) void mj_k_scan(RowsParams P) {
    __shared__ int s_sum[2][1024];
    const int tid = threadIdx.x;
    int carry[2] = {0, 0};
    for (int base = 0; base < P.n_blocks; base += 1024) {
        const int b = base + tid;
        int v0 = b < P.n_blocks ? P.block_rows[2 * b] : 0, v1 = b < P.n_blocks ? P.block_rows[2 * b + 1] : 0;
        s_sum[0][tid] = v0;
        s_sum[1][tid] = v1;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
            int a0 = tid >= off ? s_sum[0][tid - off] : 0, a1 = tid >= off ? s_sum[1][tid - off] : 0;
            __syncthreads();
            s_sum[0][tid] += a0;
            s_sum[1][tid] += a1;
            __syncthreads();
        }
        if (b < P.n_blocks) {
            P.block_rows[2 * b] = carry[0] + s_sum[0][tid] - v0;
            P.block_rows[2 * b + 1] = carry[1] + s_sum[1][tid] - v1;
        }
        carry[0] += s_sum[0][1023];
        carry[1] += s_sum[1][1023];
        __syncthreads();
    }
    if (tid == 0) {
        P.n_rows_out[0] = carry[0];
        P.n_rows_out[1] = carry[1];
    }
}
__global__ __launch_bounds__(64) void mj_k_assign(RowsParams P) {
    TableBlock* B = P.blocks + blockIdx.x;
    const int l = threadIdx.x, t = blockIdx.x * 64 + l;
    const int nr[2] = {B->n_rows[0][l], B->n_rows[1][l]};
    int incl[2] = {nr[0], nr[1]};
    for (int off = 1; off < 64; off <<= 1) {
        int a0 = __shfl_up(incl[0], off), a1 = __shfl_up(incl[1], off);
        if (l >= off) {
            incl[0] += a0;
            incl[1] += a1;
        }
    }
    if ((nr[0] | nr[1]) == 0) return;
    int base[2] = {P.block_rows[2 * blockIdx.x] + incl[0] - nr[0], P.block_rows[2 * blockIdx.x + 1] + incl[1] - nr[1]};
    const int aos = B->agent_of_seat[l];
    for (int s = 0; s < 4; s++) {
        const int agent = (aos >> s) & 1;
        int kr = B->kan_row[s][l], mr = B->main_row[s][l];
        if (kr >= 0) {
            kr += base[agent];
            B->kan_row[s][l] = kr;
            if (kr < P.max_rows[agent]) P.rows[agent][kr] = ROW_PACK(t, s, 1);
        }
        if (mr >= 0) {
            mr += base[agent];
            B->main_row[s][l] = mr;
            if (mr < P.max_rows[agent]) P.rows[agent][mr] = ROW_PACK(t, s, 0);
        }
    }
}

// ---------------------------------------------------------------- snapshot (SoA pool -> one contiguous record per table)
// The pool is lane-major for the step kernel; the encoder wants one table's 2.1 KB in one piece.  A workgroup transposes
// its block of 64 tables through LDS, a quarter of the record at a time: every pool element is read coalesced (64 lanes =
// 64 consecutive elements) and dropped at its record offset in the table's LDS row, then the rows go out as 16-byte
// coalesced stores (33 consecutive lanes per table).  (Round 1 wrote the ~550 elements of a record straight to HBM: 64
// different cache lines per store instruction -- 0.27 ms per cycle for 0.28 GB of traffic.)  Tables without a policy row
// this cycle are not written.
#define SNAP_NCH 4
constexpr int SNAP_CH = (int)(sizeof(TableOne) / SNAP_NCH);
static_assert(sizeof(TableOne) % (SNAP_NCH * 16) == 0, "record = SNAP_NCH chunks of whole 16-byte pieces");
struct SnapParams {
    const TableBlock* blocks;
    TableOne* snap;
    const MjGatherEnt* gather;  // ascending dst_off (field order); naturally aligned elements of <= 8 bytes never straddle a chunk
    int chunk_first[SNAP_NCH + 1];  // gather entries of chunk c: [chunk_first[c], chunk_first[c + 1])
};
__global__ __launch_bounds__(256) void mj_k_snapshot(SnapParams P) {
    __shared__ __attribute__((aligned(16))) char s_rec[MJ_LANES * SNAP_CH];
    __shared__ int s_has[MJ_LANES];
    const TableBlock* B = P.blocks + blockIdx.x;
    const int lane = threadIdx.x & 63, grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool has = (B->n_rows[0][lane] | B->n_rows[1][lane]) != 0;
    if (__ballot(has) == 0) return;  // (the same value in all four wavefronts)
    if (grp == 0) s_has[lane] = has;
    const char* src_base = reinterpret_cast<const char*>(B);
    char* dst_block = reinterpret_cast<char*>(P.snap + (size_t)blockIdx.x * MJ_LANES);
#pragma unroll 1
    for (int c = 0; c < SNAP_NCH; c++) {
        // four elements per round and wavefront: the list entries come through the scalar cache (uniform index) and every
        // element is fetched as the aligned 8-byte word around it -- no size-dependent branch between the four loads, so
        // they are in flight together (a load -> LDS store chain per element made the kernel latency-bound)
        const int end = P.chunk_first[c + 1];
        for (int g = P.chunk_first[c] + grp; g < end; g += 16) {
            MjGatherEnt e[4];
            uint64_t w[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                e[k] = P.gather[min(g + 4 * k, end - 1)];
                const uintptr_t a = reinterpret_cast<uintptr_t>(src_base + e[k].src_off) + (uintptr_t)(lane * e[k].size);
                w[k] = *reinterpret_cast<const uint64_t*>(a & ~(uintptr_t)7) >> (8 * (a & 7));
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (g + 4 * k >= end) break;
                char* d = s_rec + lane * SNAP_CH + ((int)e[k].dst_off - c * SNAP_CH);
                switch (e[k].size) {
                    case 1: *reinterpret_cast<uint8_t*>(d) = (uint8_t)w[k]; break;
                    case 2: *reinterpret_cast<uint16_t*>(d) = (uint16_t)w[k]; break;
                    case 4: *reinterpret_cast<uint32_t*>(d) = (uint32_t)w[k]; break;
                    default: *reinterpret_cast<uint64_t*>(d) = w[k]; break;
                }
            }
        }
        __syncthreads();
        constexpr int PPT = SNAP_CH / 16;  // 16-byte pieces per table and chunk
        for (int p = threadIdx.x; p < MJ_LANES * PPT; p += 256) {
            const int t = p / PPT, o = (p - t * PPT) * 16;
            if (s_has[t])
                *reinterpret_cast<float4*>(dst_block + (size_t)t * sizeof(TableOne) + c * SNAP_CH + o) =
                    *reinterpret_cast<const float4*>(s_rec + t * SNAP_CH + o);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- random policy (config 2)
// action = k-th set bit of the 46-bit mask, k = splitmix64(seed ^ game*K1 ^ seat*K2 ^ kan*K3 ^ cycle*K4) >> 33 mod popcount
__global__ void mj_k_random_policy(const TableBlock* blocks, const uint32_t* rows, const uint8_t* masks, int n_rows,
                                   uint64_t seed, uint64_t cycle, int* actions) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    uint32_t d = rows[r];
    uint32_t t = ROW_TABLE(d);
    uint64_t game = blocks[t >> 6].game_id[t & 63];
    uint64_t x = seed ^ (game * 0xD1B54A32D192ED03ull) ^ ((uint64_t)ROW_SEAT(d) * 0x8CB92BA72F3D8DD7ull) ^
                 ((uint64_t)ROW_KAN(d) * 0xAEF17502108EF2D9ull) ^ (cycle * 0x94D049BB133111EBull);
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x = x ^ (x >> 31);
    const uint8_t* m = masks + (size_t)r * 46;
    uint64_t bits = 0;
    for (int i = 0; i < 46; i++) bits |= (uint64_t)(m[i] != 0) << i;
    int cnt = __popcll(bits);
    int a = 45;
    if (cnt > 0) {
        int k = (int)((x >> 33) % (uint64_t)cnt);
        for (int i = 0; i < k; i++) bits &= bits - 1;
        a = __ffsll((long long)bits) - 1;
    }
    actions[r] = a;
}

// ---------------------------------------------------------------- tenpai-seeking policy (benchmark workload / tests)
// The policy of tests/parity_util.py greedy_actions, on device: always agari, usually riichi, discards that lower (else
// keep) the shanten number — read from the encoded obs' discard block (obs_repr.rs:431-476: rows d0+1 keep-shanten, d0+2
// next-shanten) — and occasional calls; choices come from a counter-based hash of (table, seat, kan flag, cycle).  It keeps
// hands at 0..3 shanten, the regime in which the obs v4 SP tables (mj_sp.hip) carry real state graphs.
__global__ void mj_k_greedy_policy(const uint32_t* rows, const uint8_t* masks, const float* obs, int C, int d0, int n_rows,
                                   uint64_t seed, uint64_t cycle, int* actions) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const uint32_t d = rows[r];
    uint64_t x = seed ^ ((uint64_t)ROW_TABLE(d) * 0xD1B54A32D192ED03ull) ^ ((uint64_t)ROW_SEAT(d) * 0x8CB92BA72F3D8DD7ull) ^
                 ((uint64_t)ROW_KAN(d) * 0xAEF17502108EF2D9ull) ^ (cycle * 0x94D049BB133111EBull);
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x = x ^ (x >> 31);
    const uint64_t hv = x >> 20;
    const uint8_t* mk = masks + (size_t)r * 46;
    uint64_t m = 0;
    for (int i = 0; i < 46; i++) m |= (uint64_t)(mk[i] != 0) << i;
    auto pick = [](uint64_t cands, uint64_t h) {  // the (h mod popcount)-th set bit
        int k = (int)(h % (uint64_t)__popcll(cands));
        for (int i = 0; i < k; i++) cands &= cands - 1;
        return __ffsll((long long)cands) - 1;
    };
    const uint64_t tiles = m & ((1ull << 37) - 1);
    int a = 45;
    if (ROW_KAN(d)) {
        a = pick(m, hv);
    } else if ((m >> 43) & 1) {
        a = 43;
    } else if (((m >> 37) & 1) && hv % 8 != 0) {
        a = 37;
    } else if (((m >> 44) & 1) && hv % 2 == 0) {
        a = 44;
    } else if (((m >> 42) & 1) && tiles && hv % 3 == 0) {
        a = 42;
    } else if (tiles) {
        const float* keep_row = obs + ((size_t)r * C + d0 + 1) * 34;
        const float* next_row = keep_row + 34;
        uint64_t nxt = 0, keep = 0;
        for (int t = 0; t < 37; t++) {
            if (!((tiles >> t) & 1)) continue;
            const int k = deaka(t);
            if (next_row[k] > 0.f) nxt |= 1ull << t;
            if (keep_row[k] > 0.f) keep |= 1ull << t;
        }
        a = pick(nxt ? nxt : keep ? keep : tiles, hv / 7);
    } else {  // reaction to a discard
        const uint64_t chi = (m >> 38) & 7;
        if (((m >> 41) & 1) && hv % 3 == 0) a = 41;
        else if (chi && hv % 4 == 0) a = 38 + pick(chi, hv / 5);
        else if (((m >> 42) & 1) && hv % 2 == 0) a = 42;
        else a = 45;
    }
    actions[r] = a;
}
