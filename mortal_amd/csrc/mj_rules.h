// Per-lane rules engine: one table per lane, state in the SoA pool (mj_state.h).
//
// An mjai event is applied ONCE to the public part of a table and then to the private slice of each of
// the four seats (the reference instead broadcasts it to four full PlayerState copies,
// arena/board.rs:199-204 -> state/update.rs:41-122).
// Functional specs: state/update.rs (event handlers, shanten/wait sets), state/action.rs (legal actions),
// state/agent_helper.rs (discard candidates, agari points), arena/board.rs (BoardState::step/poll),
// arena/game.rs (Game::poll/commit), agent/mortal.rs:200-250,292-573 (quick-eval, kan-select, action decode).
#pragma once
#include "mj_algo.h"
#include "mj_deal.h"
#include "mj_sptab.h"

__constant__ SpTabDev c_sp_tab;  // table-id shanten (mj_sptab.h; set once by mj_tables_upload): the SP kernel and, since round 5, the seats' discard / wait sets

// Section timers of the step kernel (variant build -DMJ_STEP_PROF only: tools/build_variant.sh stepprof -DMJ_STEP_PROF;
// MJ_STEP_PROF=1 in the environment makes mj_counters print them).  A section's clocks are read once per wavefront by
// whichever lanes are inside it, so a sum is "wave time with at least one lane in the section".
#ifdef MJ_STEP_PROF
__device__ unsigned long long g_step_prof[32];
#define SPROF_T(t) const long long t = clock64()
#define SPROF_ADD(k, t)                                                                                   \
    do {                                                                                                  \
        const long long d_ = clock64() - (t);                                                             \
        if ((int)__lane_id() == __ffsll((long long)__ballot(1)) - 1) atomicAdd(&g_step_prof[k], (unsigned long long)d_); \
    } while (0)
#define SPROF_CNT(k)                                                                                      \
    do {                                                                                                  \
        if ((int)__lane_id() == __ffsll((long long)__ballot(1)) - 1) atomicAdd(&g_step_prof[k], 1ull);    \
    } while (0)
#else
#define SPROF_T(t)
#define SPROF_ADD(k, t)
#define SPROF_CNT(k)
#endif

// Lookup tables of the process (set once by mj_tables_upload).  A __constant__ symbol rather than a kernel argument:
// taking the address of a by-value kernel parameter makes hipcc copy the WHOLE parameter struct to scratch in every
// thread (160 B x 256 threads per encoded row showed up as +34 % HBM write traffic in the PMC counters).
__constant__ MjTablesDev c_mj_tables;

template <class BlockT> struct LaneBlockPtr { typedef BlockT* type; };
template <class BlockT>
struct LaneT {  // a table viewed through lane `l` of a block (pool block in HBM, or the 1-lane LDS copy)
    typename LaneBlockPtr<BlockT>::type B;
    int l;
    const MjTablesDev* T;
    uint64_t* log = nullptr;   // this table's event log (NULL = logging off), see mj_state.h LG_*
    uint32_t* log_len = nullptr;
    uint32_t log_cap = 0;
    DealScratch* deal = nullptr;  // the wavefront's LDS work area for deal_wall (kernels that may start a kyoku set it)
};
typedef LaneT<TableBlock> Lane;
#define MJ_POOL_PTR(p) ((LaneBlockPtr<TableBlock>::type)(p))
#define F(f) (L.B->f[L.l])
#define F1(f, i) (L.B->f[i][L.l])
#define F2(f, i, j) (L.B->f[i][j][L.l])
#define F3(f, i, j, k) (L.B->f[i][j][k][L.l])
#define BIT(t) (1ull << (t))

template <class LN> MJD void log_push(const LN& L, uint64_t w) {
    if (!L.log) return;
    const uint32_t n = *L.log_len;
    if (n < L.log_cap) L.log[n] = w;
    else L.B->err[L.l] = L.B->err[L.l] ? L.B->err[L.l] : (uint8_t)MJ_ERR_LOG_OVERFLOW;
    *L.log_len = n + 1;
}
template <class LN> MJD void log_push_i32x4(const LN& L, const int v[4]) {
    log_push(L, (uint64_t)(uint32_t)v[0] | ((uint64_t)(uint32_t)v[1] << 32));
    log_push(L, (uint64_t)(uint32_t)v[2] | ((uint64_t)(uint32_t)v[3] << 32));
}

template <class LN> MJD Hand load_hand(const LN& L, int s) {
    Hand h;
    h.mp = F1(hand_mp, s);
    h.sz = F1(hand_sz, s);
    return h;
}
template <class LN> MJD void store_hand(const LN& L, int s, Hand h) {
    F1(hand_mp, s) = h.mp;
    F1(hand_sz, s) = h.sz;
}
template <class LN> MJD void set_err(const LN& L, int code) {
    if (F(err) == MJ_OK) F(err) = (u8)code;
}
template <class LN> MJD int dora_factor(const LN& L, int t) {  // update.rs:787-788 (derived: #indicators whose next() == t)
    int n = F(n_dora_ind), f = 0;
    for (int i = 0; i < n; i++) f += tile_next(F1(dora_ind, i)) == t;
    return f;
}
template <class LN> MJD void pub_witness(const LN& L, int tile) {  // the public half of witness_tile (update.rs:695-726)
    F1(pub_seen, deaka(tile)) += 1;
    if (is_aka(tile)) F(pub_aka_seen) |= 1 << (tile - T_5MR);
}
template <class LN> MJD bool accepted(const LN& L, int s) { return (F(riichi_accepted) >> s) & 1; }
template <class LN> MJD bool declared(const LN& L, int s) { return (F(riichi_declared) >> s) & 1; }
template <class LN> MJD Melds load_melds(const LN& L, int s) {
    Melds m;
    m.n_chis = F2(n_melds, s, 0);
    m.n_pons = F2(n_melds, s, 1);
    m.n_minkans = F2(n_melds, s, 2);
    m.n_ankans = F2(n_melds, s, 3);
    m.chis = m.pons = m.minkans = m.ankans = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        melds_put(m.chis, i, F2(chis, s, i));
        melds_put(m.pons, i, F2(pons, s, i));
        melds_put(m.minkans, i, F2(minkans, s, i));
        melds_put(m.ankans, i, F2(ankans, s, i));
    }
    return m;
}
template <class LN> MJD int seat_jikaze(const LN& L, int s) { return T_E + ((s + 4 - (F(kyoku) & 3)) & 3); }  // update.rs:154-155
template <class LN> MJD int table_bakaze(const LN& L) { return T_E + F(kyoku) / 4; }

template <class LN> MJDN bool seat_has_yaku(const LN& L, int s, Hand with_tile, int winning_tile, bool is_ron) {
    AgariIn in;
    in.tehai = with_tile;
    in.m = load_melds(L, s);
    in.is_menzen = (F1(pflags, s) & PF_IS_MENZEN) != 0;
    in.bakaze = table_bakaze(L);
    in.jikaze = seat_jikaze(L, s);
    in.winning_tile = winning_tile;
    in.is_ron = is_ron;
    return agari_search(*L.T, in, true).kind != 0;
}

// ---------------------------------------------------------------- shanten / waits (update.rs:870-953)
template <class LN> MJD void update_shanten(const LN& L, int s) {
    int v = calc_all(*L.T, load_hand(L, s), F1(len_div3, s));
    F1(shanten, s) = (int8_t)max(v, 0);
}
// Round 5: both sets come from the table-id formulation of mj_sptab.h (the one mj_k_sp expands its state graphs with): the discards
// of a 3n+2 hand that keep its shanten number and the draws that complete a tenpai hand are a handful of table walks plus mask
// arithmetic, whatever the number of tile kinds.  Rounds 1-4 probed every held kind (<= 14) resp. all 34 kinds with one table gather
// + merge each -- incremental (mj_algo.h sh_others), but still 60 % of mj_k_step's wave time, and a wavefront runs the 34-kind loop
// whenever ONE of its 64 tables has a tenpai seat to refresh.
template <class LN> MJDN void update_shanten_discards(const LN& L, int s) {  // 3n+2
    const Hand h = load_hand(L, s);
    const int ld3 = F1(len_div3, s), sh = F1(shanten, s);
    const SpTabG TG = sp_tab_g(c_sp_tab);
    // the hand with the drawn tile has the number sh14 (one less than before the draw, or the same); a discard leaves sh14 or sh14 + 1
    const int sh14 = calc_all(*L.T, h, ld3);
    const u64 held = h.nonzero_mask();
    const u64 same = sh14 >= 0 ? sp_keep_of_hand(TG, *L.T, h, ld3, sh14) & held : 0ull;  // discards that leave sh14
    const u64 up = held & ~same;                                                          // discards that leave sh14 + 1
#ifdef MJ_EMU
    // host emulation only: `up` is derived (held & ~same), so an error in the table-id set would corrupt the seat's discard
    // sets silently.  Every held kind probed the old way -- calc_all of the hand without it -- must land in the set the masks say.
    for (u64 m = held; m; m &= m - 1) {
        const int t = __ffsll((long long)m) - 1;
        Hand g = h;
        g.dec(t);
        const int v = calc_all(*L.T, g, ld3);
        if (v != (((same >> t) & 1) ? sh14 : sh14 + 1)) set_err(L, MJ_ERR_INTERNAL);
    }
#endif
    const u64 next = (sh14 < sh ? same : 0ull) | (sh14 + 1 < sh ? up : 0ull);
    const u64 keep = (sh14 == sh ? same : 0ull) | (sh14 + 1 == sh ? up : 0ull);
    F1(next_shanten, s) = next;
    F1(keep_shanten, s) = keep;
    F1(has_next_shanten, s) = next != 0;
}
template <class LN> MJDN void update_waits_and_furiten(const LN& L, int s) {  // 3n+1
    u8 pf = F1(pflags, s) & ~PF_AT_FURITEN;
    u64 waits = 0;
    if (F1(shanten, s) <= 0) {
        const Hand h = load_hand(L, s);
        const int ld3 = F1(len_div3, s);
        const u64 disc = F1(discarded, s);
        const SpTabG TG = sp_tab_g(c_sp_tab);
        // the tiles that complete the hand (a fifth copy is not a tile)
        u64 win = 0;
        if (calc_all(*L.T, h, ld3) == 0) win = sp_req_of_hand(TG, *L.T, h, ld3, 0);
#ifdef MJ_EMU
        for (int t = 0; t < 34; t++) {  // host emulation only: the winning tiles by the reference's loop (update.rs:930-951)
            if (h.get(t) == 4) continue;
            Hand g = h;
            g.inc(t);
            if ((calc_all(*L.T, g, ld3) == -1) != (bool)((win >> t) & 1)) set_err(L, MJ_ERR_INTERNAL);
        }
#endif
        for (u64 m = win; m; m &= m - 1) {
            const int t = __ffsll((long long)m) - 1, c = h.get(t);
            if (c == 4) continue;
            if ((disc >> t) & 1) pf |= PF_AT_FURITEN;
            if (F1(pub_seen, t) + c < 4) waits |= BIT(t);  // tiles_seen = pub_seen + own hand
        }
    }
    F1(pflags, s) = pf;
    F1(waits, s) = waits;
}

// ---------------------------------------------------------------- event prologue (update.rs:46-60)
template <class LN> MJD void ev_prologue(const LN& L, int actor /* -1: event without actor */) {
#pragma unroll
    for (int s = 0; s < 4; s++) {
        F1(cans, s) = 0;
        F1(cans_target, s) = (u8)(actor >= 0 ? actor : s);
        F1(ankan_cand, s) = 0;
        F1(kakan_cand, s) = 0;
        u8 pf = F1(pflags, s);
        if (pf & PF_MARK_FURITEN) pf = (pf & ~PF_MARK_FURITEN) | PF_AT_FURITEN;
        if (pf & PF_CHANKAN_CHANCE) pf &= ~(PF_CHANKAN_CHANCE | PF_AT_IPPATSU);
        F1(pflags, s) = pf;
    }
}

template <class LN> MJD void kawa_push(const LN& L, int seat, u64 entry) {
    int n = F1(kawa_len, seat);
    if (n >= MJ_KAWA_MAX) {
        set_err(L, MJ_ERR_KAWA_OVERFLOW);
        return;
    }
    F2(kawa, seat, n) = entry;
    F1(kawa_len, seat) = (u8)(n + 1);
}
template <class LN> MJD void pad_kawa_for_pon_or_daiminkan(const LN& L, int actor, int target) {  // update.rs:810-817
    int i = (target + 1) & 3;
    while (i != actor) {
        kawa_push(L, i, 0);
        i = (i + 1) & 3;
    }
}
template <class LN> MJD void hand_remove(const LN& L, int s, Hand& h, int tile) {  // move_tile Discard/FuuroConsume (update.rs:733-775)
    // the count guard only matters for seats whose hand is hidden (single-perspective replays: "?" tiles); a legal
    // event never removes a tile the seat does not hold
    if (h.get(deaka(tile)) > 0) h.dec(deaka(tile));
    if (is_aka(tile)) F1(akas_in_hand, s) &= ~(1 << (tile - T_5MR));
}

// ---------------------------------------------------------------- events
template <class LN> MJDN void ev_tsumo(const LN& L, int actor, int pai) {  // update.rs:219-309
    ev_prologue(L, actor);
    if (pai >= T_UNK) return;  // another seat's hidden draw ("?"): nothing to track (update.rs:226-229)
    int tiles_left = F(tiles_left);  // already decremented by the board
    const int s = actor;
    F1(at_turn, s) += 1;
    u32 cans = CAN_DISCARD;
    F1(last_self_tsumo, s) = (u8)pai;
    Hand h = load_hand(L, s);
    const int dp = deaka(pai);
    h.inc(dp);
    if (is_aka(pai)) F1(akas_in_hand, s) |= 1 << (pai - T_5MR);
    store_hand(L, s, h);
    const u8 pf = F1(pflags, s);
    const bool acc = accepted(L, s);

    if (pf & PF_CAN_W_RIICHI) {
        if (h.n_yao_kinds() >= 9) cans |= CAN_RYUKYOKU;
    }
    if (!acc) {
        F1(cans, s) = (uint16_t)cans;  // (update_shanten_discards asserts can_discard in the reference)
        update_shanten_discards(L, s);
    }
    if ((F1(waits, s) >> dp) & 1) {
        if ((pf & PF_IS_MENZEN) || acc || tiles_left == 0 || (pf & PF_AT_RINSHAN) || (pf & PF_CAN_W_RIICHI)) {
            cans |= CAN_TSUMO_AGARI;
        } else if (seat_has_yaku(L, s, h, dp, false)) {
            cans |= CAN_TSUMO_AGARI;
        }
    }
    if (tiles_left != 0) {
        const int kob = F(kans_on_board);
        if (acc) {
            if (kob < 4 && check_ankan_after_riichi(*L.T, h, F1(len_div3, s), pai)) {
                cans |= CAN_ANKAN;
                F1(ankan_cand, s) = BIT(dp);
            }
        } else {
            if (kob < 4) {
                u64 ac = 0, kc = 0;
                int np = F2(n_melds, s, 1);
                for (int t = 0; t < 34; t++) {
                    int c = h.get(t);
                    if (c == 0) continue;
                    if (c == 4) ac |= BIT(t);
                    else {
                        bool in_pons = false;
                        for (int i = 0; i < np; i++) in_pons |= F2(pons, s, i) == t;
                        if (in_pons) kc |= BIT(t);
                    }
                }
                if (ac) cans |= CAN_ANKAN;
                if (kc) cans |= CAN_KAKAN;
                F1(ankan_cand, s) = ac;
                F1(kakan_cand, s) = kc;
            }
            int sh = F1(shanten, s);
            if ((pf & PF_IS_MENZEN) && tiles_left >= 4 && F1(scores, s) >= 1000 &&
                (sh == 0 || (sh == 1 && F1(has_next_shanten, s))))
                cans |= CAN_RIICHI;
        }
    }
    F1(cans, s) = (uint16_t)cans;
}

MJD u32 chi_flags_from_tile(Hand h, int tile) {  // update.rs:826-868
    u32 r = 0;
    int tid = deaka(tile), lit = tid % 9 + 1;
    if (lit <= 7 && h.get(tid + 1) > 0 && h.get(tid + 2) > 0) {
        Hand a = h;
        a.clear(tid);
        a.dec(tid + 1);
        a.dec(tid + 2);
        if (lit < 7) a.clear(tid + 3);
        if (!a.empty()) r |= CAN_CHI_LOW;
    }
    if (lit >= 2 && lit <= 8 && h.get(tid - 1) > 0 && h.get(tid + 1) > 0) {
        Hand a = h;
        a.clear(tid);
        a.dec(tid - 1);
        a.dec(tid + 1);
        if (!a.empty()) r |= CAN_CHI_MID;
    }
    if (lit >= 3 && h.get(tid - 2) > 0 && h.get(tid - 1) > 0) {
        Hand a = h;
        a.clear(tid);
        a.dec(tid - 2);
        a.dec(tid - 1);
        if (lit > 3) a.clear(tid - 3);
        if (!a.empty()) r |= CAN_CHI_HIGH;
    }
    return r;
}

template <class LN> MJDN void ev_dahai(const LN& L, int actor, int pai, bool tsumogiri) {  // update.rs:311-427
    ev_prologue(L, actor);
    const int dp = deaka(pai);
    const int tiles_left = F(tiles_left);
    // ---- public
    const bool is_riichi = declared(L, actor) && !accepted(L, actor);
    const bool is_dora = dora_factor(L, dp) > 0;
    {
        u64 e = KW_VALID | ((u64)pai << 1) | ((u64)is_dora << 7) | ((u64)(!tsumogiri) << 8) | ((u64)is_riichi << 9);
        if (F1(inter_cp, 0)) e |= (1ull << 10) | ((u64)F1(inter_cp, 1) << 11) | ((u64)F1(inter_cp, 2) << 17);
        int nk = F(inter_kan_n);
        e |= (u64)nk << 23;
        for (int k = 0; k < nk; k++) e |= (u64)F1(inter_kan, k) << (26 + 6 * k);
        F(inter_kan_n) = 0;
        F1(inter_cp, 0) = 0;
        kawa_push(L, actor, e);
    }
    const u8 su = (u8)(SU_VALID | (is_dora ? SU_DORA : 0) | pai);
    if (!tsumogiri) F1(last_tedashi, actor) = su;
    if (is_riichi) F1(riichi_sutehai, actor) = su;
    pub_witness(L, pai);
#pragma unroll
    for (int s = 0; s < 4; s++) F1(last_kawa_tile, s) = (u8)pai;

    // ---- actor
    {
        const int s = actor;
        Hand h = load_hand(L, s);
        hand_remove(L, s, h, pai);
        store_hand(L, s, h);
        F1(forbidden, s) = 0;
        u8 pf = F1(pflags, s) & ~(PF_AT_RINSHAN | PF_AT_IPPATSU | PF_CAN_W_RIICHI);
        F1(pflags, s) = pf;
        F1(discarded, s) |= BIT(dp);
        if (!accepted(L, s)) {
            if ((F1(next_shanten, s) >> dp) & 1) F1(shanten, s) -= 1;
            else if (!((F1(keep_shanten, s) >> dp) & 1)) update_shanten(L, s);
            update_waits_and_furiten(L, s);
        } else if (!(pf & PF_AT_FURITEN) && ((F1(waits, s) >> dp) & 1)) {
            F1(pflags, s) = pf | PF_AT_FURITEN;
        }
    }
    // ---- the other three
    for (int k = 1; k < 4; k++) {
        const int s = (actor + k) & 3;
        u8 pf = F1(pflags, s);
        u32 cans = 0;
        Hand h = load_hand(L, s);
        const bool acc = accepted(L, s);
        if (!(pf & PF_AT_FURITEN) && ((F1(waits, s) >> dp) & 1)) {
            bool ron;
            if (acc || tiles_left == 0) ron = true;
            else {
                Hand g = h;
                g.inc(dp);
                ron = seat_has_yaku(L, s, g, dp, true);
            }
            if (ron) {
                cans |= CAN_RON_AGARI;
                pf |= PF_MARK_FURITEN;
            } else {
                pf |= PF_AT_FURITEN;
            }
            F1(pflags, s) = pf;
        }
        if (!(acc || tiles_left == 0)) {
            if (k == 1 && !is_jihai(pai) && F1(len_div3, s) > 0) cans |= chi_flags_from_tile(h, pai);  // kamicha discard
            int c = h.get(dp);
            if (c >= 2) cans |= CAN_PON;
            if (F(kans_on_board) < 4 && c == 3) cans |= CAN_DAIMINKAN;
        }
        F1(cans, s) = (uint16_t)cans;
    }
}

template <class LN> MJD void fuuro_push(const LN& L, int seat, const int* tiles, int n) {
    int k = F1(fuuro_n, seat);
    for (int i = 0; i < 4; i++) F3(fuuro, seat, k, i) = (u8)(i < n ? tiles[i] : MJ_NONE);
    F1(fuuro_n, seat) = (u8)(k + 1);
}
template <class LN> MJD void others_on_call(const LN& L, int actor) {  // update.rs:441-449 etc.
    for (int k = 1; k < 4; k++) {
        int s = (actor + k) & 3;
        F1(pflags, s) &= ~(PF_CAN_W_RIICHI | PF_AT_IPPATSU);
    }
}

template <class LN> MJDN void ev_chi_pon(const LN& L, bool is_pon, int actor, int target, int pai, int c0, int c1) {  // update.rs:429-542
    ev_prologue(L, actor);
    int set[3] = {c0, c1, pai};
    fuuro_push(L, actor, set, 3);
    int a = deaka(c0), b = deaka(c1), mn = min(a, b), mx = max(a, b);
    F1(inter_cp, 0) = 1;
    F1(inter_cp, 1) = (u8)mn;
    F1(inter_cp, 2) = (u8)mx;
    if (is_pon) pad_kawa_for_pon_or_daiminkan(L, actor, target);
    pub_witness(L, c0);
    pub_witness(L, c1);
    others_on_call(L, actor);

    const int s = actor;
    F1(cans, s) = CAN_DISCARD;
    F1(pflags, s) &= ~PF_IS_MENZEN;
    F1(len_div3, s) -= 1;
    F1(last_self_tsumo, s) = MJ_NONE;
    Hand h = load_hand(L, s);
    hand_remove(L, s, h, c0);
    hand_remove(L, s, h, c1);
    store_hand(L, s, h);
    const int tid = deaka(pai);
    u64 forb = F1(forbidden, s);
    if (is_pon) {
        int n = F2(n_melds, s, 1);
        F2(pons, s, n) = (u8)tid;
        F2(n_melds, s, 1) = (u8)(n + 1);
        if (h.get(tid) > 0) forb |= BIT(tid);
    } else {
        int n = F2(n_melds, s, 0);
        F2(chis, s, n) = (u8)min(mn, tid);
        F2(n_melds, s, 0) = (u8)(n + 1);
        if (h.get(tid) > 0) forb |= BIT(tid);  // kuikae
        if (tid < mn) {
            if (mx % 9 < 8 && h.get(mx + 1) > 0) forb |= BIT(mx + 1);
        } else if (tid > mx && mn % 9 > 0) {
            if (h.get(mn - 1) > 0) forb |= BIT(mn - 1);
        }
    }
    F1(forbidden, s) = forb;
    update_shanten(L, s);
    update_shanten_discards(L, s);
}

template <class LN> MJDN void ev_daiminkan(const LN& L, int actor, int target, int pai, int c0, int c1, int c2) {  // update.rs:544-582
    ev_prologue(L, actor);
    int set[4] = {c0, c1, c2, pai};
    fuuro_push(L, actor, set, 4);
    int nk = F(inter_kan_n);
    if (nk < 4) F1(inter_kan, nk) = (u8)pai;
    F(inter_kan_n) = (u8)(nk + 1);
    pad_kawa_for_pon_or_daiminkan(L, actor, target);
    F(kans_on_board) += 1;
    pub_witness(L, c0);
    pub_witness(L, c1);
    pub_witness(L, c2);
    others_on_call(L, actor);

    const int s = actor;
    F1(pflags, s) = (F1(pflags, s) | PF_AT_RINSHAN) & ~PF_IS_MENZEN;
    F1(len_div3, s) -= 1;
    Hand h = load_hand(L, s);
    hand_remove(L, s, h, c0);
    hand_remove(L, s, h, c1);
    hand_remove(L, s, h, c2);
    store_hand(L, s, h);
    int n = F2(n_melds, s, 2);
    F2(minkans, s, n) = (u8)deaka(pai);
    F2(n_melds, s, 2) = (u8)(n + 1);
    update_shanten(L, s);
    update_waits_and_furiten(L, s);
}

template <class LN> MJDN void ev_kakan(const LN& L, int actor, int pai) {  // update.rs:584-628
    ev_prologue(L, actor);
    const int dp = deaka(pai);
    int nf = F1(fuuro_n, actor);
    for (int k = 0; k < nf; k++) {
        if (deaka(F3(fuuro, actor, k, 0)) == dp) {
            F3(fuuro, actor, k, 3) = (u8)pai;  // a pon has 3 tiles: the added tile takes slot 3
            break;
        }
    }
    int nk = F(inter_kan_n);
    if (nk < 4) F1(inter_kan, nk) = (u8)pai;
    F(inter_kan_n) = (u8)(nk + 1);
    F(kans_on_board) += 1;
    pub_witness(L, pai);
    for (int k = 1; k < 4; k++) {
        int s = (actor + k) & 3;
        F1(last_kawa_tile, s) = (u8)pai;
        u8 pf = F1(pflags, s);
        if (!(pf & PF_AT_FURITEN) && ((F1(waits, s) >> dp) & 1)) {  // chankan
            F1(cans, s) = CAN_RON_AGARI;
            pf |= PF_MARK_FURITEN | PF_CHANKAN_CHANCE;
        } else {
            pf &= ~PF_AT_IPPATSU;
        }
        F1(pflags, s) = pf;
    }
    const int s = actor;
    F1(pflags, s) |= PF_AT_RINSHAN;
    Hand h = load_hand(L, s);
    hand_remove(L, s, h, pai);
    store_hand(L, s, h);
    {  // pons.retain(!= dp); minkans.push(dp)
        int np = F2(n_melds, s, 1), w = 0;
        for (int i = 0; i < np; i++) {
            u8 t = F2(pons, s, i);
            if (t != dp) F2(pons, s, w++) = t;
        }
        F2(n_melds, s, 1) = (u8)w;
        int n = F2(n_melds, s, 2);
        F2(minkans, s, n) = (u8)dp;
        F2(n_melds, s, 2) = (u8)(n + 1);
    }
    if ((F1(next_shanten, s) >> dp) & 1) F1(shanten, s) -= 1;
    else if (!((F1(keep_shanten, s) >> dp) & 1)) update_shanten(L, s);
    update_waits_and_furiten(L, s);
}

template <class LN> MJDN void ev_ankan(const LN& L, int actor, int tile /* deaka'd */) {  // update.rs:630-663; consumed = [akaize(t), t, t, t]
    ev_prologue(L, actor);
    int n = F1(ankan_n, actor);
    F2(ankan, actor, n) = (u8)tile;
    F1(ankan_n, actor) = (u8)(n + 1);
    int nk = F(inter_kan_n);
    if (nk < 4) F1(inter_kan, nk) = (u8)tile;
    F(inter_kan_n) = (u8)(nk + 1);
    F(kans_on_board) += 1;
#pragma unroll
    for (int s = 0; s < 4; s++) F1(pflags, s) &= ~(PF_CAN_W_RIICHI | PF_AT_IPPATSU);
    pub_witness(L, akaize(tile));
    pub_witness(L, tile);
    pub_witness(L, tile);
    pub_witness(L, tile);

    const int s = actor;
    F1(pflags, s) |= PF_AT_RINSHAN;
    F1(len_div3, s) -= 1;
    Hand h = load_hand(L, s);
    h.clear(tile);
    if (akaize(tile) != tile) F1(akas_in_hand, s) &= ~(1 << (akaize(tile) - T_5MR));
    store_hand(L, s, h);
    int m = F2(n_melds, s, 3);
    F2(ankans, s, m) = (u8)tile;
    F2(n_melds, s, 3) = (u8)(m + 1);
    if (!accepted(L, s)) {
        update_shanten(L, s);
        update_waits_and_furiten(L, s);
    }
}

template <class LN> MJD void ev_dora(const LN& L, int marker) {  // update.rs:780-808 (factor/owned/seen are derived, see mj_state.h)
    ev_prologue(L, -1);
    int n = F(n_dora_ind);
    F1(dora_ind, n) = (u8)marker;
    F(n_dora_ind) = (u8)(n + 1);
    pub_witness(L, marker);
}
template <class LN> MJD void ev_reach(const LN& L, int actor) {  // update.rs:665-675
    ev_prologue(L, actor);
    F(riichi_declared) |= 1 << actor;
    u8 pf = F1(pflags, actor);
    pf = (pf & ~PF_IS_W_RIICHI) | ((pf & PF_CAN_W_RIICHI) ? PF_IS_W_RIICHI : 0);
    F1(pflags, actor) = pf;
    F1(cans, actor) = CAN_DISCARD;
}
template <class LN> MJD void ev_reach_accepted(const LN& L, int actor) {  // update.rs:677-686 + board.rs:342-351
    ev_prologue(L, actor);
    F(riichi_accepted) |= 1 << actor;
    F1(scores, actor) -= 1000;
    F(kyotaku) += 1;
    F(accepted_riichis) += 1;
    F1(pflags, actor) |= PF_AT_IPPATSU;
}

// ---------------------------------------------------------------- kyoku start (board.rs:99-136,206-239; update.rs:125-217)
// kyoku_init: everything a StartKyoku event does to the table, given wall[0..52) (haipai) and wall[60] (first dora
// indicator); shared by the arena (deal from the seed) and the log replay (tiles from the logged event).
// `lw`: the freshly dealt wall as this lane's column of the wavefront's DealScratch (stride DEAL_LANES), or NULL: the wall is read from the pool.
// The four hands, their shanten numbers and the dora marker are taken BEFORE the ~150 field stores below: a load issued after them waits for
// every one (vmcnt counts stores), and start_kyoku runs on ONE lane of a wavefront that is mj_k_step's critical path.
// `pre`: the four hands with their shanten numbers as the wavefront's deal service left them (deal_wall_coop), or NULL: computed here.
template <class LN> MJDN void kyoku_init(const LN& L, const u8* lw = nullptr, const DealPre* pre = nullptr) {
    auto wall_at = [&](int i) -> int {
        if (lw) {
            MJ_ASSUME_LDS(lw);  // (only on this branch: the pointer may be NULL)
            return (int)lw[i * DEAL_LANES];
        }
        return (int)F1(wall, i);
    };
    const int marker = wall_at(56 + 4);
    Hand hh[4];
    u8 ak[4];
    int sv[4];
    if (pre) {
        MJ_ASSUME_LDS(pre);
#pragma unroll
        for (int s = 0; s < 4; s++) {
            hh[s].mp = (u64)pre->hand[s][0] | ((u64)pre->hand[s][1] << 32);
            hh[s].sz = (u64)pre->hand[s][2] | ((u64)pre->hand[s][3] << 32);
            ak[s] = pre->akas[s];
            sv[s] = pre->shanten[s];
        }
    } else {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            Hand h = {0, 0};
            u8 akas = 0;
            for (int i = 0; i < 13; i++) {
                const int t = wall_at(s * 13 + i);
                if (t >= T_UNK) continue;  // hidden hand of a single-perspective log
                h.inc(deaka(t));
                if (is_aka(t)) akas |= 1 << (t - T_5MR);
            }
            hh[s] = h;
            ak[s] = akas;
            sv[s] = calc_all(*L.T, h, 4);  // (update_shanten of the hand just built: len_div3 = 4)
        }
    }
    F(yama_n) = 70;
    F(rinshan_n) = 4;
    F(dora_n) = 5;
    F(tiles_left) = 70;
    F(tsumo_actor) = 0;
    F(riichi_to_be_accepted) = MJ_NONE;
    F(four_wind_tile) = MJ_NONE;
    F(accepted_riichis) = 0;
    F(kans) = 0;
    u32 fl = F(flags);
    fl &= (TF_ENDED | TF_IN_RENCHAN | TF_DONE | TF_INACTIVE);
    fl |= TF_KYOKU_STARTED | TF_CAN_FOUR_WIND | (0xFu * TF_NAGASHI0);
    F(flags) = fl;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        F1(paos, s) = MJ_NONE;
        F1(kyoku_deltas, s) = 0;
        F1(kawa_len, s) = 0;
        F1(last_tedashi, s) = 0;
        F1(riichi_sutehai, s) = 0;
        F1(fuuro_n, s) = 0;
        F1(ankan_n, s) = 0;
        F1(akas_in_hand, s) = 0;
        F1(waits, s) = 0;
        F1(keep_shanten, s) = 0;
        F1(next_shanten, s) = 0;
        F1(forbidden, s) = 0;
        F1(discarded, s) = 0;
        F1(ankan_cand, s) = 0;
        F1(kakan_cand, s) = 0;
        F1(has_next_shanten, s) = 0;
        F1(len_div3, s) = 4;
        F1(at_turn, s) = 0;
        F1(last_self_tsumo, s) = MJ_NONE;
        F1(last_kawa_tile, s) = MJ_NONE;
        F1(cans, s) = 0;
        F1(cans_target, s) = (u8)s;
        F1(pflags, s) = PF_IS_MENZEN | PF_CAN_W_RIICHI;
#pragma unroll
        for (int k = 0; k < 4; k++) F2(n_melds, s, k) = 0;
    }
    F(n_dora_ind) = 0;
    F(riichi_declared) = 0;
    F(riichi_accepted) = 0;
    for (int t = 0; t < 34; t++) F1(pub_seen, t) = 0;
    F(pub_aka_seen) = 0;
    F(kans_on_board) = 0;
    F(inter_kan_n) = 0;
    F1(inter_cp, 0) = 0;

    // StartKyoku: first dora indicator, then 13 tiles per seat
    F(dora_n) = 4;
    F1(dora_ind, 0) = (u8)marker;
    F(n_dora_ind) = 1;
    F1(pub_seen, deaka(marker)) = 1;  // pub_witness of the marker on the counters zeroed above
    if (is_aka(marker)) F(pub_aka_seen) = (u8)(1 << (marker - T_5MR));
#pragma unroll
    for (int s = 0; s < 4; s++) {
        store_hand(L, s, hh[s]);
        F1(akas_in_hand, s) = ak[s];
        F1(shanten, s) = (int8_t)max(sv[s], 0);
    }
    // waits of a hand dealt tenpai (rare); every other seat keeps the zeroed set and its flags (update_waits_and_furiten would only
    // clear a furiten flag that is not set)
#pragma unroll
    for (int s = 0; s < 4; s++)
        if (sv[s] <= 0) update_waits_and_furiten(L, s);
}
// `dealt`: the wavefront's deal service (deal_wall_coop, mj_k_step) has put the wall into the pool, into this lane's column of the LDS
// scratch and the four hands into its DealPre; otherwise the lane deals by itself.
template <class LN> MJDN void start_kyoku(const LN& L, int deal_algo, bool dealt = false) {
    static_assert(MJ_LANES == DEAL_LANES, "one DealScratch column per pool lane");
    const int kyoku = F(kyoku), honba = F(honba);
    SPROF_T(t_d);
    if (!dealt) deal_wall((uint8_t*)&L.B->wall[0][L.l], MJ_LANES, L.deal, L.l, F(seed_nonce), F(seed_key), kyoku, honba, deal_algo);
    SPROF_ADD(19, t_d);
    SPROF_T(t_i);
    const u8* const lw = L.deal ? &L.deal->wall[0][L.l] : nullptr;  // the wall just dealt, still in the wavefront's LDS scratch
    kyoku_init(L, lw, dealt && L.deal ? &L.deal->pre[L.l] : nullptr);
    SPROF_ADD(20, t_i);
    int marker, tile;  // dora marker, the oya's first tsumo
    if (lw) {
        MJ_ASSUME_LDS(lw);
        marker = (int)lw[(56 + 4) * DEAL_LANES];
        tile = (int)lw[(66 + 69) * DEAL_LANES];
    } else {
        marker = (int)F1(wall, 56 + 4);
        tile = (int)F1(wall, 66 + 69);
    }
    const int oya = kyoku & 3;
    F(yama_n) = 69;
    F(tiles_left) = 69;
    F(flags) |= TF_HAIPAI_DONE;
    if (L.log) {  // StartKyoku + the oya's first Tsumo (board.rs:206-239)
        log_push(L, LG_WORD(LG_START_KYOKU, 0, 0, marker, kyoku, 0, 0, 0, 0) | ((uint64_t)honba << LG_HONBA_SHIFT) |
                        ((uint64_t)F(kyotaku) << LG_KYOTAKU_SHIFT));
        int sc[4];
        for (int i = 0; i < 4; i++) sc[i] = F1(scores, i);
        log_push_i32x4(L, sc);
        for (int w = 0; w < 7; w++) {
            uint64_t v = 0;
            for (int k = 0; k < 8; k++) {
                const int i = w * 8 + k;
                if (i < 52) v |= (uint64_t)F1(wall, i) << (8 * k);
            }
            log_push(L, v);
        }
        log_push(L, LG_WORD(LG_TSUMO, oya, 0, tile, 0, 0, 0, 0, 0));
    }
    SPROF_T(t_t);
    ev_tsumo(L, oya, tile);
    SPROF_ADD(21, t_t);
}

// ---------------------------------------------------------------- scoring (agent_helper.rs:377-462)
// Returns false (and flags an error) when the hand is not a hora hand.
// `ura` = explicit ura indicators (rule-based agari guard); NULL = the table's own ura markers wall[61..].
template <class LN> MJDN bool seat_agari_points(const LN& L, int s, bool is_ron, int n_ura, Point& out, const u8* ura = nullptr) {
    const u8 pf = F1(pflags, s);
    const bool is_oya = s == (F(kyoku) & 3);
    if (!is_ron && (pf & PF_CAN_W_RIICHI)) {  // tenhou / chiihou: single yakuman, no stacking
        out = point_yakuman(is_oya, 1);
        return true;
    }
    int wt = is_ron ? F1(last_kawa_tile, s) : F1(last_self_tsumo, s);
    if (wt == MJ_NONE) return false;
    const bool acc = accepted(L, s);
    const int tiles_left = F(tiles_left);
    int add;
    if (is_ron)
        add = (int)acc + ((pf & PF_IS_W_RIICHI) != 0) + ((pf & PF_AT_IPPATSU) != 0) + (tiles_left == 0) +
              ((pf & PF_CHANKAN_CHANCE) != 0);
    else
        add = (int)acc + ((pf & PF_IS_W_RIICHI) != 0) + ((pf & PF_AT_IPPATSU) != 0) + ((pf & PF_IS_MENZEN) != 0) +
              (tiles_left == 0 && !(pf & PF_AT_RINSHAN)) + ((pf & PF_AT_RINSHAN) != 0);
    Hand h = load_hand(L, s);
    // doras_owned[0] = hand + own melds (derived; see mj_state.h)
    int doras = 0;
    const int nd = F(n_dora_ind);
    const int akas = F1(akas_in_hand, s);
    doras += __popc(akas);
    for (int i = 0; i < nd; i++) {
        int d = tile_next(F1(dora_ind, i));
        doras += h.get(d);
        int nf = F1(fuuro_n, s);
        for (int k = 0; k < nf; k++)
            for (int j = 0; j < 4; j++) {
                int t = F3(fuuro, s, k, j);
                if (t != MJ_NONE && deaka(t) == d) doras++;
            }
        int na = F1(ankan_n, s);
        for (int k = 0; k < na; k++)
            if (F2(ankan, s, k) == d) doras += 4;
    }
    {  // akas inside own melds
        int nf = F1(fuuro_n, s);
        for (int k = 0; k < nf; k++)
            for (int j = 0; j < 4; j++) {
                int t = F3(fuuro, s, k, j);
                if (t != MJ_NONE && is_aka(t)) doras++;
            }
        int na = F1(ankan_n, s);
        for (int k = 0; k < na; k++) {
            int t = F2(ankan, s, k);
            if (t == T_5M || t == T_5P || t == T_5S) doras++;
        }
    }
    const int wtd = deaka(wt);
    if (is_ron) {
        h.inc(wtd);
        doras += dora_factor(L, wtd);
        if (is_aka(wt)) doras++;
    }
    if (acc) {
        for (int i = 0; i < n_ura; i++) {
            int nx = tile_next(ura ? ura[i] : F1(wall, 61 + i));
            int c = h.get(nx);
            int na = F1(ankan_n, s);
            for (int k = 0; k < na; k++)
                if (F2(ankan, s, k) == nx) c += 4;
            doras += c;
        }
    }
    AgariIn in;
    in.tehai = h;
    in.m = load_melds(L, s);
    in.is_menzen = (pf & PF_IS_MENZEN) != 0;
    in.bakaze = table_bakaze(L);
    in.jikaze = seat_jikaze(L, s);
    in.winning_tile = wtd;
    in.is_ron = is_ron;
    Agari a = agari_full(*L.T, in, add, doras & 0xFF);
    if (a.kind == 0) return false;
    out = agari_point(a, is_oya);
    return true;
}

// ---------------------------------------------------------------- rule-based agari guard (agent_helper.rs:251-368)
// Whether seat s should take the agari it has been offered: always, except at all-last as a non-oya in 4th place when the
// (optimistically ura-boosted) win would neither lift it out of last place nor end with everybody below 30000.
template <class LN> MJDN bool rule_based_agari(const LN& L, int s) {
    const u32 cans = F1(cans, s);
    if (!(cans & (CAN_TSUMO_AGARI | CAN_RON_AGARI))) return false;
    const bool is_ron = (cans & CAN_RON_AGARI) != 0;
    const int target = F1(cans_target, s);
    const int kyoku_abs = F(kyoku), bak = kyoku_abs >> 2, kyoku = kyoku_abs & 3, oya = kyoku;
    const bool is_all_last = bak == 0 ? false : bak == 1 ? kyoku == 3 : true;
    int sc[4];
    for (int i = 0; i < 4; i++) sc[i] = F1(scores, i);
    auto rank_of = [&](const int* v) {
        int r = 0;
        for (int a = 0; a < 4; a++)
            if (v[a] > v[s] || (v[a] == v[s] && a < s)) r++;
        return r;
    };
    auto all_below_30k = [](const int* v) { return v[0] < 30000 && v[1] < 30000 && v[2] < 30000 && v[3] < 30000; };
    if (!is_all_last || oya == s || rank_of(sc) < 3) return true;
    if (bak == 2) {
        if (kyoku < 3) return true;
    } else if (all_below_30k(sc)) {
        return true;
    }
    Point pt;
    bool ok;
    if (accepted(L, s)) {
        // optimistic ura: indicators chosen to hit the most numerous kinds of the hand first (:287-318)
        Hand h = load_hand(L, s);
        u8 full[34], seen[34];
        for (int t = 0; t < 34; t++) {
            full[t] = (u8)h.get(t);
            seen[t] = (u8)(F1(pub_seen, t) + full[t]);
        }
        const int na = F1(ankan_n, s);
        for (int k = 0; k < na; k++) full[F2(ankan, s, k)] += 4;
        u8 ura[5];
        int n_ura = 0;
        const int n_ind = F(n_dora_ind);
        bool done = false;
        // stable order by count descending == scan counts 8..1 (a kind holds at most 4 + 4), ascending tile id inside
        for (int c = 8; c >= 1 && !done; c--)
            for (int t = 0; t < 34 && !done; t++) {
                if (full[t] != c) continue;
                const int ind = tile_prev(t);
                for (;;) {
                    if (n_ura >= n_ind) {
                        done = true;
                        break;
                    }
                    if (seen[ind] >= 4) break;
                    ura[n_ura++] = (u8)ind;
                    seen[ind]++;
                }
            }
        ok = seat_agari_points(L, s, is_ron, n_ura, pt, ura);
    } else {
        ok = seat_agari_points(L, s, is_ron, 0, pt);
    }
    if (!ok) {
        set_err(L, MJ_ERR_NOT_HORA);
        return true;
    }
    const int honba = F(honba), kyotaku = F(kyotaku);
    if (is_ron) {
        sc[s] += pt.ron + kyotaku * 1000 + honba * 300;
        sc[target] -= pt.ron + honba * 300;
    } else {
        sc[s] += pt.tsumo_oya + 2 * pt.tsumo_ko + kyotaku * 1000 + honba * 300;
        for (int a = 0; a < 4; a++) {
            if (a == s) continue;
            sc[a] -= (a == oya ? pt.tsumo_oya : pt.tsumo_ko) + honba * 100;
        }
    }
    if (all_below_30k(sc)) return true;
    return rank_of(sc) < 3;
}
