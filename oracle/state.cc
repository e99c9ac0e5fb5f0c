// ORACLE — TEST INFRASTRUCTURE ONLY (see mjo.h).
// PlayerState: state/update.rs, state/action.rs, state/agent_helper.rs
#include <algorithm>
#include <sstream>

#include "mjo.h"

namespace mjo {

static AgariCalc make_calc(const PlayerState& s, const u8* tehai, u8 winning_tile, bool is_ron) {
    AgariCalc c;
    c.tehai = tehai;
    c.is_menzen = s.is_menzen;
    c.chis = s.chis.data(); c.n_chis = (int)s.chis.size();
    c.pons = s.pons.data(); c.n_pons = (int)s.pons.size();
    c.minkans = s.minkans.data(); c.n_minkans = (int)s.minkans.size();
    c.ankans = s.ankans.data(); c.n_ankans = (int)s.ankans.size();
    c.bakaze = s.bakaze;
    c.jikaze = s.jikaze;
    c.winning_tile = winning_tile;
    c.is_ron = is_ron;
    return c;
}
static bool contains(const std::vector<u8>& v, u8 t) { return std::find(v.begin(), v.end(), t) != v.end(); }

// ---------------------------------------------------------------- update.rs:41-122
ActionCandidate PlayerState::update(const Event& ev, bool keep_cans_on_announce) {
    if (!keep_cans_on_announce || !ev.is_in_game_announce()) {
        last_cans = ActionCandidate();
        last_cans.target_actor = ev.has_actor() ? ev.actor : player_id;
        ankan_candidates.clear();
        kakan_candidates.clear();
    }
    if (to_mark_same_cycle_furiten) {
        to_mark_same_cycle_furiten = false;
        at_furiten = true;
    }
    if (chankan_chance) {
        chankan_chance = false;
        at_ippatsu = false;
    }
    switch (ev.type) {
        case EV_START_KYOKU: ev_start_kyoku(ev); break;
        case EV_TSUMO: ev_tsumo(ev.actor, ev.pai); break;
        case EV_DAHAI: ev_dahai(ev.actor, ev.pai, ev.tsumogiri); break;
        case EV_CHI: ev_chi(ev.actor, ev.pai, ev.consumed); break;
        case EV_PON: ev_pon(ev.actor, ev.target, ev.pai, ev.consumed); break;
        case EV_DAIMINKAN: ev_daiminkan(ev.actor, ev.target, ev.pai, ev.consumed); break;
        case EV_KAKAN: ev_kakan(ev.actor, ev.pai); break;
        case EV_ANKAN: ev_ankan(ev.actor, ev.consumed); break;
        case EV_DORA: add_dora_indicator(ev.dora_marker); break;
        case EV_REACH: ev_reach(ev.actor); break;
        case EV_REACH_ACCEPTED: ev_reach_accepted(ev.actor); break;
        default: break;
    }
    return last_cans;
}

void PlayerState::ev_start_kyoku(const Event& ev) {  // update.rs:125-217
    memset(tehai, 0, sizeof tehai);
    memset(waits, 0, sizeof waits);
    memset(dora_factor, 0, sizeof dora_factor);
    memset(tiles_seen, 0, sizeof tiles_seen);
    memset(akas_seen, 0, sizeof akas_seen);
    memset(keep_shanten_discards, 0, sizeof keep_shanten_discards);
    memset(next_shanten_discards, 0, sizeof next_shanten_discards);
    memset(forbidden_tiles, 0, sizeof forbidden_tiles);
    memset(discarded_tiles, 0, sizeof discarded_tiles);

    bakaze = ev.bakaze;
    honba = ev.honba;
    kyotaku = ev.kyotaku;
    oya = (u8)rel(ev.oya);
    jikaze = T_E + (4 - oya) % 4;
    kyoku = ev.kyoku - 1;
    is_all_last = bakaze == T_E ? false : bakaze == T_S ? kyoku == 3 : true;

    // scores.rotate_left(player_id)
    for (int i = 0; i < 4; i++) scores[i] = ev.scores[(i + player_id) % 4];

    dora_indicators.clear();
    memset(doras_owned, 0, sizeof doras_owned);
    doras_seen = 0;
    memset(akas_in_hand, 0, sizeof akas_in_hand);

    ankan_candidates.clear();
    kakan_candidates.clear();
    chankan_chance = false;

    at_ippatsu = false;
    at_rinshan = false;
    at_furiten = false;
    to_mark_same_cycle_furiten = false;

    is_menzen = true;
    can_w_riichi = true;
    is_w_riichi = false;
    chis.clear();
    pons.clear();
    minkans.clear();
    ankans.clear();

    kans_on_board = 0;
    tehai_len_div3 = 4;
    has_next_shanten_discard = false;
    tiles_left = 70;
    at_turn = 0;

    for (int i = 0; i < 4; i++) {
        kawa[i].clear();
        last_tedashis[i].reset();
        kawa_overview[i].clear();
        fuuro_overview[i].clear();
        ankan_overview[i].clear();
        riichi_declared[i] = false;
        riichi_accepted[i] = false;
        riichi_sutehais[i].reset();
    }
    intermediate_kan.clear();
    intermediate_chi_pon.reset();

    last_self_tsumo.reset();
    last_kawa_tile.reset();

    update_rank();
    add_dora_indicator(ev.dora_marker);
    for (int i = 0; i < 13; i++) {
        u8 t = ev.tehais[player_id][i];
        witness_tile(t);
        move_tile(t, MV_TSUMO);
    }
    update_shanten();
    update_waits_and_furiten();
    pad_kawa_at_start();
}

void PlayerState::ev_tsumo(u8 actor, u8 pai) {  // update.rs:219-309
    MJO_ENSURE(tiles_left > 0, "rule violation: attempt to tsumo from exhausted yama");
    tiles_left -= 1;
    if (actor != player_id) return;
    at_turn += 1;

    last_cans.can_discard = true;
    last_self_tsumo = pai;
    witness_tile(pai);
    move_tile(pai, MV_TSUMO);

    if (can_w_riichi) last_cans.can_ryukyoku = yaokyuu_kind_count() >= 9;

    if (!riichi_accepted[0]) update_shanten_discards();

    if (waits[deaka(pai)]) {
        if (is_menzen || riichi_accepted[0] || tiles_left == 0 || at_rinshan || can_w_riichi) {
            last_cans.can_tsumo_agari = true;
        } else {
            AgariCalc c = make_calc(*this, tehai, deaka(pai), false);
            last_cans.can_tsumo_agari = c.has_yaku();
        }
    }

    if (tiles_left == 0) return;

    if (riichi_accepted[0]) {
        if (kans_on_board < 4) {
            last_cans.can_ankan = check_ankan_after_riichi(tehai, tehai_len_div3, pai, false);
            if (last_cans.can_ankan) ankan_candidates.push_back(deaka(pai));
        }
        return;
    }

    if (kans_on_board < 4) {
        for (int tid = 0; tid < 34; tid++) {
            u8 count = tehai[tid];
            if (count == 0) continue;
            if (count == 4) {
                last_cans.can_ankan = true;
                ankan_candidates.push_back(tid);
            } else if (contains(pons, (u8)tid)) {
                last_cans.can_kakan = true;
                kakan_candidates.push_back(tid);
            }
        }
    }

    last_cans.can_riichi = is_menzen && tiles_left >= 4 && scores[0] >= 1000 &&
                           (shanten == 0 || (shanten == 1 && has_next_shanten_discard));
}

void PlayerState::ev_dahai(u8 actor, u8 pai, bool tsumogiri) {  // update.rs:311-427
    int actor_rel = rel(actor);
    if (actor_rel == 0) move_tile(pai, MV_DISCARD);
    else witness_tile(pai);

    bool is_riichi = riichi_declared[actor_rel] && !riichi_accepted[actor_rel];
    Sutehai sutehai;
    sutehai.tile = pai;
    sutehai.is_dora = dora_factor[deaka(pai)] > 0;
    sutehai.is_tedashi = !tsumogiri;
    sutehai.is_riichi = is_riichi;
    KawaItem item;
    item.kan = std::move(intermediate_kan);
    intermediate_kan.clear();
    item.chi_pon = intermediate_chi_pon;
    intermediate_chi_pon.reset();
    item.sutehai = sutehai;
    kawa[actor_rel].push_back(item);
    kawa_overview[actor_rel].push_back(pai);
    last_kawa_tile = pai;

    if (!tsumogiri) last_tedashis[actor_rel] = sutehai;
    if (is_riichi) riichi_sutehais[actor_rel] = sutehai;

    u8 dp = deaka(pai);
    if (actor_rel == 0) {
        memset(forbidden_tiles, 0, sizeof forbidden_tiles);
        at_rinshan = false;
        at_ippatsu = false;
        can_w_riichi = false;
        discarded_tiles[dp] = true;

        if (!riichi_accepted[0]) {
            if (next_shanten_discards[dp]) shanten -= 1;
            else if (!keep_shanten_discards[dp]) update_shanten();
            update_waits_and_furiten();
        } else if (!at_furiten && waits[dp]) {
            at_furiten = true;
        }
        return;
    }

    if (!at_furiten && waits[dp]) {
        if (riichi_accepted[0] || tiles_left == 0) {
            last_cans.can_ron_agari = true;
        } else {
            u8 t2[34];
            memcpy(t2, tehai, 34);
            t2[dp] += 1;
            AgariCalc c = make_calc(*this, t2, dp, true);
            last_cans.can_ron_agari = c.has_yaku();
        }
        if (last_cans.can_ron_agari) to_mark_same_cycle_furiten = true;
        else at_furiten = true;
    }

    if (riichi_accepted[0] || tiles_left == 0) return;

    if (actor_rel == 3 && !is_jihai(pai) && tehai_len_div3 > 0) set_can_chi_from_tile(pai);
    last_cans.can_pon = tehai[dp] >= 2;
    last_cans.can_daiminkan = kans_on_board < 4 && tehai[dp] == 3;
}

void PlayerState::ev_chi(u8 actor, u8 pai, const u8 consumed[2]) {  // update.rs:429-495
    int actor_rel = rel(actor);
    std::vector<u8> full_set = {consumed[0], consumed[1], pai};
    fuuro_overview[actor_rel].push_back(full_set);
    ChiPon cp;
    cp.consumed[0] = consumed[0];
    cp.consumed[1] = consumed[1];
    cp.target_tile = pai;
    intermediate_chi_pon = cp;

    if (actor_rel != 0) {
        for (int i = 0; i < 2; i++) witness_tile(consumed[i]);
        for (u8 t : full_set) update_doras_owned(actor_rel, t);
        can_w_riichi = false;
        at_ippatsu = false;
        return;
    }

    last_cans.can_discard = true;
    is_menzen = false;
    tehai_len_div3 -= 1;
    last_self_tsumo.reset();

    update_doras_owned(0, pai);
    for (int i = 0; i < 2; i++) move_tile(consumed[i], MV_FUURO_CONSUME);

    int a = deaka(consumed[0]), b = deaka(consumed[1]);
    int mn = std::min(a, b), mx = std::max(a, b);
    int tid = deaka(pai);
    chis.push_back((u8)std::min(mn, tid));

    if (tehai[tid] > 0) forbidden_tiles[tid] = true;
    if (tid < mn) {
        if (mx % 9 < 8) {
            int bigger = mx + 1;
            if (tehai[bigger] > 0) forbidden_tiles[bigger] = true;
        }
    } else if (tid > mx && mn % 9 > 0) {
        int smaller = mn - 1;
        if (tehai[smaller] > 0) forbidden_tiles[smaller] = true;
    }

    update_shanten();
    update_shanten_discards();
}

void PlayerState::ev_pon(u8 actor, u8 target, u8 pai, const u8 consumed[2]) {  // update.rs:497-542
    int actor_rel = rel(actor);
    std::vector<u8> full_set = {consumed[0], consumed[1], pai};
    fuuro_overview[actor_rel].push_back(full_set);
    ChiPon cp;
    cp.consumed[0] = consumed[0];
    cp.consumed[1] = consumed[1];
    cp.target_tile = pai;
    intermediate_chi_pon = cp;
    pad_kawa_for_pon_or_daiminkan(actor, target);

    if (actor_rel != 0) {
        for (int i = 0; i < 2; i++) witness_tile(consumed[i]);
        for (u8 t : full_set) update_doras_owned(actor_rel, t);
        can_w_riichi = false;
        at_ippatsu = false;
        return;
    }

    last_cans.can_discard = true;
    is_menzen = false;
    tehai_len_div3 -= 1;
    last_self_tsumo.reset();

    update_doras_owned(0, pai);
    for (int i = 0; i < 2; i++) move_tile(consumed[i], MV_FUURO_CONSUME);
    pons.push_back(deaka(pai));

    if (tehai[deaka(pai)] > 0) forbidden_tiles[deaka(pai)] = true;

    update_shanten();
    update_shanten_discards();
}

void PlayerState::ev_daiminkan(u8 actor, u8 target, u8 pai, const u8 consumed[3]) {  // update.rs:544-582
    int actor_rel = rel(actor);
    std::vector<u8> full_set = {consumed[0], consumed[1], consumed[2], pai};
    fuuro_overview[actor_rel].push_back(full_set);
    intermediate_kan.push_back(pai);
    pad_kawa_for_pon_or_daiminkan(actor, target);
    kans_on_board += 1;

    if (actor_rel != 0) {
        for (int i = 0; i < 3; i++) witness_tile(consumed[i]);
        for (u8 t : full_set) update_doras_owned(actor_rel, t);
        can_w_riichi = false;
        at_ippatsu = false;
        return;
    }

    at_rinshan = true;
    is_menzen = false;
    tehai_len_div3 -= 1;

    update_doras_owned(0, pai);
    for (int i = 0; i < 3; i++) move_tile(consumed[i], MV_FUURO_CONSUME);
    minkans.push_back(deaka(pai));

    update_shanten();
    update_waits_and_furiten();
}

void PlayerState::ev_kakan(u8 actor, u8 pai) {  // update.rs:584-628
    int actor_rel = rel(actor);
    for (auto& fuuro : fuuro_overview[actor_rel]) {
        if (deaka(fuuro[0]) == deaka(pai)) {
            fuuro.push_back(pai);
            break;
        }
    }
    intermediate_kan.push_back(pai);
    kans_on_board += 1;

    if (actor_rel != 0) {
        witness_tile(pai);
        update_doras_owned(actor_rel, pai);
        last_kawa_tile = pai;
        if (!at_furiten && waits[deaka(pai)]) {
            last_cans.can_ron_agari = true;
            to_mark_same_cycle_furiten = true;
            chankan_chance = true;
        } else {
            at_ippatsu = false;
        }
        return;
    }

    at_rinshan = true;
    move_tile(pai, MV_FUURO_CONSUME);
    u8 dp = deaka(pai);
    pons.erase(std::remove(pons.begin(), pons.end(), dp), pons.end());
    minkans.push_back(dp);

    if (next_shanten_discards[dp]) shanten -= 1;
    else if (!keep_shanten_discards[dp]) update_shanten();
    update_waits_and_furiten();
}

void PlayerState::ev_ankan(u8 actor, const u8 consumed[4]) {  // update.rs:630-663
    int actor_rel = rel(actor);
    u8 tile = deaka(consumed[0]);
    ankan_overview[actor_rel].push_back(tile);
    intermediate_kan.push_back(tile);
    kans_on_board += 1;

    can_w_riichi = false;
    at_ippatsu = false;

    if (actor_rel != 0) {
        for (int i = 0; i < 4; i++) {
            witness_tile(consumed[i]);
            update_doras_owned(actor_rel, consumed[i]);
        }
        return;
    }

    at_rinshan = true;
    tehai_len_div3 -= 1;
    for (int i = 0; i < 4; i++) move_tile(consumed[i], MV_FUURO_CONSUME);
    ankans.push_back(tile);

    if (!riichi_accepted[0]) {
        update_shanten();
        update_waits_and_furiten();
    }
}

void PlayerState::ev_reach(u8 actor) {  // update.rs:665-675
    int actor_rel = rel(actor);
    riichi_declared[actor_rel] = true;
    if (actor_rel == 0) {
        is_w_riichi = can_w_riichi;
        last_cans.can_discard = true;
    }
}

void PlayerState::ev_reach_accepted(u8 actor) {  // update.rs:677-686
    int actor_rel = rel(actor);
    riichi_accepted[actor_rel] = true;
    scores[actor_rel] -= 1000;
    kyotaku += 1;
    update_rank();
    if (actor_rel == 0) at_ippatsu = true;
}

void PlayerState::witness_tile(u8 tile) {  // update.rs:695-726
    MJO_ENSURE(!is_unknown(tile), "rule violation: attempt to witness an unknown tile");
    int tile_id = deaka(tile);
    MJO_ENSURE(tiles_seen[tile_id] < 4, "rule violation: attempt to witness the fifth " + tile_name(tile));
    tiles_seen[tile_id] += 1;
    doras_seen += dora_factor[tile_id];
    if (is_aka(tile)) {
        akas_seen[tile - T_5MR] = true;
        doras_seen += 1;
    }
}

void PlayerState::move_tile(u8 tile, MoveType mt) {  // update.rs:733-775
    int tile_id = deaka(tile);
    switch (mt) {
        case MV_TSUMO:
            tehai[tile_id] += 1;
            doras_owned[0] += dora_factor[tile_id];
            break;
        case MV_DISCARD:
            MJO_ENSURE(tehai[tile_id] > 0, "rule violation: attempt to discard " + tile_name(tile) + " from void");
            tehai[tile_id] -= 1;
            doras_owned[0] -= dora_factor[tile_id];
            break;
        case MV_FUURO_CONSUME:
            MJO_ENSURE(tehai[tile_id] > 0, "rule violation: attempt to consume " + tile_name(tile) + " from void");
            tehai[tile_id] -= 1;
            break;
    }
    if (is_aka(tile)) {
        int aka_id = tile - T_5MR;
        switch (mt) {
            case MV_TSUMO:
                akas_in_hand[aka_id] = true;
                doras_owned[0] += 1;
                break;
            case MV_DISCARD:
                akas_in_hand[aka_id] = false;
                doras_owned[0] -= 1;
                break;
            case MV_FUURO_CONSUME:
                akas_in_hand[aka_id] = false;
                break;
        }
    }
}

void PlayerState::add_dora_indicator(u8 tile) {  // update.rs:780-808
    dora_indicators.push_back(tile);
    witness_tile(tile);
    u8 next = tile_next(tile);
    dora_factor[next] += 1;
    doras_owned[0] += tehai[next];
    for (int i = 0; i < 4; i++) {
        int cnt = 0;
        for (auto& f : fuuro_overview[i])
            for (u8 t : f)
                if (deaka(t) == next) cnt++;
        doras_owned[i] += cnt;
        if (contains(ankan_overview[i], next)) doras_owned[i] += 4;
    }
    doras_seen += tiles_seen[next];
}

void PlayerState::pad_kawa_for_pon_or_daiminkan(u8 abs_actor, u8 abs_target) {  // update.rs:810-817
    u8 i = (abs_target + 1) % 4;
    while (i != abs_actor) {
        kawa[rel(i)].push_back(std::nullopt);
        i = (i + 1) % 4;
    }
}
void PlayerState::pad_kawa_at_start() {  // update.rs:819-824
    for (int i = 0; i < oya; i++) kawa[i].push_back(std::nullopt);
}

void PlayerState::set_can_chi_from_tile(u8 tile) {  // update.rs:826-868
    last_cans.can_chi_low = last_cans.can_chi_mid = last_cans.can_chi_high = false;
    int tile_id = deaka(tile);
    int literal_num = tile_id % 9 + 1;
    auto any_left = [](const u8* t) {
        for (int i = 0; i < 34; i++)
            if (t[i] > 0) return true;
        return false;
    };
    if (literal_num <= 7 && tehai[tile_id + 1] > 0 && tehai[tile_id + 2] > 0) {
        u8 after[34];
        memcpy(after, tehai, 34);
        after[tile_id] = 0;
        after[tile_id + 1] -= 1;
        after[tile_id + 2] -= 1;
        if (literal_num < 7) after[tile_id + 3] = 0;
        last_cans.can_chi_low = any_left(after);
    }
    if (literal_num >= 2 && literal_num <= 8 && tehai[tile_id - 1] > 0 && tehai[tile_id + 1] > 0) {
        u8 after[34];
        memcpy(after, tehai, 34);
        after[tile_id] = 0;
        after[tile_id - 1] -= 1;
        after[tile_id + 1] -= 1;
        last_cans.can_chi_mid = any_left(after);
    }
    if (literal_num >= 3 && tehai[tile_id - 2] > 0 && tehai[tile_id - 1] > 0) {
        u8 after[34];
        memcpy(after, tehai, 34);
        after[tile_id] = 0;
        after[tile_id - 2] -= 1;
        after[tile_id - 1] -= 1;
        if (literal_num > 3) after[tile_id - 3] = 0;
        last_cans.can_chi_high = any_left(after);
    }
}

void PlayerState::update_shanten() {  // update.rs:875-878
    shanten = (i8)std::max(calc_all(tehai, tehai_len_div3), 0);
}

void PlayerState::update_shanten_discards() {  // update.rs:881-912
    MJO_ENSURE(last_cans.can_discard, "tehai is not 3n+2");
    memset(next_shanten_discards, 0, sizeof next_shanten_discards);
    memset(keep_shanten_discards, 0, sizeof keep_shanten_discards);
    has_next_shanten_discard = false;
    u8 t[34];
    memcpy(t, tehai, 34);
    for (int tid = 0; tid < 34; tid++) {
        if (tehai[tid] == 0) continue;
        t[tid] -= 1;
        int after = calc_all(t, tehai_len_div3);
        t[tid] += 1;
        if (after < shanten) {
            next_shanten_discards[tid] = true;
            has_next_shanten_discard = true;
        } else if (after == shanten) {
            keep_shanten_discards[tid] = true;
        }
    }
}

void PlayerState::update_waits_and_furiten() {  // update.rs:916-953
    MJO_ENSURE(!last_cans.can_discard, "tehai is not 3n+1");
    at_furiten = false;
    memset(waits, 0, sizeof waits);
    if (shanten > 0) return;
    for (int t = 0; t < 34; t++) {
        if (tehai[t] == 4) continue;
        u8 after[34];
        memcpy(after, tehai, 34);
        after[t] += 1;
        if (calc_all(after, tehai_len_div3) == -1) {
            if (discarded_tiles[t]) at_furiten = true;
            waits[t] = tiles_seen[t] < 4;
        }
    }
}

void PlayerState::update_doras_owned(int actor_rel, u8 tile) {  // update.rs:955-960
    doras_owned[actor_rel] += dora_factor[deaka(tile)];
    if (is_aka(tile)) doras_owned[actor_rel] += 1;
}

u8 PlayerState::get_rank(const int scores_rel[4]) const {  // update.rs:966-972 + rankings.rs:8-21
    int abs[4];
    // rotate_right(player_id): abs[(i + player_id) % 4] = rel[i]
    for (int i = 0; i < 4; i++) abs[(i + player_id) % 4] = scores_rel[i];
    int order[4] = {0, 1, 2, 3};
    std::stable_sort(order, order + 4, [&](int a, int b) { return -abs[a] < -abs[b]; });
    for (int r = 0; r < 4; r++)
        if (order[r] == player_id) return (u8)r;
    return 0;
}

// ---------------------------------------------------------------- action.rs:93-228
void PlayerState::ensure_tiles_in_hand(const u8* tiles, int n) const {
    for (int i = 0; i < n; i++) {
        u8 tile = tiles[i];
        MJO_ENSURE(tile < 37 && tehai[deaka(tile)] > 0, tile_name(tile) + " is not in hand");
        if (is_aka(tile)) MJO_ENSURE(akas_in_hand[tile - T_5MR], tile_name(tile) + " is not in hand");
    }
}

static int chi_type(const u8 consumed[2], u8 tile) {  // chi_type.rs:11-25: 0 low 1 mid 2 high
    u8 a = deaka(consumed[0]), b = deaka(consumed[1]);
    u8 mn = std::min(a, b), mx = std::max(a, b);
    u8 t = deaka(tile);
    if (t < mn) return 0;
    if (t < mx) return 1;
    return 2;
}

void PlayerState::validate_reaction(const Event& a) const {
    const ActionCandidate& cans = last_cans;
    if (a.type == EV_RYUKYOKU) {
        MJO_ENSURE(cans.can_ryukyoku, "cannot ryukyoku");
        return;
    }
    if (a.type == EV_NONE) return;
    if (a.has_actor()) {
        MJO_ENSURE(a.actor == player_id, "actor is not self");
    } else {
        throw Error("action does not have actor and is not ryukyoku");
    }
    switch (a.type) {
        case EV_DAHAI:
            MJO_ENSURE(cans.can_discard, "cannot discard");
            ensure_tiles_in_hand(&a.pai, 1);
            if (a.tsumogiri) {
                MJO_ENSURE(last_self_tsumo.has_value(), "tsumogiri but the player has not dealt any tile yet");
                MJO_ENSURE(*last_self_tsumo == a.pai, "cannot tsumogiri");
            }
            break;
        case EV_REACH: MJO_ENSURE(cans.can_riichi, "cannot riichi"); break;
        case EV_CHI: {
            MJO_ENSURE((a.target + 1) % 4 == a.actor, "chi from non-kamicha");
            MJO_ENSURE(last_kawa_tile && *last_kawa_tile == a.pai, "chi target is not the last kawa tile");
            ensure_tiles_in_hand(a.consumed, 2);
            int ct = chi_type(a.consumed, a.pai);
            if (ct == 0) MJO_ENSURE(cans.can_chi_low, "cannot chi low");
            else if (ct == 1) MJO_ENSURE(cans.can_chi_mid, "cannot chi mid");
            else MJO_ENSURE(cans.can_chi_high, "cannot chi high");
            break;
        }
        case EV_PON:
            MJO_ENSURE(a.target != a.actor, "pon from itself");
            MJO_ENSURE(last_kawa_tile && *last_kawa_tile == a.pai, "pon target is not the last kawa tile");
            MJO_ENSURE(cans.can_pon, "cannot pon");
            ensure_tiles_in_hand(a.consumed, 2);
            break;
        case EV_DAIMINKAN:
            MJO_ENSURE(a.target != a.actor, "daiminkan from itself");
            MJO_ENSURE(last_kawa_tile && *last_kawa_tile == a.pai, "daiminkan target is not the last kawa tile");
            MJO_ENSURE(cans.can_daiminkan, "cannot daiminkan");
            ensure_tiles_in_hand(a.consumed, 3);
            break;
        case EV_KAKAN:
            MJO_ENSURE(cans.can_kakan, "cannot kakan");
            MJO_ENSURE(contains(kakan_candidates, deaka(a.pai)), "cannot kakan " + tile_name(a.pai));
            ensure_tiles_in_hand(&a.pai, 1);
            break;
        case EV_ANKAN: {
            MJO_ENSURE(cans.can_ankan, "cannot ankan");
            u8 tile = deaka(a.consumed[0]);
            MJO_ENSURE(contains(ankan_candidates, tile), "cannot ankan " + tile_name(tile));
            ensure_tiles_in_hand(a.consumed, 4);
            break;
        }
        case EV_HORA:
            if (a.target == player_id) MJO_ENSURE(cans.can_tsumo_agari, "cannot tsumo agari");
            else MJO_ENSURE(cans.can_ron_agari, "cannot ron agari");
            break;
        default: throw Error("unexpected action");
    }
}

// ---------------------------------------------------------------- agent_helper.rs
void PlayerState::discard_candidates_aka(bool ret[37]) const {  // :35-79
    MJO_ENSURE(last_cans.can_discard, "tehai is not 3n+2");
    memset(ret, 0, 37);
    if (riichi_accepted[0]) {
        MJO_ENSURE(last_self_tsumo.has_value(), "riichi accepted without last self tsumo");
        ret[*last_self_tsumo] = true;
        return;
    }
    for (int i = 0; i < 34; i++) {
        if (tehai[i] == 0) continue;
        if (riichi_declared[0]) {
            ret[i] = shanten == 1 ? next_shanten_discards[i] : keep_shanten_discards[i];
        } else {
            ret[i] = !forbidden_tiles[i];
        }
    }
    if (ret[T_5M] && akas_in_hand[0]) { ret[T_5MR] = true; ret[T_5M] = tehai[T_5M] > 1; }
    if (ret[T_5P] && akas_in_hand[1]) { ret[T_5PR] = true; ret[T_5P] = tehai[T_5P] > 1; }
    if (ret[T_5S] && akas_in_hand[2]) { ret[T_5SR] = true; ret[T_5S] = tehai[T_5S] > 1; }
}

void PlayerState::discard_candidates_with_unconditional_tenpai(bool ret[34]) const {  // :88-97
    bool full[37];
    discard_candidates_with_unconditional_tenpai_aka(full);
    memcpy(ret, full, 34);
    ret[T_5M] |= full[T_5MR];
    ret[T_5S] |= full[T_5SR];
    ret[T_5P] |= full[T_5PR];
}

void PlayerState::discard_candidates_with_unconditional_tenpai_aka(bool ret[37]) const {  // :100-197
    MJO_ENSURE(last_cans.can_discard, "tehai is not 3n+2");
    memset(ret, 0, 37);
    if (tiles_left == 0 || shanten > 1 || (shanten == 1 && !has_next_shanten_discard)) return;

    if (last_self_tsumo) {
        if (waits[deaka(*last_self_tsumo)]) return;
        if (riichi_accepted[0]) {
            if (!at_furiten) ret[*last_self_tsumo] = true;
            return;
        }
    } else if (calc_all(tehai, tehai_len_div3) == -1) {
        return;
    }

    const bool* tenpai_discards = shanten == 1 ? next_shanten_discards : keep_shanten_discards;
    for (int discard = 0; discard < 34; discard++) {
        if (!(tenpai_discards[discard] && !forbidden_tiles[discard])) continue;
        u8 t1[34];
        memcpy(t1, tehai, 34);
        t1[discard] -= 1;
        for (int tsumo = 0; tsumo < 34; tsumo++) {
            u8 seen = tiles_seen[tsumo];
            if (tsumo == discard || t1[tsumo] == 4) continue;
            u8 t2[34];
            memcpy(t2, t1, 34);
            t2[tsumo] += 1;
            if (calc_all(t2, tehai_len_div3) > -1) continue;
            if (discarded_tiles[tsumo]) {
                ret[discard] = false;
                break;
            }
            if (seen == 4 || ret[discard]) continue;
            AgariCalc c = make_calc(*this, t2, (u8)tsumo, true);
            ret[discard] = c.has_yaku();
        }
    }
    if (ret[T_5M] && akas_in_hand[0]) { ret[T_5MR] = true; ret[T_5M] = tehai[T_5M] > 1; }
    if (ret[T_5P] && akas_in_hand[1]) { ret[T_5PR] = true; ret[T_5P] = tehai[T_5P] > 1; }
    if (ret[T_5S] && akas_in_hand[2]) { ret[T_5SR] = true; ret[T_5S] = tehai[T_5S] > 1; }
}

int PlayerState::yaokyuu_kind_count() const {  // :201-206
    static const u8 Y[13] = {0, 8, 9, 17, 18, 26, 27, 28, 29, 30, 31, 32, 33};
    int n = 0;
    for (u8 i : Y) n += std::min<u8>(tehai[i], 1);
    return n;
}

bool PlayerState::rule_based_agari() const {  // :251-260
    if (!last_cans.can_agari()) return false;
    return rule_based_agari_slow(last_cans.can_ron_agari, rel(last_cans.target_actor));
}

bool PlayerState::rule_based_agari_slow(bool is_ron, int target_rel) const {  // :262-368
    if (!is_all_last || oya == 0 || rank < 3) return true;
    auto all_below_30k = [](const int* s) {
        for (int i = 0; i < 4; i++)
            if (s[i] >= 30000) return false;
        return true;
    };
    if (bakaze == T_W) {
        if (kyoku < 3) return true;
    } else if (all_below_30k(scores)) {
        return true;
    }

    Point max_win_point;
    if (riichi_accepted[0]) {
        u8 tehai_full[34];
        memcpy(tehai_full, tehai, 34);
        for (u8 t : ankan_overview[0]) tehai_full[t] += 4;
        // tehai_ordered_by_count: sort_unstable_by count desc.  Rust's sort_unstable is
        // deterministic (pdqsort / ipnsort) but not stable; for <= 20 elements both use insertion
        // sort, which IS stable, and a hand has at most 14 kinds. So a stable sort reproduces it.
        std::vector<std::pair<int, u8>> ordered;
        for (int t = 0; t < 34; t++)
            if (tehai_full[t] > 0) ordered.push_back({t, tehai_full[t]});
        std::stable_sort(ordered.begin(), ordered.end(), [](auto& l, auto& r) { return l.second > r.second; });
        u8 seen[34];
        memcpy(seen, tiles_seen, 34);
        std::vector<u8> ura;
        bool done = false;
        for (auto& [t, _c] : ordered) {
            if (done) break;
            u8 ura_ind = tile_prev((u8)t);
            for (;;) {
                if (ura.size() >= dora_indicators.size()) {
                    done = true;
                    break;
                }
                if (seen[ura_ind] >= 4) break;
                ura.push_back(ura_ind);
                seen[ura_ind] += 1;
            }
        }
        max_win_point = agari_points(is_ron, ura.data(), (int)ura.size());
    } else {
        max_win_point = agari_points(is_ron, nullptr, 0);
    }

    int exp_scores[4];
    memcpy(exp_scores, scores, sizeof exp_scores);
    if (is_ron) {
        exp_scores[0] += max_win_point.ron + kyotaku * 1000 + honba * 300;
        exp_scores[target_rel] -= max_win_point.ron + honba * 300;
    } else {
        exp_scores[0] += max_win_point.tsumo_total(false) + kyotaku * 1000 + honba * 300;
        for (int idx = 1; idx < 4; idx++) {
            if (idx == oya) exp_scores[idx] -= max_win_point.tsumo_oya + honba * 100;
            else exp_scores[idx] -= max_win_point.tsumo_ko + honba * 100;
        }
    }
    if (all_below_30k(exp_scores)) return true;
    return get_rank(exp_scores) < 3;
}

Point PlayerState::agari_points(bool is_ron, const u8* ura, int n_ura) const {  // :377-462
    MJO_ENSURE((is_ron && last_cans.can_ron_agari) || last_cans.can_tsumo_agari, "cannot agari");
    if (!is_ron && can_w_riichi) return point_yakuman(oya == 0, 1);

    std::optional<u8> wt = is_ron ? last_kawa_tile : last_self_tsumo;
    MJO_ENSURE(wt.has_value(), "cannot find the winning tile");
    u8 winning_tile = *wt;

    int additional_hans;
    if (is_ron) {
        additional_hans = (int)riichi_accepted[0] + (int)is_w_riichi + (int)at_ippatsu + (int)(tiles_left == 0) +
                          (int)chankan_chance;
    } else {
        additional_hans = (int)riichi_accepted[0] + (int)is_w_riichi + (int)at_ippatsu + (int)is_menzen +
                          (int)(tiles_left == 0 && !at_rinshan) + (int)at_rinshan;
    }

    u8 t[34];
    memcpy(t, tehai, 34);
    u8 final_doras_owned = doras_owned[0];
    if (is_ron) {
        int tid = deaka(winning_tile);
        t[tid] += 1;
        final_doras_owned += dora_factor[tid];
        if (is_aka(winning_tile)) final_doras_owned += 1;
    }
    if (riichi_accepted[0]) {
        for (int i = 0; i < n_ura; i++) {
            u8 next = tile_next(ura[i]);
            u8 count = t[next];
            if (contains(ankan_overview[0], next)) count += 4;
            final_doras_owned += count;
        }
    }
    AgariCalc c = make_calc(*this, t, deaka(winning_tile), is_ron);
    auto a = c.agari(additional_hans, final_doras_owned);
    MJO_ENSURE(a.has_value(), "not a hora hand");
    return a->point(oya == 0);
}

int PlayerState::real_time_shanten() const {  // :467-503
    if (!last_cans.can_discard) return shanten;
    if (shanten > 0) return has_next_shanten_discard ? shanten - 1 : shanten;
    if (last_self_tsumo) return waits[deaka(*last_self_tsumo)] ? -1 : 0;
    return calc_all(tehai, tehai_len_div3);
}

std::vector<SPCandidate> PlayerState::single_player_tables() const {  // :509-593
    MJO_ENSURE(tiles_left >= 4, "need at least one more tsumo");
    int cur_shanten = real_time_shanten();
    MJO_ENSURE(cur_shanten >= 0, "can't calculate an agari hand");

    bool can_discard = last_cans.can_discard;
    int tsumos_left;
    bool calc_haitei;
    if (can_discard) {
        tsumos_left = tiles_left / 4;
        calc_haitei = tiles_left % 4 == 0;
    } else {
        int target = rel(last_cans.target_actor);
        int sub = 4 - target;
        int at_next = tiles_left >= sub ? tiles_left - sub : 0;  // saturating_sub
        tsumos_left = at_next / 4;
        calc_haitei = at_next % 4 == 0;
    }
    MJO_ENSURE(tsumos_left >= 1, "need at least one more tsumo");

    u8 num_doras_in_fuuro;
    if (is_menzen && ankan_overview[0].empty()) {
        num_doras_in_fuuro = 0;
    } else {
        u8 in_tehai = 0;
        for (u8 ind : dora_indicators) in_tehai += tehai[tile_next(ind)];
        u8 num_akas = (u8)akas_in_hand[0] + (u8)akas_in_hand[1] + (u8)akas_in_hand[2];
        num_doras_in_fuuro = doras_owned[0] - in_tehai - num_akas;
    }
    bool prefer_riichi = scores[0] >= 1000;
    bool calc_double_riichi = can_discard && can_w_riichi;

    SPInitState init;
    memcpy(init.tehai, tehai, 34);
    memcpy(init.akas_in_hand, akas_in_hand, 3);
    bool is_discard_after_riichi = can_discard && riichi_accepted[0];
    if (is_discard_after_riichi) {
        u8 lt = *last_self_tsumo;
        init.tehai[deaka(lt)] -= 1;
        if (is_aka(lt)) init.akas_in_hand[lt - T_5MR] = false;
        can_discard = false;
    }
    memcpy(init.tiles_seen, tiles_seen, 34);
    memcpy(init.akas_seen, akas_seen, 3);

    SPCalculator sp;
    sp.tehai_len_div3 = tehai_len_div3;
    sp.is_menzen = is_menzen;
    sp.chis = chis.data(); sp.n_chis = (int)chis.size();
    sp.pons = pons.data(); sp.n_pons = (int)pons.size();
    sp.minkans = minkans.data(); sp.n_minkans = (int)minkans.size();
    sp.ankans = ankans.data(); sp.n_ankans = (int)ankans.size();
    sp.bakaze = bakaze;
    sp.jikaze = jikaze;
    sp.num_doras_in_fuuro = num_doras_in_fuuro;
    sp.prefer_riichi = prefer_riichi;
    sp.dora_indicators = dora_indicators.data();
    sp.n_dora_indicators = (int)dora_indicators.size();
    sp.calc_double_riichi = calc_double_riichi;
    sp.calc_haitei = calc_haitei;
    sp.sort_result = true;
    sp.maximize_win_prob = false;
    sp.calc_tegawari = false;
    sp.calc_shanten_down = false;

    auto table = sp.calc(init, can_discard, tsumos_left, cur_shanten);
    if (is_discard_after_riichi) table[0].tile = *last_self_tsumo;
    return table;
}

std::string PlayerState::brief() const {
    std::ostringstream o;
    o << "player " << (int)player_id << " oya(rel) " << (int)oya << " kyoku " << tile_name(bakaze) << (int)kyoku + 1
      << "-" << (int)honba << " turn " << (int)at_turn << " tehai:";
    for (int t = 0; t < 34; t++)
        for (int k = 0; k < tehai[t]; k++) o << " " << tile_name(t);
    o << " akas " << akas_in_hand[0] << akas_in_hand[1] << akas_in_hand[2] << " shanten " << (int)shanten
      << " tiles_left " << (int)tiles_left << " cans d" << last_cans.can_discard << " cl" << last_cans.can_chi_low
      << " cm" << last_cans.can_chi_mid << " ch" << last_cans.can_chi_high << " p" << last_cans.can_pon << " dk"
      << last_cans.can_daiminkan << " kk" << last_cans.can_kakan << " ak" << last_cans.can_ankan << " r"
      << last_cans.can_riichi << " ts" << last_cans.can_tsumo_agari << " ro" << last_cans.can_ron_agari << " ry"
      << last_cans.can_ryukyoku;
    return o.str();
}

}  // namespace mjo
