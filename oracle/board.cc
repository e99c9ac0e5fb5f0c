// ORACLE — TEST INFRASTRUCTURE ONLY (see mjo.h).
// BoardState (arena/board.rs), Game (arena/game.rs), MortalBatchAgent glue (agent/mortal.rs).
#include <algorithm>

#include "mjo.h"

namespace mjo {

// ---------------------------------------------------------------- Board
void Board::init_from_seq(const u8 seq[136]) {  // board.rs:111-122
    for (int i = 0; i < 4; i++) memcpy(haipai[i], seq + i * 13, 13);
    int idx = 52;
    rinshan.assign(seq + idx, seq + idx + 4);
    idx += 4;
    dora_indicators.assign(seq + idx, seq + idx + 5);
    idx += 5;
    ura_indicators.assign(seq + idx, seq + idx + 5);
    idx += 5;
    yama.assign(seq + idx, seq + idx + 70);
}

BoardState::BoardState(const Board& b) : board(b) {  // board.rs:125-136
    oya = board.kyoku % 4;
    dora_indicators_full = board.dora_indicators;
    for (int i = 0; i < 4; i++) player_states[i] = PlayerState((u8)i);
}

KyokuResult BoardState::end() const {  // board.rs:172-182
    KyokuResult r;
    r.kyoku = board.kyoku;
    r.can_renchan = can_renchan;
    r.has_hora = has_hora;
    r.has_abortive_ryukyoku = has_abortive_ryukyoku;
    r.kyotaku_left = board.kyotaku;
    memcpy(r.scores, board.scores, sizeof r.scores);
    return r;
}

void BoardState::broadcast(const Event& ev) {  // board.rs:199-204
    for (auto& s : player_states) s.update(ev);
}

Poll BoardState::poll(std::array<Event, 4> reactions) {  // board.rs:141-161
    for (;;) {
        Poll p = step(reactions);
        if (p == POLL_IN_GAME) {
            for (auto& s : player_states)
                if (s.last_cans.can_act()) return p;
        } else {
            Event e;
            e.type = EV_END_KYOKU;
            log.push_back(e);
            for (int i = 0; i < 4; i++) board.scores[i] += kyoku_deltas[i];
            if (has_abortive_ryukyoku) can_renchan = true;
            return p;
        }
        reactions = std::array<Event, 4>();
    }
}

void BoardState::haipai() {  // board.rs:206-239
    Event sk;
    sk.type = EV_START_KYOKU;
    sk.bakaze = T_E + board.kyoku / 4;
    MJO_ENSURE(!board.dora_indicators.empty(), "insufficient dora indicators");
    sk.dora_marker = board.dora_indicators.back();
    board.dora_indicators.pop_back();
    sk.kyoku = oya + 1;
    sk.honba = board.honba;
    sk.kyotaku = board.kyotaku;
    sk.oya = oya;
    memcpy(sk.scores, board.scores, sizeof sk.scores);
    memcpy(sk.tehais, board.haipai, sizeof sk.tehais);
    broadcast(sk);
    log.push_back(sk);

    MJO_ENSURE(!board.yama.empty(), "invalid yama: empty at init");
    u8 tile = board.yama.back();
    board.yama.pop_back();
    tiles_left -= 1;
    Event ts;
    ts.type = EV_TSUMO;
    ts.actor = oya;
    ts.pai = tile;
    broadcast(ts);
    log.push_back(ts);
}

void BoardState::exhaustive_ryukyoku() {  // board.rs:241-294
    int deltas[4] = {0, 0, 0, 0};
    can_renchan = player_states[oya].shanten == 0;
    bool has_nagashi = false;
    for (int i = 0; i < 4; i++) {
        if (!can_nagashi_mangan[i]) continue;
        has_nagashi = true;
        if (i == oya) {
            for (int k = 0; k < 4; k++) deltas[k] += (k == i) ? 12000 : -4000;
        } else {
            for (int k = 0; k < 4; k++) {
                int d = -2000;
                if (k == i) d = 8000;
                if (k == oya) d = -4000;  // `dod[oya] = -4000` is assigned after `dod[i] = 8000`, i != oya
                deltas[k] += d;
            }
        }
    }
    if (!has_nagashi) {
        int tenpai[4], n = 0;
        for (int i = 0; i < 4; i++)
            if (player_states[i].shanten == 0) tenpai[n++] = i;
        int plus = 0, minus = 0;
        if (n == 1) { plus = 3000; minus = -1000; }
        else if (n == 2) { plus = 1500; minus = -1500; }
        else if (n == 3) { plus = 1000; minus = -3000; }
        if (plus > 0) {
            int dod[4] = {minus, minus, minus, minus};
            for (int k = 0; k < n; k++) dod[tenpai[k]] = plus;
            for (int k = 0; k < 4; k++) deltas[k] += dod[k];
        }
    }
    for (int k = 0; k < 4; k++) kyoku_deltas[k] += deltas[k];
    Event e;
    e.type = EV_RYUKYOKU;
    e.has_deltas = true;
    memcpy(e.deltas, deltas, sizeof deltas);
    log.push_back(e);
}

void BoardState::update_nagashi_mangan_and_four_wind(const Event& ev) {  // board.rs:296-312
    switch (ev.type) {
        case EV_DAHAI:
            if (!is_yaokyuu(ev.pai)) can_nagashi_mangan[ev.actor] = false;
            break;
        case EV_CHI: case EV_PON: case EV_DAIMINKAN:
            can_nagashi_mangan[ev.target] = false;
            can_four_wind = false;
            break;
        case EV_ANKAN: can_four_wind = false; break;
        default: break;
    }
}

bool BoardState::check_four_wind(u8 pai) {  // board.rs:314-340
    if (!(pai >= T_E && pai <= T_N)) {
        can_four_wind = false;
    } else if (player_states[tsumo_actor].can_w_riichi) {
        if (four_wind_tile) can_four_wind = *four_wind_tile == pai;
        else four_wind_tile = pai;
    } else if (four_wind_tile) {
        if (*four_wind_tile == pai) return true;
        can_four_wind = false;
    } else {
        throw Error("unexpected state when calculating four winds");
    }
    return false;
}

void BoardState::check_riichi_accepted() {  // board.rs:342-351
    if (riichi_to_be_accepted) {
        u8 actor = *riichi_to_be_accepted;
        riichi_to_be_accepted.reset();
        Event e;
        e.type = EV_REACH_ACCEPTED;
        e.actor = actor;
        broadcast(e);
        log.push_back(e);
        board.scores[actor] -= 1000;
        board.kyotaku += 1;
        accepted_riichis += 1;
    }
}

void BoardState::add_new_dora() {  // board.rs:353-364
    MJO_ENSURE(!board.dora_indicators.empty(), "illegal kan: already 4 kans and this is the 5th");
    u8 dora = board.dora_indicators.back();
    board.dora_indicators.pop_back();
    Event e;
    e.type = EV_DORA;
    e.dora_marker = dora;
    broadcast(e);
    log.push_back(e);
}

void BoardState::handle_hora(u8 single_actor, u8 single_target, const std::array<Event, 4>& reactions) {  // :366-471
    has_hora = true;
    bool is_ron = single_actor != single_target;
    int honba_left = board.honba;
    int kyotaku_point = board.kyotaku * 1000;
    board.kyotaku = 0;

    int n_ura = 5 - (int)board.dora_indicators.size();
    const u8* ura = board.ura_indicators.data();

    std::optional<Point> points[4];
    for (int i = 0; i < 4; i++) {
        const Event& ev = reactions[i];
        if (ev.type == EV_HORA) {
            can_renchan |= ev.actor == oya;
            points[i] = player_states[ev.actor].agari_points(is_ron, ura, n_ura);
        }
    }

    auto make_hora = [&](int actor, const int deltas[4]) {
        Event e;
        e.type = EV_HORA;
        e.actor = (u8)actor;
        e.target = single_target;
        e.has_deltas = true;
        memcpy(e.deltas, deltas, 4 * sizeof(int));
        if (player_states[actor].riichi_accepted[0]) {
            e.n_ura = n_ura;
            memcpy(e.ura_markers, ura, n_ura);
        } else {
            e.n_ura = 0;
        }
        log.push_back(e);
    };

    if (is_ron) {
        for (int k = 1; k <= 3; k++) {
            int actor = (single_target + k) % 4;
            if (!points[actor]) continue;
            Point point = *points[actor];
            int deltas[4] = {0, 0, 0, 0};
            if (paos[actor]) {
                u8 pao_target = *paos[actor];
                deltas[pao_target] = -point.ron / 2 - honba_left * 300;
                deltas[single_target] -= point.ron / 2;
            } else {
                deltas[single_target] = -point.ron - honba_left * 300;
            }
            deltas[actor] = point.ron + kyotaku_point + honba_left * 300;
            kyotaku_point = 0;
            honba_left = 0;
            for (int i = 0; i < 4; i++) kyoku_deltas[i] += deltas[i];
            make_hora(actor, deltas);
        }
        return;
    }

    MJO_ENSURE(points[single_actor].has_value(), "tsumo hora without points");
    Point point = *points[single_actor];
    int deltas[4] = {0, 0, 0, 0};
    if (paos[single_actor]) {
        deltas[*paos[single_actor]] = -point.ron - honba_left * 300;
    } else {
        for (int i = 0; i < 4; i++) deltas[i] = -point.tsumo_ko - honba_left * 100;
        if (single_actor != oya) deltas[oya] = -point.tsumo_oya - honba_left * 100;
    }
    deltas[single_actor] = point.tsumo_total(single_actor == oya) + kyotaku_point + honba_left * 300;
    for (int i = 0; i < 4; i++) kyoku_deltas[i] += deltas[i];
    make_hora(single_actor, deltas);
}

void BoardState::update_paos(const Event& ev) {  // board.rs:473-499
    if ((ev.type == EV_PON || ev.type == EV_DAIMINKAN) && is_jihai(ev.pai)) {
        u8 jihais = 0;
        const PlayerState& ps = player_states[ev.actor];
        for (u8 t : ps.pons)
            if (t >= T_E) jihais |= 1 << (t - T_E);
        for (u8 t : ps.minkans)
            if (t >= T_E) jihais |= 1 << (t - T_E);
        bool daisangen = (jihais & 0b1110000) == 0b1110000;
        bool daisuushi = (jihais & 0b0001111) == 0b0001111;
        bool is_dragon = ev.pai >= T_P && ev.pai <= T_C;
        bool is_wind = ev.pai >= T_E && ev.pai <= T_N;
        if ((daisangen && is_dragon) || (daisuushi && is_wind)) paos[ev.actor] = ev.target;
    }
}

void BoardState::abortive_ryukyoku() {  // board.rs:502-509
    Event e;
    e.type = EV_RYUKYOKU;
    e.has_deltas = true;
    log.push_back(e);
    has_abortive_ryukyoku = true;
}

Poll BoardState::step(const std::array<Event, 4>& reactions) {  // board.rs:511-678
    if (tiles_left == 70) {
        haipai();
        return POLL_IN_GAME;
    }
    if (accepted_riichis == 4) {
        abortive_ryukyoku();
        return POLL_END;
    }
    for (int actor = 0; actor < 4; actor++) {
        try {
            player_states[actor].validate_reaction(reactions[actor]);
        } catch (const Error& e) {
            throw Error(std::string("invalid action: ") + e.what() + "\nstate:\n" + player_states[actor].brief());
        }
    }
    // min_by_key -> first minimum
    auto prio = [](const Event& e) {
        switch (e.type) {
            case EV_HORA: return 0;
            case EV_DAIMINKAN: case EV_PON: return 1;
            case EV_NONE: return 3;
            default: return 2;
        }
    };
    int best = 0;
    for (int i = 1; i < 4; i++)
        if (prio(reactions[i]) < prio(reactions[best])) best = i;
    const Event& ev = reactions[best];

    if (check_four_kan && ev.type != EV_HORA) {
        abortive_ryukyoku();
        return POLL_END;
    }

    update_nagashi_mangan_and_four_wind(ev);

    switch (ev.type) {
        case EV_NONE: {
            if (tiles_left == 0) {
                exhaustive_ryukyoku();
                return POLL_END;
            }
            check_riichi_accepted();
            u8 tile;
            if (deal_from_rinshan) {
                deal_from_rinshan = false;
                MJO_ENSURE(!board.rinshan.empty(), "illegal kan: already 4 kans and this is the 5th");
                tile = board.rinshan.back();
                board.rinshan.pop_back();
            } else {
                MJO_ENSURE(!board.yama.empty(), "tiles left > 0 but yama is empty");
                tile = board.yama.back();
                board.yama.pop_back();
            }
            tiles_left -= 1;
            Event ts;
            ts.type = EV_TSUMO;
            ts.actor = tsumo_actor;
            ts.pai = tile;
            if (need_new_dora_at_tsumo) {
                need_new_dora_at_tsumo = false;
                add_new_dora();
            }
            broadcast(ts);
            log.push_back(ts);
            break;
        }
        case EV_DAHAI: {
            if (need_new_dora_at_discard) {
                need_new_dora_at_discard = false;
                add_new_dora();
            }
            broadcast(ev);
            log.push_back(ev);
            tsumo_actor = (ev.actor + 1) % 4;
            if (can_four_wind && check_four_wind(ev.pai)) {
                abortive_ryukyoku();
                return POLL_END;
            }
            if (kans == 4) {
                bool all_lt4 = true;
                for (auto& s : player_states)
                    if (s.kans_count() >= 4) all_lt4 = false;
                if (all_lt4) check_four_kan = true;
            }
            break;
        }
        case EV_CHI:
        case EV_PON:
            check_riichi_accepted();
            broadcast(ev);
            log.push_back(ev);
            break;
        case EV_ANKAN:
            if (need_new_dora_at_discard) {
                need_new_dora_at_discard = false;
                add_new_dora();
            }
            broadcast(ev);
            log.push_back(ev);
            add_new_dora();
            tsumo_actor = ev.actor;
            deal_from_rinshan = true;
            kans += 1;
            break;
        case EV_DAIMINKAN:
        case EV_KAKAN:
            if (need_new_dora_at_discard) need_new_dora_at_tsumo = true;
            check_riichi_accepted();
            broadcast(ev);
            log.push_back(ev);
            need_new_dora_at_discard = true;
            tsumo_actor = ev.actor;
            deal_from_rinshan = true;
            kans += 1;
            break;
        case EV_REACH:
            broadcast(ev);
            log.push_back(ev);
            riichi_to_be_accepted = ev.actor;
            break;
        case EV_HORA:
            handle_hora(ev.actor, ev.target, reactions);
            return POLL_END;
        case EV_RYUKYOKU:
            abortive_ryukyoku();
            return POLL_END;
        default: throw Error("unexpected event");
    }
    update_paos(ev);
    return POLL_IN_GAME;
}

// ---------------------------------------------------------------- agent glue (agent/mortal.rs)
SceneInfo agent_scene(const PlayerState& st, u8 actor, bool enable_quick_eval) {  // mortal.rs:200-250
    SceneInfo info;
    const ActionCandidate& cans = st.last_cans;
    info.can_act = cans.can_act();
    if (!info.can_act) return info;
    if (enable_quick_eval && cans.can_discard && !cans.can_riichi && !cans.can_tsumo_agari && !cans.can_ankan &&
        !cans.can_kakan && !cans.can_ryukyoku) {
        bool cand[37];
        st.discard_candidates_aka(cand);
        int only = -1, n = 0;
        for (int t = 0; t < 37; t++)
            if (cand[t]) {
                n++;
                if (n == 1) only = t;
            }
        // mortal.rs:220-228: `only_candidate.take()` then break on the 2nd -> None when >= 2
        if (n == 1) {
            info.quick_eval = true;
            Event e;
            e.type = EV_DAHAI;
            e.actor = actor;
            e.pai = (u8)only;
            e.tsumogiri = st.last_self_tsumo && *st.last_self_tsumo == (u8)only;
            info.quick_event = e;
            return info;
        }
    }
    if (!cans.can_ankan && !cans.can_kakan) info.need_kan_select = false;
    else if (!enable_quick_eval) info.need_kan_select = true;
    else info.need_kan_select = st.ankan_candidates.size() + st.kakan_candidates.size() > 1;
    return info;
}

Event agent_decode_action(const PlayerState& st, u8 actor, int action, int kan_tile) {  // mortal.rs:338-573
    const ActionCandidate& cans = st.last_cans;
    const bool* akas = st.akas_in_hand;
    Event e;
    auto contains = [](const std::vector<u8>& v, u8 t) { return std::find(v.begin(), v.end(), t) != v.end(); };
    auto aka_for = [&](u8 pai, u8 a, u8 b) {  // matches `match pai.as_u8()` on exact (non-deaka'd) ids
        if (pai == a || pai == b) return akas[0];
        if (pai == a + 9 || pai == b + 9) return akas[1];
        if (pai == a + 18 || pai == b + 18) return akas[2];
        return false;
    };
    if (action >= 0 && action <= 36) {
        MJO_ENSURE(cans.can_discard, "failed discard check: " + st.brief());
        e.type = EV_DAHAI;
        e.actor = actor;
        e.pai = (u8)action;
        e.tsumogiri = st.last_self_tsumo && *st.last_self_tsumo == (u8)action;
        return e;
    }
    switch (action) {
        case 37:
            MJO_ENSURE(cans.can_riichi, "failed riichi check: " + st.brief());
            e.type = EV_REACH;
            e.actor = actor;
            return e;
        case 38: {
            MJO_ENSURE(cans.can_chi_low, "failed chi low check: " + st.brief());
            MJO_ENSURE(st.last_kawa_tile.has_value(), "invalid state: no last kawa tile");
            u8 pai = *st.last_kawa_tile;
            u8 first = tile_next(pai);
            bool can_aka = aka_for(pai, 2, 3);  // 3m|4m ...
            e.type = EV_CHI;
            e.actor = actor;
            e.target = cans.target_actor;
            e.pai = pai;
            if (can_aka) { e.consumed[0] = akaize(first); e.consumed[1] = akaize(tile_next(first)); }
            else { e.consumed[0] = first; e.consumed[1] = tile_next(first); }
            return e;
        }
        case 39: {
            MJO_ENSURE(cans.can_chi_mid, "failed chi mid check: " + st.brief());
            MJO_ENSURE(st.last_kawa_tile.has_value(), "invalid state: no last kawa tile");
            u8 pai = *st.last_kawa_tile;
            bool can_aka = aka_for(pai, 3, 5);  // 4m|6m ...
            e.type = EV_CHI;
            e.actor = actor;
            e.target = cans.target_actor;
            e.pai = pai;
            if (can_aka) { e.consumed[0] = akaize(tile_prev(pai)); e.consumed[1] = akaize(tile_next(pai)); }
            else { e.consumed[0] = tile_prev(pai); e.consumed[1] = tile_next(pai); }
            return e;
        }
        case 40: {
            MJO_ENSURE(cans.can_chi_high, "failed chi high check: " + st.brief());
            MJO_ENSURE(st.last_kawa_tile.has_value(), "invalid state: no last kawa tile");
            u8 pai = *st.last_kawa_tile;
            u8 last = tile_prev(pai);
            bool can_aka = aka_for(pai, 5, 6);  // 6m|7m ...
            e.type = EV_CHI;
            e.actor = actor;
            e.target = cans.target_actor;
            e.pai = pai;
            if (can_aka) { e.consumed[0] = akaize(tile_prev(last)); e.consumed[1] = akaize(last); }
            else { e.consumed[0] = tile_prev(last); e.consumed[1] = last; }
            return e;
        }
        case 41: {
            MJO_ENSURE(cans.can_pon, "failed pon check: " + st.brief());
            MJO_ENSURE(st.last_kawa_tile.has_value(), "invalid state: no last kawa tile");
            u8 pai = *st.last_kawa_tile;
            bool can_aka = pai == T_5M ? akas[0] : pai == T_5P ? akas[1] : pai == T_5S ? akas[2] : false;
            e.type = EV_PON;
            e.actor = actor;
            e.target = cans.target_actor;
            e.pai = pai;
            if (can_aka) { e.consumed[0] = akaize(pai); e.consumed[1] = deaka(pai); }
            else { e.consumed[0] = e.consumed[1] = deaka(pai); }
            return e;
        }
        case 42: {
            MJO_ENSURE(cans.can_daiminkan || cans.can_ankan || cans.can_kakan, "failed kan check: " + st.brief());
            u8 tile;
            if (kan_tile >= 0) {
                tile = (u8)kan_tile;
                MJO_ENSURE(contains(st.ankan_candidates, tile) || contains(st.kakan_candidates, tile),
                           "kan choice not in kan candidates: " + st.brief());
            } else if (cans.can_daiminkan) {
                MJO_ENSURE(st.last_kawa_tile.has_value(), "invalid state: no last kawa tile");
                tile = *st.last_kawa_tile;
            } else if (cans.can_ankan) {
                tile = st.ankan_candidates.at(0);
            } else {
                tile = st.kakan_candidates.at(0);
            }
            e.actor = actor;
            if (cans.can_daiminkan) {
                e.type = EV_DAIMINKAN;
                e.target = cans.target_actor;
                e.pai = tile;
                if (is_aka(tile)) { e.consumed[0] = e.consumed[1] = e.consumed[2] = deaka(tile); }
                else { e.consumed[0] = akaize(tile); e.consumed[1] = e.consumed[2] = tile; }
            } else if (cans.can_ankan && contains(st.ankan_candidates, deaka(tile))) {
                e.type = EV_ANKAN;
                e.consumed[0] = akaize(tile);
                e.consumed[1] = e.consumed[2] = e.consumed[3] = tile;
            } else {
                bool can_aka = tile == T_5M ? akas[0] : tile == T_5P ? akas[1] : tile == T_5S ? akas[2] : false;
                e.type = EV_KAKAN;
                if (can_aka) {
                    e.pai = akaize(tile);
                    e.consumed[0] = e.consumed[1] = e.consumed[2] = deaka(tile);
                } else {
                    e.pai = deaka(tile);
                    e.consumed[0] = akaize(tile);
                    e.consumed[1] = e.consumed[2] = deaka(tile);
                }
            }
            return e;
        }
        case 43:
            MJO_ENSURE(cans.can_agari(), "failed hora check: " + st.brief());
            e.type = EV_HORA;
            e.actor = actor;
            e.target = cans.target_actor;
            return e;
        case 44:
            MJO_ENSURE(cans.can_ryukyoku, "failed ryukyoku check: " + st.brief());
            e.type = EV_RYUKYOKU;
            return e;
        default: return e;  // 45: Event::None
    }
}

// ---------------------------------------------------------------- Game (arena/game.rs)
void Game::poll() {  // game.rs:59-178
    if (ended) return;
    if (!kyoku_started) {
        bool any30k = false;
        for (int i = 0; i < 4; i++)
            if (scores[i] >= 30000) any30k = true;
        if (kyoku >= length + 4 || (kyoku >= length && !in_renchan && any30k)) {
            ended = true;
            return;
        }
        Board nb;
        nb.kyoku = kyoku;
        nb.honba = honba;
        nb.kyotaku = kyotaku;
        memcpy(nb.scores, scores, sizeof scores);
        u8 seq[136];
        deal_from_seed(seed_nonce, seed_key, kyoku, honba, deal_algo, seq);
        nb.init_from_seq(seq);
        board.emplace(nb);
        kyoku_started = true;
    }
    std::array<Event, 4> reactions = last_reactions;
    last_reactions = std::array<Event, 4>();
    Poll p = board->poll(reactions);
    if (p == POLL_IN_GAME) return;

    kyoku_started = false;
    in_renchan = false;
    KyokuResult kr = board->end();
    kyotaku = kr.kyotaku_left;
    memcpy(scores, kr.scores, sizeof scores);
    if (keep_log) game_log.push_back(std::move(board->log));

    for (int i = 0; i < 4; i++)
        if (scores[i] < 0) {
            ended = true;
            return;
        }
    if (kr.has_abortive_ryukyoku) {
        honba += 1;
        return poll();
    }
    if (!kr.can_renchan) {
        kyoku += 1;
        if (kr.has_hora) honba = 0;
        else honba += 1;
        return poll();
    }
    int oya = kr.kyoku % 4;
    if (kr.kyoku >= length - 1 && scores[oya] >= 30000) {
        int top = 0;  // min_by_key(-s): first max
        for (int i = 1; i < 4; i++)
            if (kr.scores[i] > kr.scores[top]) top = i;
        if (top == oya) {
            ended = true;
            return;
        }
    }
    in_renchan = true;
    honba += 1;
    return poll();
}

bool Game::commit_end() {  // game.rs:180-197
    if (!ended) return false;
    if (kyotaku > 0) {
        int top = 0;
        for (int i = 1; i < 4; i++)
            if (scores[i] > scores[top]) top = i;
        scores[top] += kyotaku * 1000;
    }
    return true;
}


// ---------------------------------------------------------------- board.rs:679-782
void BoardState::encode_oracle_obs(u8 perspective, int version, float* out) const {
    const int rows = oracle_obs_rows(version);
    memset(out, 0, sizeof(float) * rows * 34);
    auto assign = [&](int r, int c, float v) { out[r * 34 + c] = v; };
    auto fill = [&](int r, float v) {
        for (int c = 0; c < 34; c++) out[r * 34 + c] = v;
    };
    int idx = 0;
    for (int k = 1; k <= 3; k++) {
        const PlayerState& st = player_states[(perspective + k) % 4];
        for (int t = 0; t < 34; t++)
            for (int c = 0; c < st.tehai[t]; c++) assign(idx + c, t, 1.f);  // assign_rows(idx, tile, count, 1)
        idx += 4;
        for (int i = 0; i < 3; i++)
            if (st.akas_in_hand[i]) fill(idx + i, 1.f);
        idx += 3;
        const int n = st.shanten;
        MJO_ENSURE(n >= 0 && n <= 6, "oracle obs: shanten out of range");
        if (version == 1) {
            for (int i = 0; i < n; i++) fill(idx + i, 1.f);
            idx += 6;
        } else {
            fill(idx + n, 1.f);
            idx += 7;
            fill(idx, (float)n / 6.f);
            idx += 1;
        }
        for (int t = 0; t < 34; t++)
            if (st.waits[t]) assign(idx, t, 1.f);
        idx += 1;
        if (st.at_furiten) fill(idx, 1.f);
        idx += 1;
    }
    auto encode_tile = [&](int r, u8 tile) {
        assign(r, deaka(tile), 1.f);
        if (is_aka(tile)) fill(r + 1, 1.f);
    };
    {
        int taken = 0;
        for (auto it = board.yama.rbegin(); it != board.yama.rend() && taken < tiles_left; ++it, ++taken) {
            encode_tile(idx, *it);
            idx += 2;
        }
        MJO_ENSURE(taken == tiles_left, "oracle obs: yama shorter than tiles_left");
        idx += (69 - tiles_left) * 2;
    }
    for (auto it = board.rinshan.rbegin(); it != board.rinshan.rend(); ++it) {
        encode_tile(idx, *it);
        idx += 2;
    }
    idx += (4 - (int)board.rinshan.size()) * 2;
    for (auto it = dora_indicators_full.rbegin(); it != dora_indicators_full.rend(); ++it) {
        encode_tile(idx, *it);
        idx += 2;
    }
    for (u8 t : board.ura_indicators) {
        encode_tile(idx, t);
        idx += 2;
    }
    MJO_ENSURE(idx == rows, "oracle obs: row count mismatch");
}

}  // namespace mjo
