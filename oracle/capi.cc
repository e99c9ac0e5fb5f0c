// ORACLE — TEST INFRASTRUCTURE ONLY (see mjo.h).
// Flat C API for ctypes (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
#include <algorithm>
#include <memory>
#include <string>

#include "mjo.h"

using namespace mjo;

namespace {
thread_local std::string g_err;
std::vector<u8> g_table_storage;

// Packed event layout (int32[MJO_EV_INTS]); mirrored in tests/oracle_lib.py
enum { EV_INTS = 82 };
Event unpack_event(const int* p) {
    Event e;
    e.type = (u8)p[0];
    e.actor = (u8)p[1];
    e.target = (u8)p[2];
    e.pai = (u8)p[3];
    for (int i = 0; i < 4; i++) e.consumed[i] = (u8)p[4 + i];
    e.tsumogiri = p[8] != 0;
    e.bakaze = (u8)p[9];
    e.dora_marker = (u8)p[10];
    e.kyoku = (u8)p[11];
    e.honba = (u8)p[12];
    e.kyotaku = (u8)p[13];
    e.oya = (u8)p[14];
    for (int i = 0; i < 4; i++) e.scores[i] = p[15 + i];
    for (int i = 0; i < 52; i++) e.tehais[i / 13][i % 13] = (u8)p[19 + i];
    e.has_deltas = p[71] != 0;
    for (int i = 0; i < 4; i++) e.deltas[i] = p[72 + i];
    e.n_ura = p[76];
    for (int i = 0; i < 5; i++) e.ura_markers[i] = (u8)p[77 + i];
    return e;
}
void pack_event(const Event& e, int* p) {
    p[0] = e.type; p[1] = e.actor; p[2] = e.target; p[3] = e.pai;
    for (int i = 0; i < 4; i++) p[4 + i] = e.consumed[i];
    p[8] = e.tsumogiri; p[9] = e.bakaze; p[10] = e.dora_marker; p[11] = e.kyoku; p[12] = e.honba;
    p[13] = e.kyotaku; p[14] = e.oya;
    for (int i = 0; i < 4; i++) p[15 + i] = e.scores[i];
    for (int i = 0; i < 52; i++) p[19 + i] = e.tehais[i / 13][i % 13];
    p[71] = e.has_deltas;
    for (int i = 0; i < 4; i++) p[72 + i] = e.deltas[i];
    p[76] = e.n_ura;
    for (int i = 0; i < 5; i++) p[77 + i] = e.ura_markers[i];
}
void pack_cans(const ActionCandidate& c, int* o) {
    o[0] = c.can_discard; o[1] = c.can_chi_low; o[2] = c.can_chi_mid; o[3] = c.can_chi_high; o[4] = c.can_pon;
    o[5] = c.can_daiminkan; o[6] = c.can_kakan; o[7] = c.can_ankan; o[8] = c.can_riichi; o[9] = c.can_tsumo_agari;
    o[10] = c.can_ron_agari; o[11] = c.can_ryukyoku; o[12] = c.target_actor;
}

template <class F> int guard(F f) {
    try {
        return f();
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

// ---- arena: a batch of hanchan driven in lock-step (arena/game.rs:230-316)
struct Row {
    int game, seat, is_kan;
};
struct Arena {
    std::vector<std::unique_ptr<Game>> games;
    std::vector<int> live;  // indices of games still running, in swap_remove order (game.rs:298-300)
    std::vector<u8> done;   // per game
    std::vector<int> final_scores;  // 4 per game
    long guard_hits = 0;            // agari actions overridden by the rule-based guard
    std::vector<Row> rows;
    struct Pending {
        bool can_act = false, quick = false;
        Event quick_ev;
        int main_row = -1, kan_row = -1;
    };
    std::vector<std::array<Pending, 4>> pending;  // per game
    bool enable_quick_eval = true;
    int version = 4;
    long cycles = 0, steps = 0;
};
}  // namespace

namespace mjo { extern long g_sp_stats[16]; }

extern "C" {

void mjo_sp_stats(long* out, int reset) { for (int i = 0; i < 16; i++) { out[i] = mjo::g_sp_stats[i]; if (reset) mjo::g_sp_stats[i] = 0; } }
const char* mjo_last_error() { return g_err.c_str(); }
int mjo_ev_ints() { return EV_INTS; }

int mjo_set_tables(const u8* payload, size_t n) {
    return guard([&] {
        g_table_storage.assign(payload, payload + n);
        set_tables(g_table_storage.data(), g_table_storage.size());
        return 0;
    });
}

int mjo_calc_shanten(const u8* tehai, int len_div3, int which) {
    switch (which) {
        case 1: return calc_normal(tehai, len_div3);
        case 2: return calc_chitoi(tehai);
        case 3: return calc_kokushi(tehai);
        default: return calc_all(tehai, len_div3);
    }
}

// melds: chis,pons,minkans,ankans as 4 arrays of up to 4 with counts in n[4]
// mode 0: agari(additional_hans, doras); mode 1: search_yakus; mode 2: has_yaku
// out: [kind (0 none, 1 normal, 2 yakuman), fu, han_or_n]
int mjo_agari(const u8* tehai, const u8* melds, const int* n, int bakaze, int jikaze, int winning_tile, int is_ron,
              int mode, int additional_hans, int doras, int* out) {
    return guard([&] {
        AgariCalc c;
        c.tehai = tehai;
        c.chis = melds; c.n_chis = n[0];
        c.pons = melds + 4; c.n_pons = n[1];
        c.minkans = melds + 8; c.n_minkans = n[2];
        c.ankans = melds + 12; c.n_ankans = n[3];
        c.is_menzen = n[0] == 0 && n[1] == 0 && n[2] == 0;
        c.bakaze = (u8)bakaze;
        c.jikaze = (u8)jikaze;
        c.winning_tile = (u8)winning_tile;
        c.is_ron = is_ron != 0;
        std::optional<Agari> a;
        if (mode == 0) a = c.agari(additional_hans, doras);
        else if (mode == 1) a = c.search_yakus();
        else {
            out[0] = c.has_yaku();
            out[1] = out[2] = 0;
            return 0;
        }
        if (!a) { out[0] = out[1] = out[2] = 0; }
        else if (a->is_yakuman) { out[0] = 2; out[1] = 0; out[2] = a->n; }
        else { out[0] = 1; out[1] = a->fu; out[2] = a->han; }
        return 0;
    });
}
int mjo_check_ankan_after_riichi(const u8* tehai, int len_div3, int tile, int strict) {
    return guard([&] { return (int)check_ankan_after_riichi(tehai, len_div3, (u8)tile, strict != 0); });
}
int mjo_point(int is_oya, int fu, int han, int* out) {
    return guard([&] {
        Point p = point_calc(is_oya != 0, fu, han);
        out[0] = p.ron; out[1] = p.tsumo_ko; out[2] = p.tsumo_oya;
        return 0;
    });
}
int mjo_tile_next(int t) { return tile_next((u8)t); }
int mjo_tile_prev(int t) { return tile_prev((u8)t); }

void mjo_sha3_256(const u8* data, size_t len, u8* out) { sha3_256(data, len, out); }
void mjo_chacha12(const u8* seed, int n, uint32_t* out) {
    ChaCha12 r(seed);
    for (int i = 0; i < n; i++) out[i] = r.next_u32();
}
void mjo_deal(u64 nonce, u64 key, int kyoku, int honba, int algo, u8* seq) {
    deal_from_seed(nonce, key, (u8)kyoku, (u8)honba, (DealAlgo)algo, seq);
}

// ---- PlayerState handle
void* mjo_ps_new(int player_id) { return new PlayerState((u8)player_id); }
void mjo_ps_free(void* h) { delete (PlayerState*)h; }
void* mjo_ps_clone(void* h) { return new PlayerState(*(PlayerState*)h); }
int mjo_ps_update(void* h, const int* ev, int* cans_out) {
    return guard([&] {
        ActionCandidate c = ((PlayerState*)h)->update(unpack_event(ev));
        pack_cans(c, cans_out);
        return 0;
    });
}
int mjo_ps_validate_reaction(void* h, const int* ev) {
    return guard([&] {
        ((PlayerState*)h)->validate_reaction(unpack_event(ev));
        return 0;
    });
}
int mjo_ps_encode_obs(void* h, int version, int at_kan_select, float* obs, u8* mask) {
    return guard([&] {
        ((PlayerState*)h)->encode_obs(version, at_kan_select != 0, obs, mask);
        return 0;
    });
}
// test hooks: set tehai / call internals directly (state/test.rs:71-221 constructs states by hand)
int mjo_ps_set_tehai(void* h, const u8* tehai, int len_div3) {
    PlayerState* s = (PlayerState*)h;
    memcpy(s->tehai, tehai, 34);
    s->tehai_len_div3 = (u8)len_div3;
    return 0;
}
int mjo_ps_call(void* h, int what, int arg) {
    return guard([&] {
        PlayerState* s = (PlayerState*)h;
        switch (what) {
            case 0: s->update_shanten(); s->update_waits_and_furiten(); return 0;
            case 1: s->set_can_chi_from_tile((u8)arg); return 0;
            case 2: return (int)s->rule_based_agari();
            case 3: return s->real_time_shanten();
            case 4: return s->yaokyuu_kind_count();
            case 5: s->add_dora_indicator((u8)arg); return 0;
            default: return -2;
        }
    });
}
int mjo_ps_set_scores(void* h, const int* scores) {
    PlayerState* s = (PlayerState*)h;
    for (int i = 0; i < 4; i++) s->scores[i] = scores[i];
    return 0;
}
int mjo_ps_get_rank(void* h, const int* scores_rel) { return ((PlayerState*)h)->get_rank(scores_rel); }
int mjo_ps_agari_points(void* h, int is_ron, const u8* ura, int n_ura, int* out) {
    return guard([&] {
        Point p = ((PlayerState*)h)->agari_points(is_ron != 0, ura, n_ura);
        out[0] = p.ron; out[1] = p.tsumo_ko; out[2] = p.tsumo_oya;
        return 0;
    });
}
// Snapshot of the fields the reference's tests inspect.  out: int32[256]
//  0 shanten, 1 at_furiten, 2 has_next_shanten_discard, 3 tehai_len_div3, 4 is_menzen, 5 tiles_left, 6 at_turn,
//  7 rank, 8 doras_seen, 9 kans_on_board, 10..13 doras_owned, 14..47 waits, 48..81 tehai, 82..115 tiles_seen,
//  116..149 keep_shanten, 150..183 next_shanten, 184..217 forbidden, 218..229 cans(12), 230 target_actor,
//  231..233 akas_in_hand, 234 real_time_shanten, 235 can_w_riichi, 236 at_ippatsu, 237 at_rinshan, 238..241 scores
int mjo_ps_snapshot(void* h, int* o) {
    PlayerState* s = (PlayerState*)h;
    memset(o, 0, 256 * sizeof(int));
    o[0] = s->shanten; o[1] = s->at_furiten; o[2] = s->has_next_shanten_discard; o[3] = s->tehai_len_div3;
    o[4] = s->is_menzen; o[5] = s->tiles_left; o[6] = s->at_turn; o[7] = s->rank; o[8] = s->doras_seen;
    o[9] = s->kans_on_board;
    for (int i = 0; i < 4; i++) o[10 + i] = s->doras_owned[i];
    for (int i = 0; i < 34; i++) {
        o[14 + i] = s->waits[i];
        o[48 + i] = s->tehai[i];
        o[82 + i] = s->tiles_seen[i];
        o[116 + i] = s->keep_shanten_discards[i];
        o[150 + i] = s->next_shanten_discards[i];
        o[184 + i] = s->forbidden_tiles[i];
    }
    int c[13];
    pack_cans(s->last_cans, c);
    for (int i = 0; i < 13; i++) o[218 + i] = c[i];
    for (int i = 0; i < 3; i++) o[231 + i] = s->akas_in_hand[i];
    o[234] = s->real_time_shanten();
    o[235] = s->can_w_riichi; o[236] = s->at_ippatsu; o[237] = s->at_rinshan;
    for (int i = 0; i < 4; i++) o[238 + i] = s->scores[i];
    o[242] = (int)s->ankan_candidates.size();
    o[243] = (int)s->kakan_candidates.size();
    o[244] = s->last_self_tsumo ? (int)*s->last_self_tsumo : -1;
    return 0;
}
// kawa of relative seat `rel` in the pool's u64 entry format (mortal_amd/csrc/mj_state.h KW_*); returns length
int mjo_ps_kawa(void* h, int rel, unsigned long long* out, int max_n) {
    PlayerState* s = (PlayerState*)h;
    int n = 0;
    for (auto& it : s->kawa[rel]) {
        unsigned long long e = 0;
        if (it) {
            e = 1ull | ((unsigned long long)it->sutehai.tile << 1) | ((unsigned long long)it->sutehai.is_dora << 7) |
                ((unsigned long long)it->sutehai.is_tedashi << 8) | ((unsigned long long)it->sutehai.is_riichi << 9);
            if (it->chi_pon) {
                int a = deaka(it->chi_pon->consumed[0]), b = deaka(it->chi_pon->consumed[1]);
                e |= (1ull << 10) | ((unsigned long long)std::min(a, b) << 11) | ((unsigned long long)std::max(a, b) << 17);
            }
            e |= (unsigned long long)it->kan.size() << 23;
            for (size_t k = 0; k < it->kan.size() && k < 4; k++) e |= (unsigned long long)it->kan[k] << (26 + 6 * k);
        }
        if (n < max_n) out[n] = e;
        n++;
    }
    return n;
}
int mjo_ps_uncond_tenpai(void* h, u8* out34) {
    return guard([&] {
        bool b[34];
        ((PlayerState*)h)->discard_candidates_with_unconditional_tenpai(b);
        for (int i = 0; i < 34; i++) out34[i] = b[i];
        return 0;
    });
}
int mjo_ps_scene(void* h, int enable_quick_eval, int* out) {  // out: can_act, quick, quick_pai, quick_tsumogiri, need_kan
    return guard([&] {
        PlayerState* s = (PlayerState*)h;
        SceneInfo i = agent_scene(*s, s->player_id, enable_quick_eval != 0);
        out[0] = i.can_act; out[1] = i.quick_eval; out[2] = i.quick_event.pai; out[3] = i.quick_event.tsumogiri;
        out[4] = i.need_kan_select;
        return 0;
    });
}
int mjo_ps_decode_action(void* h, int action, int kan_tile, int* ev_out) {
    return guard([&] {
        PlayerState* s = (PlayerState*)h;
        pack_event(agent_decode_action(*s, s->player_id, action, kan_tile), ev_out);
        return 0;
    });
}
// SP tables: out_f: per candidate 3*17 floats; out_i: per candidate [tile, n_probs, shanten_down, num_required, n_req, (tile,count)*34]
int mjo_ps_sp_tables(void* h, float* out_f, int* out_i, int max_cands) {
    return guard([&] {
        auto t = ((PlayerState*)h)->single_player_tables();
        int n = std::min<int>((int)t.size(), max_cands);
        for (int k = 0; k < n; k++) {
            auto& c = t[k];
            float* f = out_f + k * 51;
            for (int i = 0; i < 51; i++) f[i] = 0.f;
            for (size_t i = 0; i < c.tenpai_probs.size(); i++) {
                f[i] = c.tenpai_probs[i];
                f[17 + i] = c.win_probs[i];
                f[34 + i] = c.exp_values[i];
            }
            int* o = out_i + k * 73;
            o[0] = c.tile; o[1] = (int)c.tenpai_probs.size(); o[2] = c.shanten_down; o[3] = c.num_required_tiles;
            o[4] = (int)c.required_tiles.size();
            for (size_t i = 0; i < c.required_tiles.size() && i < 34; i++) {
                o[5 + 2 * i] = c.required_tiles[i].tile;
                o[6 + 2 * i] = c.required_tiles[i].count;
            }
        }
        return (int)t.size();
    });
}
// Direct SP calculator access for the calc.rs KATs.  cfg: [len_div3, is_menzen, bakaze, jikaze, num_doras_in_fuuro,
//  calc_double_riichi, calc_haitei, prefer_riichi, sort_result, maximize_win_prob, calc_tegawari, calc_shanten_down,
//  can_discard, tsumos_left, cur_shanten, n_dora_indicators, dora_ind[5], n_chis, n_pons, n_minkans, n_ankans, melds[16]]
int mjo_sp_calc(const int* cfg, const u8* tehai, const u8* akas_in_hand, const u8* tiles_seen, const u8* akas_seen,
                float* out_f, int* out_i, int max_cands) {
    return guard([&] {
        SPCalculator sp;
        sp.tehai_len_div3 = (u8)cfg[0];
        sp.is_menzen = cfg[1];
        sp.bakaze = (u8)cfg[2];
        sp.jikaze = (u8)cfg[3];
        sp.num_doras_in_fuuro = (u8)cfg[4];
        sp.calc_double_riichi = cfg[5]; sp.calc_haitei = cfg[6]; sp.prefer_riichi = cfg[7]; sp.sort_result = cfg[8];
        sp.maximize_win_prob = cfg[9]; sp.calc_tegawari = cfg[10]; sp.calc_shanten_down = cfg[11];
        u8 dora[5], melds[16];
        for (int i = 0; i < 5; i++) dora[i] = (u8)cfg[16 + i];
        for (int i = 0; i < 16; i++) melds[i] = (u8)cfg[25 + i];
        sp.dora_indicators = dora; sp.n_dora_indicators = cfg[15];
        sp.chis = melds; sp.n_chis = cfg[21];
        sp.pons = melds + 4; sp.n_pons = cfg[22];
        sp.minkans = melds + 8; sp.n_minkans = cfg[23];
        sp.ankans = melds + 12; sp.n_ankans = cfg[24];
        SPInitState init;
        memcpy(init.tehai, tehai, 34);
        memcpy(init.tiles_seen, tiles_seen, 34);
        for (int i = 0; i < 3; i++) { init.akas_in_hand[i] = akas_in_hand[i]; init.akas_seen[i] = akas_seen[i]; }
        auto t = sp.calc(init, cfg[12] != 0, cfg[13], cfg[14]);
        int n = std::min<int>((int)t.size(), max_cands);
        for (int k = 0; k < n; k++) {
            auto& c = t[k];
            float* f = out_f + k * 51;
            for (int i = 0; i < 51; i++) f[i] = 0.f;
            for (size_t i = 0; i < c.tenpai_probs.size(); i++) {
                f[i] = c.tenpai_probs[i]; f[17 + i] = c.win_probs[i]; f[34 + i] = c.exp_values[i];
            }
            int* o = out_i + k * 73;
            o[0] = c.tile; o[1] = (int)c.tenpai_probs.size(); o[2] = c.shanten_down; o[3] = c.num_required_tiles;
            o[4] = (int)c.required_tiles.size();
            for (size_t i = 0; i < c.required_tiles.size() && i < 34; i++) {
                o[5 + 2 * i] = c.required_tiles[i].tile; o[6 + 2 * i] = c.required_tiles[i].count;
            }
        }
        return (int)t.size();
    });
}

// ---- arena
void* mjo_arena_new(int n_games, const u64* nonces, const u64* keys, int deal_algo, int enable_quick_eval, int version,
                    int keep_log) {
    Arena* a = new Arena;
    a->enable_quick_eval = enable_quick_eval != 0;
    a->version = version;
    for (int g = 0; g < n_games; g++) {
        auto gm = std::make_unique<Game>();
        gm->seed_nonce = nonces[g];
        gm->seed_key = keys[g];
        gm->deal_algo = (DealAlgo)deal_algo;
        gm->keep_log = keep_log != 0;
        a->games.push_back(std::move(gm));
        a->live.push_back(g);
    }
    a->done.assign(n_games, 0);
    a->final_scores.assign(n_games * 4, 0);
    a->pending.resize(n_games);
    return a;
}
void mjo_arena_free(void* h) { delete (Arena*)h; }
// Steady-state mode of the throughput benchmark (no reference counterpart: BatchGame::run never restarts a game): slot `game`,
// whose hanchan is finished, starts a fresh hanchan on (nonce, same key) — what the device's mj_k_refill does.
int mjo_arena_restart(void* h, int game, u64 nonce) {
    return guard([&] {
        Arena* a = (Arena*)h;
        if (!a->done.at(game)) throw std::runtime_error("restart of a game that has not finished");
        auto gm = std::make_unique<Game>();
        gm->seed_nonce = nonce;
        gm->seed_key = a->games[game]->seed_key;
        gm->deal_algo = a->games[game]->deal_algo;
        gm->keep_log = a->games[game]->keep_log;
        a->games[game] = std::move(gm);
        a->done[game] = 0;
        for (auto& p : a->pending[game]) p = Arena::Pending();
        a->live.push_back(game);
        return 0;
    });
}

// Staggered first start of the benchmark's steady-state mode (the device's mj_pool_set_start_stagger / mj_k_park): every slot is
// marked finished before the first cycle without having played; the harness then restarts slot t at its own cycle.
int mjo_arena_park(void* h) {
    return guard([&] {
        Arena* a = (Arena*)h;
        if (a->cycles != 0) throw std::runtime_error("park after the first cycle");
        for (size_t g = 0; g < a->done.size(); g++) a->done[g] = 1;
        a->live.clear();
        return 0;
    });
}

// Poll phase (game.rs:287-289).  Returns the number of policy rows, -1 on error.
int mjo_arena_poll(void* h) {
    return guard([&] {
        Arena* a = (Arena*)h;
        a->rows.clear();
        // Row order: ascending game index, then seat, kan-select row before the main row.
        std::vector<int> order = a->live;
        std::sort(order.begin(), order.end());
        for (int g : order) {
            Game& gm = *a->games[g];
            gm.poll();
            auto& pend = a->pending[g];
            for (auto& p : pend) p = Arena::Pending();
            if (gm.ended) continue;
            for (int seat = 0; seat < 4; seat++) {
                const PlayerState& st = gm.board->player_states[seat];
                SceneInfo si = agent_scene(st, (u8)seat, a->enable_quick_eval);
                if (!si.can_act) continue;
                auto& p = pend[seat];
                p.can_act = true;
                if (si.quick_eval) {
                    p.quick = true;
                    p.quick_ev = si.quick_event;
                    continue;
                }
                if (si.need_kan_select) {
                    p.kan_row = (int)a->rows.size();
                    a->rows.push_back({g, seat, 1});
                }
                p.main_row = (int)a->rows.size();
                a->rows.push_back({g, seat, 0});
            }
        }
        return (int)a->rows.size();
    });
}
int mjo_arena_n_live(void* h) { return (int)((Arena*)h)->live.size(); }
long mjo_arena_steps(void* h) { return ((Arena*)h)->steps; }
long mjo_arena_guard_hits(void* h) { return ((Arena*)h)->guard_hits; }
long mjo_arena_cycles(void* h) { return ((Arena*)h)->cycles; }
// rows_out: int32[n_rows*3] (game, seat, is_kan)
int mjo_arena_rows(void* h, int* rows_out) {
    Arena* a = (Arena*)h;
    for (size_t i = 0; i < a->rows.size(); i++) {
        rows_out[i * 3] = a->rows[i].game;
        rows_out[i * 3 + 1] = a->rows[i].seat;
        rows_out[i * 3 + 2] = a->rows[i].is_kan;
    }
    return (int)a->rows.size();
}
// Encode rows [row0, row1).  obs may be NULL (masks only).
int mjo_arena_encode(void* h, int row0, int row1, float* obs, u8* masks) {
    return guard([&] {
        Arena* a = (Arena*)h;
        // masks are version-independent; without an obs buffer use v3 (no SP tables) to get them cheaply
        int version = obs ? a->version : 3;
        size_t stride = (size_t)obs_rows(version) * 34;
        std::vector<float> tmp;
        if (!obs) tmp.resize(stride);
        for (int r = row0; r < row1; r++) {
            const Row& row = a->rows.at(r);
            const PlayerState& st = a->games[row.game]->board->player_states[row.seat];
            float* o = obs ? obs + (size_t)(r - row0) * stride : tmp.data();
            st.encode_obs(version, row.is_kan != 0, o, masks + (size_t)(r - row0) * 46);
        }
        return 0;
    });
}
// Invisible ("oracle") obs of rows [row0,row1) (game.rs:100-101): out = f32 [n][oracle_obs_rows(version)][34]
int mjo_arena_encode_oracle(void* h, int row0, int row1, int version, float* out) {
    return guard([&] {
        Arena* a = (Arena*)h;
        size_t stride = (size_t)BoardState::oracle_obs_rows(version) * 34;
        for (int r = row0; r < row1; r++) {
            const Row& row = a->rows.at(r);
            a->games[row.game]->board->encode_oracle_obs((u8)row.seat, version, out + (size_t)(r - row0) * stride);
        }
        return 0;
    });
}
// Commit phase (game.rs:291-304): actions[n_rows].  Returns number of games finished in this cycle.
// q != NULL enables the rule-based agari guard (agent/mortal.rs:319-336) with q = f32 [rows][46] of the batch
int mjo_arena_commit_q(void* h, const int* actions, const float* q);
int mjo_arena_commit(void* h, const int* actions) { return mjo_arena_commit_q(h, actions, nullptr); }
int mjo_arena_commit_q(void* h, const int* actions, const float* q) {
    return guard([&] {
        Arena* a = (Arena*)h;
        int finished = 0;
        std::vector<int> still;
        for (int g : a->live) {
            Game& gm = *a->games[g];
            if (gm.ended) {
                gm.commit_end();
                for (int i = 0; i < 4; i++) a->final_scores[g * 4 + i] = gm.scores[i];
                a->done[g] = 1;
                finished++;
                continue;
            }
            auto& pend = a->pending[g];
            for (int seat = 0; seat < 4; seat++) {
                auto& p = pend[seat];
                if (!p.can_act) continue;
                if (p.quick) {
                    gm.last_reactions[seat] = p.quick_ev;
                    continue;
                }
                const PlayerState& st = gm.board->player_states[seat];
                int action = actions[p.main_row];
                if (q && action == 43 && !st.rule_based_agari()) {
                    // q_values[43] = f32::MIN; iter().enumerate().max_by(total_cmp) -> last of the equal maxima
                    int best = 0;
                    int32_t best_key = INT32_MIN;
                    for (int k = 0; k < 46; k++) {
                        float v = k == 43 ? -3.40282347e+38f : q[(size_t)p.main_row * 46 + k];
                        int32_t bits;
                        memcpy(&bits, &v, 4);
                        bits ^= (int32_t)((uint32_t)(bits >> 31) >> 1);
                        if (bits >= best_key) {
                            best_key = bits;
                            best = k;
                        }
                    }
                    action = best;
                    a->guard_hits += 1;
                }
                int kan_tile = p.kan_row >= 0 ? actions[p.kan_row] : -1;
                gm.last_reactions[seat] = agent_decode_action(st, (u8)seat, action, kan_tile);
            }
            still.push_back(g);
        }
        a->live.swap(still);
        a->cycles += 1;
        a->steps += (long)a->live.size();
        return finished;
    });
}
int mjo_arena_result(void* h, int game, int* scores4, int* done) {
    Arena* a = (Arena*)h;
    for (int i = 0; i < 4; i++) scores4[i] = a->final_scores[game * 4 + i];
    *done = a->done[game];
    return 0;
}
// The done flag of every slot at once (the lock-step harness of a big pool scans for finished slots every cycle).
int mjo_arena_done_flags(void* h, unsigned char* out) {
    Arena* a = (Arena*)h;
    for (size_t g = 0; g < a->done.size(); g++) out[g] = (unsigned char)(a->done[g] != 0);
    return (int)a->done.size();
}
// Live view of one game for lock-step comparison: out int32[16] =
//  [ended, kyoku, honba, kyotaku, scores[4], kyoku_started, tiles_left, oya, in_renchan, 0...]
int mjo_arena_game_view(void* h, int game, int* out) {
    Arena* a = (Arena*)h;
    Game& gm = *a->games[game];
    memset(out, 0, 16 * sizeof(int));
    out[0] = gm.ended; out[1] = gm.kyoku; out[2] = gm.honba; out[3] = gm.kyotaku;
    for (int i = 0; i < 4; i++) out[4 + i] = gm.scores[i];
    out[8] = gm.kyoku_started;
    if (gm.board) { out[9] = gm.board->tiles_left; out[10] = gm.board->oya; }
    out[11] = gm.in_renchan;
    return 0;
}
void* mjo_arena_player_state(void* h, int game, int seat) {  // borrowed pointer
    Arena* a = (Arena*)h;
    Game& gm = *a->games[game];
    if (!gm.board) return nullptr;
    return &gm.board->player_states[seat];
}
// Event log of a game: returns number of events; out may be NULL to query the count. Kyoku logs are concatenated;
// the log of the kyoku in progress is included last.
int mjo_arena_log(void* h, int game, int* out, int max_events) {
    Arena* a = (Arena*)h;
    Game& gm = *a->games[game];
    int n = 0;
    auto emit = [&](const std::vector<Event>& v) {
        for (auto& e : v) {
            if (out && n < max_events) pack_event(e, out + (size_t)n * EV_INTS);
            n++;
        }
    };
    for (auto& k : gm.game_log) emit(k);
    if (gm.board && gm.kyoku_started) emit(gm.board->log);
    return n;
}

}  // extern "C"
