// ORACLE — TEST INFRASTRUCTURE ONLY (see mjo.h).
// Single-player EV / win-prob / tenpai-prob tables: algo/sp/{calc,state,candidate,tile}.rs
// Compiled with -ffp-contract=off: Rust never fuses a*b+c, and the f32 sums are order-sensitive.
#include <algorithm>
#include <memory>
#include <unordered_map>

#include "mjo.h"

namespace mjo {

long g_sp_stats[16] = {};  // debug: draw-cache inserts per shanten level [0..3], discard-cache inserts [4..7], draw calls [8..11], discard calls [12..15]

namespace {

constexpr int SHANTEN_THRES = 3;                 // calc.rs:13
constexpr int MAX_TILES_LEFT = 34 * 4 - 1 - 13;  // calc.rs:14

// algo/data/uradora_prob_table.txt (values restated; calc.rs:17)
const float URADORA_PROB_TABLE[5][13] = {
    {0.639485f, 0.327801f, 0.0327134f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f},
    {0.406736f, 0.42281f, 0.147966f, 0.021674f, 0.0008142f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f},
    {0.257516f, 0.406819f, 0.246851f, 0.0757724f, 0.0122266f, 0.0008004f, 1.43e-5f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f},
    {0.162199f, 0.346513f, 0.301539f, 0.142396f, 0.0401276f, 0.0066491f, 0.0005575f, 1.85e-5f, 0.f, 0.f, 0.f, 0.f, 0.f},
    {0.101768f, 0.275319f, 0.313742f, 0.20189f, 0.081774f, 0.0215394f, 0.0035918f, 0.0003607f, 1.52e-5f, 3e-7f, 0.f, 0.f, 0.f}};

struct State {  // sp/state.rs:9-20
    u8 tehai[34];
    u8 akas_in_hand[3];
    u8 tiles_in_wall[34];
    u8 akas_in_wall[3];
    u8 n_extra_tsumo;
    bool operator==(const State& o) const { return memcmp(this, &o, sizeof(State)) == 0; }
};
struct StateHash {
    size_t operator()(const State& s) const {
        const u8* p = reinterpret_cast<const u8*>(&s);
        u64 h = 1469598103934665603ull;
        for (size_t i = 0; i < sizeof(State); i++) h = (h ^ p[i]) * 1099511628211ull;
        return (size_t)h;
    }
};

struct DiscardTile { u8 tile; i8 shanten_diff; };
struct DrawTile { u8 tile; u8 count; i8 shanten_diff; };

void st_discard(State& s, u8 tile) {  // state.rs:57-65
    s.tehai[deaka(tile)] -= 1;
    if (is_aka(tile)) s.akas_in_hand[tile - T_5MR] = 0;
}
void st_undo_discard(State& s, u8 tile) {  // state.rs:67-75
    s.tehai[deaka(tile)] += 1;
    if (is_aka(tile)) s.akas_in_hand[tile - T_5MR] = 1;
}
void st_deal(State& s, u8 tile) {  // state.rs:77-86
    s.tiles_in_wall[deaka(tile)] -= 1;
    if (is_aka(tile)) s.akas_in_wall[tile - T_5MR] = 0;
    st_undo_discard(s, tile);
}
void st_undo_deal(State& s, u8 tile) {  // state.rs:88-97
    st_discard(s, tile);
    s.tiles_in_wall[deaka(tile)] += 1;
    if (is_aka(tile)) s.akas_in_wall[tile - T_5MR] = 1;
}
std::vector<DiscardTile> get_discard_tiles(const State& s, int shanten, int len_div3) {  // state.rs:99-130
    std::vector<DiscardTile> out;
    u8 tehai[34];
    memcpy(tehai, s.tehai, 34);
    for (int tid = 0; tid < 34; tid++) {
        if (tehai[tid] == 0) continue;
        tehai[tid] -= 1;
        int after = calc_all(tehai, len_div3);
        tehai[tid] += 1;
        u8 tile = (u8)tid;
        if (tid == T_5M && s.akas_in_hand[0] && tehai[tid] == 1) tile = T_5MR;
        else if (tid == T_5P && s.akas_in_hand[1] && tehai[tid] == 1) tile = T_5PR;
        else if (tid == T_5S && s.akas_in_hand[2] && tehai[tid] == 1) tile = T_5SR;
        out.push_back({tile, (i8)(after - shanten)});
    }
    return out;
}
std::vector<DrawTile> get_draw_tiles(const State& s, int shanten, int len_div3) {  // state.rs:132-174
    std::vector<DrawTile> out;
    u8 tehai[34];
    memcpy(tehai, s.tehai, 34);
    for (int tid = 0; tid < 34; tid++) {
        u8 count = s.tiles_in_wall[tid];
        if (count == 0) continue;
        tehai[tid] += 1;
        int after = calc_all(tehai, len_div3);
        tehai[tid] -= 1;
        i8 diff = (i8)(after - shanten);
        bool aka_in_wall = (tid == T_5M && s.akas_in_wall[0]) || (tid == T_5P && s.akas_in_wall[1]) ||
                           (tid == T_5S && s.akas_in_wall[2]);
        if (aka_in_wall) {
            if (count >= 2) out.push_back({(u8)tid, (u8)(count - 1), diff});
            out.push_back({akaize((u8)tid), 1, diff});
        } else {
            out.push_back({(u8)tid, count, diff});
        }
    }
    return out;
}
std::vector<RequiredTile> get_required_tiles(const State& s, int len_div3) {  // state.rs:176-200
    u8 tehai[34];
    memcpy(tehai, s.tehai, 34);
    int shanten = calc_all(tehai, len_div3);
    std::vector<RequiredTile> out;
    for (int tid = 0; tid < 34; tid++) {
        u8 count = s.tiles_in_wall[tid];
        if (count == 0) continue;
        tehai[tid] += 1;
        int after = calc_all(tehai, len_div3);
        tehai[tid] -= 1;
        if (after < shanten) out.push_back({(u8)tid, count});
    }
    return out;
}
int sum_left_tiles(const State& s) {  // state.rs:202-204 (u8 sum, wraps)
    u8 n = 0;
    for (int i = 0; i < 34; i++) n += s.tiles_in_wall[i];
    return n;
}

struct Values {  // calc.rs:22-26
    float tenpai_probs[MAX_TSUMOS_LEFT];
    float win_probs[MAX_TSUMOS_LEFT];
    float exp_values[MAX_TSUMOS_LEFT];
};
typedef std::shared_ptr<Values> ValuesP;

SPCandidate make_candidate(u8 tile, const float* tp, const float* wp, const float* ev, int n,
                           std::vector<RequiredTile> req, bool shanten_down) {  // candidate.rs:46-70
    SPCandidate c;
    c.tile = tile;
    u8 sum = 0;
    for (auto& r : req) sum += r.count;
    c.num_required_tiles = sum;
    for (int i = 0; i < n; i++) {
        // f32::clamp(0,1): NaN stays NaN; max(0): NaN -> 0.  No NaNs arise here.
        c.tenpai_probs.push_back(std::min(std::max(tp[i], 0.f), 1.f));
        c.win_probs.push_back(std::min(std::max(wp[i], 0.f), 1.f));
        c.exp_values.push_back(std::max(ev[i], 0.f));
    }
    c.required_tiles = std::move(req);
    c.shanten_down = shanten_down;
    return c;
}

struct Calc {  // calc.rs:64-78
    const SPCalculator& sup;
    State state;
    int T;  // MAX_TSUMO
    float tsumo_prob_table[4][MAX_TSUMOS_LEFT];
    std::vector<std::array<float, MAX_TSUMOS_LEFT>> not_tsumo_prob_table;
    std::unordered_map<State, ValuesP, StateHash> discard_cache[SHANTEN_THRES + 1], draw_cache[SHANTEN_THRES + 1];

    Calc(const SPCalculator& s, const State& st, int max_tsumo) : sup(s), state(st), T(max_tsumo) {
        int n_left = sum_left_tiles(state);
        // build_tsumo_prob_table (calc.rs:135-146)
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < T; j++) tsumo_prob_table[i][j] = (float)(i + 1) / (float)(n_left - j);
        // build_not_tsumo_prob_table (calc.rs:148-167)
        not_tsumo_prob_table.assign(MAX_TILES_LEFT + 1, {});
        for (int i = 0; i <= MAX_TILES_LEFT && i < n_left + 1; i++) {
            auto& row = not_tsumo_prob_table[i];
            row[0] = 1.f;
            int lim = std::min(T - 1, n_left - i);
            for (int j = 0; j < lim; j++) row[j + 1] = row[j] * (float)(n_left - i - j) / (float)(n_left - j);
        }
    }

    std::vector<SPCandidate> calc(bool can_discard, int cur_shanten) {  // calc.rs:170-203
        std::vector<SPCandidate> cands;
        if (cur_shanten <= SHANTEN_THRES) {
            cands = can_discard ? analyze_discard(cur_shanten) : analyze_draw(cur_shanten);
            if (sup.sort_result && !cands.empty()) {
                SPColumn by = sup.maximize_win_prob ? COL_WIN_PROB : COL_EV;
                // slice::sort_by is stable
                std::stable_sort(cands.begin(), cands.end(),
                                 [&](const SPCandidate& l, const SPCandidate& r) { return sp_candidate_cmp(r, l, by) < 0; });
            }
        } else {
            cands = can_discard ? analyze_discard_simple(cur_shanten) : analyze_draw_simple();
            if (sup.sort_result && !cands.empty()) {
                std::stable_sort(cands.begin(), cands.end(), [&](const SPCandidate& l, const SPCandidate& r) {
                    return sp_candidate_cmp(r, l, COL_NOT_SHANTEN_DOWN) < 0;
                });
            }
        }
        return cands;
    }

    std::vector<SPCandidate> analyze_discard(int shanten) {  // calc.rs:205-256
        auto discard_tiles = get_discard_tiles(state, shanten, sup.tehai_len_div3);
        std::vector<SPCandidate> cands;
        for (auto& d : discard_tiles) {
            if (d.shanten_diff == 0) {
                st_discard(state, d.tile);
                auto req = get_required_tiles(state, sup.tehai_len_div3);
                ValuesP v = draw(shanten);
                st_undo_discard(state, d.tile);
                float tp[MAX_TSUMOS_LEFT];
                memcpy(tp, v->tenpai_probs, sizeof tp);
                if (shanten == 0)
                    for (int i = 0; i < T; i++) tp[i] = 1.f;
                cands.push_back(make_candidate(d.tile, tp, v->win_probs, v->exp_values, T, std::move(req), false));
            } else if (sup.calc_shanten_down && d.shanten_diff == 1 && shanten < SHANTEN_THRES) {
                st_discard(state, d.tile);
                auto req = get_required_tiles(state, sup.tehai_len_div3);
                state.n_extra_tsumo += 1;
                ValuesP v = draw(shanten + 1);
                state.n_extra_tsumo -= 1;
                st_undo_discard(state, d.tile);
                cands.push_back(
                    make_candidate(d.tile, v->tenpai_probs, v->win_probs, v->exp_values, T, std::move(req), true));
            }
        }
        return cands;
    }
    std::vector<SPCandidate> analyze_draw(int shanten) {  // calc.rs:258-279
        auto req = get_required_tiles(state, sup.tehai_len_div3);
        ValuesP v = draw(shanten);
        float tp[MAX_TSUMOS_LEFT];
        memcpy(tp, v->tenpai_probs, sizeof tp);
        if (shanten == 0)
            for (int i = 0; i < T; i++) tp[i] = 1.f;
        std::vector<SPCandidate> out;
        out.push_back(make_candidate(T_UNK, tp, v->win_probs, v->exp_values, T, std::move(req), false));
        return out;
    }
    std::vector<SPCandidate> analyze_discard_simple(int shanten) {  // calc.rs:281-301
        auto discard_tiles = get_discard_tiles(state, shanten, sup.tehai_len_div3);
        std::vector<SPCandidate> out;
        for (auto& d : discard_tiles) {
            st_discard(state, d.tile);
            auto req = get_required_tiles(state, sup.tehai_len_div3);
            st_undo_discard(state, d.tile);
            out.push_back(make_candidate(d.tile, nullptr, nullptr, nullptr, 0, std::move(req), d.shanten_diff == 1));
        }
        return out;
    }
    std::vector<SPCandidate> analyze_draw_simple() {  // calc.rs:303-312
        auto req = get_required_tiles(state, sup.tehai_len_div3);
        std::vector<SPCandidate> out;
        out.push_back(make_candidate(T_UNK, nullptr, nullptr, nullptr, 0, std::move(req), false));
        return out;
    }

    ValuesP draw(int shanten) {  // calc.rs:314-320
        g_sp_stats[8 + shanten]++;
        if (sup.calc_tegawari && state.n_extra_tsumo == 0) return draw_with_tegawari(shanten);
        return draw_without_tegawari(shanten);
    }

    void accumulate_score(const float scores[4], int i, int j, bool ippatsu_any, float prob, float* win_probs,
                          float* exp_values) {
        bool assume_riichi = sup.is_menzen && sup.prefer_riichi;
        bool win_double_riichi = assume_riichi && sup.calc_double_riichi && i == 0;
        bool win_ippatsu = ippatsu_any ? assume_riichi : (assume_riichi && j == i);
        bool win_haitei = sup.calc_haitei && j == T - 1;
        int han_plus = (int)win_double_riichi + (int)win_ippatsu + (int)win_haitei;
        win_probs[i] += prob;
        exp_values[i] += prob * scores[han_plus];
    }

    ValuesP draw_with_tegawari(int shanten) {  // calc.rs:322-439
        auto it = draw_cache[shanten].find(state);
        if (it != draw_cache[shanten].end()) return it->second;
        auto vals = std::make_shared<Values>();
        memset(vals.get(), 0, sizeof(Values));
        float* tenpai_probs = vals->tenpai_probs;
        float* win_probs = vals->win_probs;
        float* exp_values = vals->exp_values;
        auto draw_tiles = get_draw_tiles(state, shanten, sup.tehai_len_div3);
        int sum_left = sum_left_tiles(state);
        for (auto& d : draw_tiles) {
            if (d.shanten_diff != -1) continue;
            st_deal(state, d.tile);
            ValuesP next;
            float scores[4];
            bool is_scores = false;
            if (shanten > 0) {
                next = discard(shanten - 1);
            } else if (get_score(d.tile, scores)) {
                is_scores = true;
            } else {
                st_undo_deal(state, d.tile);
                continue;
            }
            st_undo_deal(state, d.tile);
            for (int i = 0; i < T; i++) {
                float tump_prob = (float)d.count / (float)sum_left;
                if (is_scores) {
                    // calc.rs:372-386: haitei uses i == MAX_TSUMO-1, ippatsu always
                    accumulate_score(scores, i, i, true, tump_prob, win_probs, exp_values);
                } else {
                    if (shanten == 1) tenpai_probs[i] += tump_prob;
                    if (i < T - 1) {
                        if (shanten > 1) tenpai_probs[i] += tump_prob * next->tenpai_probs[i + 1];
                        win_probs[i] += tump_prob * next->win_probs[i + 1];
                        exp_values[i] += tump_prob * next->exp_values[i + 1];
                    }
                }
            }
        }
        for (auto& d : draw_tiles) {
            if (d.shanten_diff != 0) continue;
            st_deal(state, d.tile);
            state.n_extra_tsumo += 1;
            ValuesP next = discard(shanten);
            state.n_extra_tsumo -= 1;
            st_undo_deal(state, d.tile);
            for (int i = 0; i < T - 1; i++) {
                float tump_prob = (float)d.count / (float)sum_left;
                tenpai_probs[i] += tump_prob * next->tenpai_probs[i + 1];
                win_probs[i] += tump_prob * next->win_probs[i + 1];
                exp_values[i] += tump_prob * next->exp_values[i + 1];
            }
        }
        draw_cache[shanten][state] = vals;
        return vals;
    }

    ValuesP draw_without_tegawari(int shanten) {  // calc.rs:447-561
        auto it = draw_cache[shanten].find(state);
        if (it != draw_cache[shanten].end()) return it->second;
        auto vals = std::make_shared<Values>();
        memset(vals.get(), 0, sizeof(Values));
        float* tenpai_probs = vals->tenpai_probs;
        float* win_probs = vals->win_probs;
        float* exp_values = vals->exp_values;
        auto draw_tiles = get_draw_tiles(state, shanten, sup.tehai_len_div3);
        u8 sum_required = 0;
        for (auto& d : draw_tiles)
            if (d.shanten_diff == -1) sum_required += d.count;
        const auto& not_tsumo_probs = not_tsumo_prob_table.at(sum_required);

        for (auto& d : draw_tiles) {
            if (d.shanten_diff != -1) continue;
            st_deal(state, d.tile);
            ValuesP next;
            float scores[4];
            bool is_scores = false;
            if (shanten > 0) {
                next = discard(shanten - 1);
            } else if (get_score(d.tile, scores)) {
                is_scores = true;
            } else {
                st_undo_deal(state, d.tile);
                continue;
            }
            st_undo_deal(state, d.tile);

            const float* tsumo_probs = tsumo_prob_table[d.count - 1];
            for (int i = 0; i < T; i++) {
                float m = not_tsumo_probs[i];
                if (m == 0.f) break;
                for (int j = i; j < T; j++) {
                    float n = not_tsumo_probs[j];
                    if (n == 0.f) break;
                    float prob = tsumo_probs[j] * n / m;
                    if (is_scores) {
                        accumulate_score(scores, i, j, false, prob, win_probs, exp_values);
                    } else {
                        if (shanten == 1) tenpai_probs[i] += prob;
                        if (j < T - 1) {
                            if (shanten > 1) tenpai_probs[i] += prob * next->tenpai_probs[j + 1];
                            win_probs[i] += prob * next->win_probs[j + 1];
                            exp_values[i] += prob * next->exp_values[j + 1];
                        }
                    }
                }
            }
        }
        draw_cache[shanten][state] = vals;
        g_sp_stats[shanten]++;
        return vals;
    }

    ValuesP discard(int shanten) {  // calc.rs:563-637
        g_sp_stats[12 + shanten]++;
        auto it = discard_cache[shanten].find(state);
        if (it != discard_cache[shanten].end()) return it->second;
        auto discard_tiles = get_discard_tiles(state, shanten, sup.tehai_len_div3);
        auto vals = std::make_shared<Values>();
        float* max_tenpai = vals->tenpai_probs;
        float* max_win = vals->win_probs;
        float* max_ev = vals->exp_values;
        u8 max_tiles[MAX_TSUMOS_LEFT];
        int32_t max_values[MAX_TSUMOS_LEFT];
        for (int i = 0; i < MAX_TSUMOS_LEFT; i++) {
            max_tenpai[i] = max_win[i] = max_ev[i] = -3.40282347e+38f;  // f32::MIN
            max_tiles[i] = T_UNK;
            max_values[i] = INT32_MIN;
        }
        for (auto& d : discard_tiles) {
            ValuesP v;
            if (d.shanten_diff == 0) {
                st_discard(state, d.tile);
                v = draw(shanten);
                st_undo_discard(state, d.tile);
            } else if (sup.calc_shanten_down && state.n_extra_tsumo == 0 && d.shanten_diff == 1 &&
                       shanten < SHANTEN_THRES) {
                st_discard(state, d.tile);
                state.n_extra_tsumo += 1;
                v = draw(shanten + 1);
                state.n_extra_tsumo -= 1;
                st_undo_discard(state, d.tile);
            } else {
                continue;
            }
            for (int i = 0; i < T; i++) {
                float fv = sup.maximize_win_prob ? v->win_probs[i] * 1e5f : v->exp_values[i];
                // Rust `as i32` saturates; values here are far inside the range.
                int32_t value = (int32_t)fv;
                if (value > max_values[i] ||
                    (value == max_values[i] && cmp_discard_priority(d.tile, max_tiles[i]) > 0)) {
                    max_tenpai[i] = v->tenpai_probs[i];
                    max_win[i] = v->win_probs[i];
                    max_ev[i] = v->exp_values[i];
                    max_values[i] = value;
                    max_tiles[i] = d.tile;
                }
            }
        }
        discard_cache[shanten][state] = vals;
        g_sp_stats[4 + shanten]++;
        return vals;
    }

    bool get_score(u8 win_tile, float scores[4]) {  // calc.rs:640-758
        AgariCalc calc;
        calc.tehai = state.tehai;
        calc.is_menzen = sup.is_menzen;
        calc.chis = sup.chis; calc.n_chis = sup.n_chis;
        calc.pons = sup.pons; calc.n_pons = sup.n_pons;
        calc.minkans = sup.minkans; calc.n_minkans = sup.n_minkans;
        calc.ankans = sup.ankans; calc.n_ankans = sup.n_ankans;
        calc.bakaze = sup.bakaze;
        calc.jikaze = sup.jikaze;
        calc.winning_tile = deaka(win_tile);
        calc.is_ron = false;
        bool is_oya = sup.jikaze == T_E;
        int additional_yakus = sup.is_menzen ? (sup.prefer_riichi ? 2 : 1) : 0;
        u8 num_doras = 0;
        for (int i = 0; i < sup.n_dora_indicators; i++) num_doras += state.tehai[tile_next(sup.dora_indicators[i])];
        num_doras += (u8)(state.akas_in_hand[0] + state.akas_in_hand[1] + state.akas_in_hand[2]);
        num_doras += sup.num_doras_in_fuuro;

        auto a = calc.agari(additional_yakus, num_doras);
        if (!a) return false;
        if (a->is_yakuman) {
            float v = (float)a->point(is_oya).tsumo_total(is_oya);
            for (int i = 0; i < 4; i++) scores[i] = v;
            return true;
        }
        u8 fu = a->fu, han = a->han;
        for (int i = 0; i < 4; i++) scores[i] = 0.f;
        auto pt = [&](int h) {
            Agari x;
            x.fu = fu;
            x.han = (u8)h;
            return (float)x.point(is_oya).tsumo_total(is_oya);
        };
        bool assume_riichi = sup.is_menzen && sup.prefer_riichi;
        if (assume_riichi && sup.n_dora_indicators == 1) {
            u8 n_indicators[5] = {};
            u8 sum_indicators = 0;
            for (int tid = 0; tid < 34; tid++) {
                u8 count = state.tehai[tid];
                if (count == 0) continue;
                u8 ind_count = state.tiles_in_wall[tile_prev((u8)tid)];
                n_indicators[count] += ind_count;
                sum_indicators += ind_count;
            }
            float uradora_probs[5];
            int n_left = sum_left_tiles(state);
            uradora_probs[0] = (float)(u8)(n_left - sum_indicators) / (float)n_left;
            for (int i = 1; i < 5; i++) uradora_probs[i] = (float)n_indicators[i] / (float)n_left;
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 5; j++) {
                    float p = uradora_probs[j];
                    if (p == 0.f) continue;
                    scores[i] += pt(han + i + j) * p;
                }
        } else if (assume_riichi && sup.n_dora_indicators > 1) {
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 13; j++) {
                    float p = URADORA_PROB_TABLE[sup.n_dora_indicators - 1][j];
                    if (p == 0.f) continue;
                    scores[i] += pt(han + i + j) * p;
                }
        } else {
            for (int i = 0; i < 4; i++) scores[i] = pt(han + i);
        }
        return true;
    }
};

}  // namespace

int sp_candidate_cmp(const SPCandidate& l, const SPCandidate& r, SPColumn by) {  // candidate.rs:73-106
    if (l.tile == r.tile) return 0;
    auto total_cmp = [](float a, float b) {
        int32_t x, y;
        memcpy(&x, &a, 4);
        memcpy(&y, &b, 4);
        x ^= (int32_t)((uint32_t)(x >> 31) >> 1);
        y ^= (int32_t)((uint32_t)(y >> 31) >> 1);
        return (x > y) - (x < y);
    };
    switch (by) {
        case COL_EV: {
            int o = total_cmp(l.exp_values.at(0), r.exp_values.at(0));
            return o ? o : sp_candidate_cmp(l, r, COL_WIN_PROB);
        }
        case COL_WIN_PROB: {
            int o = total_cmp(l.win_probs.at(0), r.win_probs.at(0));
            return o ? o : sp_candidate_cmp(l, r, COL_TENPAI_PROB);
        }
        case COL_TENPAI_PROB: {
            int o = total_cmp(l.tenpai_probs.at(0), r.tenpai_probs.at(0));
            return o ? o : sp_candidate_cmp(l, r, COL_NOT_SHANTEN_DOWN);
        }
        case COL_NOT_SHANTEN_DOWN:
            if (!l.shanten_down && r.shanten_down) return 1;
            if (l.shanten_down && !r.shanten_down) return -1;
            return sp_candidate_cmp(l, r, COL_NUM_REQUIRED);
        case COL_NUM_REQUIRED:
            if (l.num_required_tiles != r.num_required_tiles) return l.num_required_tiles < r.num_required_tiles ? -1 : 1;
            return sp_candidate_cmp(l, r, COL_DISCARD_PRIORITY);
        case COL_DISCARD_PRIORITY: return cmp_discard_priority(l.tile, r.tile);
    }
    return 0;
}

std::vector<SPCandidate> SPCalculator::calc(const SPInitState& init, bool can_discard, int tsumos_left,
                                            int cur_shanten) const {  // calc.rs:84-133
    MJO_ENSURE(cur_shanten >= 0, "can't calculate an agari hand");
    MJO_ENSURE(tsumos_left >= 1, "need at least one more tsumo");
    MJO_ENSURE(tsumos_left <= MAX_TSUMOS_LEFT, "too many tsumos left");
    State st;
    memset(&st, 0, sizeof st);
    memcpy(st.tehai, init.tehai, 34);
    for (int i = 0; i < 3; i++) st.akas_in_hand[i] = init.akas_in_hand[i];
    for (int i = 0; i < 34; i++) st.tiles_in_wall[i] = 4 - init.tiles_seen[i];
    for (int i = 0; i < 3; i++) st.akas_in_wall[i] = !init.akas_seen[i];
    st.n_extra_tsumo = 0;
    Calc c(*this, st, tsumos_left);
    return c.calc(can_discard, cur_shanten);
}

}  // namespace mjo
