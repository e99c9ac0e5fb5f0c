// ORACLE — TEST INFRASTRUCTURE ONLY (see mjo.h).
// Observation + mask encoder: state/obs_repr.rs:18-799, array.rs:1-64, consts.rs:20-28
#include <algorithm>
#include <cmath>

#include "mjo.h"

namespace mjo {

int obs_rows(int version) {  // consts.rs:20-28
    switch (version) {
        case 1: return 938;
        case 2: return 942;
        case 3: return 934;
        case 4: return 1012;
        default: throw Error("bad obs version");
    }
}

namespace {

constexpr int SELF_KAWA_ITEM_CHANNELS = 4;
constexpr int KAWA_ITEM_CHANNELS = 8;
constexpr int MAX_NUM_TURNS = 17;

struct Ctx {  // obs_repr.rs:18-25 + array.rs
    const PlayerState& state;
    float* arr;
    u8* mask;
    int rows;
    int idx = 0;
    bool at_kan_select;
    int version;

    float get(int row, int col) const { return arr[row * 34 + col]; }
    void fill(int row, float v) { fill_rows(row, 1, v); }
    void fill_rows(int row, int n, float v) {
        if (row + n > rows) throw Error("obs row overflow");
        for (int i = row * 34; i < (row + n) * 34; i++) arr[i] = v;
    }
    void assign(int row, int col, float v) {
        if (row >= rows) throw Error("obs row overflow");
        arr[row * 34 + col] = v;
    }
    void assign_rows(int row, int col, int n, float v) {
        for (int k = 0; k < n; k++) assign(row + k, col, v);
    }

    // IntegerEncoder (obs_repr.rs:27-108)
    void int_encode(size_t n_in, size_t cap, bool one_hot, bool rescale, size_t rbf_intervals) {
        size_t n = std::min(n_in, cap);
        switch (version) {
            case 1:
                fill_rows(idx, (int)n, 1.f);
                idx += (int)cap;
                break;
            case 2:
            case 3:
                if (one_hot) {
                    fill(idx + (int)n, 1.f);
                    idx += (int)cap + 1;
                }
                if (rescale) {
                    float v = (float)n / (float)cap;
                    fill(idx, v);
                    idx += 1;
                }
                if (rbf_intervals) {
                    size_t intervals = rbf_intervals;
                    float interval_size = (float)cap / (float)intervals;
                    for (size_t i = 1; i < intervals; i++) {
                        float x = (float)n_in;  // the original value, not the clamped
                        float mu = (float)i * interval_size;
                        float sigma = interval_size;
                        float d = x - mu;
                        float v = expf(-(d * d) / (2.f * (sigma * sigma)));
                        fill(idx + (int)i - 1, v);
                    }
                    idx += (int)intervals - 1;
                }
                break;
            case 4:
                if (one_hot) {
                    fill(idx + (int)n, 1.f);
                    idx += (int)cap + 1;
                }
                if (rescale) {
                    float v = (float)n / (float)cap;
                    fill(idx, v);
                    idx += 1;
                }
                break;
        }
    }

    void encode_tile_set(const std::vector<u8>& tiles) {  // obs_repr.rs:694-712
        int counts[34] = {};
        for (u8 tile : tiles) {
            int tid = deaka(tile);
            assign(idx + counts[tid], tid, 1.f);
            counts[tid]++;
            if (is_aka(tile)) fill(idx + 4 + (tile - T_5MR), 1.f);
        }
        idx += 7;
    }
    void encode_self_kawa(const std::optional<KawaItem>& item) {  // obs_repr.rs:714-734
        if (item) {
            for (u8 kan : item->kan) assign(idx, deaka(kan), 1.f);
            const Sutehai& s = item->sutehai;
            assign(idx + 1, deaka(s.tile), 1.f);
            if (is_aka(s.tile)) fill(idx + 2, 1.f);
            if (s.is_dora) fill(idx + 3, 1.f);
        }
        idx += SELF_KAWA_ITEM_CHANNELS;
    }
    void encode_kawa(const std::optional<KawaItem>& item) {  // obs_repr.rs:736-773
        if (item) {
            if (item->chi_pon) {
                int a = deaka(item->chi_pon->consumed[0]), b = deaka(item->chi_pon->consumed[1]);
                assign(idx, std::min(a, b), 1.f);
                assign(idx + 1, std::max(a, b), 1.f);
            }
            for (u8 kan : item->kan) assign(idx + 2, deaka(kan), 1.f);
            const Sutehai& s = item->sutehai;
            assign(idx + 3, deaka(s.tile), 1.f);
            if (is_aka(s.tile)) fill(idx + 4, 1.f);
            if (s.is_dora) fill(idx + 5, 1.f);
            if (s.is_tedashi) fill(idx + 6, 1.f);
            if (s.is_riichi) fill(idx + 7, 1.f);
        }
        idx += KAWA_ITEM_CHANNELS;
    }
    void encode_ev(float value) {  // obs_repr.rs:632-638
        float v = std::min(std::max(value, 0.f), 100000.f) / 100000.f;
        fill(idx, v);
        v = std::min(std::max(value, 0.f), 30000.f) / 30000.f;
        fill(idx + 1, v);
        idx += 2;
    }
    void encode_sp_table(const std::vector<SPCandidate>& cands, bool can_discard, float ev_scale) {  // :644-692
        if (cands.empty() || cands[0].tenpai_probs.empty() || !(cands[0].tenpai_probs[0] > 0.f)) {
            idx += 3 * MAX_NUM_TURNS;
            return;
        }
        auto each = [&](const SPCandidate& c, bool whole_row, int tid) {
            size_t n = std::min({c.tenpai_probs.size(), c.win_probs.size(), c.exp_values.size()});
            for (size_t turn = 0; turn < n; turn++) {
                float tp = c.tenpai_probs[turn];
                if (!(tp > 0.f)) break;  // take_while(p > 0)
                float wp = c.win_probs[turn];
                float ev = c.exp_values[turn];
                int i = idx + (int)turn;
                float evs = std::min(ev * ev_scale, 1.f);
                if (whole_row) {
                    fill(i, tp);
                    fill(i + MAX_NUM_TURNS, wp);
                    fill(i + 2 * MAX_NUM_TURNS, evs);
                } else {
                    assign(i, tid, tp);
                    assign(i + MAX_NUM_TURNS, tid, wp);
                    assign(i + 2 * MAX_NUM_TURNS, tid, evs);
                }
            }
        };
        if (can_discard) {
            for (auto& c : cands) each(c, false, deaka(c.tile));
        } else {
            each(cands[0], true, 0);
        }
        idx += 3 * MAX_NUM_TURNS;
    }

    void run() {  // obs_repr.rs:126-630
        const PlayerState& st = state;
        const ActionCandidate& cans = st.last_cans;

        for (int t = 0; t < 34; t++)
            if (st.tehai[t] > 0) assign_rows(idx, t, st.tehai[t], 1.f);
        idx += 4;
        for (int i = 0; i < 3; i++)
            if (st.akas_in_hand[i]) fill(idx + i, 1.f);
        idx += 3;

        for (int i = 0; i < 4; i++) {
            int score = st.scores[i];
            float v = (float)std::min(std::max(score, 0), 100000) / 100000.f;
            fill(idx, v);
            idx += 1;
            if (version == 2 || version == 3) {
                // `score as usize / 100`: a negative i32 sign-extends to a huge usize
                size_t n = (size_t)(int64_t)score / 100;
                int_encode(n, 500, false, false, 10);
            } else if (version == 4) {
                float v2 = (float)std::min(std::max(score, 0), 30000) / 30000.f;
                fill(idx, v2);
                idx += 1;
            }
        }

        fill(idx + st.rank, 1.f);
        idx += 4;

        if (version == 1) fill_rows(idx, st.kyoku, 1.f);
        else fill(idx + st.kyoku, 1.f);
        idx += 4;

        size_t cap = (version == 1 || version == 4) ? 10 : 6;
        int_encode(st.honba, cap, false, version == 4, 3);
        int_encode(st.kyotaku, cap, false, version == 4, 3);

        assign(idx, st.bakaze, 1.f);
        assign(idx + 1, st.jikaze, 1.f);
        idx += 2;

        if (version >= 2) {
            int n = std::min<int>(st.bakaze - T_E, 1) * 4 + st.kyoku;
            int_encode(n, 7, false, true, 0);
        }

        encode_tile_set(st.dora_indicators);

        {
            const auto& k0 = st.kawa[0];
            size_t n = std::min<size_t>(k0.size(), 6);
            for (size_t i = 0; i < n; i++) encode_self_kawa(k0[i]);
            idx += (6 - (int)n) * SELF_KAWA_ITEM_CHANNELS;
            n = std::min<size_t>(k0.size(), 18);
            for (size_t i = 0; i < n; i++) encode_self_kawa(k0[k0.size() - 1 - i]);
            idx += (18 - (int)n) * SELF_KAWA_ITEM_CHANNELS;
        }

        size_t max_kawa_len = 0;
        for (int i = 0; i < 4; i++) max_kawa_len = std::max(max_kawa_len, st.kawa[i].size());
        if (version == 3 || version == 4) {
            for (size_t turn = 0; turn < st.kawa[0].size(); turn++) {
                if (st.kawa[0][turn]) {
                    int tid = deaka(st.kawa[0][turn]->sutehai.tile);
                    float v = expf(-0.2f * (float)(max_kawa_len - 1 - turn));
                    assign(idx, tid, v);
                }
            }
            idx += 1;
        }

        for (int p = 1; p < 4; p++) {
            const auto& pk = st.kawa[p];
            size_t n = std::min<size_t>(pk.size(), 6);
            for (size_t i = 0; i < n; i++) encode_kawa(pk[i]);
            idx += (6 - (int)n) * KAWA_ITEM_CHANNELS;
            n = std::min<size_t>(pk.size(), 18);
            for (size_t i = 0; i < n; i++) encode_kawa(pk[pk.size() - 1 - i]);
            idx += (18 - (int)n) * KAWA_ITEM_CHANNELS;

            if (version == 2) {
                size_t turn = 0;  // enumerate AFTER flatten
                for (auto& item : pk) {
                    if (!item) continue;
                    int row = (int)std::min<size_t>(turn / 6, 2);
                    int tid = deaka(item->sutehai.tile);
                    assign(idx + row, tid, 1.f);
                    if (item->sutehai.is_tedashi) assign(idx + 3 + row, tid, 1.f);
                    turn++;
                }
                idx += 6;
            } else if (version == 3 || version == 4) {
                for (size_t turn = 0; turn < pk.size(); turn++) {
                    if (!pk[turn]) continue;
                    const Sutehai& s = pk[turn]->sutehai;
                    int tid = deaka(s.tile);
                    float v = expf(-0.2f * (float)(max_kawa_len - 1 - turn));
                    assign(idx, tid, v);
                    if (s.is_tedashi) assign(idx + 1, tid, v);
                    if (s.is_riichi) assign(idx + 2, tid, v);
                }
                idx += 3;
            }
        }

        fill(idx, (float)st.tiles_left / 69.f);
        idx += 1;

        for (int i = 0; i < 4; i++) int_encode(st.doras_owned[i], 12, false, true, 3);

        u8 doras_unseen = (u8)((u8)st.dora_indicators.size() * 4 + 3 - st.doras_seen);
        int_encode(doras_unseen, 5 * 4 + 3, false, true, 4);

        for (int i = 0; i < 4; i++) encode_tile_set(st.kawa_overview[i]);

        for (int p = 0; p < 4; p++) {
            for (auto& f : st.fuuro_overview[p]) {
                for (u8 tile : f) {
                    int tid = deaka(tile);
                    int i = 0;
                    while (i < 4 && get(idx + i, tid) != 0.f) i++;
                    if (i == 4) throw Error("fuuro encode overflow");
                    assign(idx + i, tid, 1.f);
                    if (is_aka(tile)) fill(idx + 4, 1.f);
                }
                idx += 5;
            }
            idx += (4 - (int)st.fuuro_overview[p].size()) * 5;
        }

        for (int p = 0; p < 4; p++) {
            for (u8 tile : st.ankan_overview[p]) assign(idx, tile, 1.f);
            idx += 1;
        }

        if (version >= 2) {
            for (int t = 0; t < 34; t++) assign(idx, t, (float)st.tiles_seen[t] / 4.f);
            idx += 1;
            for (int p = 1; p < 4; p++) {
                if (st.last_tedashis[p]) {
                    const Sutehai& s = *st.last_tedashis[p];
                    assign(idx, deaka(s.tile), 1.f);
                    if (is_aka(s.tile)) fill(idx + 1, 1.f);
                    if (s.is_dora) fill(idx + 2, 1.f);
                }
                idx += 3;
            }
            for (int p = 1; p < 4; p++) {
                if (st.riichi_sutehais[p]) {
                    const Sutehai& s = *st.riichi_sutehais[p];
                    assign(idx, deaka(s.tile), 1.f);
                    if (is_aka(s.tile)) fill(idx + 1, 1.f);
                    if (s.is_dora) fill(idx + 2, 1.f);
                }
                idx += 3;
            }
        }

        for (int i = 1; i < 4; i++)
            if (st.riichi_declared[i]) fill(idx + i - 1, 1.f);
        idx += 3;
        for (int i = 1; i < 4; i++)
            if (st.riichi_accepted[i]) fill(idx + i - 1, 1.f);
        idx += 3;

        for (int t = 0; t < 34; t++)
            if (st.waits[t]) assign(idx, t, 1.f);
        idx += 1;

        if (st.at_furiten) fill(idx, 1.f);
        idx += 1;

        int_encode((size_t)st.shanten, 6, true, false, 0);

        if (st.riichi_accepted[0]) fill(idx, 1.f);
        idx += 1;

        if (at_kan_select) fill(idx, 1.f);
        idx += 1;

        if (cans.can_pass()) {
            if (!st.last_kawa_tile) throw Error("building chi/pon/daiminkan/ron feature without any kawa tile");
            u8 tile = *st.last_kawa_tile;
            int tid = deaka(tile);
            assign(idx, tid, 1.f);
            if (is_aka(tile)) fill(idx + 1, 1.f);
            if (st.dora_factor[tid] > 0) fill(idx + 2, 1.f);
            if (!at_kan_select) mask[45] = 1;
            else if (cans.can_daiminkan) mask[tid] = 1;
        }
        idx += 3;

        if (cans.can_discard) {
            bool dc[37];
            st.discard_candidates_aka(dc);
            for (int t = 0; t < 37; t++) {
                if (!dc[t]) continue;
                assign(idx, deaka((u8)t), 1.f);
                if (!at_kan_select) mask[t] = 1;
            }
            for (int t = 0; t < 34; t++)
                if (st.keep_shanten_discards[t]) assign(idx + 1, t, 1.f);
            for (int t = 0; t < 34; t++)
                if (st.next_shanten_discards[t]) assign(idx + 2, t, 1.f);
            if (st.shanten <= 1) {
                bool ut[34];
                st.discard_candidates_with_unconditional_tenpai(ut);
                for (int t = 0; t < 34; t++)
                    if (ut[t]) assign(idx + 3, t, 1.f);
            }
            if (st.riichi_declared[0]) fill(idx + 4, 1.f);
        }
        idx += 5;

        if (cans.can_riichi) {
            fill(idx, 1.f);
            if (!at_kan_select) mask[37] = 1;
        }
        idx += 1;

        if (cans.can_chi_low) {
            fill(idx, 1.f);
            if (!at_kan_select) mask[38] = 1;
        }
        if (cans.can_chi_mid) {
            fill(idx + 1, 1.f);
            if (!at_kan_select) mask[39] = 1;
        }
        if (cans.can_chi_high) {
            fill(idx + 2, 1.f);
            if (!at_kan_select) mask[40] = 1;
        }
        idx += 3;

        if (cans.can_pon) {
            fill(idx, 1.f);
            if (!at_kan_select) mask[41] = 1;
        }
        idx += 1;

        if (cans.can_daiminkan) {
            fill(idx, 1.f);
            if (!at_kan_select) mask[42] = 1;
        }
        idx += 1;

        if (cans.can_ankan) {
            for (u8 tile : st.ankan_candidates) {
                assign(idx, tile, 1.f);
                if (at_kan_select) mask[tile] = 1;
            }
            if (!at_kan_select) mask[42] = 1;
        }
        idx += 1;

        if (cans.can_kakan) {
            for (u8 tile : st.kakan_candidates) {
                assign(idx, tile, 1.f);
                if (at_kan_select) mask[tile] = 1;
            }
            if (!at_kan_select) mask[42] = 1;
        }
        idx += 1;

        if (cans.can_agari()) {
            fill(idx, 1.f);
            if (!at_kan_select) mask[43] = 1;
        }
        idx += 1;

        if (cans.can_ryukyoku) {
            fill(idx, 1.f);
            if (!at_kan_select) mask[44] = 1;
        }
        idx += 1;

        if (version == 4) {
            std::vector<SPCandidate> table;
            bool ok = true;
            try {
                table = st.single_player_tables();
            } catch (const Error&) {
                ok = false;
            }
            if (ok) {
                float max_ev = 0.f;
                if (!table.empty() && !table[0].exp_values.empty()) max_ev = table[0].exp_values[0];
                encode_ev(max_ev);

                if (cans.can_discard) {
                    for (auto& c : table) {
                        int discard_tid = deaka(c.tile);
                        for (auto& r : c.required_tiles) {
                            int req = deaka(r.tile);
                            if (c.shanten_down) assign(idx + 34 + discard_tid, req, 1.f);
                            else assign(idx + discard_tid, req, 1.f);
                        }
                    }
                    idx += 2 * 34;
                    // Iterator::max_by returns the LAST maximum
                    const SPCandidate* best = nullptr;
                    for (auto& c : table)
                        if (!best || sp_candidate_cmp(c, *best, COL_NOT_SHANTEN_DOWN) >= 0) best = &c;
                    if (!best) throw Error("empty sp table");
                    assign(idx, deaka(best->tile), 1.f);
                    idx += 2;
                } else {
                    idx += 2 * 34 + 1;
                    for (auto& r : table.at(0).required_tiles) assign(idx, deaka(r.tile), 1.f);
                    idx += 1;
                }
                float ev_scale = max_ev < 1.f ? 0.f : 1.f / max_ev;
                encode_sp_table(table, cans.can_discard, ev_scale);
            } else {
                float min_tsumo_agari = 0.f;
                try {
                    Point p = st.agari_points(cans.can_ron_agari, nullptr, 0);
                    min_tsumo_agari = (float)p.tsumo_total(st.is_oya());
                } catch (const Error&) {
                }
                encode_ev(min_tsumo_agari);
                idx += 2 * 34 + 2 + 3 * MAX_NUM_TURNS;
            }
        }

        if (idx != rows) throw Error("obs row count mismatch: " + std::to_string(idx));
    }
};

}  // namespace

void PlayerState::encode_obs(int version, bool at_kan_select, float* obs, u8* mask) const {
    int rows = obs_rows(version);
    std::fill(obs, obs + (size_t)rows * 34, 0.f);
    memset(mask, 0, 46);
    Ctx ctx{*this, obs, mask, rows, 0, at_kan_select, version};
    ctx.run();
}

}  // namespace mjo
