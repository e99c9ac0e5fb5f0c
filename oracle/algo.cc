// ORACLE — TEST INFRASTRUCTURE ONLY (see mjo.h).
// tile helpers, lookup tables, shanten, point, agari.
#include <algorithm>

#include "mjo.h"

namespace mjo {

Tables g_tables;

// ---------------------------------------------------------------- tiles
u8 tile_next(u8 t) {  // tile.rs:117-132
    if (is_unknown(t)) return t;
    u8 tile = deaka(t);
    u8 kind = tile / 9, num = tile % 9;
    if (kind < 3) return kind * 9 + (num + 1) % 9;
    if (num < 4) return 27 + (num + 1) % 4;
    return 27 + 4 + (num - 4 + 1) % 3;
}
u8 tile_prev(u8 t) {  // tile.rs:134-150
    if (is_unknown(t)) return t;
    u8 tile = deaka(t);
    u8 kind = tile / 9, num = tile % 9;
    if (kind < 3) return kind * 9 + (num + 9 - 1) % 9;
    if (num < 4) return 27 + (num + 4 - 1) % 4;
    return 27 + 4 + (num - 4 + 3 - 1) % 3;
}
static const u8 DISCARD_PRIORITIES[38] = {  // tile.rs:21-28
    6, 5, 4, 3, 2, 3, 4, 5, 6, 6, 5, 4, 3, 2, 3, 4, 5, 6, 6, 5, 4, 3, 2, 3, 4, 5, 6, 7, 7, 7, 7, 7, 7, 7, 1, 1, 1, 0};
int cmp_discard_priority(u8 l, u8 r) {  // tile.rs:169-177
    u8 pl = DISCARD_PRIORITIES[l], pr = DISCARD_PRIORITIES[r];
    if (pl != pr) return pl < pr ? -1 : 1;
    // Ordering::Equal => r.cmp(&l)
    if (r != l) return r < l ? -1 : 1;
    return 0;
}
static const char* NAMES[38] = {"1m", "2m", "3m", "4m", "5m", "6m", "7m", "8m", "9m", "1p", "2p", "3p", "4p",
                                "5p", "6p", "7p", "8p", "9p", "1s", "2s", "3s", "4s", "5s", "6s", "7s", "8s",
                                "9s", "E",  "S",  "W",  "N",  "P",  "F",  "C",  "5mr", "5pr", "5sr", "?"};
std::string tile_name(u8 t) { return t < 38 ? NAMES[t] : "??"; }
int tile_from_name(const std::string& s) {
    for (int i = 0; i < 38; i++)
        if (s == NAMES[i]) return i;
    return -1;
}

// ---------------------------------------------------------------- tables
void set_tables(const u8* p, size_t size) {
    MJO_ENSURE(size >= 16 && memcmp(p, "MJT1", 4) == 0, "bad table payload");
    u32 ns, nj, na;
    memcpy(&ns, p + 4, 4);
    memcpy(&nj, p + 8, 4);
    memcpy(&na, p + 12, 4);
    MJO_ENSURE(size == 16 + (size_t)ns * 5 + (size_t)nj * 5 + (size_t)na * 24, "bad table payload size");
    g_tables.suhai = p + 16;
    g_tables.n_suhai = ns;
    g_tables.jihai = g_tables.suhai + (size_t)ns * 5;
    g_tables.n_jihai = nj;
    g_tables.agari = reinterpret_cast<const AgariRec*>(g_tables.jihai + (size_t)nj * 5);
    g_tables.n_agari = na;
}

// shanten.rs:27-44: 5 bytes per row, low nibble first; out-of-range -> default (zeros) (:52,:72,:92-95)
static void table_row(const u8* tab, u32 n, size_t index, u8 out[10]) {
    if (index >= n) {
        memset(out, 0, 10);
        return;
    }
    const u8* b = tab + index * 5;
    for (int i = 0; i < 5; i++) {
        out[i * 2] = b[i] & 15;
        out[i * 2 + 1] = (b[i] >> 4) & 15;
    }
}
static size_t sum_tiles(const u8* tiles, int n) {  // shanten.rs:82-84
    size_t acc = 0;
    for (int i = 0; i < n; i++) acc = acc * 5 + tiles[i];
    return acc;
}
static void add_suhai(u8 lhs[10], size_t index, int m) {  // shanten.rs:51-69
    u8 tab[10];
    table_row(g_tables.suhai, g_tables.n_suhai, index, tab);
    for (int j = 5 + m; j >= 5; j--) {
        u8 sht = std::min<u8>(lhs[j] + tab[0], lhs[0] + tab[j]);
        for (int k = 5; k < j; k++) sht = std::min<u8>(std::min<u8>(sht, lhs[k] + tab[j - k]), lhs[j - k] + tab[k]);
        lhs[j] = sht;
    }
    for (int j = m; j >= 0; j--) {
        u8 sht = lhs[j] + tab[0];
        for (int k = 0; k < j; k++) sht = std::min<u8>(sht, lhs[k] + tab[j - k]);
        lhs[j] = sht;
    }
}
static void add_jihai(u8 lhs[10], size_t index, int m) {  // shanten.rs:71-80
    u8 tab[10];
    table_row(g_tables.jihai, g_tables.n_jihai, index, tab);
    int j = m + 5;
    u8 sht = std::min<u8>(lhs[j] + tab[0], lhs[0] + tab[j]);
    for (int k = 5; k < j; k++) sht = std::min<u8>(std::min<u8>(sht, lhs[k] + tab[j - k]), lhs[j - k] + tab[k]);
    lhs[j] = sht;
}
int calc_normal(const u8* tiles, int len_div3) {  // shanten.rs:88-102
    u8 ret[10];
    table_row(g_tables.suhai, g_tables.n_suhai, sum_tiles(tiles, 9), ret);
    add_suhai(ret, sum_tiles(tiles + 9, 9), len_div3);
    add_suhai(ret, sum_tiles(tiles + 18, 9), len_div3);
    add_jihai(ret, sum_tiles(tiles + 27, 7), len_div3);
    return (int)(i8)ret[5 + len_div3] - 1;
}
int calc_chitoi(const u8* tiles) {  // shanten.rs:104-118
    int pairs = 0, kinds = 0;
    for (int i = 0; i < 34; i++)
        if (tiles[i] > 0) {
            kinds++;
            if (tiles[i] >= 2) pairs++;
        }
    int redunct = kinds >= 7 ? 0 : 7 - kinds;
    return 7 - pairs + redunct - 1;
}
static const u8 YAOKYUU[13] = {0, 8, 9, 17, 18, 26, 27, 28, 29, 30, 31, 32, 33};
int calc_kokushi(const u8* tiles) {  // shanten.rs:120-137
    int pairs = 0, kinds = 0;
    for (u8 i : YAOKYUU)
        if (tiles[i] > 0) {
            kinds++;
            if (tiles[i] >= 2) pairs++;
        }
    int redunct = pairs > 0;
    return 14 - kinds - redunct - 1;
}
int calc_all(const u8* tiles, int len_div3) {  // shanten.rs:139-150
    int shanten = calc_normal(tiles, len_div3);
    if (shanten <= 0 || len_div3 < 4) return shanten;
    shanten = std::min(shanten, calc_chitoi(tiles));
    if (shanten > 0) return std::min(shanten, calc_kokushi(tiles));
    return shanten;
}

// ---------------------------------------------------------------- point (point.rs:13-103)
Point point_calc(bool is_oya, int fu, int han) {
    struct Row { int ron, ko, oya; };
    auto mk = [](Row r) { Point p; p.ron = r.ron; p.tsumo_ko = r.ko; p.tsumo_oya = r.oya; return p; };
    // The reference's match is ordered: explicit (fu,han) arms first, then the mangan+ arms.
    auto key = [&](int f, int h) { return fu == f && han == h; };
    if (is_oya) {
        if (key(20, 2) || key(40, 1)) return mk({2000, 700, 0});
        if (key(20, 3) || key(40, 2) || key(80, 1)) return mk({3900, 1300, 0});
        if (key(20, 4) || key(40, 3) || key(80, 2)) return mk({7700, 2600, 0});
        if (key(25, 2) || key(50, 1)) return mk({2400, 800, 0});
        if (key(25, 3) || key(50, 2) || key(100, 1)) return mk({4800, 1600, 0});
        if (key(25, 4) || key(50, 3) || key(100, 2)) return mk({9600, 3200, 0});
        if (key(30, 1)) return mk({1500, 500, 0});
        if (key(30, 2) || key(60, 1)) return mk({2900, 1000, 0});
        if (key(30, 3) || key(60, 2)) return mk({5800, 2000, 0});
        if (key(30, 4) || key(60, 3)) return mk({11600, 3900, 0});
        if (key(70, 1)) return mk({3400, 1200, 0});
        if (key(70, 2)) return mk({6800, 2300, 0});
        if (key(90, 1)) return mk({4400, 1500, 0});
        if (key(90, 2)) return mk({8700, 2900, 0});
        if (key(110, 1)) return mk({5300, 1800, 0});
        if (key(110, 2)) return mk({10600, 3600, 0});
        if (han == 5 || (fu >= 40 && han == 4) || (fu >= 70 && han == 3)) return mk({12000, 4000, 0});
        if (han >= 6 && han <= 7) return mk({18000, 6000, 0});
        if (han >= 8 && han <= 10) return mk({24000, 8000, 0});
        if (han >= 11 && han <= 12) return mk({36000, 12000, 0});
        if (han >= 13) return mk({48000, 16000, 0});
    } else {
        if (key(20, 2) || key(40, 1)) return mk({1300, 400, 700});
        if (key(20, 3) || key(40, 2) || key(80, 1)) return mk({2600, 700, 1300});
        if (key(20, 4) || key(40, 3) || key(80, 2)) return mk({5200, 1300, 2600});
        if (key(25, 2) || key(50, 1)) return mk({1600, 400, 800});
        if (key(25, 3) || key(50, 2) || key(100, 1)) return mk({3200, 800, 1600});
        if (key(25, 4) || key(50, 3) || key(100, 2)) return mk({6400, 1600, 3200});
        if (key(30, 1)) return mk({1000, 300, 500});
        if (key(30, 2) || key(60, 1)) return mk({2000, 500, 1000});
        if (key(30, 3) || key(60, 2)) return mk({3900, 1000, 2000});
        if (key(30, 4) || key(60, 3)) return mk({7700, 2000, 3900});
        if (key(70, 1)) return mk({2300, 600, 1200});
        if (key(70, 2)) return mk({4500, 1200, 2300});
        if (key(90, 1)) return mk({2900, 800, 1500});
        if (key(90, 2)) return mk({5800, 1500, 2900});
        if (key(110, 1)) return mk({3600, 900, 1800});
        if (key(110, 2)) return mk({7100, 1800, 3600});
        if (han == 5 || (fu >= 40 && han == 4) || (fu >= 70 && han == 3)) return mk({8000, 2000, 4000});
        if (han >= 6 && han <= 7) return mk({12000, 3000, 6000});
        if (han >= 8 && han <= 10) return mk({16000, 4000, 8000});
        if (han >= 11 && han <= 12) return mk({24000, 6000, 12000});
        if (han >= 13) return mk({32000, 8000, 16000});
    }
    throw Error("impossible combinition of " + std::to_string(fu) + " fu and " + std::to_string(han) + " han");
}
Point point_yakuman(bool is_oya, int count) {  // point.rs:87-103
    Point p;
    if (is_oya) {
        p.ron = 48000 * count;
        p.tsumo_ko = 16000 * count;
        p.tsumo_oya = 0;
    } else {
        p.ron = 32000 * count;
        p.tsumo_ko = 8000 * count;
        p.tsumo_oya = 16000 * count;
    }
    return p;
}

// ---------------------------------------------------------------- agari
int agari_cmp(const Agari& l, const Agari& r) {  // agari.rs:180-195
    if (l.is_yakuman && r.is_yakuman) return (l.n > r.n) - (l.n < r.n);
    if (l.is_yakuman) return 1;
    if (r.is_yakuman) return -1;
    if (l.han != r.han) return l.han < r.han ? -1 : 1;
    return (l.fu > r.fu) - (l.fu < r.fu);
}

const AgariRec* agari_lookup(u32 key) {
    const AgariRec* b = g_tables.agari;
    const AgariRec* e = b + g_tables.n_agari;
    const AgariRec* it = std::lower_bound(b, e, key, [](const AgariRec& r, u32 k) { return r.key < k; });
    if (it != e && it->key == key) return it;
    return nullptr;
}

u32 get_tile14_and_key(const u8* tiles, u8 tile14[14]) {  // agari.rs:767-838
    memset(tile14, 0, 14);
    int n14 = 0;
    u32 key = 0;
    int bit_idx = -1;
    bool prev_in_hand = false;
    for (int kind = 0; kind < 3; kind++) {
        for (int num = 0; num < 9; num++) {
            u8 c = tiles[kind * 9 + num];
            if (c > 0) {
                prev_in_hand = true;
                tile14[n14++] = kind * 9 + num;
                bit_idx += 1;
                switch (c) {
                    case 2: key |= 0b11u << bit_idx; bit_idx += 2; break;
                    case 3: key |= 0b1111u << bit_idx; bit_idx += 4; break;
                    case 4: key |= 0b111111u << bit_idx; bit_idx += 6; break;
                    default: break;
                }
            } else if (prev_in_hand) {
                prev_in_hand = false;
                key |= 1u << bit_idx;
                bit_idx += 1;
            }
        }
        if (prev_in_hand) {
            prev_in_hand = false;
            key |= 1u << bit_idx;
            bit_idx += 1;
        }
    }
    for (int tid = 27; tid < 34; tid++) {
        u8 c = tiles[tid];
        if (c == 0) continue;
        tile14[n14++] = tid;
        bit_idx += 1;
        switch (c) {
            case 2: key |= 0b11u << bit_idx; bit_idx += 2; break;
            case 3: key |= 0b1111u << bit_idx; bit_idx += 4; break;
            case 4: key |= 0b111111u << bit_idx; bit_idx += 6; break;
            default: break;
        }
        key |= 1u << bit_idx;
        bit_idx += 1;
    }
    return key;
}

namespace {

struct Div {  // agari.rs:53-64, 126-157
    u8 pair_idx;
    u8 kotsu_idxs[4]; int n_kotsu;
    u8 shuntsu_idxs[4]; int n_shuntsu;
    bool has_chitoi, has_chuuren, has_ittsuu, has_ryanpeikou, has_ipeikou;
    explicit Div(u32 v) {
        pair_idx = (v >> 6) & 0b1111;
        n_kotsu = v & 0b111;
        for (int i = 0; i < n_kotsu; i++) kotsu_idxs[i] = (v >> (10 + i * 4)) & 0b1111;
        n_shuntsu = (v >> 3) & 0b111;
        for (int i = 0; i < n_shuntsu; i++) shuntsu_idxs[i] = (v >> (10 + (n_kotsu + i) * 4)) & 0b1111;
        has_chitoi = (v >> 26) & 1;
        has_chuuren = (v >> 27) & 1;
        has_ittsuu = (v >> 28) & 1;
        has_ryanpeikou = (v >> 29) & 1;
        has_ipeikou = (v >> 30) & 1;
    }
};

struct DivWorker {  // agari.rs:100-124, 287-761
    const AgariCalc& sup;
    const u8* tile14;
    const Div& div;
    u8 pair_tile;
    u8 menzen_kotsu[4]; int n_mk;
    u8 menzen_shuntsu[4]; int n_ms;
    bool winning_tile_makes_minkou;

    DivWorker(const AgariCalc& c, const u8* t14, const Div& d) : sup(c), tile14(t14), div(d) {  // :288-311
        pair_tile = tile14[div.pair_idx];
        n_mk = div.n_kotsu;
        for (int i = 0; i < n_mk; i++) menzen_kotsu[i] = tile14[div.kotsu_idxs[i]];
        n_ms = div.n_shuntsu;
        for (int i = 0; i < n_ms; i++) menzen_shuntsu[i] = tile14[div.shuntsu_idxs[i]];
        winning_tile_makes_minkou = calc_wtmm();
    }
    bool mk_contains(u8 t) const {
        for (int i = 0; i < n_mk; i++)
            if (menzen_kotsu[i] == t) return true;
        return false;
    }
    bool ms_contains(u8 t) const {
        for (int i = 0; i < n_ms; i++)
            if (menzen_shuntsu[i] == t) return true;
        return false;
    }
    bool calc_wtmm() const {  // :314-338
        if (!sup.is_ron) return false;
        if (!mk_contains(sup.winning_tile)) return false;
        if (sup.winning_tile >= 27) return true;
        u8 kind = sup.winning_tile / 9, num = sup.winning_tile % 9;
        u8 low = kind * 9 + (num >= 2 ? num - 2 : 0);
        u8 high = kind * 9 + std::min<u8>(num, 6);
        for (u8 t = low; t <= high; t++)
            if (ms_contains(t)) return false;
        return true;
    }
    // iteration helpers (agari.rs:341-365): order = menzen_kotsu, pons, minkans, ankans | menzen_shuntsu, chis
    template <class F> void for_kotsu_kantsu(F f) const {
        for (int i = 0; i < n_mk; i++) f(menzen_kotsu[i]);
        for (int i = 0; i < sup.n_pons; i++) f(sup.pons[i]);
        for (int i = 0; i < sup.n_minkans; i++) f(sup.minkans[i]);
        for (int i = 0; i < sup.n_ankans; i++) f(sup.ankans[i]);
    }
    template <class F> void for_shuntsu(F f) const {
        for (int i = 0; i < n_ms; i++) f(menzen_shuntsu[i]);
        for (int i = 0; i < sup.n_chis; i++) f(sup.chis[i]);
    }

    u8 calc_fu(bool has_pinfu) const {  // :367-452
        if (div.has_chitoi) return 25;
        int fu = 20;
        for (int i = 0; i < n_mk; i++) {
            u8 t = menzen_kotsu[i];
            bool is_minkou = winning_tile_makes_minkou && t == sup.winning_tile;
            bool yao = is_yaokyuu(t);
            if (!is_minkou && yao) fu += 8;
            else if ((!is_minkou && !yao) || (is_minkou && yao)) fu += 4;
            else fu += 2;
        }
        for (int i = 0; i < sup.n_pons; i++) fu += is_yaokyuu(sup.pons[i]) ? 4 : 2;
        for (int i = 0; i < sup.n_ankans; i++) fu += is_yaokyuu(sup.ankans[i]) ? 32 : 16;
        for (int i = 0; i < sup.n_minkans; i++) fu += is_yaokyuu(sup.minkans[i]) ? 16 : 8;
        if (pair_tile == T_P || pair_tile == T_F || pair_tile == T_C) {
            fu += 2;
        } else {
            if (pair_tile == sup.bakaze) fu += 2;
            if (pair_tile == sup.jikaze) fu += 2;
        }
        if (fu == 20) {
            if (!sup.is_menzen) return 30;
            if (has_pinfu) return sup.is_ron ? 30 : 20;
            return sup.is_ron ? 40 : 30;
        }
        if (!sup.is_ron) fu += 2;
        else if (sup.is_menzen) fu += 10;
        if (!winning_tile_makes_minkou) {
            if (pair_tile == sup.winning_tile) {
                fu += 2;
            } else {
                bool kp = false;
                for (int i = 0; i < n_ms; i++) {
                    u8 s = menzen_shuntsu[i];
                    if (s + 1 == sup.winning_tile || (s % 9 == 0 && s + 2 == sup.winning_tile) ||
                        (s % 9 == 6 && s == sup.winning_tile))
                        kp = true;
                }
                if (kp) fu += 2;
            }
        }
        return (u8)(((fu - 1) / 10 + 1) * 10);
    }

    std::optional<Agari> search_yakus(bool RETURN_IF_ANY) const {  // :454-761
        int han = 0, yakuman = 0;
        bool has_pinfu = n_ms == 4 && !(pair_tile == T_P || pair_tile == T_F || pair_tile == T_C) &&
                         pair_tile != sup.bakaze && pair_tile != sup.jikaze;
        if (has_pinfu) {
            bool any = false;
            for (int i = 0; i < n_ms; i++) {
                u8 s = menzen_shuntsu[i];
                int num = s % 9 + 1;
                if ((num <= 6 && s == sup.winning_tile) || (num >= 2 && s + 2 == sup.winning_tile)) any = true;
            }
            has_pinfu = any;
        }
        auto make_return = [&]() -> std::optional<Agari> {
            Agari a;
            if (yakuman > 0) {
                a.is_yakuman = true;
                a.n = yakuman;
                return a;
            }
            if (han > 0) {
                a.han = han;
                a.fu = (RETURN_IF_ANY || han >= 5) ? 0 : calc_fu(has_pinfu);
                return a;
            }
            return std::nullopt;
        };
#define CHECK_EARLY_RETURN(stmt) \
    do {                         \
        stmt;                    \
        if (RETURN_IF_ANY) return make_return(); \
    } while (0)

        if (has_pinfu) CHECK_EARLY_RETURN(han += 1);
        if (div.has_chitoi) CHECK_EARLY_RETURN(han += 2);
        if (div.has_ryanpeikou) CHECK_EARLY_RETURN(han += 3);
        if (div.has_chuuren) CHECK_EARLY_RETURN(yakuman += 1);

        auto simple = [](u8 t) {
            u8 kind = t / 9, num = t % 9;
            return kind < 3 && num > 0 && num < 8;
        };
        bool has_tanyao;
        if (div.has_chitoi) {
            has_tanyao = true;
            for (int i = 0; i < 7; i++)
                if (!simple(tile14[i])) has_tanyao = false;
        } else {
            has_tanyao = true;
            for_shuntsu([&](u8 s) {
                u8 num = s % 9;
                if (!(num > 0 && num < 6)) has_tanyao = false;
            });
            for_kotsu_kantsu([&](u8 k) {
                if (!simple(k)) has_tanyao = false;
            });
            if (!simple(pair_tile)) has_tanyao = false;
        }
        if (has_tanyao) CHECK_EARLY_RETURN(han += 1);

        bool has_toitoi = !div.has_chitoi && n_ms == 0 && sup.n_chis == 0;
        if (has_toitoi) CHECK_EARLY_RETURN(han += 2);

        {  // isou (:533-571)
            int isou_kind = -1;
            bool has_jihai = false, is_chin_or_hon = true, stop = false;
            auto iter_fn = [&](u8 m) {
                if (stop) return;
                u8 kind = m / 9;
                if (kind >= 3) {
                    has_jihai = true;
                    return;
                }
                if (isou_kind >= 0) {
                    if (isou_kind != kind) {
                        is_chin_or_hon = false;
                        stop = true;
                    }
                } else {
                    isou_kind = kind;
                }
            };
            if (div.has_chitoi) {
                for (int i = 0; i < 7; i++) iter_fn(tile14[i]);
            } else {
                for_kotsu_kantsu(iter_fn);
                for_shuntsu(iter_fn);
                iter_fn(pair_tile);
            }
            if (isou_kind < 0) {
                CHECK_EARLY_RETURN(yakuman += 1);  // 字一色
            } else if (is_chin_or_hon) {
                int n = (has_jihai ? 2 : 5) + (sup.is_menzen ? 1 : 0);
                CHECK_EARLY_RETURN(han += n);
            }
        }

        if (!div.has_chitoi) {
            // 一盃口 (:573-596)
            if (div.has_ipeikou) {
                CHECK_EARLY_RETURN(han += 1);
            } else if (sup.n_ankans > 0 && sup.is_menzen && n_ms >= 2) {
                u8 marks[3] = {0, 0, 0};
                bool has_ipeikou = false;
                for (int i = 0; i < n_ms && !has_ipeikou; i++) {
                    u8 t = menzen_shuntsu[i];
                    int kind = t / 9, num = t % 9;
                    if ((marks[kind] >> num) & 1) has_ipeikou = true;
                    else marks[kind] |= 1 << num;
                }
                if (has_ipeikou) CHECK_EARLY_RETURN(han += 1);
            }
            // 一気通貫 (:598-619)
            if (sup.is_menzen && div.has_ittsuu) {
                CHECK_EARLY_RETURN(han += 2);
            } else if (sup.n_chis == 0 && div.has_ittsuu) {
                CHECK_EARLY_RETURN(han += 1);
            } else if (n_ms + sup.n_chis >= 3) {
                int kinds[3] = {0, 0, 0};
                for_shuntsu([&](u8 s) {
                    int kind = s / 9, num = s % 9;
                    if (num == 0) kinds[kind] |= 0b001;
                    else if (num == 3) kinds[kind] |= 0b010;
                    else if (num == 6) kinds[kind] |= 0b100;
                });
                if (kinds[0] == 0b111 || kinds[1] == 0b111 || kinds[2] == 0b111) CHECK_EARLY_RETURN(han += 1);
            }
            // 三色 (:621-647)
            int s_counter[9] = {};
            for_shuntsu([&](u8 s) { s_counter[s % 9] |= 1 << (s / 9); });
            bool sanshoku = false;
            for (int i = 0; i < 9; i++)
                if (s_counter[i] == 0b111) sanshoku = true;
            if (sanshoku) {
                int n = sup.is_menzen ? 2 : 1;
                CHECK_EARLY_RETURN(han += n);
            } else {
                int k_counter[9] = {};
                for_kotsu_kantsu([&](u8 k) {
                    if (k / 9 < 3) k_counter[k % 9] |= 1 << (k / 9);
                });
                bool doukou = false;
                for (int i = 0; i < 9; i++)
                    if (k_counter[i] == 0b111) doukou = true;
                if (doukou) CHECK_EARLY_RETURN(han += 2);
            }
            // 暗刻 (:649-657)
            int ankous_count = sup.n_ankans + n_mk - (winning_tile_makes_minkou ? 1 : 0);
            if (ankous_count == 4) CHECK_EARLY_RETURN(yakuman += 1);
            else if (ankous_count == 3) CHECK_EARLY_RETURN(han += 2);
            // 槓子 (:659-666)
            int kans_count = sup.n_ankans + sup.n_minkans;
            if (kans_count == 4) CHECK_EARLY_RETURN(yakuman += 1);
            else if (kans_count == 3) CHECK_EARLY_RETURN(han += 2);
            // 緑一色 (:668-676)
            auto green = [](u8 k) { return k == 19 || k == 20 || k == 21 || k == 23 || k == 25 || k == T_F; };
            bool has_ryuisou = green(pair_tile);
            for_kotsu_kantsu([&](u8 k) {
                if (!green(k)) has_ryuisou = false;
            });
            for_shuntsu([&](u8 s) {
                if (s != 19) has_ryuisou = false;
            });
            if (has_ryuisou) CHECK_EARLY_RETURN(yakuman += 1);

            if (!has_tanyao) {  // :678-721
                bool has_jihai[7] = {};
                for_kotsu_kantsu([&](u8 k) {
                    if (k >= 27) has_jihai[k - 27] = true;
                });
                if (has_jihai[sup.bakaze - 27]) CHECK_EARLY_RETURN(han += 1);
                if (has_jihai[sup.jikaze - 27]) CHECK_EARLY_RETURN(han += 1);
                int saneins = has_jihai[4] + has_jihai[5] + has_jihai[6];
                if (saneins > 0) {
                    CHECK_EARLY_RETURN(han += saneins);
                    if (saneins == 3) CHECK_EARLY_RETURN(yakuman += 1);
                    else if (saneins == 2 && (pair_tile == T_P || pair_tile == T_F || pair_tile == T_C))
                        CHECK_EARLY_RETURN(han += 2);
                }
                int winds = has_jihai[0] + has_jihai[1] + has_jihai[2] + has_jihai[3];
                if (winds == 4) CHECK_EARLY_RETURN(yakuman += 1);
                else if (winds == 3 && pair_tile >= T_E && pair_tile <= T_N) CHECK_EARLY_RETURN(yakuman += 1);
            }
        }

        if (!has_tanyao) {  // :724-757
            bool has_jihai = false;
            // `.all()` short-circuits: has_jihai only reflects elements visited before the first failure,
            // but it is only read when all elements passed.
            bool all_yao = true;
            auto is_yao = [&](u8 k) {
                if (!all_yao) return;
                u8 kind = k / 9;
                if (kind >= 3) {
                    has_jihai = true;
                } else {
                    u8 num = k % 9;
                    if (!(num == 0 || num == 8)) all_yao = false;
                }
            };
            if (div.has_chitoi) {
                for (int i = 0; i < 7; i++) is_yao(tile14[i]);
            } else {
                for_kotsu_kantsu(is_yao);
                is_yao(pair_tile);
            }
            if (all_yao) {
                if (div.has_chitoi || has_toitoi) {
                    if (has_jihai) CHECK_EARLY_RETURN(han += 2);
                    else CHECK_EARLY_RETURN(yakuman += 1);
                } else {
                    bool jc = true;
                    for_shuntsu([&](u8 s) {
                        u8 num = s % 9;
                        if (!(num == 0 || num == 6)) jc = false;
                    });
                    if (jc) {
                        int n = (has_jihai ? 1 : 2) + (sup.is_menzen ? 1 : 0);
                        CHECK_EARLY_RETURN(han += n);
                    }
                }
            }
        }
        return make_return();
#undef CHECK_EARLY_RETURN
    }
};

}  // namespace

bool AgariCalc::has_yaku() const { return search_yakus_impl(true).has_value(); }
std::optional<Agari> AgariCalc::search_yakus() const { return search_yakus_impl(false); }

std::optional<Agari> AgariCalc::search_yakus_impl(bool return_if_any) const {  // agari.rs:260-288
    MJO_ENSURE(is_menzen == (n_chis == 0 && n_pons == 0 && n_minkans == 0), "is_menzen mismatch");
    if (is_menzen && calc_kokushi(tehai) == -1) {
        Agari a;
        a.is_yakuman = true;
        a.n = 1;
        return a;
    }
    u8 tile14[14];
    u32 key = get_tile14_and_key(tehai, tile14);
    const AgariRec* rec = agari_lookup(key);
    if (!rec) return std::nullopt;
    std::optional<Agari> best;
    for (u32 i = 0; i < rec->n; i++) {
        Div d(rec->div[i]);
        DivWorker w(*this, tile14, d);
        auto r = w.search_yakus(return_if_any);
        if (!r) continue;
        if (return_if_any) return r;
        // Iterator::max returns the last maximum
        if (!best || agari_cmp(*r, *best) >= 0) best = r;
    }
    return best;
}

std::optional<Agari> AgariCalc::agari(int additional_hans, int doras) const {  // agari.rs:228-258
    if (auto a = search_yakus()) {
        if (!a->is_yakuman) a->han = a->han + additional_hans + doras;
        return a;
    }
    if (additional_hans == 0) return std::nullopt;
    if (additional_hans + doras >= 5) {
        Agari a;
        a.fu = 0;
        a.han = additional_hans + doras;
        return a;
    }
    u8 tile14[14];
    u32 key = get_tile14_and_key(tehai, tile14);
    const AgariRec* rec = agari_lookup(key);
    if (!rec) return std::nullopt;
    int fu = -1;
    for (u32 i = 0; i < rec->n; i++) {
        Div d(rec->div[i]);
        DivWorker w(*this, tile14, d);
        fu = std::max<int>(fu, w.calc_fu(false));
    }
    if (fu < 0) return std::nullopt;
    Agari a;
    a.fu = fu;
    a.han = additional_hans + doras;
    return a;
}

bool check_ankan_after_riichi(const u8* tehai, int len_div3, u8 tile, bool strict) {  // agari.rs:854-912
    int tile_id = deaka(tile);
    if (tehai[tile_id] != 4) return false;
    if (tile_id >= 27) return true;
    u8 before[34];
    memcpy(before, tehai, 34);
    before[tile_id] -= 1;
    for (int t = 0; t < 34; t++) {
        if (before[t] == 4) continue;
        u8 tmp[34];
        memcpy(tmp, before, 34);
        tmp[t] += 1;
        if (calc_all(tmp, len_div3) != -1) continue;
        // t is a wait
        if (t == tile_id) return false;
        u8 after[34];
        memcpy(after, tehai, 34);
        after[tile_id] = 0;
        after[t] += 1;
        u8 t14[14];
        const AgariRec* divs_after = agari_lookup(get_tile14_and_key(after, t14));
        if (!divs_after) return false;
        if (strict) {
            u8 tb[34];
            memcpy(tb, before, 34);
            tb[t] += 1;
            const AgariRec* divs_before = agari_lookup(get_tile14_and_key(tb, t14));
            MJO_ENSURE(divs_before, "invalid riichi detected when testing ankan after riichi");
            if (divs_after->n != divs_before->n) return false;
        }
    }
    return true;
}

}  // namespace mjo
