// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement (C++17, single table, clarity over speed) of the reference's
// batched self-play hot path (Equim-chan/Mortal `libriichi`).  It exists to be
// the parity checker for the HIP path in mortal_amd/csrc and the `cpu_baseline`
// leg of bench.py.  Nothing under mortal_amd/ may include, link or call it.
//
// Parity status: pinned against the reference's own KATs / fixtures (see
// tests/test_oracle_*.py and tests/golden/): shanten (shanten.rs:158-201),
// agari (agari.rs:920-1379), point (point.rs:121-153), state scenarios
// (state/test.rs), SP calc (sp/calc.rs:773-1007) and the seeded 3-kyoku game log
// (log-viewer/index.example.html:10-264, rand-0.8 shuffle).  "Parity unpinned":
// obs tensor values, whole-hanchan trajectories and the rand-0.9.1 deal — the
// reference has no vectors for them and cannot be built here (no rustc).
//
// Every function cites the reference file:line it follows (paths relative to
// /root/reference/libriichi/src).
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

namespace mjo {

typedef uint8_t u8;
typedef int8_t i8;
typedef uint32_t u32;
typedef uint64_t u64;

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};
#define MJO_ENSURE(cond, msg)                                   \
    do {                                                        \
        if (!(cond)) throw ::mjo::Error(std::string("") + msg); \
    } while (0)

// ---------------------------------------------------------------- tiles
// macros.rs:9-128: 0-8 m, 9-17 p, 18-26 s, 27-33 ESWNPFC, 34-36 aka 5m/5p/5s, 37 '?'
enum : u8 { T_5M = 4, T_5P = 13, T_5S = 22, T_E = 27, T_S = 28, T_W = 29, T_N = 30, T_P = 31, T_F = 32, T_C = 33, T_5MR = 34, T_5PR = 35, T_5SR = 36, T_UNK = 37 };

inline u8 deaka(u8 t) { return t == T_5MR ? (u8)T_5M : t == T_5PR ? (u8)T_5P : t == T_5SR ? (u8)T_5S : t; }       // tile.rs:68-77
inline u8 akaize(u8 t) { return t == T_5M ? (u8)T_5MR : t == T_5P ? (u8)T_5PR : t == T_5S ? (u8)T_5SR : t; }      // tile.rs:79-88
inline bool is_aka(u8 t) { return t >= T_5MR && t <= T_5SR; }                                          // tile.rs:90-94
inline bool is_jihai(u8 t) { return t >= T_E && t <= T_C; }                                            // tile.rs:96-100
inline bool is_yaokyuu(u8 t) {                                                                         // tile.rs:102-109
    if (t >= 34) return false;
    if (t >= 27) return true;
    return t % 9 == 0 || t % 9 == 8;
}
inline bool is_unknown(u8 t) { return t >= T_UNK; }
u8 tile_next(u8 t);  // tile.rs:117-132
u8 tile_prev(u8 t);  // tile.rs:134-150
int cmp_discard_priority(u8 l, u8 r);  // tile.rs:169-177 (returns -1/0/1)
std::string tile_name(u8 t);
int tile_from_name(const std::string& s);  // -1 if invalid

// ---------------------------------------------------------------- tables
struct AgariRec {
    u32 key;
    u32 n;
    u32 div[4];
};
struct Tables {
    const u8* suhai = nullptr;  // n_suhai x 5 packed nibbles
    u32 n_suhai = 0;
    const u8* jihai = nullptr;
    u32 n_jihai = 0;
    const AgariRec* agari = nullptr;  // sorted by key
    u32 n_agari = 0;
};
extern Tables g_tables;
void set_tables(const u8* payload, size_t size);  // payload format: tools/build_tables.py

// ---------------------------------------------------------------- algo
int calc_normal(const u8* tehai, int len_div3);   // shanten.rs:88-102
int calc_chitoi(const u8* tehai);                 // shanten.rs:104-118
int calc_kokushi(const u8* tehai);                // shanten.rs:120-137
int calc_all(const u8* tehai, int len_div3);      // shanten.rs:139-150

struct Point {  // point.rs:1-6
    int ron = 0, tsumo_ko = 0, tsumo_oya = 0;
    int tsumo_total(bool is_oya) const { return is_oya ? tsumo_ko * 3 : tsumo_ko * 2 + tsumo_oya; }  // point.rs:105-112
};
Point point_calc(bool is_oya, int fu, int han);   // point.rs:13-85
Point point_yakuman(bool is_oya, int count);      // point.rs:87-103

struct Agari {  // agari.rs:66-74
    bool is_yakuman = false;
    u8 fu = 0, han = 0;  // Normal
    u8 n = 0;            // Yakuman(n)
    Point point(bool is_oya) const { return is_yakuman ? point_yakuman(is_oya, n) : point_calc(is_oya, fu, han); }
};
int agari_cmp(const Agari& l, const Agari& r);  // agari.rs:180-195

struct AgariCalc {  // agari.rs:76-98
    const u8* tehai;  // 34, must include the winning tile
    bool is_menzen;
    const u8* chis; int n_chis;
    const u8* pons; int n_pons;
    const u8* minkans; int n_minkans;
    const u8* ankans; int n_ankans;
    u8 bakaze, jikaze;
    u8 winning_tile;  // deaka'd
    bool is_ron;

    bool has_yaku() const;                                       // agari.rs:207-211
    std::optional<Agari> search_yakus() const;                   // agari.rs:213-217
    std::optional<Agari> agari(int additional_hans, int doras) const;  // agari.rs:228-258
    std::optional<Agari> search_yakus_impl(bool return_if_any) const;  // agari.rs:260-288
};
bool check_ankan_after_riichi(const u8* tehai, int len_div3, u8 tile, bool strict);  // agari.rs:854-912
u32 get_tile14_and_key(const u8* tiles, u8 tile14[14]);                               // agari.rs:767-838
const AgariRec* agari_lookup(u32 key);

// ---------------------------------------------------------------- events (mjai/event.rs:20-120)
enum EvType : u8 {
    EV_NONE = 0, EV_START_GAME, EV_START_KYOKU, EV_TSUMO, EV_DAHAI, EV_CHI, EV_PON, EV_DAIMINKAN, EV_KAKAN,
    EV_ANKAN, EV_DORA, EV_REACH, EV_REACH_ACCEPTED, EV_HORA, EV_RYUKYOKU, EV_END_KYOKU, EV_END_GAME
};
struct Event {
    u8 type = EV_NONE;
    u8 actor = 0, target = 0;
    u8 pai = T_UNK;
    u8 consumed[4] = {T_UNK, T_UNK, T_UNK, T_UNK};
    bool tsumogiri = false;
    // StartKyoku
    u8 bakaze = T_E, dora_marker = T_UNK, kyoku = 1, honba = 0, kyotaku = 0, oya = 0;
    int scores[4] = {0, 0, 0, 0};
    u8 tehais[4][13];
    // Hora / Ryukyoku
    bool has_deltas = false;
    int deltas[4] = {0, 0, 0, 0};
    u8 ura_markers[5];
    int n_ura = -1;  // -1 = None
    Event() {
        memset(tehais, T_UNK, sizeof tehais);
        memset(ura_markers, T_UNK, sizeof ura_markers);
    }
    bool has_actor() const {  // event.rs:166-181
        switch (type) {
            case EV_TSUMO: case EV_DAHAI: case EV_CHI: case EV_PON: case EV_DAIMINKAN: case EV_KAKAN:
            case EV_ANKAN: case EV_REACH: case EV_REACH_ACCEPTED: case EV_HORA: return true;
            default: return false;
        }
    }
    bool is_in_game_announce() const { return type == EV_REACH_ACCEPTED || type == EV_DORA || type == EV_HORA; }  // event.rs:185-190
};

// ---------------------------------------------------------------- state
struct ActionCandidate {  // state/action.rs:13-41
    bool can_discard = false, can_chi_low = false, can_chi_mid = false, can_chi_high = false, can_pon = false,
         can_daiminkan = false, can_kakan = false, can_ankan = false, can_riichi = false, can_tsumo_agari = false,
         can_ron_agari = false, can_ryukyoku = false;
    u8 target_actor = 0;
    bool can_chi() const { return can_chi_low || can_chi_mid || can_chi_high; }
    bool can_kan() const { return can_daiminkan || can_kakan || can_ankan; }
    bool can_agari() const { return can_tsumo_agari || can_ron_agari; }
    bool can_pass() const { return can_chi() || can_pon || can_daiminkan || can_ron_agari; }
    bool can_act() const { return can_discard || can_chi() || can_pon || can_kan() || can_riichi || can_agari() || can_ryukyoku; }
};

struct Sutehai {  // state/item.rs:14-21
    u8 tile = T_UNK;
    bool is_dora = false, is_tedashi = false, is_riichi = false;
};
struct ChiPon {  // state/item.rs:23-27
    u8 consumed[2];
    u8 target_tile;
};
struct KawaItem {  // state/item.rs:7-12
    std::optional<ChiPon> chi_pon;
    std::vector<u8> kan;
    Sutehai sutehai;
};

struct SPCandidate;  // sp

struct PlayerState {  // state/player_state.rs:24-140
    u8 player_id = 0;
    u8 tehai[34] = {};
    bool waits[34] = {};
    u8 dora_factor[34] = {};
    u8 tiles_seen[34] = {};
    bool akas_seen[3] = {};
    bool keep_shanten_discards[34] = {};
    bool next_shanten_discards[34] = {};
    bool forbidden_tiles[34] = {};
    bool discarded_tiles[34] = {};
    u8 bakaze = T_UNK, jikaze = T_UNK;
    u8 kyoku = 0, honba = 0, kyotaku = 0;
    int scores[4] = {};
    u8 rank = 0, oya = 0;
    bool is_all_last = false;
    std::vector<u8> dora_indicators;
    std::vector<std::optional<KawaItem>> kawa[4];
    std::optional<Sutehai> last_tedashis[4];
    std::optional<Sutehai> riichi_sutehais[4];
    std::vector<u8> kawa_overview[4];
    std::vector<std::vector<u8>> fuuro_overview[4];
    std::vector<u8> ankan_overview[4];
    bool riichi_declared[4] = {};
    bool riichi_accepted[4] = {};
    u8 at_turn = 0, tiles_left = 0;
    std::vector<u8> intermediate_kan;
    std::optional<ChiPon> intermediate_chi_pon;
    i8 shanten = 0;
    std::optional<u8> last_self_tsumo;
    std::optional<u8> last_kawa_tile;
    ActionCandidate last_cans;
    std::vector<u8> ankan_candidates, kakan_candidates;
    bool chankan_chance = false;
    bool can_w_riichi = false, is_w_riichi = false, at_rinshan = false, at_ippatsu = false, at_furiten = false;
    bool to_mark_same_cycle_furiten = false;
    u8 kans_on_board = 0;
    bool is_menzen = false;
    std::vector<u8> chis, pons, minkans, ankans;
    u8 doras_owned[4] = {};
    u8 doras_seen = 0;
    bool akas_in_hand[3] = {};
    u8 tehai_len_div3 = 0;
    bool has_next_shanten_discard = false;

    explicit PlayerState(u8 id = 0) : player_id(id) {}

    // state/update.rs
    ActionCandidate update(const Event& ev, bool keep_cans_on_announce = false);  // :22-122
    int rel(u8 actor) const { return (actor + 4 - player_id) % 4; }              // :688-690
    void witness_tile(u8 tile);                                                   // :695-726
    enum MoveType { MV_TSUMO, MV_DISCARD, MV_FUURO_CONSUME };
    void move_tile(u8 tile, MoveType mt);                                         // :733-775
    void add_dora_indicator(u8 tile);                                             // :780-808
    void set_can_chi_from_tile(u8 tile);                                          // :826-868
    void update_shanten();                                                        // :875-878
    void update_shanten_discards();                                               // :881-912
    void update_waits_and_furiten();                                              // :916-953
    void update_doras_owned(int actor_rel, u8 tile);                              // :955-960
    void update_rank() { rank = get_rank(scores); }                               // :962-964
    u8 get_rank(const int scores_rel[4]) const;                                   // :966-972
    // state/action.rs:93-228
    void validate_reaction(const Event& action) const;
    // state/agent_helper.rs
    int kans_count() const { return (int)(minkans.size() + ankans.size()); }     // :16-18
    void discard_candidates_aka(bool out[37]) const;                              // :35-79
    void discard_candidates_with_unconditional_tenpai(bool out[34]) const;        // :88-97
    void discard_candidates_with_unconditional_tenpai_aka(bool out[37]) const;    // :100-197
    int yaokyuu_kind_count() const;                                               // :201-206
    bool rule_based_agari() const;                                                // :251-368
    Point agari_points(bool is_ron, const u8* ura, int n_ura) const;              // :377-462 (throws on error)
    int real_time_shanten() const;                                                // :467-503
    std::vector<SPCandidate> single_player_tables() const;                        // :509-593 (throws on error)
    bool is_oya() const { return oya == 0; }
    // state/obs_repr.rs:126-630.  obs: rows*34 floats (zero-filled by callee), mask: 46 bytes
    void encode_obs(int version, bool at_kan_select, float* obs, u8* mask) const;
    std::string brief() const;

  private:
    void ev_start_kyoku(const Event& ev);
    void ev_tsumo(u8 actor, u8 pai);
    void ev_dahai(u8 actor, u8 pai, bool tsumogiri);
    void ev_chi(u8 actor, u8 pai, const u8 consumed[2]);
    void ev_pon(u8 actor, u8 target, u8 pai, const u8 consumed[2]);
    void ev_daiminkan(u8 actor, u8 target, u8 pai, const u8 consumed[3]);
    void ev_kakan(u8 actor, u8 pai);
    void ev_ankan(u8 actor, const u8 consumed[4]);
    void ev_reach(u8 actor);
    void ev_reach_accepted(u8 actor);
    void pad_kawa_for_pon_or_daiminkan(u8 abs_actor, u8 abs_target);  // :810-817
    void pad_kawa_at_start();                                         // :819-824
    bool rule_based_agari_slow(bool is_ron, int target_rel) const;
    void ensure_tiles_in_hand(const u8* tiles, int n) const;
};

int obs_rows(int version);  // consts.rs:20-28

// ---------------------------------------------------------------- SP (algo/sp/*)
constexpr int MAX_TSUMOS_LEFT = 17;  // sp/mod.rs:40
struct RequiredTile {
    u8 tile;
    u8 count;
};
struct SPCandidate {  // sp/candidate.rs:9-23
    u8 tile = T_UNK;
    std::vector<float> tenpai_probs, win_probs, exp_values;
    std::vector<RequiredTile> required_tiles;
    u8 num_required_tiles = 0;
    bool shanten_down = false;
};
enum SPColumn { COL_EV, COL_WIN_PROB, COL_TENPAI_PROB, COL_NOT_SHANTEN_DOWN, COL_NUM_REQUIRED, COL_DISCARD_PRIORITY };
int sp_candidate_cmp(const SPCandidate& l, const SPCandidate& r, SPColumn by);  // sp/candidate.rs:73-106
struct SPInitState {  // sp/state.rs:22-31
    u8 tehai[34];
    bool akas_in_hand[3];
    u8 tiles_seen[34];
    bool akas_seen[3];
};
struct SPCalculator {  // sp/calc.rs:36-62
    u8 tehai_len_div3;
    const u8* chis; int n_chis;
    const u8* pons; int n_pons;
    const u8* minkans; int n_minkans;
    const u8* ankans; int n_ankans;
    u8 bakaze, jikaze;
    bool is_menzen;
    u8 num_doras_in_fuuro;
    const u8* dora_indicators; int n_dora_indicators;
    bool calc_double_riichi, calc_haitei, prefer_riichi, sort_result;
    bool maximize_win_prob, calc_tegawari, calc_shanten_down;
    std::vector<SPCandidate> calc(const SPInitState& init, bool can_discard, int tsumos_left, int cur_shanten) const;  // calc.rs:84-133
};

// ---------------------------------------------------------------- deal (arena/board.rs:99-123, 786-824)
enum DealAlgo { DEAL_RAND08 = 0, DEAL_RAND09 = 1 };
void sha3_256(const u8* data, size_t len, u8 out[32]);
struct ChaCha12 {
    u32 key[8];
    u64 counter = 0;
    u32 buf[16];
    int idx = 16;
    explicit ChaCha12(const u8 seed[32]);
    u32 next_u32();
};
void deal_from_seed(u64 nonce, u64 key, u8 kyoku, u8 honba, DealAlgo algo, u8 seq[136]);

// ---------------------------------------------------------------- board / game (arena/board.rs, arena/game.rs)
struct Board {  // board.rs:30-48
    u8 kyoku = 0, honba = 0, kyotaku = 0;
    int scores[4] = {25000, 25000, 25000, 25000};
    u8 haipai[4][13];
    std::vector<u8> yama, rinshan, dora_indicators, ura_indicators;
    void init_from_seq(const u8 seq[136]);  // board.rs:111-122
};
enum Poll { POLL_IN_GAME, POLL_END };
struct KyokuResult {  // result.rs:8-16
    u8 kyoku;
    bool can_renchan, has_hora, has_abortive_ryukyoku;
    u8 kyotaku_left;
    int scores[4];
};
struct BoardState {  // board.rs:52-85
    Board board;
    u8 oya = 0;
    PlayerState player_states[4];
    bool can_renchan = false, has_hora = false, has_abortive_ryukyoku = false;
    int kyoku_deltas[4] = {};
    u8 tiles_left = 70;
    u8 tsumo_actor = 0;
    bool deal_from_rinshan = false, need_new_dora_at_discard = false, need_new_dora_at_tsumo = false;
    std::optional<u8> riichi_to_be_accepted;
    bool can_nagashi_mangan[4] = {true, true, true, true};
    bool can_four_wind = true;
    std::optional<u8> four_wind_tile;
    u8 accepted_riichis = 0, kans = 0;
    bool check_four_kan = false;
    std::optional<u8> paos[4];
    std::vector<Event> log;
    std::vector<u8> dora_indicators_full;          // board.rs:84,127 (copy taken before any indicator is popped)

    explicit BoardState(const Board& b);           // board.rs:125-136
    // invisible ("oracle") observation of seat `perspective`: out = zeroed-and-filled [oracle_obs_rows(version)][34]
    void encode_oracle_obs(u8 perspective, int version, float* out) const;  // board.rs:679-782
    static int oracle_obs_rows(int version) { return version == 1 ? 211 : 217; }  // consts.rs:32-38
    Poll poll(std::array<Event, 4> reactions);     // board.rs:141-161
    Poll step(const std::array<Event, 4>& reactions);  // board.rs:511-678
    KyokuResult end() const;                       // board.rs:172-182

  private:
    void broadcast(const Event& ev);               // board.rs:199-204
    void haipai();                                 // board.rs:206-239
    void exhaustive_ryukyoku();                    // board.rs:241-294
    void update_nagashi_mangan_and_four_wind(const Event& ev);  // board.rs:296-312
    bool check_four_wind(u8 pai);                  // board.rs:314-340
    void check_riichi_accepted();                  // board.rs:342-351
    void add_new_dora();                           // board.rs:353-364
    void handle_hora(u8 actor, u8 target, const std::array<Event, 4>& reactions);  // board.rs:366-471
    void update_paos(const Event& ev);             // board.rs:473-499
    void abortive_ryukyoku();                      // board.rs:502-509
};

// The reference's MortalBatchAgent glue (agent/mortal.rs:200-250, 292-573) as pure functions.
struct SceneInfo {
    bool can_act = false;
    bool quick_eval = false;     // single legal discard & nothing else (mortal.rs:210-242)
    Event quick_event;           // valid if quick_eval
    bool need_kan_select = false;  // mortal.rs:244-250
};
SceneInfo agent_scene(const PlayerState& st, u8 actor, bool enable_quick_eval);
// action id (0..45) -> mjai event (mortal.rs:338-573); kan_tile = kan-select choice or -1
Event agent_decode_action(const PlayerState& st, u8 actor, int action, int kan_tile);

struct Game {  // arena/game.rs:28-55
    u8 length = 8;
    u64 seed_nonce = 0, seed_key = 0;
    DealAlgo deal_algo = DEAL_RAND08;
    std::array<Event, 4> last_reactions;
    std::optional<BoardState> board;
    u8 kyoku = 0, honba = 0, kyotaku = 0;
    int scores[4] = {25000, 25000, 25000, 25000};
    std::vector<std::vector<Event>> game_log;
    bool kyoku_started = false, ended = false, in_renchan = false;
    bool keep_log = true;

    void poll();                 // game.rs:59-178 (without the agent calls)
    // game.rs:180-218: returns true when the game has just been finalised (scores final)
    bool commit_end();
};

}  // namespace mjo
