// SoA table pool for the batched riichi arena (MI355X / gfx950).
//
// Layout: tables are grouped in blocks of 64 (one wavefront); inside a block every field is an
// array whose LAST dimension is the lane (`field[...][64]`), so a wave touching the same field of
// its 64 tables issues one coalesced 64/128/256-byte access.  One table == one lane of the step
// kernel; the encode kernel reads a table's fields through the same accessors.
//
// What is stored ONCE per table (the reference keeps four PlayerState copies of it,
// state/player_state.rs:24-140): kawa, fuuro/ankan overview, dora indicators, riichi flags,
// public tile counts.  Per seat only the private slice is kept: hand counts, aka flags, waits,
// shanten sets, furiten/ippatsu/rinshan flags, own meld lists, last_cans.
// Derived on demand instead of stored (proved equal to the reference's incremental values in
// DESIGN.md §state): tiles_seen = pub_seen + tehai, dora_factor, doras_owned, doras_seen,
// kawa_overview (= the Some() entries of kawa), rank, scores/kyotaku per perspective.
#pragma once
#include <stdint.h>

#define MJ_LANES 64
#define MJ_KAWA_MAX 40   // reference kawa is unbounded (None padding, player_state.rs:74-80); overflow sets err
#define MJ_NONE 0xFF

// last_cans bits (state/action.rs:13-41)
enum {
    CAN_DISCARD = 1 << 0, CAN_CHI_LOW = 1 << 1, CAN_CHI_MID = 1 << 2, CAN_CHI_HIGH = 1 << 3, CAN_PON = 1 << 4,
    CAN_DAIMINKAN = 1 << 5, CAN_KAKAN = 1 << 6, CAN_ANKAN = 1 << 7, CAN_RIICHI = 1 << 8, CAN_TSUMO_AGARI = 1 << 9,
    CAN_RON_AGARI = 1 << 10, CAN_RYUKYOKU = 1 << 11,
    CAN_CHI = CAN_CHI_LOW | CAN_CHI_MID | CAN_CHI_HIGH,
    CAN_KAN = CAN_DAIMINKAN | CAN_KAKAN | CAN_ANKAN,
    CAN_AGARI = CAN_TSUMO_AGARI | CAN_RON_AGARI,
    CAN_PASS = CAN_CHI | CAN_PON | CAN_DAIMINKAN | CAN_RON_AGARI,
    CAN_ACT = CAN_DISCARD | CAN_CHI | CAN_PON | CAN_KAN | CAN_RIICHI | CAN_AGARI | CAN_RYUKYOKU,
};

// per-seat private flag bits
enum {
    PF_CHANKAN_CHANCE = 1 << 0, PF_CAN_W_RIICHI = 1 << 1, PF_IS_W_RIICHI = 1 << 2, PF_AT_RINSHAN = 1 << 3,
    PF_AT_IPPATSU = 1 << 4, PF_AT_FURITEN = 1 << 5, PF_MARK_FURITEN = 1 << 6, PF_IS_MENZEN = 1 << 7,
};

// table-level flag bits (game + board)
enum {
    TF_KYOKU_STARTED = 1 << 0, TF_ENDED = 1 << 1, TF_IN_RENCHAN = 1 << 2, TF_DONE = 1 << 3,
    TF_CAN_RENCHAN = 1 << 4, TF_HAS_HORA = 1 << 5, TF_HAS_ABORTIVE = 1 << 6, TF_DEAL_FROM_RINSHAN = 1 << 7,
    TF_NEW_DORA_AT_DISCARD = 1 << 8, TF_NEW_DORA_AT_TSUMO = 1 << 9, TF_CAN_FOUR_WIND = 1 << 10,
    TF_CHECK_FOUR_KAN = 1 << 11, TF_NAGASHI0 = 1 << 12 /* ..15: can_nagashi_mangan[4] */, TF_HAIPAI_DONE = 1 << 16,
    TF_INACTIVE = 1 << 17,  // slot beyond n_tables
};

// error codes (sticky per table; mj_pool_errors reports the first)
enum {
    MJ_OK = 0, MJ_ERR_ILLEGAL_ACTION = 1, MJ_ERR_KAWA_OVERFLOW = 2, MJ_ERR_WALL = 3, MJ_ERR_INTERNAL = 4,
    MJ_ERR_NOT_HORA = 5, MJ_ERR_FOUR_WIND = 6, MJ_ERR_LOG_OVERFLOW = 7,
};

// kawa entry (u64):  bit0 valid(Some) | 1..6 tile(0..36) | 7 is_dora | 8 is_tedashi | 9 is_riichi
//   | 10 has_chi_pon | 11..16 cp_min | 17..22 cp_max (deaka'd consumed) | 23..25 n_kan | 26+6k kan tile k (raw id)
// A None padding entry is 0.
#define KW_VALID 1ull
#define KW_TILE(e) ((uint32_t)(((e) >> 1) & 63))
#define KW_DORA(e) (((e) >> 7) & 1)
#define KW_TEDASHI(e) (((e) >> 8) & 1)
#define KW_RIICHI(e) (((e) >> 9) & 1)
#define KW_HAS_CP(e) (((e) >> 10) & 1)
#define KW_CP_MIN(e) ((uint32_t)(((e) >> 11) & 63))
#define KW_CP_MAX(e) ((uint32_t)(((e) >> 17) & 63))
#define KW_NKAN(e) ((uint32_t)(((e) >> 23) & 7))
#define KW_KAN(e, k) ((uint32_t)(((e) >> (26 + 6 * (k))) & 63))

// sutehai byte (last_tedashis / riichi_sutehais): bit7 valid | bit6 is_dora | 0..5 tile
#define SU_VALID 0x80
#define SU_DORA 0x40

// Field list: X(type, name, dims, count) declares `type name dims [LANES]`; count = number of elements per table.
// The same list instantiates the 64-lane pool block and the single-table copy the encode kernel gathers into LDS.
#define MJ_FIELDS(X)                                                                                          \
    /* ---- game level (arena/game.rs:28-55) */                                                               \
    X(uint64_t, seed_nonce, , 1)                                                                              \
    X(uint64_t, seed_key, , 1)                                                                                \
    X(int32_t, scores, [4], 4)           /* == board.scores == every PlayerState.scores (rotated) */          \
    X(uint32_t, flags, , 1)              /* TF_* */                                                           \
    X(uint32_t, game_id, , 1)            /* global game index (seat plan / RNG key / results) */              \
    X(uint8_t, kyoku, , 1)               /* 0 (E1) .. 11 (W4) */                                              \
    X(uint8_t, honba, , 1)                                                                                    \
    X(uint8_t, kyotaku, , 1)                                                                                  \
    X(uint8_t, err, , 1)                                                                                      \
    /* ---- board level (arena/board.rs:30-85) */                                                             \
    X(uint8_t, wall, [136], 136)         /* shuffled sequence; slices per board.rs:111-122 */                 \
    X(uint8_t, yama_n, , 1)              /* pop: wall[66 + --yama_n] */                                       \
    X(uint8_t, rinshan_n, , 1)           /* pop: wall[52 + --rinshan_n] */                                    \
    X(uint8_t, dora_n, , 1)              /* unrevealed indicators; pop: wall[56 + --dora_n]; ura = wall[61..] */ \
    X(uint8_t, tiles_left, , 1)          /* board.tiles_left == PlayerState.tiles_left */                     \
    X(uint8_t, tsumo_actor, , 1)                                                                              \
    X(uint8_t, riichi_to_be_accepted, , 1) /* MJ_NONE */                                                      \
    X(uint8_t, four_wind_tile, , 1)      /* MJ_NONE */                                                        \
    X(uint8_t, accepted_riichis, , 1)                                                                         \
    X(uint8_t, kans, , 1)                                                                                     \
    X(uint8_t, paos, [4], 4)             /* MJ_NONE */                                                        \
    X(int32_t, kyoku_deltas, [4], 4)                                                                          \
    /* ---- public player state, absolute seats */                                                            \
    X(uint8_t, dora_ind, [5], 5)         /* revealed indicators in reveal order */                            \
    X(uint8_t, n_dora_ind, , 1)                                                                               \
    X(uint64_t, kawa, [4][MJ_KAWA_MAX], 4 * MJ_KAWA_MAX)                                                      \
    X(uint8_t, kawa_len, [4], 4)                                                                              \
    X(uint8_t, last_tedashi, [4], 4)     /* sutehai byte */                                                   \
    X(uint8_t, riichi_sutehai, [4], 4)                                                                        \
    X(uint8_t, fuuro, [4][4][4], 64)     /* [seat][set][tile] raw ids, MJ_NONE empty; consumed.., called */   \
    X(uint8_t, fuuro_n, [4], 4)                                                                               \
    X(uint8_t, ankan, [4][4], 16)        /* deaka'd tile of each ankan */                                     \
    X(uint8_t, ankan_n, [4], 4)                                                                               \
    X(uint8_t, riichi_declared, , 1)     /* bit per seat */                                                   \
    X(uint8_t, riichi_accepted, , 1)     /* bit per seat */                                                   \
    X(uint8_t, pub_seen, [34], 34)       /* tiles visible to every seat */                                    \
    X(uint8_t, pub_aka_seen, , 1)        /* 3 bits */                                                         \
    X(uint8_t, kans_on_board, , 1)                                                                            \
    X(uint8_t, inter_kan, [4], 4)        /* intermediate_kan (raw ids) of the seat in turn */                 \
    X(uint8_t, inter_kan_n, , 1)                                                                              \
    X(uint8_t, inter_cp, [3], 3)         /* intermediate_chi_pon: valid, cp_min, cp_max */                    \
    /* ---- private per seat */                                                                               \
    X(uint64_t, hand_mp, [4], 4)         /* 3-bit counts, tiles 0..17 */                                      \
    X(uint64_t, hand_sz, [4], 4)         /* 3-bit counts, tiles 18..33 */                                     \
    X(uint8_t, akas_in_hand, [4], 4)     /* 3 bits */                                                         \
    X(uint64_t, waits, [4], 4)           /* 34-bit sets */                                                    \
    X(uint64_t, keep_shanten, [4], 4)                                                                         \
    X(uint64_t, next_shanten, [4], 4)                                                                         \
    X(uint64_t, forbidden, [4], 4)                                                                            \
    X(uint64_t, discarded, [4], 4)                                                                            \
    X(uint64_t, ankan_cand, [4], 4)                                                                           \
    X(uint64_t, kakan_cand, [4], 4)                                                                           \
    X(int8_t, shanten, [4], 4)                                                                                \
    X(uint8_t, has_next_shanten, [4], 4)                                                                      \
    X(uint8_t, len_div3, [4], 4)                                                                              \
    X(uint8_t, at_turn, [4], 4)                                                                               \
    X(uint8_t, last_self_tsumo, [4], 4)  /* MJ_NONE */                                                        \
    X(uint8_t, last_kawa_tile, [4], 4)   /* MJ_NONE; per seat: kakan sets it for the other three only */      \
    X(uint16_t, cans, [4], 4)                                                                                 \
    X(uint8_t, cans_target, [4], 4)                                                                           \
    X(uint8_t, pflags, [4], 4)           /* PF_* */                                                           \
    X(uint8_t, chis, [4][4], 16)                                                                              \
    X(uint8_t, pons, [4][4], 16)                                                                              \
    X(uint8_t, minkans, [4][4], 16)                                                                           \
    X(uint8_t, ankans, [4][4], 16)                                                                            \
    X(uint8_t, n_melds, [4][4], 16)      /* [seat][0 chi,1 pon,2 minkan,3 ankan] */                           \
    /* ---- decision bookkeeping step -> encode -> step (agent/mortal.rs SyncFields) */                       \
    X(int32_t, main_row, [4], 4)         /* policy row of the seat's decision, -1 none */                     \
    X(int32_t, kan_row, [4], 4)          /* kan-select row, -1 none */                                        \
    X(uint8_t, quick_pai, [4], 4)        /* quick-eval discard, MJ_NONE none */                               \
    X(uint8_t, pending, , 1)             /* bit per seat: has a decision this cycle */                        \
    X(uint8_t, n_rows, [2], 2)           /* rows wanted per agent this cycle */                               \
    X(uint8_t, agent_of_seat, , 1)       /* bit per seat: 0 = agent 0 (challenger), 1 = agent 1 (champion) */

template <int LANES>
struct alignas(16) TableT {
#define MJ_X_DECL(type, name, dims, count) type name dims[LANES];
    MJ_FIELDS(MJ_X_DECL)
#undef MJ_X_DECL
};
typedef TableT<MJ_LANES> TableBlock;  // 64 tables, field-major (the HBM pool)
typedef TableT<1> TableOne;           // one table, gathered into LDS by the encode kernel

struct MjGatherEnt {  // one element to copy from a pool block (lane 0 address) to a TableOne
    uint32_t src_off;
    uint16_t dst_off;
    uint16_t size;
};

// Row descriptor written by the step kernel for the encode kernel: table | seat<<28 | is_kan<<31
#define ROW_PACK(table, seat, kan) ((uint32_t)(table) | ((uint32_t)(seat) << 28) | ((uint32_t)(kan) << 31))
#define ROW_TABLE(r) ((r) & 0x0FFFFFFFu)
#define ROW_SEAT(r) (((r) >> 28) & 3u)
#define ROW_KAN(r) ((r) >> 31)

struct MjTablesDev {
    const uint64_t* suhai;  // 10 nibbles per row in the low 40 bits
    uint32_t n_suhai;
    const uint64_t* jihai;
    uint32_t n_jihai;
    const uint32_t* agari_keys;  // sorted
    const uint64_t* agari_hash;  // open addressing, 2^15 slots: (key << 32) | (index + 1), 0 = empty
    const uint32_t* agari_divs;  // n x 5: n_div, div[4]
    uint32_t n_agari;
};
// ---- optional per-table mjai event log (arena/board.rs:189-197 add_log; result.rs:32-51 dump_json_log).
// One u64 header word per event, followed by payload words for start_kyoku (2 score words, 7 haipai words = 52 tiles,
// 8 per word, seat-major), hora (2 delta words, 1 ura word = 6 bits per indicator) and ryukyoku (2 delta words).
enum MjLogType : uint32_t {
    LG_START_KYOKU = 1, LG_TSUMO, LG_DAHAI, LG_CHI, LG_PON, LG_DAIMINKAN, LG_KAKAN, LG_ANKAN, LG_DORA, LG_REACH,
    LG_REACH_ACCEPTED, LG_HORA, LG_RYUKYOKU, LG_END_KYOKU,
};
#define LG_WORD(type, actor, target, pai, c0, c1, c2, c3, tsumogiri)                                                   \
    ((uint64_t)(type) | ((uint64_t)(actor) << 4) | ((uint64_t)(target) << 6) | ((uint64_t)((pai) & 63) << 8) |         \
     ((uint64_t)((c0) & 63) << 14) | ((uint64_t)((c1) & 63) << 20) | ((uint64_t)((c2) & 63) << 26) |                   \
     ((uint64_t)((c3) & 63) << 32) | ((uint64_t)((tsumogiri) & 1) << 38))
#define LG_NURA_SHIFT 39   /* hora: number of ura indicators (0..5) */
#define LG_TAG_BIT 43      /* the event is an agent's reaction: one tag word follows the header (before other payload):
                              cycle (20 bits) | main row (18) | kan-select row + 1 (18) | shanten + 1 (4) | furiten (1) | bit 63
                              -> the host attaches the per-decision `meta` object (agent/mortal.rs:161-186,575-591) */
#define LG_HONBA_SHIFT 44  /* start_kyoku: honba (8 bits); kyoku (0..11) travels in the c0 field, dora marker in pai */
#define LG_KYOTAKU_SHIFT 52
/* start_kyoku words of a REPLAY script only (dataset loader with oracle=True, dataset/invisible.rs): */
#define LG_SK_AUG_BIT 61   /* with LG_SK_DEAL_BIT: the script's tiles are suit-augmented (manzu <-> pinzu), the seed's wall is not */
#define LG_SK_DEAL_BIT 62  /* rebuild the whole wall from the table's seed (trust_seed) */
#define LG_SK_WALL_BIT 63  /* 17 more payload words follow: the full 136-tile wall, 8 tiles per word */


