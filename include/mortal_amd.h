/* mortal_amd — C-ABI of the MI355X-native batched riichi arena (drop-in for the hot path of Mortal's `libriichi`).
 *
 * Plain C: pointers and sizes only, no torch / C++ types.  Device pointers are raw HIP device addresses
 * (e.g. torch.Tensor.data_ptr()); `stream` is a hipStream_t passed as void* (0 = default stream).
 * Every function returns 0 on success or a negative code; mj_last_error() describes the failure.
 *
 * Each entry point names the reference interface it replaces (paths relative to libriichi/src):
 *
 *   mj_tables_upload      algo/shanten.rs:11-44 + algo/agari.rs:24-51   (lazy-static table loading)
 *   mj_pool_create/reset  arena/game.rs:230-266  BatchGame::run set-up: one Game per (seed, seat plan)
 *                         arena/one_vs_three.rs:140-191  seed list / agent index planning (done by the caller)
 *   mj_pool_configure     agent/mortal.rs:53-74  engine attributes read once per agent
 *   mj_step               arena/game.rs:286-304  one poll/commit cycle over every live game:
 *                           Game::commit (game.rs:180-218) + agent/mortal.rs:292-573 get_reaction (action decode)
 *                           Game::poll   (game.rs:59-178)  + BoardState::poll/step (board.rs:141-161,511-678)
 *                           agent/mortal.rs:200-250 set_scene (quick-eval, kan-select classification)
 *   mj_rows_count         agent/mortal.rs:118-123  batch size of the coming react_batch call
 *   mj_encode             agent/mortal.rs:252-287 -> state/obs_repr.rs:126-630  obs + mask for every row of the batch
 *   mj_random_policy      (no reference counterpart: BASELINE config 2's uniform-random legal policy)
 *   mj_greedy_policy      (no reference counterpart: tenpai-seeking benchmark / test policy)
 *   mj_results            arena/result.rs:19-30 GameResult.scores; arena/one_vs_three.rs:55-60 ranking input
 *   mj_counters           arena/game.rs:298-311 cycles/actions progress counters
 */
#ifndef MORTAL_AMD_H
#define MORTAL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct MjPool MjPool;

enum { MJ_DEAL_RAND08 = 0, MJ_DEAL_RAND09 = 1 };

const char* mj_last_error(void);
int mj_abi_version(void);

/* Upload the lookup tables (payload format: mortal_amd/tables.py, 'MJT1').  Once per process and device. */
int mj_tables_upload(const void* payload, size_t size);

/* n_tables concurrent tables; obs `version` 1..4 (consts.rs:20-28); max_rows = capacity of the row lists per agent
 * (0 -> 8 * n_tables, the theoretical maximum). */
MjPool* mj_pool_create(int n_tables, int version, int deal_algo, int max_rows);
void mj_pool_destroy(MjPool* pool);

/* Start n_tables games.  Host arrays of length n_tables: seed nonce/key (board.rs:99-107), global game id, and
 * agent_of_seat (bit s = agent index 0/1 of seat s).  n_games_total sizes the result arrays (>= max game id + 1). */
int mj_pool_reset(MjPool* pool, const uint64_t* nonces, const uint64_t* keys, const uint32_t* game_ids,
                  const uint8_t* agent_of_seat, int n_games_total);

/* Per-agent engine attributes (agent/mortal.rs:53-74): obs version (0 = keep), quick-eval, agari guard. */
int mj_pool_configure(MjPool* pool, int agent, int version, int enable_quick_eval,
                      int enable_rule_based_agari_guard);
/* Per-table mjai event log (arena/board.rs:189-197 add_log -> arena/result.rs:32-51 dump_json_log), off by default.
 * words_per_table u64 words are reserved per table (a hanchan needs <= ~250 words per kyoku); a table that overflows
 * ends with error MJ_ERR_LOG_OVERFLOW (7).  Word format: mortal_amd/csrc/mj_state.h LG_*; decoded by mortal_amd/mjai_log.py. */
int mj_pool_enable_log(MjPool* pool, uint32_t words_per_table);
int mj_log_lengths(MjPool* pool, uint32_t* len_out /* [n_tables] host */, void* stream);
int mj_log_read(MjPool* pool, int table0, int n, uint64_t* words_out /* [n][words_per_table] host */, void* stream);

/* Steady-state mode for throughput runs: finished tables restart with nonce += stride. 0 disables. */
int mj_pool_set_refill(MjPool* pool, uint64_t nonce_stride);
/* Steady-state throughput mode only (after mj_pool_reset + mj_pool_set_refill, before the first mj_step): instead of all tables
 * starting their hanchan on the same cycle, table t enters play at cycle hash(t) % cycles (it is parked as finished until then
 * and started by the refill path), so a pool is spread over every phase of a hanchan from the start — what a long-running
 * self-play server looks like.  No reference counterpart (BatchGame::run starts all games at once and never restarts one). */
int mj_pool_set_start_stagger(MjPool* pool, uint32_t cycles, void* stream);

/* One arena cycle.  actions_dev[a] = int32 device array with one action id (0..45) per row of agent a's previous
 * batch (NULL on the first cycle or when that agent had no rows). */
int mj_step(MjPool* pool, const int32_t* actions_dev0, const int32_t* actions_dev1, void* stream);
/* Same, with the q-values of each batch (f32 device array [n_rows][46], agent/mortal.rs:126,150).  Needed when an agent
 * was configured with enable_rule_based_agari_guard: an agari (action 43) that PlayerState::rule_based_agari
 * (state/agent_helper.rs:251-368) rejects is replaced by the best other action, exactly as agent/mortal.rs:319-336. */
int mj_step_q(MjPool* pool, const int32_t* actions_dev0, const int32_t* actions_dev1, const float* q_dev0,
              const float* q_dev1, void* stream);

/* The same with EXPLICIT reactions for an agent: ev_devN = one packed mjai event word per row of agent N's batch (the
 * header word of the LG_* log format, 0 = {"type":"none"}), applied verbatim instead of an action id — the reference's
 * MjaiLogBatchAgent (agent/mjai_log.rs:12-150, py_agent.rs:24-37) answers with events.  The host validates them first
 * (state/action.rs:91-228). */
int mj_step_ev(MjPool* pool, const int32_t* actions_dev0, const int32_t* actions_dev1, const float* q_dev0,
               const float* q_dev1, const uint64_t* ev_dev0, const uint64_t* ev_dev1, void* stream);

/* ---- Log replay for the dataset loader (dataset/gameplay.rs:239-443 Gameplay::load_events_by_player).
 * One game log per table, as packed event words (LG_* format, the same words mj_log_read returns; host encoder:
 * mortal_amd/mjai_log.py encode_events).  script = all logs concatenated, off[n_logs + 1] = word offsets, tracked[t] bit s =
 * samples wanted for seat s of log t.  mj_replay_step applies events until some tracked seat has a sample, then
 * mj_rows_count / mj_encode(agent 0) / mj_encode_oracle deliver obs + masks exactly as in the arena, and mj_replay_meta the
 * per-row int32[8] = {label, log, seat, kyoku index, turn, shanten, is kan-select row, event index}.
 * counters()[1] = logs fully replayed.
 * For the invisible obs (GameplayLoader(oracle=True), dataset/invisible.rs) a start_kyoku word may carry the whole wall
 * (LG_SK_WALL_BIT) or ask for it to be rebuilt from the table's seed (LG_SK_DEAL_BIT, nonces/keys given; trust_seed);
 * in replay mode mj_encode_oracle lists every undrawn yama tile like Invisible::encode. */
int mj_replay_load(MjPool* pool, const uint64_t* script_host, const uint32_t* off_host, const uint8_t* tracked_host,
                   int n_logs, int always_include_kan_select, const uint64_t* nonces_host /* NULL ok */,
                   const uint64_t* keys_host /* NULL ok */);
int mj_replay_step(MjPool* pool, void* stream);
int mj_replay_meta(MjPool* pool, int32_t* meta_dev, void* stream);

/* ---- Single-table access behind libriichi.state.PlayerState (state/player_state.rs:142-167, tests / debugging):
 * apply one event (LG_* words, "?" tiles = 37 allowed for hidden hands), make (table, seat) the only policy row so that
 * mj_rows_count + mj_encode return its obs/mask (state/obs_repr.rs:776-791 encode_obs), and a few queries:
 * what = 0 agari_points(args: is_ron, n_ura, ura[5]) -> ok, ron, tsumo_ko, tsumo_oya | 1 rule_based_agari |
 *        2 real_time_shanten | 3 doras_owned[4] | 4 add_dora_indicator(args[0]) | 5 set scores(args[0..3]) |
 *        6 set_scene (agent/mortal.rs:200-250) | 7 action id -> event word (agent/mortal.rs:338-573) |
 *        8 the step kernel's check of an explicit reaction word (args: word lo, hi; state/action.rs:91-228 plus the seat
 *          the call is made on) -> error code (0 = accepted), reaction type; the table is left untouched. */
int mj_table_apply_event(MjPool* pool, int table, const uint64_t* words_host, int n_words, void* stream);
int mj_table_mark_row(MjPool* pool, int table, int seat, int at_kan_select, void* stream);
int mj_table_query(MjPool* pool, int table, int seat, int what, const int32_t* args8_host, int32_t* out8_host, void* stream);

/* Number of policy rows per agent produced by the last mj_step.  Waits for the row counts of THAT step (recorded on the stream the
 * step was launched on; `stream` is only used before any step has run) -- it does NOT drain the stream: the snapshot kernel queued
 * behind the counts may still be running when the call returns.  mj_encode / mj_encode_oracle order themselves behind it: on the
 * step's stream by stream order, on any other stream by waiting for the snapshot's event. */
int mj_rows_count(MjPool* pool, int32_t n_rows_out[2], void* stream);
/* Device array of row descriptors of agent a: table | seat << 28 | is_kan_select << 31. */
const uint32_t* mj_rows_dev(MjPool* pool, int agent);

/* Encode agent a's rows: obs_dev [n_rows][C][34] f32, masks_dev [n_rows][46] u8 (bool). */
int mj_encode(MjPool* pool, int agent, float* obs_dev, uint8_t* masks_dev, void* stream);
/* Invisible ("oracle") observation of agent a's rows, for engines with is_oracle=True (arena/game.rs:100-101 ->
 * BoardState::encode_oracle_obs, arena/board.rs:679-782): out_dev [n_rows][mj_oracle_obs_rows(version)][34] f32. */
int mj_encode_oracle(MjPool* pool, int agent, float* out_dev, void* stream);
/* consts.rs:32-38 oracle_obs_shape(version).0: 211 for v1, 217 for v2..v4, -1 otherwise. */
int mj_oracle_obs_rows(int version);
/* Average duration (ms) of the encode kernel launches timed with HIP events since the last call, and their count. */
int mj_encode_timing(MjPool* pool, int enable, double* total_ms_out, int64_t* launches_out);
/* Same for the SP-table kernel (obs v4 rows 889..1011) launched by mj_encode; collected while encode timing is enabled. */
int mj_sp_timing(MjPool* pool, double* total_ms_out, int64_t* launches_out);
/* Cumulative phase timers of mj_k_sp since the pool was created (workgroup wall-clock ticks at 100 MHz, summed over all
 * workgroups): out[0] hash/pool overflows, [1] rows, [2] set-up, [3] expansion, [4] level 0 (probe + scoring + sum),
 * [5] evaluation of levels > 0, [6] writing the rows, [7] states visited.  Measurement only (bench.py `sp_phases`). */
int mj_sp_phase_ticks(MjPool* pool, uint64_t* out8, void* stream);
/* Small-pool schedule of the SP kernel (round 6).  The reference spreads the decisions of a batch over every core and pins nothing
 * (agent/mortal.rs:252-287: rayon par_iter over the batch's states); here a decision row is pinned to one 256-thread workgroup of
 * mj_k_sp, and with few rows per launch the launch lasts as long as its heaviest row.  With the schedule on, a workgroup that finds a
 * row's state graph large (the level it is about to expand has >= min_level1 / min_level2 states) parks the row and a 1,024-thread
 * workgroup of the kernel mj_k_sp_wide [wide_grid of them, one per CU, running beside mj_k_sp] finishes it; results are bit-identical either way.
 * In auto mode the schedule switches itself off for a pool whose two kernels turn out not to overlap (more concurrent streams than hardware
 * queues: the wide workgroups give up after 50 ms, the sweep launch still finishes every parked row, one line on stderr).
 * mode: -1 auto (launches of at most max_rows rows; the default, also settable through MJ_SP_WIDE / MJ_SP_WIDE_MAX_ROWS /
 * MJ_SP_WIDE_GRID / MJ_SP_PROMO_MIN1 / _MIN2), 0 never, 1 every launch.  Arguments <= 0 keep the current value (mode: < -1).
 * Call it before the pool's first obs-v4 mj_encode (the spare work areas are sized then); afterwards only mode 0 / the thresholds change. */
int mj_pool_set_sp_schedule(MjPool* pool, int mode, int max_rows, int wide_grid, int min_level1, int min_level2);
/* out[0] launches that ran both kernels, [1] rows parked and finished by mj_k_sp_wide, [2] rows the sweep launch had to take (the two
 * kernels did not overlap), [3] wide workgroups that gave up waiting (same), since the pool was created.  Measurement only. */
int mj_sp_schedule_stats(MjPool* pool, uint64_t* out4, void* stream);

/* Uniform-random legal action per row, counter-based (seed, game id, seat, kan flag, cycle). */
int mj_random_policy(MjPool* pool, int agent, const uint8_t* masks_dev, uint64_t seed, uint64_t cycle,
                     int32_t* actions_dev, void* stream);

/* Tenpai-seeking policy per row (always agari, mostly riichi, shanten-lowering discards read from the encoded obs' discard
 * block, occasional calls), counter-based like mj_random_policy but keyed by the table index.  No reference counterpart:
 * the benchmark's realistic-hand workload and the parity tests' policy (tests/parity_util.py greedy_actions). */
int mj_greedy_policy(MjPool* pool, int agent, const uint8_t* masks_dev, const float* obs_dev, uint64_t seed, uint64_t cycle,
                     int32_t* actions_dev, void* stream);

/* counters: [0] env steps (live tables summed over cycles), [1] games finished, [2] tables in error,
 *           [3] decisions, [4] quick-eval decisions, [5] cycles, [6] SP hash-set overflows (must stay 0) */
int mj_counters(MjPool* pool, uint64_t out[8], void* stream);
/* Final scores [n_games_total][4] and done flags (0 running, 1 finished, 2 aborted on error) to host memory. */
int mj_results(MjPool* pool, int32_t* scores_out, uint8_t* done_out, void* stream);
/* First table in error: returns its error code (>0) and index, or 0. */
int mj_pool_first_error(MjPool* pool, int* table_out, void* stream);

/* Debug/test: copy one table's state (struct TableOne, mortal_amd/csrc/mj_state.h) to host memory. */
int mj_debug_table(MjPool* pool, int table, void* out, size_t out_size, void* stream);
size_t mj_debug_table_size(void);
/* "name:elem_size:count:offset;..." describing the struct returned by mj_debug_table. */
const char* mj_debug_layout(void);
int mj_obs_rows(int version);

/* Debug/test: the pure rule functions of the device code applied to explicit inputs, one device thread per query (no pool
 * needed, only mj_tables_upload).  This is how the reference's own known-answer tests reach the HIP code
 * (algo/shanten.rs:158-201, algo/agari.rs:920-1379, algo/point.rs:121-153; tests/test_gpu_kats.py).
 *   op 0 calc_shanten(tehai, len_div3)                       -> r0 = shanten            (algo/shanten.rs:139-150)
 *   op 1 AgariCalculator::search_yakus                        -> r0 = kind (0 none, 1 fu/han, 2 yakuman), r1 = fu, r2 = han or
 *                                                                yakuman count           (algo/agari.rs:260-288)
 *   op 2 AgariCalculator::has_yaku                            -> r0 = 0 / 1
 *   op 3 AgariCalculator::agari(additional_hans, doras)       -> r0..r2 as op 1; r3 = 1   (algo/agari.rs:228-258)
 *        followed by Agari::point(arg0 = is_oya)              -> p0 = ron, p1 = tsumo_ko, p2 = tsumo_oya
 *   op 4 check_ankan_after_riichi(tehai incl. the drawn tile, len_div3, arg0 = tile), non-strict -> r0 (algo/agari.rs:854-912)
 *   op 5 Point::calc(arg0 = is_oya, arg1 = fu, arg2 = han)    -> p0..p2                   (algo/point.rs:13-112)
 *   op 6 the wall shuffle's u32 division (x = tehai[0..3] little endian, n = arg0 in 1..136) -> r0 = x / n, r1 = x % n
 *        (the `chunk % next_n`, `chunk /= next_n` steps of rand 0.9.1's IncreasingUniform behind arena/board.rs:107-109)
 * Queries and results are host arrays. */
typedef struct MjAlgoQuery {
    uint8_t tehai[34];                                  /* tile counts, red fives counted as fives (34-tile form) */
    uint8_t chis[4], pons[4], minkans[4], ankans[4];    /* deaka'd tile ids (lowest tile of a chi) */
    uint8_t n_chis, n_pons, n_minkans, n_ankans;
    uint8_t len_div3, is_menzen, bakaze, jikaze, winning_tile, is_ron, additional_hans, doras;
    uint8_t op, arg0, arg1, arg2;
    uint8_t pad[6];
} MjAlgoQuery;                                          /* 72 bytes */
typedef struct MjAlgoResult {
    int32_t r0, r1, r2, r3, p0, p1, p2, p3;
} MjAlgoResult;
int mj_algo_query(const MjAlgoQuery* queries_host, int n, MjAlgoResult* results_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif
