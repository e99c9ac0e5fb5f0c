// C-ABI host side (include/mortal_amd.h): pool life-cycle and kernel launches.  One translation unit for the whole
// library; the kernels live in mj_step.hip / mj_replay.hip / mj_encode.hip / mj_sp.hip.
// Host float math below builds bit-exact LUTs: compile with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mortal_amd.h"
#include "mj_step.hip"
#include "mj_replay.hip"
#include "mj_encode.hip"
#include "mj_sp.hip"

static_assert(sizeof(MjAlgoQuery) == 72, "MjAlgoQuery layout");
// include/mortal_amd.h mj_algo_query: one thread per query, the same device functions the step / encode / SP kernels call
__global__ __launch_bounds__(64) void mj_k_algo_query(const MjAlgoQuery* q, int n, MjAlgoResult* out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const MjAlgoQuery Q = q[i];
    MjAlgoResult R = {0, 0, 0, 0, 0, 0, 0, 0};
    Hand h = {0, 0};
    for (int t = 0; t < 34; t++)
        for (int c = 0; c < (int)Q.tehai[t] && c < 4; c++) h.inc(t);
    AgariIn in;
    in.tehai = h;
    in.m.chis = in.m.pons = in.m.minkans = in.m.ankans = 0;
    for (int k = 0; k < 4; k++) {
        melds_put(in.m.chis, k, Q.chis[k]);
        melds_put(in.m.pons, k, Q.pons[k]);
        melds_put(in.m.minkans, k, Q.minkans[k]);
        melds_put(in.m.ankans, k, Q.ankans[k]);
    }
    in.m.n_chis = Q.n_chis;
    in.m.n_pons = Q.n_pons;
    in.m.n_minkans = Q.n_minkans;
    in.m.n_ankans = Q.n_ankans;
    in.is_menzen = Q.is_menzen != 0;
    in.bakaze = Q.bakaze;
    in.jikaze = Q.jikaze;
    in.winning_tile = Q.winning_tile;
    in.is_ron = Q.is_ron != 0;
    auto put = [&](const Agari& a) { R.r0 = a.kind; R.r1 = a.fu; R.r2 = a.han; };
    auto put_point = [&](const Point& p) { R.p0 = p.ron; R.p1 = p.tsumo_ko; R.p2 = p.tsumo_oya; };
    switch (Q.op) {
        case 0: R.r0 = calc_all(c_mj_tables, h, Q.len_div3); break;
        case 1: put(agari_search(c_mj_tables, in, false)); break;
        case 2: R.r0 = agari_search(c_mj_tables, in, true).kind != 0; break;
        case 3: {
            const Agari a = agari_full(c_mj_tables, in, Q.additional_hans, Q.doras);
            put(a);
            R.r3 = 1;
            if (a.kind) put_point(agari_point(a, Q.arg0 != 0));
            break;
        }
        case 4: R.r0 = check_ankan_after_riichi(c_mj_tables, h, Q.len_div3, Q.arg0); break;
        case 5: put_point(point_calc(Q.arg0 != 0, Q.arg1, Q.arg2)); break;
        case 6: {  // the shuffle's division by multiply-high (mj_deal.h: deal_divmod), x = tehai[0..3] little endian, n = arg0
            const u32 x = (u32)Q.tehai[0] | ((u32)Q.tehai[1] << 8) | ((u32)Q.tehai[2] << 16) | ((u32)Q.tehai[3] << 24);
            u32 qq = 0, rr = 0;
            if (Q.arg0 >= 1 && Q.arg0 <= 136) deal_divmod(x, (u32)Q.arg0, qq, rr);
            R.r0 = (int32_t)qq;
            R.r1 = (int32_t)rr;
            break;
        }
        default: R.r3 = -1; break;
    }
    out[i] = R;
}

namespace {

thread_local std::string g_err;
int fail(const std::string& msg) {
    g_err = msg;
    return -1;
}
#define HIP_OK(expr)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct DevTables {
    bool ready = false;
    int device = -1;  // the HIP device the tables (and the __constant__ copy of their pointers) live on
    MjTablesDev dev{};
    MjGatherEnt* gather = nullptr;
    int n_gather = 0;
    int gather_chunk[SNAP_NCH + 1] = {0};  // first gather entry of each record chunk (mj_k_snapshot)
    float *decay = nullptr, *rbf_score = nullptr, *rbf_6 = nullptr, *rbf_12 = nullptr, *rbf_23 = nullptr;
} g_tables;

template <class T> int upload(const std::vector<T>& v, T** out) {
    HIP_OK(hipMalloc(out, v.size() * sizeof(T)));
    HIP_OK(hipMemcpy(*out, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

// obs_repr.rs:79-90 with f32 arithmetic in the reference's order
std::vector<float> build_rbf(int n_max, int cap, int intervals) {
    std::vector<float> out((size_t)n_max * (intervals - 1));
    float interval_size = (float)cap / (float)intervals;
    for (int n = 0; n < n_max; n++)
        for (int i = 1; i < intervals; i++) {
            float x = (float)n;
            float mu = (float)i * interval_size;
            float sigma = interval_size;
            float d = x - mu;
            out[(size_t)n * (intervals - 1) + i - 1] = expf(-(d * d) / (2.f * (sigma * sigma)));
        }
    return out;
}

size_t enc_lds_bytes(int version) {
    int C = enc_rows_written(version);
    int tile_rows = ((C + ENC_PASSES - 1) / ENC_PASSES + 1) & ~1;
    // MJ_ENC_LDS_PAD (measurement only): extra dynamic LDS per workgroup = fewer resident workgroups per CU (round 6: the encoder is FASTER
    // with four than with five or six -- concurrent write streams, not occupancy, limit it)
    static const size_t pad = getenv("MJ_ENC_LDS_PAD") ? (size_t)atoi(getenv("MJ_ENC_LDS_PAD")) : 0;
    return (size_t)tile_rows * 34 * 4 + ((sizeof(TableOne) + 15) & ~(size_t)15) + sizeof(EncDerived) + pad;
}

std::vector<MjGatherEnt> build_gather() {
    std::vector<MjGatherEnt> v;
#define MJ_X_GATHER(type, name, dims, count)                                                         \
    for (int k = 0; k < (count); k++)                                                                \
        v.push_back({(uint32_t)(offsetof(TableBlock, name) + (size_t)k * MJ_LANES * sizeof(type)),   \
                     (uint16_t)(offsetof(TableOne, name) + (size_t)k * sizeof(type)), (uint16_t)sizeof(type)});
    MJ_FIELDS(MJ_X_GATHER)
#undef MJ_X_GATHER
    return v;
}

}  // namespace

struct MjPool {
    int n_tables = 0, n_blocks = 0, deal_algo = 0, max_rows = 0;
    // log replay (dataset loader)
    uint64_t* rp_script = nullptr;
    uint32_t *rp_off = nullptr, *rp_cursor = nullptr, *rp_ev_index = nullptr;
    uint8_t *rp_kyoku = nullptr, *rp_tracked = nullptr;
    int32_t *rp_label = nullptr, *rp_kan_label = nullptr;
    int rp_always_kan = 1;
    bool rp_active = false;    // replay mode: the invisible obs lists every undrawn yama tile
    uint64_t* log = nullptr;   // optional event log [n_tables][log_cap]
    uint32_t* log_len = nullptr;
    uint32_t log_cap = 0;
    int version[2] = {4, 4};  // obs version per agent (engine.version, agent/mortal.rs:57)
    TableBlock* blocks = nullptr;
    uint32_t* rows[2] = {nullptr, nullptr};
    int* n_rows_dev = nullptr;
    int* block_rows = nullptr;
    TableOne* snap = nullptr;
    SpWork* sp_work = nullptr;      // lazily allocated on the first v4 encode: sp_grid areas (one per workgroup of mj_k_sp) + sp_spare spare ones
    int sp_grid = 0;
    int sp_wide_areas = 0;          // work areas of mj_k_sp_wide's own workgroups (= its largest grid)
    int sp_spare = 0;               // spare work areas = promotions per launch (small pools: mj_sp.hip "promotion"; 0 = this pool never promotes)
    int sp_wide_mode = -1;          // -1 auto (launches of at most sp_wide_max_rows rows), 0 never, 1 always
    int sp_wide_max_rows = 20000, sp_wide_grid = 0, sp_promo_min[4] = {0, 0, 0, 0};  // grid / thresholds 0 = by the launch's row count (mj_encode)
    hipStream_t sp_stream2 = nullptr;   // mj_k_sp's stream while mj_k_sp_wide runs on the caller's
    hipEvent_t sp_ev_fork = nullptr, sp_ev_join = nullptr;
    uint64_t sp_hybrid_launches = 0;
    unsigned long long* sp_gaveup_host = nullptr;  // pinned: the device's count of wide workgroups that gave up waiting, copied behind every sweep
    bool sp_wide_off = false;       // the two kernels did not overlap on this system: the schedule switched itself off (see mj_encode)
    bool sp_sched_set = false;      // mj_pool_set_sp_schedule was called: the environment does not override it
    int* sp_queue = nullptr;        // [0] row queue head, [1..8] / [9..16] class counts / cursors of the row sort, [SP_Q_TAIL] head of the tail
    uint32_t* sp_order = nullptr;   // [max_rows] queue position -> row
    uint8_t* sp_cls = nullptr;      // [max_rows] cost class of a row
    unsigned long long* sp_err = nullptr;
    int* enc_flag = nullptr;        // [1] an encoder op list overflowed (reported with the SP overflows)
    int* n_rows_host = nullptr;  // pinned
    hipEvent_t ev_rows = nullptr;  // recorded right after the row counts' copy: mj_rows_count waits for it, not for the snapshot behind it
    hipEvent_t ev_snap = nullptr;  // recorded after mj_k_snapshot: a reader of P->snap on ANOTHER stream than the step's waits for it
    hipStream_t step_stream = nullptr;  // the stream the last mj_step / mj_table_* launched on
    unsigned long long* counters = nullptr;
    int* final_scores = nullptr;
    uint8_t* final_done = nullptr;
    int n_games_total = 0;
    int enable_quick_eval[2] = {1, 1};
    int enable_agari_guard[2] = {0, 0};
    uint64_t refill_stride = 0;
    uint32_t start_stagger = 0;
    uint64_t cycles = 0;
    int last_rows[2] = {0, 0};
    bool rows_valid = false;
    // encode timing
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events, sp_events;
    double timed_ms = 0;
    int64_t timed_launches = 0;
};

extern "C" {

const char* mj_last_error(void) { return g_err.c_str(); }
int mj_abi_version(void) { return 1; }
int mj_obs_rows(int version) { return version == 1 ? 938 : version == 2 ? 942 : version == 3 ? 934 : version == 4 ? 1012 : -1; }
size_t mj_debug_table_size(void) { return sizeof(TableOne); }
int mj_algo_query(const MjAlgoQuery* queries_host, int n, MjAlgoResult* results_host, void* stream) {
    if (!g_tables.ready) return fail("mj_tables_upload has not been called");
    if (n <= 0) return 0;
    if (!queries_host || !results_host) return fail("null query / result buffer");
    struct DevBuf {  // freed on every return path
        void* p = nullptr;
        ~DevBuf() { if (p) hipFree(p); }
    } bq, br;
    hipStream_t s = (hipStream_t)stream;
    HIP_OK(hipMalloc(&bq.p, (size_t)n * sizeof(MjAlgoQuery)));
    HIP_OK(hipMalloc(&br.p, (size_t)n * sizeof(MjAlgoResult)));
    MjAlgoQuery* dq = (MjAlgoQuery*)bq.p;
    MjAlgoResult* dr = (MjAlgoResult*)br.p;
    HIP_OK(hipMemcpyAsync(dq, queries_host, (size_t)n * sizeof(MjAlgoQuery), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(mj_k_algo_query, dim3((n + 63) / 64), dim3(64), 0, s, dq, n, dr);
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpyAsync(results_host, dr, (size_t)n * sizeof(MjAlgoResult), hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    return 0;
}
// "name:elem_size:count:offset;..." of struct TableOne, so a host tool can decode mj_debug_table() generically
const char* mj_debug_layout(void) {
    static std::string s;
    if (s.empty()) {
#define MJ_X_LAYOUT(type, name, dims, count) \
    s += std::string(#name) + ":" + std::to_string(sizeof(type)) + ":" + std::to_string(count) + ":" + \
         std::to_string(offsetof(TableOne, name)) + ";";
        MJ_FIELDS(MJ_X_LAYOUT)
#undef MJ_X_LAYOUT
    }
    return s.c_str();
}

int mj_tables_upload(const void* payload, size_t size) {
    if (g_tables.ready) return 0;
    const uint8_t* p = (const uint8_t*)payload;
    if (size < 16 || memcmp(p, "MJT1", 4) != 0) return fail("bad table payload");
    uint32_t ns, nj, na;
    memcpy(&ns, p + 4, 4);
    memcpy(&nj, p + 8, 4);
    memcpy(&na, p + 12, 4);
    if (size != 16 + (size_t)ns * 5 + (size_t)nj * 5 + (size_t)na * 24) return fail("bad table payload size");
    auto rows = [](const uint8_t* b, uint32_t n) {
        std::vector<uint64_t> v(n);
        for (uint32_t i = 0; i < n; i++) {
            uint64_t r = 0;
            for (int k = 0; k < 5; k++) r |= (uint64_t)b[(size_t)i * 5 + k] << (8 * k);
            v[i] = r;
        }
        return v;
    };
    std::vector<uint64_t> suhai = rows(p + 16, ns), jihai = rows(p + 16 + (size_t)ns * 5, nj);
    const uint8_t* ag = p + 16 + (size_t)ns * 5 + (size_t)nj * 5;
    std::vector<uint32_t> keys(na), divs((size_t)na * 5);
    for (uint32_t i = 0; i < na; i++) {
        uint32_t rec[6];
        memcpy(rec, ag + (size_t)i * 24, 24);
        keys[i] = rec[0];
        for (int k = 0; k < 5; k++) divs[(size_t)i * 5 + k] = rec[1 + k];
    }
    std::vector<uint64_t> ahash(32768, 0);
    for (uint32_t i = 0; i < na; i++) {
        uint32_t pos = (keys[i] * 0x9E3779B1u) >> 17;
        while (ahash[pos]) pos = (pos + 1) & 32767;
        ahash[pos] = ((uint64_t)keys[i] << 32) | (uint64_t)(i + 1);
    }
    uint64_t *d_s, *d_j, *d_h;
    uint32_t *d_k, *d_d;
    if (upload(suhai, &d_s) || upload(jihai, &d_j) || upload(keys, &d_k) || upload(divs, &d_d) || upload(ahash, &d_h)) return -1;
    g_tables.dev = {d_s, ns, d_j, nj, d_k, d_h, d_d, na};
    HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(c_mj_tables), &g_tables.dev, sizeof(MjTablesDev)));
    {   // table-id shanten (mj_sptab.h): row ids, merge closure, optimal-entry table, per-key wait / keep masks
        SpTabHost H;
        if (!sp_tab_build(suhai.data(), ns, jihai.data(), nj, H)) return fail("sp_tab_build: " + H.error);
        u8 *d_id, *d_m;
        SpRec *d_o, *d_wk;
        if (upload(H.id, &d_id) || upload(H.mrg, &d_m) || upload(H.opt, &d_o) || upload(H.wk, &d_wk)) return -1;
        SpTabDev st{d_id, d_m, d_o, d_wk, ns, nj, H.zero_id};
        HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(c_sp_tab), &st, sizeof(SpTabDev)));
        std::vector<float> nt((size_t)SP_NT_ROWS * SP_NT_ROWS * SP_NT_STRIDE);  // not_tsumo rows of every wall size (mj_sp.hip)
        sp_not_tsumo_build(nt.data());
        float* d_nt;
        if (upload(nt, &d_nt)) return -1;
        const float* d_ntc = d_nt;
        HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(c_sp_nt), &d_ntc, sizeof d_ntc));
    }
    auto g = build_gather();
    g_tables.n_gather = (int)g.size();
    for (int c = 0, k = 0; c <= SNAP_NCH; c++) {  // entries are in field order = ascending dst_off
        while (k < (int)g.size() && (int)g[k].dst_off < c * SNAP_CH) k++;
        g_tables.gather_chunk[c] = c == SNAP_NCH ? (int)g.size() : k;
    }
    for (size_t k = 1; k < g.size(); k++)
        if (g[k].dst_off < g[k - 1].dst_off) return fail("gather list not in record order");
    if (upload(g, &g_tables.gather)) return -1;
    std::vector<float> decay(64);
    for (int k = 0; k < 64; k++) decay[k] = expf(-0.2f * (float)k);  // obs_repr.rs:228,266
    if (upload(decay, &g_tables.decay) || upload(build_rbf(4096, 500, 10), &g_tables.rbf_score) ||
        upload(build_rbf(256, 6, 3), &g_tables.rbf_6) || upload(build_rbf(256, 12, 3), &g_tables.rbf_12) ||
        upload(build_rbf(256, 23, 4), &g_tables.rbf_23))
        return -1;
    HIP_OK(hipGetDevice(&g_tables.device));
    // ready only after c_mj_tables AND c_sp_tab / c_sp_nt are on the device: since round 5 the step kernel of EVERY obs version walks the
    // table-id sets (mj_rules.h: update_shanten_discards / update_waits_and_furiten), and no pool can be created before this flag is set
    g_tables.ready = true;
    return 0;
}

MjPool* mj_pool_create(int n_tables, int version, int deal_algo, int max_rows) {
    if (!g_tables.ready) {
        fail("mj_tables_upload has not been called");
        return nullptr;
    }
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != g_tables.device) {  // one process per GPU: pools live where the tables are
        fail("the lookup tables were uploaded to HIP device " + std::to_string(g_tables.device) + ", the current device is " +
             std::to_string(cur) + ": use one process per GPU (tables and pools of a process share one device)");
        return nullptr;
    }
    if (n_tables <= 0 || mj_obs_rows(version) < 0) {
        fail("bad n_tables / version");
        return nullptr;
    }
    MjPool* P = new MjPool;
    P->n_tables = n_tables;
    P->n_blocks = (n_tables + MJ_LANES - 1) / MJ_LANES;
    P->version[0] = P->version[1] = version;
    P->deal_algo = deal_algo;
    P->max_rows = max_rows > 0 ? max_rows : 8 * n_tables;
    bool ok = hipMalloc(&P->blocks, (size_t)P->n_blocks * sizeof(TableBlock)) == hipSuccess &&
              hipMalloc(&P->rows[0], (size_t)P->max_rows * 4) == hipSuccess &&
              hipMalloc(&P->rows[1], (size_t)P->max_rows * 4) == hipSuccess &&
              hipMalloc(&P->n_rows_dev, 2 * sizeof(int)) == hipSuccess &&
              hipMalloc(&P->block_rows, (size_t)P->n_blocks * 2 * sizeof(int)) == hipSuccess &&
              hipMalloc(&P->snap, (size_t)P->n_blocks * MJ_LANES * sizeof(TableOne)) == hipSuccess &&
              hipHostMalloc(&P->n_rows_host, 2 * sizeof(int)) == hipSuccess &&
              hipMalloc(&P->counters, 8 * sizeof(unsigned long long)) == hipSuccess;
    if (!ok) {
        fail("device allocation failed");
        mj_pool_destroy(P);
        return nullptr;
    }
    hipMemset(P->blocks, 0, (size_t)P->n_blocks * sizeof(TableBlock));
    for (int v = 1; v <= 4; v++) {
        const void* fn = v == 1 ? (const void*)mj_k_encode<1> : v == 2 ? (const void*)mj_k_encode<2>
                       : v == 3 ? (const void*)mj_k_encode<3> : (const void*)mj_k_encode<4>;
        hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)enc_lds_bytes(v));
    }
    return P;
}

void mj_pool_destroy(MjPool* P) {
    if (!P) return;
    hipFree(P->blocks);
    hipFree(P->rows[0]);
    hipFree(P->rows[1]);
    hipFree(P->n_rows_dev);
    hipFree(P->block_rows);
    hipFree(P->snap);
    hipFree(P->sp_work);
    if (P->sp_stream2) hipStreamDestroy(P->sp_stream2);
    if (P->sp_ev_fork) hipEventDestroy(P->sp_ev_fork);
    if (P->sp_ev_join) hipEventDestroy(P->sp_ev_join);
    hipFree(P->sp_queue);
    hipFree(P->sp_order);
    hipFree(P->sp_cls);
    hipFree(P->sp_err);
    hipFree(P->enc_flag);
    hipFree(P->log);
    hipFree(P->log_len);
    hipFree(P->rp_script); hipFree(P->rp_off); hipFree(P->rp_cursor); hipFree(P->rp_ev_index);
    hipFree(P->rp_kyoku); hipFree(P->rp_tracked); hipFree(P->rp_label); hipFree(P->rp_kan_label);
    if (P->n_rows_host) hipHostFree(P->n_rows_host);
    if (P->sp_gaveup_host) hipHostFree(P->sp_gaveup_host);
    if (P->ev_rows) hipEventDestroy(P->ev_rows);
    if (P->ev_snap) hipEventDestroy(P->ev_snap);
    hipFree(P->counters);
    hipFree(P->final_scores);
    hipFree(P->final_done);
    for (auto* v : {&P->events, &P->sp_events})
        for (auto& e : *v) {
            hipEventDestroy(e.first);
            hipEventDestroy(e.second);
        }
    delete P;
}

int mj_pool_reset(MjPool* P, const uint64_t* nonces, const uint64_t* keys, const uint32_t* game_ids,
                  const uint8_t* agent_of_seat, int n_games_total) {
    if (!P) return fail("null pool");
    std::vector<TableBlock> host(P->n_blocks);
    memset(host.data(), 0, host.size() * sizeof(TableBlock));
    for (int t = 0; t < P->n_blocks * MJ_LANES; t++) {
        TableBlock& B = host[t >> 6];
        int l = t & 63;
        if (t >= P->n_tables) {
            B.flags[l] = TF_INACTIVE | TF_DONE | TF_ENDED;
            continue;
        }
        B.seed_nonce[l] = nonces[t];
        B.seed_key[l] = keys[t];
        B.game_id[l] = game_ids ? game_ids[t] : (uint32_t)t;
        B.agent_of_seat[l] = agent_of_seat ? agent_of_seat[t] : 0;
        for (int i = 0; i < 4; i++) B.scores[i][l] = 25000;  // BatchGame::tenhou_hanchan (game.rs:222-228)
    }
    HIP_OK(hipMemcpy(P->blocks, host.data(), host.size() * sizeof(TableBlock), hipMemcpyHostToDevice));
    HIP_OK(hipMemset(P->counters, 0, 8 * sizeof(unsigned long long)));
    hipFree(P->final_scores);
    hipFree(P->final_done);
    P->n_games_total = n_games_total > 0 ? n_games_total : P->n_tables;
    HIP_OK(hipMalloc(&P->final_scores, (size_t)P->n_games_total * 4 * sizeof(int)));
    HIP_OK(hipMalloc(&P->final_done, (size_t)P->n_games_total));
    HIP_OK(hipMemset(P->final_scores, 0, (size_t)P->n_games_total * 4 * sizeof(int)));
    HIP_OK(hipMemset(P->final_done, 0, (size_t)P->n_games_total));
    if (P->log_len) HIP_OK(hipMemset(P->log_len, 0, (size_t)P->n_tables * sizeof(uint32_t)));
    P->rp_active = false;
    P->cycles = 0;
    P->start_stagger = 0;
    P->rows_valid = false;
    return 0;
}

int mj_pool_enable_log(MjPool* P, uint32_t words_per_table) {
    if (!P) return fail("null pool");
    if (P->log) { hipFree(P->log); hipFree(P->log_len); P->log = nullptr; P->log_len = nullptr; }
    P->log_cap = words_per_table;
    if (words_per_table == 0) return 0;
    HIP_OK(hipMalloc(&P->log, (size_t)P->n_tables * words_per_table * sizeof(uint64_t)));
    HIP_OK(hipMalloc(&P->log_len, (size_t)P->n_tables * sizeof(uint32_t)));
    HIP_OK(hipMemset(P->log_len, 0, (size_t)P->n_tables * sizeof(uint32_t)));
    return 0;
}
int mj_log_lengths(MjPool* P, uint32_t* len_out, void* stream) {
    if (!P || !P->log) return fail("event log is not enabled");
    HIP_OK(hipStreamSynchronize((hipStream_t)stream));
    HIP_OK(hipMemcpy(len_out, P->log_len, (size_t)P->n_tables * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return 0;
}
int mj_log_read(MjPool* P, int table0, int n, uint64_t* words_out, void* stream) {
    if (!P || !P->log) return fail("event log is not enabled");
    if (table0 < 0 || n < 0 || table0 + n > P->n_tables) return fail("table range out of bounds");
    HIP_OK(hipStreamSynchronize((hipStream_t)stream));
    HIP_OK(hipMemcpy(words_out, P->log + (size_t)table0 * P->log_cap, (size_t)n * P->log_cap * sizeof(uint64_t),
                     hipMemcpyDeviceToHost));
    return 0;
}

int mj_pool_configure(MjPool* P, int agent, int version, int enable_quick_eval, int enable_guard) {
    if (!P || agent < 0 || agent > 1) return fail("bad agent");
    if (version != 0) {
        if (mj_obs_rows(version) < 0) return fail("bad obs version");
        P->version[agent] = version;
    }
    P->enable_quick_eval[agent] = enable_quick_eval;
    P->enable_agari_guard[agent] = enable_guard;
    return 0;
}
int mj_pool_set_refill(MjPool* P, uint64_t stride) {
    if (!P) return fail("null pool");
    P->refill_stride = stride;
    return 0;
}

int mj_pool_set_start_stagger(MjPool* P, uint32_t cycles, void* stream) {
    if (!P) return fail("null pool");
    if (P->cycles != 0) return fail("mj_pool_set_start_stagger: call it right after mj_pool_reset, before the first mj_step");
    if (cycles && !P->refill_stride) return fail("mj_pool_set_start_stagger needs the refill mode (mj_pool_set_refill)");
    if (!cycles && P->start_stagger) return fail("mj_pool_set_start_stagger(0) after the tables were parked: reset the pool instead");
    P->start_stagger = cycles;
    if (cycles) hipLaunchKernelGGL(mj_k_park, dim3(P->n_blocks), dim3(64), 0, (hipStream_t)stream, P->blocks, P->n_tables);
    HIP_OK(hipGetLastError());
    return 0;
}

static int launch_rows(MjPool* P, hipStream_t s);
// mj_rows_count returns as soon as the row counts are on the host; mj_k_snapshot, queued behind their copy, may still be running.  A
// reader of the snapshot records on the SAME stream is ordered behind it by the stream; one on another stream waits for ev_snap.
static int wait_snapshot(MjPool* P, hipStream_t s) {
    if (P->ev_snap && s != P->step_stream) HIP_OK(hipStreamWaitEvent(s, P->ev_snap, 0));
    return 0;
}
int mj_step(MjPool* P, const int32_t* a0, const int32_t* a1, void* stream) {
    return mj_step_q(P, a0, a1, nullptr, nullptr, stream);
}
int mj_step_q(MjPool* P, const int32_t* a0, const int32_t* a1, const float* q0, const float* q1, void* stream) {
    return mj_step_ev(P, a0, a1, q0, q1, nullptr, nullptr, stream);
}
int mj_step_ev(MjPool* P, const int32_t* a0, const int32_t* a1, const float* q0, const float* q1, const uint64_t* ev0,
               const uint64_t* ev1, void* stream) {
    if (!P) return fail("null pool");
    if ((P->enable_agari_guard[0] && a0 && !q0) || (P->enable_agari_guard[1] && a1 && !q1))
        return fail("enable_rule_based_agari_guard needs the q-values of the batch (mj_step_q)");
    hipStream_t s = (hipStream_t)stream;
    StepParams sp;
    sp.blocks = P->blocks;
    sp.n_tables = P->n_tables;
    sp.tables = g_tables.dev;
    sp.actions[0] = a0;
    sp.actions[1] = a1;
    sp.q_values[0] = q0;
    sp.q_values[1] = q1;
    sp.reactions[0] = ev0;
    sp.reactions[1] = ev1;
    sp.log = P->log;
    sp.log_len = P->log_len;
    sp.log_cap = P->log_cap;
    sp.cycle = (uint32_t)P->cycles;
    sp.deal_algo = P->deal_algo;
    for (int a = 0; a < 2; a++) {
        sp.enable_quick_eval[a] = P->enable_quick_eval[a];
        sp.enable_agari_guard[a] = P->enable_agari_guard[a];
    }
    sp.game_length = 8;
    sp.refill = P->refill_stride != 0;
    sp.refill_stride = P->refill_stride;
    sp.start_stagger = P->start_stagger;
    sp.counters = P->counters;
    sp.final_scores = P->final_scores;
    sp.final_done = P->final_done;
    sp.n_games_total = P->n_games_total;
    sp.block_rows = P->block_rows;
    if (sp.refill) hipLaunchKernelGGL(mj_k_refill, dim3(P->n_blocks), dim3(64), 0, s, sp);
    hipLaunchKernelGGL(mj_k_step, dim3(P->n_blocks), dim3(MJ_LANES), 0, s, sp);
    return launch_rows(P, s);
}
static int launch_rows(MjPool* P, hipStream_t s) {
    RowsParams rp;
    rp.blocks = P->blocks;
    rp.n_blocks = P->n_blocks;
    rp.block_rows = P->block_rows;
    rp.rows[0] = P->rows[0];
    rp.rows[1] = P->rows[1];
    rp.n_rows_out = P->n_rows_dev;
    rp.max_rows[0] = rp.max_rows[1] = P->max_rows;
    hipLaunchKernelGGL(mj_k_scan, dim3(1), dim3(1024), 0, s, rp);
    hipLaunchKernelGGL(mj_k_assign, dim3(P->n_blocks), dim3(64), 0, s, rp);
    // the row counts go to the host BEFORE the snapshot is queued (round 5): the host's read of them (mj_rows_count, one per cycle) and
    // its launch of the encoder then run under the snapshot kernel's 0.1 ms instead of after it
    HIP_OK(hipMemcpyAsync(P->n_rows_host, P->n_rows_dev, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    if (!P->ev_rows) HIP_OK(hipEventCreateWithFlags(&P->ev_rows, hipEventDisableTiming));
    HIP_OK(hipEventRecord(P->ev_rows, s));
    SnapParams snp = {P->blocks, P->snap, g_tables.gather, {0}};
    for (int c = 0; c <= SNAP_NCH; c++) snp.chunk_first[c] = g_tables.gather_chunk[c];
    hipLaunchKernelGGL(mj_k_snapshot, dim3(P->n_blocks), dim3(256), 0, s, snp);
    if (!P->ev_snap) HIP_OK(hipEventCreateWithFlags(&P->ev_snap, hipEventDisableTiming));
    HIP_OK(hipEventRecord(P->ev_snap, s));
    P->step_stream = s;
    HIP_OK(hipGetLastError());
    P->cycles += 1;
    P->rows_valid = false;
    return 0;
}

// ---------------------------------------------------------------- log replay (dataset/gameplay.rs)
int mj_replay_load(MjPool* P, const uint64_t* script, const uint32_t* off, const uint8_t* tracked, int n_logs,
                   int always_include_kan_select, const uint64_t* nonces, const uint64_t* keys) {
    if (!P) return fail("null pool");
    if (n_logs != P->n_tables) return fail("mj_replay_load: one log per table (create the pool with n_tables = n_logs)");
    const size_t n_words = off[n_logs];
    hipFree(P->rp_script); hipFree(P->rp_off); hipFree(P->rp_cursor); hipFree(P->rp_ev_index);
    hipFree(P->rp_kyoku); hipFree(P->rp_tracked); hipFree(P->rp_label); hipFree(P->rp_kan_label);
    HIP_OK(hipMalloc(&P->rp_script, (n_words + 1) * sizeof(uint64_t)));
    HIP_OK(hipMalloc(&P->rp_off, (size_t)(n_logs + 1) * sizeof(uint32_t)));
    HIP_OK(hipMalloc(&P->rp_cursor, (size_t)n_logs * sizeof(uint32_t)));
    HIP_OK(hipMalloc(&P->rp_ev_index, (size_t)n_logs * sizeof(uint32_t)));
    HIP_OK(hipMalloc(&P->rp_kyoku, (size_t)n_logs));
    HIP_OK(hipMalloc(&P->rp_tracked, (size_t)n_logs));
    HIP_OK(hipMalloc(&P->rp_label, (size_t)n_logs * 4 * sizeof(int32_t)));
    HIP_OK(hipMalloc(&P->rp_kan_label, (size_t)n_logs * 4 * sizeof(int32_t)));
    HIP_OK(hipMemcpy(P->rp_script, script, n_words * sizeof(uint64_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(P->rp_off, off, (size_t)(n_logs + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(P->rp_tracked, tracked, (size_t)n_logs, hipMemcpyHostToDevice));
    HIP_OK(hipMemset(P->rp_cursor, 0, (size_t)n_logs * sizeof(uint32_t)));
    HIP_OK(hipMemset(P->rp_ev_index, 0, (size_t)n_logs * sizeof(uint32_t)));
    HIP_OK(hipMemset(P->rp_kyoku, 0, (size_t)n_logs));
    P->rp_always_kan = always_include_kan_select;
    // fresh tables (all seats agent 0); padding lanes of the last block inactive
    std::vector<TableBlock> host(P->n_blocks);
    memset(host.data(), 0, host.size() * sizeof(TableBlock));
    for (int t = P->n_tables; t < P->n_blocks * MJ_LANES; t++) host[t >> 6].flags[t & 63] = TF_INACTIVE | TF_DONE | TF_ENDED;
    if (nonces && keys)
        for (int t = 0; t < P->n_tables; t++) {
            host[t >> 6].seed_nonce[t & 63] = nonces[t];
            host[t >> 6].seed_key[t & 63] = keys[t];
        }
    P->rp_active = true;
    HIP_OK(hipMemcpy(P->blocks, host.data(), host.size() * sizeof(TableBlock), hipMemcpyHostToDevice));
    HIP_OK(hipMemset(P->counters, 0, 8 * sizeof(unsigned long long)));
    P->cycles = 0;
    P->rows_valid = false;
    return 0;
}
int mj_replay_step(MjPool* P, void* stream) {
    if (!P || !P->rp_script) return fail("mj_replay_load first");
    ReplayParams rp;
    rp.blocks = P->blocks;
    rp.n_tables = P->n_tables;
    rp.script = P->rp_script;
    rp.script_off = P->rp_off;
    rp.cursor = P->rp_cursor;
    rp.ev_index = P->rp_ev_index;
    rp.kyoku_idx = P->rp_kyoku;
    rp.tracked = P->rp_tracked;
    rp.always_include_kan_select = P->rp_always_kan;
    rp.deal_algo = P->deal_algo;
    rp.block_rows = P->block_rows;
    rp.label = P->rp_label;
    rp.kan_label = P->rp_kan_label;
    rp.counters = P->counters;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(mj_k_replay, dim3(P->n_blocks), dim3(64), 0, s, rp);
    return launch_rows(P, s);
}
int mj_replay_meta(MjPool* P, int32_t* meta_dev, void* stream) {
    if (!P || !P->rp_script) return fail("mj_replay_load first");
    if (!P->rows_valid) return fail("mj_rows_count must be called after mj_replay_step");
    const int n = P->last_rows[0];
    if (n == 0) return 0;
    ReplayMetaParams mp = {P->blocks, P->rows[0], n, P->rp_label, P->rp_kan_label, P->rp_kyoku, P->rp_ev_index, meta_dev};
    hipLaunchKernelGGL(mj_k_replay_meta, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, mp);
    HIP_OK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------- single-table access (libriichi.state.PlayerState)
int mj_table_apply_event(MjPool* P, int table, const uint64_t* words, int n_words, void* stream) {
    if (!P || table < 0 || table >= P->n_tables) return fail("bad table");
    if (n_words < 1 || n_words > 16) return fail("bad event");
    hipStream_t s = (hipStream_t)stream;
    uint64_t* dev = nullptr;
    HIP_OK(hipMalloc(&dev, 16 * sizeof(uint64_t)));
    HIP_OK(hipMemcpyAsync(dev, words, (size_t)n_words * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(mj_k_apply_event, dim3(1), dim3(1), 0, s, P->blocks, table, dev);
    HIP_OK(hipStreamSynchronize(s));
    hipFree(dev);
    HIP_OK(hipGetLastError());
    return 0;
}
int mj_table_mark_row(MjPool* P, int table, int seat, int at_kan_select, void* stream) {
    if (!P || table < 0 || table >= P->n_tables || seat < 0 || seat > 3) return fail("bad table / seat");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(mj_k_mark_row, dim3(P->n_blocks), dim3(64), 0, s, P->blocks, P->n_tables, table, seat, at_kan_select,
                       P->block_rows);
    return launch_rows(P, s);
}
int mj_table_query(MjPool* P, int table, int seat, int what, const int32_t* args8, int32_t* out8, void* stream) {
    if (!P || table < 0 || table >= P->n_tables || seat < 0 || seat > 3) return fail("bad table / seat");
    hipStream_t s = (hipStream_t)stream;
    int32_t* dev = nullptr;
    HIP_OK(hipMalloc(&dev, 16 * sizeof(int32_t)));
    HIP_OK(hipMemsetAsync(dev, 0, 16 * sizeof(int32_t), s));
    if (args8) HIP_OK(hipMemcpyAsync(dev, args8, 8 * sizeof(int32_t), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(mj_k_query, dim3(1), dim3(1), 0, s, P->blocks, table, seat, what, dev, dev + 8);
    HIP_OK(hipMemcpyAsync(out8, dev + 8, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    hipFree(dev);
    HIP_OK(hipGetLastError());
    return 0;
}

int mj_rows_count(MjPool* P, int32_t out[2], void* stream) {
    if (!P) return fail("null pool");
    if (P->ev_rows) HIP_OK(hipEventSynchronize(P->ev_rows));  // (the counts' copy; the snapshot queued behind it may still be running)
    else HIP_OK(hipStreamSynchronize((hipStream_t)stream));
    P->last_rows[0] = P->n_rows_host[0];
    P->last_rows[1] = P->n_rows_host[1];
    P->rows_valid = true;
    out[0] = P->last_rows[0];
    out[1] = P->last_rows[1];
    if (out[0] > P->max_rows || out[1] > P->max_rows) return fail("row capacity exceeded");
    return 0;
}
const uint32_t* mj_rows_dev(MjPool* P, int agent) { return P ? P->rows[agent & 1] : nullptr; }

int mj_encode(MjPool* P, int agent, float* obs, uint8_t* masks, void* stream) {
    if (!P) return fail("null pool");
    if (!P->rows_valid) return fail("mj_rows_count must be called after mj_step and before mj_encode");
    int n = P->last_rows[agent & 1];
    if (n == 0) return 0;
    EncParams ep;
    ep.blocks = P->blocks;
    ep.rows = P->rows[agent & 1];
    ep.n_rows = n;
    ep.tables = g_tables.dev;
    ep.obs = obs;
    ep.masks = masks;
    ep.version = P->version[agent & 1];
    ep.C = mj_obs_rows(ep.version);
    ep.snap = P->snap;
    ep.decay_lut = g_tables.decay;
    ep.rbf_score = g_tables.rbf_score;
    ep.rbf_6 = g_tables.rbf_6;
    ep.rbf_12 = g_tables.rbf_12;
    ep.rbf_23 = g_tables.rbf_23;
    if (!P->enc_flag) {
        HIP_OK(hipMalloc(&P->enc_flag, sizeof(int)));
        HIP_OK(hipMemset(P->enc_flag, 0, sizeof(int)));
    }
    ep.err_flag = P->enc_flag;
    size_t lds = enc_lds_bytes(ep.version);
    hipStream_t s = (hipStream_t)stream;
    if (wait_snapshot(P, s)) return -1;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (P->timing) {
        HIP_OK(hipEventCreate(&e0));
        HIP_OK(hipEventCreate(&e1));
        HIP_OK(hipEventRecord(e0, s));
    }
#if ENC_PERSIST
    static const int enc_grid_max = getenv("MJ_ENC_GRID") ? std::max(1, atoi(getenv("MJ_ENC_GRID"))) : 256 * ENC_WPS;  // persistent: ENC_WPS workgroups per CU
    const int egrid = std::min(n, enc_grid_max);
#else
    const int egrid = n;
#endif
    switch (ep.version) {
        case 1: hipLaunchKernelGGL(mj_k_encode<1>, dim3(egrid), dim3(ENC_THREADS), lds, s, ep); break;
        case 2: hipLaunchKernelGGL(mj_k_encode<2>, dim3(egrid), dim3(ENC_THREADS), lds, s, ep); break;
        case 3: hipLaunchKernelGGL(mj_k_encode<3>, dim3(egrid), dim3(ENC_THREADS), lds, s, ep); break;
        default: hipLaunchKernelGGL(mj_k_encode<4>, dim3(egrid), dim3(ENC_THREADS), lds, s, ep); break;
    }
    if (P->timing) {
        HIP_OK(hipEventRecord(e1, s));
        P->events.push_back({e0, e1});
    }
    HIP_OK(hipGetLastError());
    if (ep.version == 4) {  // SP block, rows 889..1011
        // one decision row per persistent workgroup (mj_sp.hip: mj_k_sp).  Round 4 also built a per-phase pipeline (every phase its own launch
        // over the state graphs of ALL rows: 34.8 ms against 21.9 ms, DESIGN.md section 6, profiles/r04_spg_*); it left the tree in round 5
        // (git: mortal_amd/csrc/mj_sp2.hip up to commit 73bd82e).
        if (!P->sp_err) {
            HIP_OK(hipMalloc(&P->sp_err, 32 * sizeof(unsigned long long)));
            HIP_OK(hipMemset(P->sp_err, 0, 32 * sizeof(unsigned long long)));
            HIP_OK(hipMalloc(&P->sp_order, (size_t)P->max_rows * sizeof(uint32_t)));
            HIP_OK(hipMalloc(&P->sp_cls, (size_t)P->max_rows));
        }
        hipEvent_t s0 = nullptr, s1 = nullptr;
        if (P->timing) {
            HIP_OK(hipEventCreate(&s0));
            HIP_OK(hipEventCreate(&s1));
            HIP_OK(hipEventRecord(s0, s));
        }
        {
            if (!P->sp_work) {
                P->sp_grid = 256 * SP_WGS;  // persistent workgroups: SP_WGS per CU, one decision row each at a time
                if (P->sp_grid > P->max_rows) P->sp_grid = P->max_rows;  // never more rows than that in a launch (small pools: small work area)
                if (const char* g = getenv("MJ_SP_GRID")) P->sp_grid = std::max(1, std::min(P->sp_grid, atoi(g)));  // tests: few workgroups, many rows each (the row-to-row paths)
                // promotion of large rows to mj_k_sp_wide (mj_sp.hip): MJ_SP_WIDE = 0 never / 1 every launch / unset: launches of at most
                // MJ_SP_WIDE_MAX_ROWS rows; MJ_SP_WIDE_GRID wide workgroups; MJ_SP_PROMO_MIN1 / _MIN2: the level sizes that park a row
                if (!P->sp_sched_set) {
                    if (const char* g = getenv("MJ_SP_WIDE")) P->sp_wide_mode = atoi(g);
                    if (const char* g = getenv("MJ_SP_WIDE_MAX_ROWS")) P->sp_wide_max_rows = atoi(g);
                    if (const char* g = getenv("MJ_SP_WIDE_GRID")) P->sp_wide_grid = std::max(1, atoi(g));
                    if (const char* g = getenv("MJ_SP_PROMO_MIN1")) P->sp_promo_min[1] = std::max(1, atoi(g));
                    if (const char* g = getenv("MJ_SP_PROMO_MIN2")) P->sp_promo_min[2] = std::max(1, atoi(g));
                }
                P->sp_spare = P->sp_wide_mode == 0 ? 0 : std::min(SP_PROMO_CAP, std::max(8, P->n_tables / 4) & ~1);
                P->sp_wide_areas = P->sp_spare ? std::min(256, std::max(2, P->n_tables / 16)) : 0;  // (a 64-table test pool does not need 2 GB of work areas)
                if (P->sp_wide_mode < 0 && P->n_tables > P->sp_wide_max_rows) P->sp_spare = 0;  // (a launch has about as many rows as the pool has tables)
                const int areas = P->sp_grid + P->sp_spare + P->sp_wide_areas;  // + one per workgroup of mj_k_sp_wide (its own rows)
                HIP_OK(hipMalloc(&P->sp_work, (size_t)areas * sizeof(SpWork)));
                for (int g = 0; g < areas; g++) {
                    HIP_OK(hipMemsetAsync(P->sp_work[g].tag, 0, sizeof(P->sp_work[g].tag), s));  // empty hash sets ...
                    HIP_OK(hipMemsetAsync(&P->sp_work[g].epoch, 0, sizeof(P->sp_work[g].epoch) + sizeof(P->sp_work[g].pad_), s));  // ... at epoch 0
                }
                HIP_OK(hipMalloc(&P->sp_queue, SP_Q_WORDS * sizeof(int)));
                if (P->sp_spare) {
                    HIP_OK(hipStreamCreateWithFlags(&P->sp_stream2, hipStreamNonBlocking));
                    HIP_OK(hipEventCreateWithFlags(&P->sp_ev_fork, hipEventDisableTiming));
                    HIP_OK(hipEventCreateWithFlags(&P->sp_ev_join, hipEventDisableTiming));
                    // One empty launch of the pair now: the HIP runtime sizes a queue's scratch at the first launch that needs it, and a caller
                    // whose allocator has taken the whole HBM by then (torch's caching allocator under a growing batch) turns that into
                    // HSA_STATUS_ERROR_OUT_OF_RESOURCES in the middle of a run -- at pool set-up it is an ordinary, early failure.
                    HIP_OK(hipMemsetAsync(P->sp_queue, 0, SP_Q_WORDS * sizeof(int), s));
                    HIP_OK(hipStreamSynchronize(s));
                    SpParams w{};
                    w.snap = P->snap;
                    w.rows = P->rows[agent & 1];
                    w.n_rows = 0;
                    w.tables = g_tables.dev;
                    w.obs = obs;
                    w.work = P->sp_work;
                    w.queue = P->sp_queue;
                    w.order = P->sp_order;
                    w.err = P->sp_err;
                    w.sweep = 1;
                    hipLaunchKernelGGL(mj_k_sp_wide, dim3(1), dim3(SP_WIDE_THREADS), 0, s, w);
                    hipLaunchKernelGGL(mj_k_sp_promo, dim3(1), dim3(SP_THREADS), 0, P->sp_stream2, w);
                    HIP_OK(hipStreamSynchronize(P->sp_stream2));
                    HIP_OK(hipStreamSynchronize(s));
                    HIP_OK(hipGetLastError());
                }
            }
            HIP_OK(hipMemsetAsync(P->sp_queue, 0, SP_Q_WORDS * sizeof(int), s));
            SpParams sp;
            sp.snap = P->snap;
            sp.rows = P->rows[agent & 1];
            sp.n_rows = n;
            sp.tables = g_tables.dev;
            sp.obs = obs;
            sp.work = P->sp_work;
            sp.queue = P->sp_queue;
            sp.order = P->sp_order;
            sp.err = P->sp_err;
            sp.prof = getenv("MJ_SP_PROF") ? P->sp_err : nullptr;
            sp.rowdump = nullptr;
            static const char* rowdump_path = getenv("MJ_SP_ROWDUMP");  // (debug) per-row cost records appended to this file, one synchronous copy per launch
            if (rowdump_path) {
                HIP_OK(hipMalloc(&sp.rowdump, (size_t)n * 48));
                HIP_OK(hipMemsetAsync(sp.rowdump, 0, (size_t)n * 48, s));
                HIP_OK(hipMemsetAsync(P->sp_err + 28, 0xFF, 8, s));
                HIP_OK(hipMemsetAsync(P->sp_err + 29, 0, 24, s));
            }
            int grid = n < P->sp_grid ? n : P->sp_grid;
            // The schedule needs mj_k_sp_wide and mj_k_sp_promo side by side.  Where they do not overlap -- more streams in the process than the
            // runtime has hardware queues, so that the promo kernel queues up BEHIND the spinning wide kernel -- the wide workgroups give up
            // after SP_WIDE_TIMEOUT, the sweep launch still produces the same obs, and the give-ups (copied to pinned memory behind every
            // sweep) switch the schedule off for this pool: one slow launch, then mj_k_sp alone as in round 5.
            if (P->sp_wide_mode < 0 && P->sp_gaveup_host && *P->sp_gaveup_host && !P->sp_wide_off) {  // (auto mode only: mode 1 = every launch, as asked)
                P->sp_wide_off = true;
                fprintf(stderr, "[mortal_amd] small-pool SP schedule switched off for this pool: mj_k_sp_wide and mj_k_sp_promo did not run side by side "
                                "(%llu wide workgroups gave up waiting; more concurrent streams than hardware queues?)\n", *P->sp_gaveup_host);
            }
            const bool hybrid = P->sp_spare > 0 && !P->sp_wide_off && (P->sp_wide_mode > 0 || (P->sp_wide_mode < 0 && n <= P->sp_wide_max_rows));
            sp.promo_cap = hybrid ? P->sp_spare : 0;
            // Defaults measured on MI355X (tools/r06_sweep.sh): up to ~12 k rows 64 wide workgroups (a quarter of the CUs),
            // rows parked at >= 1,200 level-1 states (or >= 400 level-2 states, before that level is expanded); up to ~20 k rows 32 wide
            // workgroups and 1,600 level-1 states; beyond that a launch keeps all CUs for mj_k_sp (sp_wide_max_rows).  The root level is never
            // parked (nothing is known yet), level 0 is not expanded.
            sp.promo_min[0] = sp.promo_min[3] = 1 << 30;
            sp.promo_min[1] = P->sp_promo_min[1] > 0 ? P->sp_promo_min[1] : n <= 12000 ? 1200 : 1600;
            sp.promo_min[2] = P->sp_promo_min[2] > 0 ? P->sp_promo_min[2] : n <= 12000 ? 400 : 1 << 30;
            sp.n_narrow = grid;
            sp.sweep = 0;
            // queue order: rows counting-sorted by cost class, heaviest first (inside the timed mj_k_sp region)
            hipLaunchKernelGGL(mj_k_order_classify, dim3((n + 255) / 256), dim3(256), 0, s, P->snap, sp.rows, n, P->sp_cls, P->sp_queue + 1);
            hipLaunchKernelGGL(mj_k_order_scatter, dim3((n + 255) / 256), dim3(256), 0, s, P->sp_cls, n, P->sp_queue + 1, P->sp_queue + 9, P->sp_order);
            if (!hybrid) {
                hipLaunchKernelGGL(mj_k_sp, dim3(grid), dim3(SP_THREADS), 0, s, sp);
            } else {
                // mj_k_sp_wide FIRST and on the caller's stream (its few workgroups take a whole CU each and must be resident before the 1,024
                // workgroups of mj_k_sp fill the chip), mj_k_sp on the second stream behind the row order, then the sweep behind both.
                // The emulator runs a launch to completion: there (and with MJ_SP_WIDE_SERIAL=1) the sweep alone takes the parked rows.
                P->sp_hybrid_launches++;
                const int wgrid = std::min(P->sp_wide_areas, P->sp_wide_grid > 0 ? P->sp_wide_grid : n <= 12000 ? 64 : 32);
#ifdef MJ_EMU
                const bool serial = true;
#else
                static const bool serial = getenv("MJ_SP_WIDE_SERIAL") != nullptr;
#endif
                if (serial) {
                    if (getenv("MJ_SP_WIDE_ALL_ROWS")) {  // (tests) the wide kernel alone first: with no producer to wait for it takes EVERY row of the queue itself
                        SpParams spw = sp;
                        spw.n_narrow = 0;
                        spw.work = sp.work + grid;  // (its own areas: work + n_narrow + promo_cap + block, as in the concurrent launch)
                        hipLaunchKernelGGL(mj_k_sp_wide, dim3(wgrid), dim3(SP_WIDE_THREADS), 0, s, spw);
                    }
                    hipLaunchKernelGGL(mj_k_sp_promo, dim3(grid), dim3(SP_THREADS), 0, s, sp);
                } else {
                    HIP_OK(hipEventRecord(P->sp_ev_fork, s));
                    hipLaunchKernelGGL(mj_k_sp_wide, dim3(wgrid), dim3(SP_WIDE_THREADS), 0, s, sp);
                    HIP_OK(hipStreamWaitEvent(P->sp_stream2, P->sp_ev_fork, 0));
                    hipLaunchKernelGGL(mj_k_sp_promo, dim3(grid), dim3(SP_THREADS), 0, P->sp_stream2, sp);
                    HIP_OK(hipEventRecord(P->sp_ev_join, P->sp_stream2));
                    HIP_OK(hipStreamWaitEvent(s, P->sp_ev_join, 0));
                }
                sp.sweep = 1;
                hipLaunchKernelGGL(mj_k_sp_wide, dim3(wgrid), dim3(SP_WIDE_THREADS), 0, s, sp);
                if (!P->sp_gaveup_host && hipHostMalloc(&P->sp_gaveup_host, sizeof(unsigned long long)) == hipSuccess) *P->sp_gaveup_host = 0ull;
                if (P->sp_gaveup_host) HIP_OK(hipMemcpyAsync(P->sp_gaveup_host, P->sp_err + 25, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
            }
            if (sp.rowdump) {
                std::vector<uint32_t> h((size_t)n * 12);
                int q[SP_Q_WORDS];
                HIP_OK(hipMemcpyAsync(h.data(), sp.rowdump, (size_t)n * 48, hipMemcpyDeviceToHost, s));
                HIP_OK(hipMemcpyAsync(q, P->sp_queue, sizeof(q), hipMemcpyDeviceToHost, s));
                HIP_OK(hipStreamSynchronize(s));
                hipFree(sp.rowdump);
                if (FILE* f = fopen(rowdump_path, "ab")) {
                    uint32_t hdr[12] = {0xFFFFFFFFu, (uint32_t)n};
                    for (int k = 0; k < 10; k++) hdr[2 + k] = 0;
                    unsigned long long tt[4];
                    HIP_OK(hipMemcpy(tt, P->sp_err + 28, sizeof tt, hipMemcpyDeviceToHost));
                    for (int k = 0; k < 4; k++) hdr[2 + k] = (uint32_t)tt[k];  // first workgroup in, last narrow / wide out of the row loop, end of the tail
                    fwrite(hdr, 4, 12, f);
                    uint32_t cc[12];
                    for (int k = 0; k < 12; k++) cc[k] = k < 8 ? (uint32_t)q[1 + k] : 0u;
                    fwrite(cc, 4, 12, f);
                    for (int i = 0; i < n; i++)
                        if (h[(size_t)i * 12 + 7]) fwrite(&h[(size_t)i * 12], 4, 12, f);
                    fclose(f);
                }
            }
        }
        if (P->timing) {
            HIP_OK(hipEventRecord(s1, s));
            P->sp_events.push_back({s0, s1});
        }
        HIP_OK(hipGetLastError());
    }
    return 0;
}

int mj_oracle_obs_rows(int version) {  // consts.rs:32-38
    if (version == 1) return 211;
    if (version >= 2 && version <= 4) return 217;
    return -1;
}

int mj_encode_oracle(MjPool* P, int agent, float* out, void* stream) {
    if (!P) return fail("null pool");
    if (!P->rows_valid) return fail("mj_rows_count must be called after mj_step and before mj_encode_oracle");
    int n = P->last_rows[agent & 1];
    if (n == 0) return 0;
    OracleEncParams ep;
    ep.rows = P->rows[agent & 1];
    ep.n_rows = n;
    ep.version = P->version[agent & 1];
    ep.snap = P->snap;
    ep.out = out;
    ep.all_yama = P->rp_active ? 1 : 0;
    size_t lds = enc_oracle_lds_bytes(ep.version);
    hipStream_t s = (hipStream_t)stream;
    if (wait_snapshot(P, s)) return -1;
    if (ep.version == 1) hipLaunchKernelGGL(mj_k_encode_oracle<true>, dim3(n), dim3(ENC_THREADS), lds, s, ep);
    else hipLaunchKernelGGL(mj_k_encode_oracle<false>, dim3(n), dim3(ENC_THREADS), lds, s, ep);
    HIP_OK(hipGetLastError());
    return 0;
}

int mj_encode_timing(MjPool* P, int enable, double* total_ms, int64_t* launches) {
    if (!P) return fail("null pool");
    for (auto& e : P->events) {
        hipEventSynchronize(e.second);
        float ms = 0;
        hipEventElapsedTime(&ms, e.first, e.second);
        P->timed_ms += ms;
        P->timed_launches += 1;
        hipEventDestroy(e.first);
        hipEventDestroy(e.second);
    }
    P->events.clear();
    if (total_ms) *total_ms = P->timed_ms;
    if (launches) *launches = P->timed_launches;
    P->timed_ms = 0;
    P->timed_launches = 0;
    P->timing = enable != 0;
    return 0;
}

int mj_sp_timing(MjPool* P, double* total_ms, int64_t* launches) {
    if (!P) return fail("null pool");
    double tot = 0;
    int64_t n = 0;
    for (auto& e : P->sp_events) {
        hipEventSynchronize(e.second);
        float ms = 0;
        hipEventElapsedTime(&ms, e.first, e.second);
        tot += ms;
        n += 1;
        hipEventDestroy(e.first);
        hipEventDestroy(e.second);
    }
    P->sp_events.clear();
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
    return 0;
}

int mj_random_policy(MjPool* P, int agent, const uint8_t* masks, uint64_t seed, uint64_t cycle, int32_t* actions,
                     void* stream) {
    if (!P) return fail("null pool");
    if (!P->rows_valid) return fail("mj_rows_count must be called first");
    int n = P->last_rows[agent & 1];
    if (n == 0) return 0;
    hipLaunchKernelGGL(mj_k_random_policy, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, P->blocks,
                       P->rows[agent & 1], masks, n, seed, cycle, actions);
    HIP_OK(hipGetLastError());
    return 0;
}

int mj_greedy_policy(MjPool* P, int agent, const uint8_t* masks, const float* obs, uint64_t seed, uint64_t cycle,
                     int32_t* actions, void* stream) {
    if (!P) return fail("null pool");
    if (!P->rows_valid) return fail("mj_rows_count must be called first");
    int n = P->last_rows[agent & 1];
    if (n == 0) return 0;
    const int v = P->version[agent & 1];
    const int d0 = v == 1 ? 923 : v == 2 ? 927 : v == 3 ? 919 : 874;  // first row of the discard block (obs_repr.rs:431-476)
    hipLaunchKernelGGL(mj_k_greedy_policy, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, P->rows[agent & 1], masks,
                       obs, mj_obs_rows(v), d0, n, seed, cycle, actions);
    HIP_OK(hipGetLastError());
    return 0;
}

int mj_counters(MjPool* P, uint64_t out[8], void* stream) {
#ifdef MJ_STEP_PROF
    if (getenv("MJ_STEP_PROF")) {
        unsigned long long sp_[32];
        hipDeviceSynchronize();
        if (hipMemcpyFromSymbol(sp_, HIP_SYMBOL(g_step_prof), sizeof sp_) == hipSuccess) {
            fprintf(stderr, "[step prof] waves %llu kernel %llu commit %llu poll %llu (start_kyoku %llu board_step %llu [n %llu] any_can_act %llu; kyoku ends %llu) classify %llu | board_step by event:",
                    sp_[8], sp_[0], sp_[1], sp_[2], sp_[3], sp_[4], sp_[9], sp_[5], sp_[10], sp_[7]);
            for (int k = 11; k < 32; k++) fprintf(stderr, " %llu", sp_[k]);
            fprintf(stderr, "\n");
        }
    }
#endif
    if (!P) return fail("null pool");
    HIP_OK(hipStreamSynchronize((hipStream_t)stream));
    unsigned long long tmp[8];
    HIP_OK(hipMemcpy(tmp, P->counters, sizeof tmp, hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; i++) out[i] = tmp[i];
    out[5] = P->cycles;
    if (P->enc_flag) {
        int f = 0;
        HIP_OK(hipMemcpy(&f, P->enc_flag, sizeof f, hipMemcpyDeviceToHost));
        out[6] = (unsigned long long)f;  // (the SP block's overflows are added below)
    }
    if (P->sp_err) {
        unsigned long long e2[32];
        HIP_OK(hipMemcpy(e2, P->sp_err, sizeof e2, hipMemcpyDeviceToHost));
        out[6] += e2[0];
        out[7] = e2[1];
        if (getenv("MJ_SP_PROF"))
            fprintf(stderr, "[sp prof] rows %llu setup %llu expand %llu evalL0 %llu evalL>0 %llu encode %llu states %llu (wall_clock64 ticks, 100 MHz) | "
                    "expand passes: probes %llu lists+V %llu td-probes %llu layout %llu inserts %llu; items %llu expanded %llu edges %llu l0-entries %llu; level-0 probe %llu scoring %llu; workgroup lifetimes: sum %llu max %llu queue pops %llu hash resets %llu; eval wavefront time %llu; shader clock %.0f MHz (s_memtime cycles %llu over the lifetimes)\n",
                    e2[1], e2[2], e2[3], e2[4], e2[5], e2[6], e2[7], e2[8], e2[9], e2[10], e2[11], e2[12], e2[13], e2[14], e2[15], e2[16], e2[17], e2[18], e2[19], e2[20], e2[21], e2[22], e2[23],
                    e2[19] ? 100.0 * (double)e2[24] / (double)e2[19] : 0.0, e2[24]);
    }
    return 0;
}

int mj_sp_phase_ticks(MjPool* P, uint64_t out[8], void* stream) {
    if (!P) return fail("null pool");
    for (int i = 0; i < 8; i++) out[i] = 0;
    if (!P->sp_err) return 0;  // no obs-v4 encode has run yet
    HIP_OK(hipStreamSynchronize((hipStream_t)stream));
    unsigned long long e2[8];
    HIP_OK(hipMemcpy(e2, P->sp_err, sizeof e2, hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; i++) out[i] = e2[i];
    return 0;
}

int mj_pool_set_sp_schedule(MjPool* P, int mode, int max_rows, int wide_grid, int min_level1, int min_level2) {
    if (!P) return fail("null pool");
    if (P->sp_work && mode != 0 && mode >= -1 && P->sp_spare == 0) return fail("mj_pool_set_sp_schedule: the work areas are allocated (call it before the first obs-v4 mj_encode)");
    if (mode >= -1) P->sp_wide_mode = mode > 0 ? 1 : mode;
    if (max_rows > 0) P->sp_wide_max_rows = max_rows;
    if (wide_grid > 0) P->sp_wide_grid = std::min(wide_grid, 256);
    if (min_level1 > 0) P->sp_promo_min[1] = min_level1;
    if (min_level2 > 0) P->sp_promo_min[2] = min_level2;
    P->sp_sched_set = true;
    return 0;
}

int mj_sp_schedule_stats(MjPool* P, uint64_t out[4], void* stream) {
    if (!P) return fail("null pool");
    for (int i = 0; i < 4; i++) out[i] = 0;
    if (!P->sp_err) return 0;
    HIP_OK(hipStreamSynchronize((hipStream_t)stream));
    unsigned long long e2[32];
    HIP_OK(hipMemcpy(e2, P->sp_err, sizeof e2, hipMemcpyDeviceToHost));
    out[0] = P->sp_hybrid_launches;
    out[1] = e2[26];
    out[2] = e2[27];
    out[3] = e2[25];
    return 0;
}

int mj_results(MjPool* P, int32_t* scores, uint8_t* done, void* stream) {
    if (!P) return fail("null pool");
    HIP_OK(hipStreamSynchronize((hipStream_t)stream));
    HIP_OK(hipMemcpy(scores, P->final_scores, (size_t)P->n_games_total * 4 * sizeof(int), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(done, P->final_done, (size_t)P->n_games_total, hipMemcpyDeviceToHost));
    return 0;
}

__global__ void mj_k_first_error(const TableBlock* blocks, int n_tables, unsigned long long* out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tables) return;
    const unsigned e = blocks[t >> 6].err[t & 63];
    if (e) atomicMin(out, ((unsigned long long)t << 8) | e);
}
int mj_pool_first_error(MjPool* P, int* table_out, void* stream) {
    if (!P) return fail("null pool");
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* slot = P->counters + 7;  // spare counter word
    HIP_OK(hipMemsetAsync(slot, 0xFF, sizeof(unsigned long long), s));
    hipLaunchKernelGGL(mj_k_first_error, dim3((P->n_tables + 255) / 256), dim3(256), 0, s, P->blocks, P->n_tables, slot);
    HIP_OK(hipGetLastError());
    unsigned long long v = 0;
    HIP_OK(hipMemcpyAsync(&v, slot, sizeof v, hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    if (v == ~0ull) return 0;
    if (table_out) *table_out = (int)(v >> 8);
    return (int)(v & 0xFF);
}

int mj_debug_table(MjPool* P, int table, void* out, size_t out_size, void* stream) {
    if (!P || table < 0 || table >= P->n_tables) return fail("bad table");
    if (out_size < sizeof(TableOne)) return fail("buffer too small");
    HIP_OK(hipStreamSynchronize((hipStream_t)stream));
    std::vector<uint8_t> blk(sizeof(TableBlock));
    HIP_OK(hipMemcpy(blk.data(), &P->blocks[table >> 6], sizeof(TableBlock), hipMemcpyDeviceToHost));
    auto g = build_gather();
    int lane = table & 63;
    for (auto& e : g) memcpy((char*)out + e.dst_off, blk.data() + e.src_off + (size_t)lane * e.size, e.size);
    return 0;
}

}  // extern "C"
