// Host-side equivalence checks of the pure device helpers in mortal_amd/csrc/mj_algo.h (no GPU needed): the helpers
// are compiled __host__ __device__ here and the optimised formulations are compared with the reference-shaped ones.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -o algo_check tests/host/algo_check.hip && ./algo_check
#include <hip/hip_runtime.h>
#define MJD __host__ __device__ inline
#define MJDN __device__ __noinline__
// the bit-count intrinsics of the device headers are __device__-only; the compiler builtins serve both passes
#define __popcll(x) __builtin_popcountll(x)
#define __popc(x) __builtin_popcount(x)
#include "../../mortal_amd/csrc/mj_sptab.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

static u64 random_row(std::mt19937_64& g, int max_nib) {
    u64 r = 0;
    for (int j = 1; j < 10; j++) r |= (u64)(g() % (max_nib + 1)) << (4 * j);  // nibble 0 (0 mentsu, no pair) is 0 in every row
    return r;
}

// Draw-candidate property (mj_algo.h sh_draw_candidate_fields): every draw that lowers calc_all of a 3k+1-tile hand is inside
// the candidate set the SP kernel probes.  Needs the shanten tables: argv[1] = the MJT1 payload (mortal_amd.tables.payload()).
#include <vector>
#ifndef DRAW_CHECK_ITERS
#define DRAW_CHECK_ITERS 1500000
#endif
#ifndef DRAW_CHECK_SEED
#define DRAW_CHECK_SEED 777
#endif
static int check_draw_candidates(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { printf("cannot open %s\n", path); return 10; }
    std::vector<unsigned char> buf;
    unsigned char tmp[65536];
    size_t k;
    while ((k = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + k);
    fclose(f);
    uint32_t ns, nj;
    memcpy(&ns, &buf[4], 4);
    memcpy(&nj, &buf[8], 4);
    std::vector<uint64_t> su(ns), ji(nj);
    for (uint32_t i = 0; i < ns; i++) for (int b = 0; b < 5; b++) su[i] |= (uint64_t)buf[16 + (size_t)i * 5 + b] << (8 * b);
    for (uint32_t i = 0; i < nj; i++) for (int b = 0; b < 5; b++) ji[i] |= (uint64_t)buf[16 + (size_t)ns * 5 + (size_t)i * 5 + b] << (8 * b);
    MjTablesDev T{};
    T.suhai = su.data();
    T.n_suhai = ns;
    T.jihai = ji.data();
    T.n_jihai = nj;
    std::mt19937_64 g(DRAW_CHECK_SEED);
    long hands = 0, lowering = 0, cand_total = 0;
    for (int it = 0; it < DRAW_CHECK_ITERS; it++) {
        const int ld3 = it % 5 == 0 ? (int)(g() % 4) : 4, n_tiles = 3 * ld3 + 1, mode = it % 7;
        int cnt[34] = {0};
        // tile pool of the mode: everything / few kinds / one suit + honours / terminals + honours / triplet-heavy
        int pool[34], np = 0;
        if (mode <= 2) for (int t = 0; t < 34; t++) pool[np++] = t;
        else if (mode == 3) { const int kk = 2 + (int)(g() % 6); while (np < kk) { int t = (int)(g() % 34), dup = 0; for (int i = 0; i < np; i++) dup |= pool[i] == t; if (!dup) pool[np++] = t; } }
        else if (mode == 4) { const int s0 = 9 * (int)(g() % 3); for (int t = s0; t < s0 + 9; t++) pool[np++] = t; for (int t = 27; t < 34; t++) if (g() & 1) pool[np++] = t; }
        else if (mode == 5) { const int yao[13] = {0, 8, 9, 17, 18, 26, 27, 28, 29, 30, 31, 32, 33}; for (int i = 0; i < 13; i++) pool[np++] = yao[i]; for (int i = 0; i < 3; i++) pool[np++] = (int)(g() % 27); }
        else { const int kk = 4 + (int)(g() % 3); while (np < kk) { int t = (int)(g() % 34), dup = 0; for (int i = 0; i < np; i++) dup |= pool[i] == t; if (!dup) pool[np++] = t; } }
        if (np * 4 < n_tiles) continue;
        for (int placed = 0; placed < n_tiles;) {
            const int t = pool[g() % np];
            if (cnt[t] < 4) { cnt[t]++; placed++; }
        }
        Hand h = {0, 0};
        for (int t = 0; t < 34; t++) for (int c = 0; c < cnt[t]; c++) h.inc(t);
        const int base = calc_all(T, h, ld3), nk = h.n_kinds();
        u32 cf[4];
        for (int i = 0; i < 4; i++) {
            const u32 hn = (u32)(Hand::nz_fields(i < 2 ? h.mp : h.sz) >> ((i & 1) * 27)) & 0x7FFFFFFu;
            cf[i] = sh_draw_candidate_fields(hn, i, sh_draw_rule(base, ld3, h.n_pairs(), nk, h.n_yao_pairs(), h.n_yao_kinds(), h.has_quad()));
            cand_total += __builtin_popcount(cf[i]);
        }
        for (int t = 0; t < 34; t++) {
            if (cnt[t] == 4) continue;
            Hand x = h;
            x.inc(t);
            if (calc_all(T, x, ld3) < base) {
                lowering++;
                const int i = t < 27 ? t / 9 : 3, j = t < 27 ? t % 9 : t - 27;
                if (!((cf[i] >> (3 * j)) & 1)) {
                    printf("draw candidate missed: tile %d lowers %d, len_div3 %d, hand", t, base, ld3);
                    for (int q = 0; q < 34; q++) printf(" %d", cnt[q]);
                    printf("\n");
                    return 11;
                }
            }
        }
        hands++;
    }
    printf("draw candidates cover every shanten-lowering draw: %ld hands, %ld lowering draws, %.1f candidates per hand\n", hands, lowering,
           (double)cand_total / (double)hands);
    return 0;
}


// Table-id shanten (mj_sptab.h): required draws / shanten-keeping discards from the per-key wait / keep masks and the
// optimal-entry table, against the reference-shaped brute force (calc_all of every h + t, g - d; state.rs:100-173).
#ifndef SPTAB_CHECK_ITERS
#define SPTAB_CHECK_ITERS 400000
#endif
static bool load_tables(const char* path, std::vector<uint64_t>& su, std::vector<uint64_t>& ji) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    std::vector<unsigned char> buf;
    unsigned char tmp[65536];
    size_t k;
    while ((k = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + k);
    fclose(f);
    uint32_t ns, nj;
    memcpy(&ns, &buf[4], 4);
    memcpy(&nj, &buf[8], 4);
    su.assign(ns, 0);
    ji.assign(nj, 0);
    for (uint32_t i = 0; i < ns; i++) for (int b = 0; b < 5; b++) su[i] |= (uint64_t)buf[16 + (size_t)i * 5 + b] << (8 * b);
    for (uint32_t i = 0; i < nj; i++) for (int b = 0; b < 5; b++) ji[i] |= (uint64_t)buf[16 + (size_t)ns * 5 + (size_t)i * 5 + b] << (8 * b);
    return true;
}
static int check_sptab(const char* path) {
    std::vector<uint64_t> su, ji;
    if (!load_tables(path, su, ji)) { printf("cannot open %s\n", path); return 20; }
    MjTablesDev MT{};
    MT.suhai = su.data();
    MT.n_suhai = (uint32_t)su.size();
    MT.jihai = ji.data();
    MT.n_jihai = (uint32_t)ji.size();
    SpTabHost H;
    if (!sp_tab_build(su.data(), MT.n_suhai, ji.data(), MT.n_jihai, H)) { printf("sp_tab_build: %s\n", H.error.c_str()); return 21; }
    SpTabDev T{H.id.data(), H.mrg.data(), H.opt.data(), H.wk.data(), MT.n_suhai, MT.n_jihai, H.zero_id};
    printf("sp tables: %u distinct rows, merge closure %zu vectors\n", H.n_rows, H.vec.size());
    // merge / final tables == the nibble arithmetic of mj_algo.h, for every (vector, row) and every len_div3
    for (size_t v = 0; v < H.vec.size(); v++)
        for (uint32_t b = 0; b < H.n_rows; b++)
            for (int m = 0; m <= 4; m++) {
                const u64 mg = sh_merge(H.vec[v], H.vec[b], m), full = H.vec[H.mrg[v * SPT_NB + b]];
                // a truncated merge (entries beyond m keep the left operand's values) agrees with the full one on the entries a final for m reads
                for (int j = 0; j <= 9; j++)
                    if ((j <= m || (j >= 5 && j <= 5 + m)) && NIB(mg, j) != NIB(full, j)) { printf("merge table mismatch\n"); return 22; }
                if (spt_fin(spt_opt(T, m, (u32)v, b)) != sh_final(H.vec[v], H.vec[b], m)) { printf("final table mismatch\n"); return 23; }
            }
    std::mt19937_64 g(4242);
    long hands = 0, n_req = 0, n_keep = 0, n_fb = 0, n_slow = 0;
    auto check_hand = [&](Hand h, int ld3) -> int {
        const int L = calc_all(MT, h, ld3);
        int fin = -1;
        const u64 got = sp_req_of_hand(T, MT, h, ld3, L, &fin), want = sp_req_brute(MT, h, ld3, L);
        u64 full = 0;
        for (int t = 0; t < 34; t++) if (h.get(t) >= 4) full |= 1ull << t;
        if ((got & ~full) != want) {
            printf("required draws differ: ld3 %d L %d got %llx want %llx hand", ld3, L, (unsigned long long)got, (unsigned long long)want);
            for (int q = 0; q < 34; q++) printf(" %d", h.get(q));
            printf("\n");
            return 24;
        }
        if (fin - 1 != calc_normal(MT, h, ld3) && fin >= 0) {
            const SpSuitView v = sp_suit_view(T, h);
            bool inside = true;
            for (int s = 0; s < 4; s++) inside &= spt_in_table(T, s, v.key[s]);
            if (inside) { printf("normal-form number differs\n"); return 25; }
        }
        n_req += __builtin_popcountll(want);
        for (int t = 0; t < 34; t++) {
            if (h.get(t) >= 4) continue;
            if (!((want >> t) & 1) && (g() % 4)) continue;  // every required draw, a quarter of the others
            Hand x = h;
            x.inc(t);
            const int Tg = calc_all(MT, x, ld3);
            const u64 k_got = sp_keep_of_hand(T, MT, x, ld3, Tg), k_want = sp_keep_brute(MT, x, ld3, Tg);
            if (k_got != k_want) {
                printf("keeping discards differ: ld3 %d T %d got %llx want %llx hand", ld3, Tg, (unsigned long long)k_got, (unsigned long long)k_want);
                for (int q = 0; q < 34; q++) printf(" %d", x.get(q));
                printf("\n");
                return 26;
            }
            n_keep += __builtin_popcountll(k_want);
        }
        hands++;
        return 0;
    };
    for (int it = 0; it < SPTAB_CHECK_ITERS; it++) {
        const int ld3 = it % 5 == 0 ? (int)(g() % 4) : 4, n_tiles = 3 * ld3 + 1, mode = it % 9;
        int cnt[34] = {0};
        int pool[34], np = 0;
        if (mode <= 2) for (int t = 0; t < 34; t++) pool[np++] = t;
        else if (mode == 3) { const int kk = 2 + (int)(g() % 6); while (np < kk) { int t = (int)(g() % 34), dup = 0; for (int i = 0; i < np; i++) dup |= pool[i] == t; if (!dup) pool[np++] = t; } }
        else if (mode == 4) { const int s0 = 9 * (int)(g() % 3); for (int t = s0; t < s0 + 9; t++) pool[np++] = t; for (int t = 27; t < 34; t++) if (g() & 1) pool[np++] = t; }
        else if (mode == 5) { const int yao[13] = {0, 8, 9, 17, 18, 26, 27, 28, 29, 30, 31, 32, 33}; for (int i = 0; i < 13; i++) pool[np++] = yao[i]; for (int i = 0; i < 3; i++) pool[np++] = (int)(g() % 27); }
        else if (mode == 6) { const int kk = 4 + (int)(g() % 3); while (np < kk) { int t = (int)(g() % 34), dup = 0; for (int i = 0; i < np; i++) dup |= pool[i] == t; if (!dup) pool[np++] = t; } }
        else if (mode == 7) { const int s0 = 9 * (int)(g() % 3); for (int t = s0; t < s0 + 9; t++) pool[np++] = t; }  // one suit only
        else { const int kk = 7 + (int)(g() % 3); while (np < kk) { int t = (int)(g() % 34), dup = 0; for (int i = 0; i < np; i++) dup |= pool[i] == t; if (!dup) pool[np++] = t; } }  // pairs-heavy
        if (np * 4 < n_tiles) continue;
        if (mode == 8) {  // seven-pairs shapes: pairs first, then singles
            int placed = 0;
            for (int i = 0; i < np && placed + 2 <= n_tiles && i < 6; i++) { cnt[pool[i]] = 2; placed += 2; }
            while (placed < n_tiles) { const int t = pool[g() % np]; if (cnt[t] < 4) { cnt[t]++; placed++; } }
        } else {
            for (int placed = 0; placed < n_tiles;) {
                const int t = pool[g() % np];
                if (cnt[t] < 4) { cnt[t]++; placed++; }
            }
        }
        Hand h = {0, 0};
        for (int t = 0; t < 34; t++) for (int c = 0; c < cnt[t]; c++) h.inc(t);
        const int rc = check_hand(h, ld3);
        if (rc) return rc;
    }
    // the 13-tile one-suit patterns whose +1 neighbour lies past the table (fallback records), in every suit
    const int fb[8][5] = {{3, 4, 4, 1, 1}, {3, 4, 4, 2, 0}, {4, 3, 4, 1, 1}, {4, 3, 4, 2, 0}, {4, 4, 3, 1, 1}, {4, 4, 3, 2, 0}, {4, 4, 4, 0, 1}, {4, 4, 4, 1, 0}};
    for (int s = 0; s < 3; s++)
        for (int q = 0; q < 8; q++) {
            Hand h = {0, 0};
            int n = 0;
            for (int i = 0; i < 5; i++) for (int c = 0; c < fb[q][i]; c++) { h.inc(9 * s + i); n++; }
            while (n < 13) { h.inc(27 + n % 7); n++; }
            const SpSuitView v = sp_suit_view(T, h);
            if (n == 13 && spt_in_table(T, s, v.key[s]) && (spt_rec(T, s, v.key[s], 0).w & SPT_FALLBACK)) n_fb++;
            const int rc = check_hand(h, 4);
            if (rc) return rc;
        }
    printf("table-id shanten sets == brute force: %ld hands, %ld required draws, %ld keeping discards, %ld fallback patterns\n", hands, n_req, n_keep, n_fb);
    (void)n_slow;
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1) {
        int rc = check_draw_candidates(argv[1]);
        if (rc) return rc;
        rc = check_sptab(argv[1]);
        if (rc) return rc;
    }
    std::mt19937_64 g(12345);
    long n = 0;
    for (int it = 0; it < 2000000; it++) {
        const int max_nib = it % 3 == 0 ? 15 : it % 3 == 1 ? 14 : 6;
        const u64 a = random_row(g, max_nib), b = random_row(g, max_nib);
        for (int m = 0; m <= 4; m++) {
            const int want = sh_final_ref(a, b, m), got = sh_final(a, b, m);
            if (want != got) {
                printf("sh_final mismatch: a=%010llx b=%010llx m=%d want %d got %d\n", (unsigned long long)a, (unsigned long long)b, m, want, got);
                return 1;
            }
            n++;
        }
    }
    // the partial merges may be taken in any order (the SP kernel merges the untouched suits with the row of h - d first and
    // finishes with the row of h + t, or the other way round)
    for (int it = 0; it < 1000000; it++) {
        const u64 a = random_row(g, 14), b = random_row(g, 14), c = random_row(g, 14), d = random_row(g, 14);
        for (int m = 0; m <= 4; m++) {
            const u64 ab = sh_merge(a, b, m);
            const int x = sh_final(sh_merge(ab, c, m), d, m), y = sh_final(sh_merge(ab, d, m), c, m);
            const int z = sh_final(sh_merge(sh_merge(b, a, m), d, m), c, m);
            if (x != y || x != z) {
                printf("merge order mismatch m=%d: %d %d %d\n", m, x, y, z);
                return 3;
            }
        }
    }
    // all-zero rows (keys past the table, `unwrap_or_default`)
    for (int m = 0; m <= 4; m++)
        if (sh_final(0, 0, m) != sh_final_ref(0, 0, m)) return 2;
    printf("sh_final == reference-shaped loop on %ld cases\n", n);
    // sp_div (the SP kernel's division with a hoisted reciprocal) == IEEE division, for the reciprocal estimate the host
    // uses (RN(1/b)) and for both of its neighbours (v_rcp_f32 is accurate to 1 ulp).
    // (1) the kernel's true operand domain, exhaustively: prob = tsumo_prob[c][j] * not_tsumo[s][j] / not_tsumo[s][i]
    //     (calc.rs:135-167,486-548) for every wall size, required-tile sum, count and turn pair;
    auto check_div = [](float a, float b) -> bool {
        const float want = a / b;
        const float r = 1.0f / b;
        for (int k = -1; k <= 1; k++) {
            const float r0 = k < 0 ? nextafterf(r, 0.f) : k > 0 ? nextafterf(r, 2.f * r) : r;
            const float e0 = __builtin_fmaf(-b, r0, 1.0f), r1 = __builtin_fmaf(e0, r0, r0);
            if (sp_div(a, b, r1) != want) {
                printf("sp_div mismatch: a=%a b=%a k=%d want %a got %a\n", a, b, k, want, sp_div(a, b, r1));
                return false;
            }
        }
        return sp_div(a, b, sp_rcp_refined(b)) == want;
    };
    auto check_div_domain = [](float a, float b) -> bool {  // the 3-instruction form the kernel uses: its true operand domain only
        const float want = a / b;
        const float r = 1.0f / b;
        for (int k = -1; k <= 1; k++) {
            const float r0 = k < 0 ? nextafterf(r, 0.f) : k > 0 ? nextafterf(r, 2.f * r) : r;
            const float e0 = __builtin_fmaf(-b, r0, 1.0f), r1 = __builtin_fmaf(e0, r0, r0);
            if (sp_div_domain(a, b, r1) != want) {
                printf("sp_div_domain mismatch: a=%a b=%a k=%d want %a got %a\n", a, b, k, want, sp_div_domain(a, b, r1));
                return false;
            }
        }
        return true;
    };
    long nd = 0;
    for (int n_left = 1; n_left <= 123; n_left++)
        for (int sumreq = 0; sumreq <= n_left; sumreq++) {
            const int T = n_left < 17 ? n_left : 17;
            float nt[17], tp[4][17];
            for (int j = 0; j < 17; j++) nt[j] = 0.f;
            nt[0] = 1.f;
            const int lim = (T - 1) < (n_left - sumreq) ? (T - 1) : (n_left - sumreq);
            for (int j = 0; j < lim; j++) nt[j + 1] = nt[j] * (float)(n_left - sumreq - j) / (float)(n_left - j);
            for (int c = 0; c < 4; c++)
                for (int j = 0; j < T; j++) tp[c][j] = (float)(c + 1) / (float)(n_left - j);
            for (int c = 0; c < 4; c++)
                for (int i = 0; i < T; i++)
                    for (int j = i; j < T; j++) {
                        if (nt[i] == 0.f) continue;
                        if (!check_div(tp[c][j] * nt[j], nt[i]) || !check_div_domain(tp[c][j] * nt[j], nt[i])) return 4;
                        nd++;
                    }
        }
    // (2) random operands over a wider range than the domain: divisor in [2^-70, 2), quotient in [2^-25, 2) — the
    //     numerator stays above 2^-96, where the hardware sequence needs no operand scaling (v_div_scale acts below
    //     2^-103) and every residual of the sequence is a normal number
    std::uniform_real_distribution<double> exb(-70.0, 0.0), exq(-25.0, 0.0);
    for (int it = 0; it < 20000000; it++) {
        const float b = (float)exp2(exb(g)) * (1.f + (float)(g() & 0xFFFFFF) / 16777216.f);
        const float a = b * (float)exp2(exq(g)) * (1.f + (float)(g() & 0xFFFFFF) / 16777216.f);
        if (!check_div(a, b)) return 5;
        nd++;
    }
    printf("sp_div == IEEE division on %ld operand pairs (sp_div_domain on the kernel's whole operand domain)\n", nd);
    return 0;
}
