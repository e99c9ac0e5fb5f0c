// Host-side equivalence checks of the pure device helpers in mortal_amd/csrc/mj_algo.h (no GPU needed): the helpers
// are compiled __host__ __device__ here and the optimised formulations are compared with the reference-shaped ones.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -o algo_check tests/host/algo_check.hip && ./algo_check
#define MJD __host__ __device__ inline
#define MJDN __device__ __noinline__
#include "../../mortal_amd/csrc/mj_algo.h"

#include <cmath>
#include <cstdio>
#include <random>

static u64 random_row(std::mt19937_64& g, int max_nib) {
    u64 r = 0;
    for (int j = 1; j < 10; j++) r |= (u64)(g() % (max_nib + 1)) << (4 * j);  // nibble 0 (0 mentsu, no pair) is 0 in every row
    return r;
}

int main() {
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
                        if (!check_div(tp[c][j] * nt[j], nt[i])) return 4;
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
    printf("sp_div == IEEE division on %ld operand pairs\n", nd);
    return 0;
}
