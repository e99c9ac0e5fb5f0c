// Host-side equivalence checks of the pure device helpers in mortal_amd/csrc/mj_algo.h (no GPU needed): the helpers
// are compiled __host__ __device__ here and the optimised formulations are compared with the reference-shaped ones.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -o algo_check tests/host/algo_check.hip && ./algo_check
#define MJD __host__ __device__ inline
#define MJDN __device__ __noinline__
#include "../../mortal_amd/csrc/mj_algo.h"

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
    return 0;
}
