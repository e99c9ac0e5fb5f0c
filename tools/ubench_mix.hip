// Issue model of a gfx950 SIMD for MIXED instruction streams (round 4): do the SALU / LDS instructions of one wave issue beside
// the VALU instructions of another (time = max of the classes), or does every instruction take its own issue slot (time = sum)?
// mj_k_sp executes ~340 VALU + ~140 SALU + ~30 LDS + ~15 VMEM wave-instructions per state with the SIMDs 88 % "active"
// (SQ_ACTIVE_INST_ANY) — this decides whether its bound is the VALU count or the total instruction count.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/ubench_mix tools/ubench_mix.hip && tools/bin/ubench_mix
// Each kernel: W = 1 / 2 / 4 waves per SIMD on every CU, a loop of 64 VALU instructions (8 independent chains) with N_S scalar (or
// LDS) instructions interleaved; reports aggregate cycles per LOOP ITERATION per SIMD at the nominal 2.4 GHz.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define V8(OPV) OPV(a0) OPV(a1) OPV(a2) OPV(a3) OPV(a4) OPV(a5) OPV(a6) OPV(a7)
#define BFE(x) asm volatile("v_bfe_u32 %0, %0, %1, 31" : "+v"(x) : "v"(sh));
#define VADD(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(k));
#define SADD(s) asm volatile("s_add_u32 %0, %0, 7" : "+s"(s) : : "scc");
#define SAND(t) asm volatile("s_and_b64 %0, %0, -3" : "+s"(t) : : "scc");
#define DSR(x) asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"(laddr) : "memory");
// one VALU + one scalar per pair
#define PAIR_BFE_SADD(x, s) BFE(x) SADD(s)
#define PAIR_ADD_SADD(x, s) VADD(x) SADD(s)
#define PAIR_BFE_SAND(x, t) BFE(x) SAND(t)

enum { M_BFE64, M_BFE64_SADD64, M_BFE64_SADD32, M_SADD64, M_ADD64, M_ADD64_SADD64, M_BFE64_SAND64, M_BFE64_DSR16, M_BFE64_BRANCH16, M_N };
static const char* names[M_N] = {"64 v_bfe", "64 v_bfe + 64 s_add", "64 v_bfe + 32 s_add", "64 s_add", "64 v_add_u32", "64 v_add_u32 + 64 s_add",
                                 "64 v_bfe + 64 s_and_b64", "64 v_bfe + 16 ds_read_b32 + waitcnt", "64 v_bfe + 16 (s_cmp + s_cbranch not taken)"};

template <int M>
__global__ __launch_bounds__(256) void k_mix(int iters, uint32_t seed, uint32_t* sink) {
    __shared__ uint32_t lds[2048];
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;
    uint32_t d0 = 0, d1 = 0;
    uint32_t k = seed | 1, sh = (seed & 7) + 1;
    uint32_t s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3, s4 = seed + 4, s5 = seed + 5, s6 = seed + 6, s7 = seed + 7;
    uint64_t t0 = seed, t1 = seed + 5, t2 = seed + 6, t3 = seed + 7;
    lds[threadIdx.x] = a0;
    const uint32_t laddr = (threadIdx.x * 4) & 8191;
    __syncthreads();
    for (int it = 0; it < iters; it++) {
        if constexpr (M == M_BFE64) {
#pragma unroll
            for (int r = 0; r < 8; r++) { V8(BFE) }
        } else if constexpr (M == M_BFE64_SADD64) {
#pragma unroll
            for (int r = 0; r < 8; r++) { BFE(a0) SADD(s0) BFE(a1) SADD(s1) BFE(a2) SADD(s2) BFE(a3) SADD(s3) BFE(a4) SADD(s4) BFE(a5) SADD(s5) BFE(a6) SADD(s6) BFE(a7) SADD(s7) }
        } else if constexpr (M == M_BFE64_SADD32) {
#pragma unroll
            for (int r = 0; r < 8; r++) { BFE(a0) SADD(s0) BFE(a1) BFE(a2) SADD(s2) BFE(a3) BFE(a4) SADD(s4) BFE(a5) BFE(a6) SADD(s6) BFE(a7) }
        } else if constexpr (M == M_SADD64) {
#pragma unroll
            for (int r = 0; r < 8; r++) { SADD(s0) SADD(s1) SADD(s2) SADD(s3) SADD(s4) SADD(s5) SADD(s6) SADD(s7) }
        } else if constexpr (M == M_ADD64) {
#pragma unroll
            for (int r = 0; r < 8; r++) { V8(VADD) }
        } else if constexpr (M == M_ADD64_SADD64) {
#pragma unroll
            for (int r = 0; r < 8; r++) { VADD(a0) SADD(s0) VADD(a1) SADD(s1) VADD(a2) SADD(s2) VADD(a3) SADD(s3) VADD(a4) SADD(s4) VADD(a5) SADD(s5) VADD(a6) SADD(s6) VADD(a7) SADD(s7) }
        } else if constexpr (M == M_BFE64_SAND64) {
#pragma unroll
            for (int r = 0; r < 8; r++) { BFE(a0) SAND(t0) BFE(a1) SAND(t1) BFE(a2) SAND(t2) BFE(a3) SAND(t3) BFE(a4) SAND(t0) BFE(a5) SAND(t1) BFE(a6) SAND(t2) BFE(a7) SAND(t3) }
        } else if constexpr (M == M_BFE64_DSR16) {
#pragma unroll
            for (int r = 0; r < 8; r++) { BFE(a0) DSR(d0) BFE(a1) BFE(a2) BFE(a3) BFE(a4) DSR(d1) BFE(a5) BFE(a6) BFE(a7) }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (M == M_BFE64_BRANCH16) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                BFE(a0) BFE(a1) BFE(a2) BFE(a3)
                asm volatile("s_cmp_eq_u32 %0, 0x7fffffff\n s_cbranch_scc1 1f\n 1:" : : "s"(s0) : "scc");
                BFE(a4) BFE(a5) BFE(a6) BFE(a7)
                asm volatile("s_cmp_eq_u32 %0, 0x7ffffffe\n s_cbranch_scc1 2f\n 2:" : : "s"(s1) : "scc");
            }
        }
    }
    uint32_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7 ^ (uint32_t)(t0 ^ t1 ^ t2 ^ t3) ^ d0 ^ d1;
    if (r == 0x12345678u) sink[0] = r;
}

template <int M>
int run(int iters, uint32_t* d_sink) {
    for (int W : {1, 2, 4}) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_mix<M>, dim3(256 * W), dim3(256), 0, 0, iters / 8, 12345u, d_sink);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_mix<M>, dim3(256 * W), dim3(256), 0, 0, iters, 12345u, d_sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("{\"mix\": \"%s\", \"waves_per_simd\": %d, \"nominal_cycles_per_iteration_per_simd\": %.2f, \"ms\": %.3f}\n", names[M], W,
               (double)ms * 2.4e6 / ((double)iters * W), ms);
    }
    return 0;
}
template <int M> int run_all(int iters, uint32_t* d) {
    if constexpr (M < M_N) { if (run<M>(iters, d)) return 1; return run_all<M + 1>(iters, d); }
    return 0;
}
int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    uint32_t* d_sink;
    CHECK(hipMalloc(&d_sink, 64));
    return run_all<0>(20000, d_sink);
}
