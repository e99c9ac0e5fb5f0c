// Issue-rate microbenchmark, round 5: the select / packed-f32 / lane-move instruction forms of mj_k_sp's accumulate and of its
// SGPR-spill traffic (gfx950).  Same harness as tools/ubench_valu.hip: 8 independent chains x 8 repeats = 64 instructions per loop
// iteration per wave, W = 1, 2, 4 waves per SIMD on every CU; reports the aggregate issue interval per SIMD at the nominal 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/ubench_sel tools/ubench_sel.hip && tools/bin/ubench_sel
// Question it answers: round 3's table priced v_cndmask_b32_e64 with an SGPR-pair mask at 23 cycles per instruction (5 x any other
// VALU class).  Is that the instruction, or the harness (the mask operand)?  -> the e32 / VCC form, the e64 form with a mask that is
// loop-invariant, v_and_b32 by an all-ones / zero mask and v_mul_f32 by 0 / 1 as replacements, the packed f32 forms next to the
// scalar ones, v_readlane / v_writelane (SGPR spills) and v_mov_b32.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define REP8(S) S S S S S S S S
#define CHAINS(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#define CHAINS64(OP) OP(b0) OP(b1) OP(b2) OP(b3) OP(b4) OP(b5) OP(b6) OP(b7)

enum Kind { K_CND_E32_VCC, K_CND_E64_SGPR, K_CND_E64_VCC, K_CND_E64_SGPR_F32ZERO, K_AND_MASK, K_MUL_MASK, K_CMP_CND_PAIR, K_PK_MUL, K_PK_ADD, K_PK_FMA,
            K_MUL_F32, K_ADD_F32, K_FMA_F32, K_MOV, K_READLANE, K_WRITELANE, K_CVT_I32_F32, K_MAX_I32, K_BFE_I32, K_N };
static const char* kind_name[K_N] = {
    "v_cndmask_b32_e32 (vcc)", "v_cndmask_b32_e64 (sgpr pair)", "v_cndmask_b32_e64 (vcc)", "v_cndmask_b32_e64 v, v, 0, sgpr (as hipcc emits it)",
    "v_and_b32 (mask in vgpr)", "v_mul_f32 (0/1 mask in vgpr)", "v_cmp_le_i32 + v_cndmask_b32_e32 (pair = 2 insts)", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32",
    "v_mul_f32", "v_add_f32", "v_fma_f32", "v_mov_b32", "v_readlane_b32", "v_writelane_b32", "v_cvt_i32_f32", "v_max_i32", "v_bfe_i32"};

template <int KIND>
__global__ __launch_bounds__(256) void k_bench(int iters, uint32_t seed, unsigned long long* out_cycles, uint32_t* sink) {
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 11 + 4, a5 = a0 * 13 + 5,
             a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;
    uint64_t b0 = a0 | ((uint64_t)a1 << 32), b1 = a1 | ((uint64_t)a2 << 32), b2 = a2 | ((uint64_t)a3 << 32), b3 = a3 | ((uint64_t)a4 << 32),
             b4 = a4 | ((uint64_t)a5 << 32), b5 = a5 | ((uint64_t)a6 << 32), b6 = a6 | ((uint64_t)a7 << 32), b7 = a7 | ((uint64_t)a0 << 32);
    uint32_t k = seed | 1;
    uint32_t vmask = (threadIdx.x & 1) ? 0xFFFFFFFFu : 0u;
    uint32_t fmask = (threadIdx.x & 1) ? 0x3F800000u : 0u;
    uint64_t pk = 0x3F8000013F800001ull;
    uint32_t fk = 0x3F800001u;
    uint64_t smask = 0x5555555555555555ull ^ seed;
    uint32_t s0 = seed;
    asm volatile("s_mov_b64 vcc, %0" : : "s"(smask) : "vcc");
    const long long c0 = clock64();
    for (int it = 0; it < iters; it++) {
        if constexpr (KIND == K_CND_E32_VCC) {
#define OP(x) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(k));
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_CND_E64_SGPR) {
#define OP(x) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(k), "s"(smask));
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_CND_E64_VCC) {
#define OP(x) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(x) : "v"(k));
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_CND_E64_SGPR_F32ZERO) {
#define OP(x) asm volatile("v_cndmask_b32_e64 %0, %0, 0, %1" : "+v"(x) : "s"(smask));
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_AND_MASK) {
#define OP(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(vmask));
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_MUL_MASK) {
#define OP(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(fmask));
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_CMP_CND_PAIR) {  // 32 pairs = 64 instructions
#define OP(x) asm volatile("v_cmp_le_i32 vcc, %1, %0\n v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(k) : "vcc");
            OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
            OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#undef OP
        } else if constexpr (KIND == K_PK_MUL) {
#define OP(x) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(pk));
            REP8(CHAINS64(OP))
#undef OP
        } else if constexpr (KIND == K_PK_ADD) {
#define OP(x) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(pk));
            REP8(CHAINS64(OP))
#undef OP
        } else if constexpr (KIND == K_PK_FMA) {
#define OP(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(pk));
            REP8(CHAINS64(OP))
#undef OP
        } else if constexpr (KIND == K_MUL_F32) {
#define OP(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(fk));
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_ADD_F32) {
#define OP(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(fk));
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_FMA_F32) {
#define OP(x) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(fk));
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_MOV) {
#define OP(x) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(k));
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_READLANE) {
#define OP(x) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s0) : "v"(x));
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_WRITELANE) {
#define OP(x) asm volatile("v_writelane_b32 %0, %1, 5" : "+v"(x) : "s"(s0));
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_CVT_I32_F32) {
#define OP(x) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x));
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_MAX_I32) {
#define OP(x) asm volatile("v_max_i32 %0, %0, %1" : "+v"(x) : "v"(k));
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_BFE_I32) {
#define OP(x) asm volatile("v_bfe_i32 %0, %0, 3, 1" : "+v"(x));
            REP8(CHAINS(OP))
#undef OP
        }
    }
    const long long c1 = clock64();
    uint32_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7) ^ s0;
    if (r == 0x12345678u) sink[0] = r;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out_cycles[threadIdx.x >> 6] = (unsigned long long)(c1 - c0);
}

template <int KIND>
int run_kind(int iters, unsigned long long* d_cyc, uint32_t* d_sink) {
    for (int W : {1, 2, 4}) {
        const int blocks = 256 * W;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_bench<KIND>, dim3(blocks), dim3(256), 0, 0, iters / 8, 12345u, d_cyc, d_sink);  // warm
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_bench<KIND>, dim3(blocks), dim3(256), 0, 0, iters, 12345u, d_cyc, d_sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long cyc[4];
        CHECK(hipMemcpy(cyc, d_cyc, sizeof(cyc), hipMemcpyDeviceToHost));
        const double n_inst = (double)iters * 64.0;
        printf("{\"inst\": \"%s\", \"waves_per_simd\": %d, \"memtime_ticks_per_wave_inst\": %.3f, \"nominal_cycles_per_inst_per_simd\": %.3f, \"ms\": %.3f}\n",
               kind_name[KIND], W, (double)cyc[0] / n_inst, (double)ms * 2.4e6 / (n_inst * W), ms);
    }
    return 0;
}

template <int K>
int run_all(int iters, unsigned long long* d_cyc, uint32_t* d_sink) {
    if constexpr (K < K_N) {
        if (run_kind<K>(iters, d_cyc, d_sink)) return 1;
        return run_all<K + 1>(iters, d_cyc, d_sink);
    }
    return 0;
}

int main() {
    unsigned long long* d_cyc;
    uint32_t* d_sink;
    CHECK(hipMalloc(&d_cyc, 64));
    CHECK(hipMalloc(&d_sink, 64));
    return run_all<0>(10000, d_cyc, d_sink);
}
