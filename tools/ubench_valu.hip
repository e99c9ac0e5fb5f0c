// Issue-rate microbenchmark of the integer / f32 VALU, SALU and LDS instructions the SP kernel is made of (gfx950).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/ubench_valu tools/ubench_valu.hip && /tmp/ubench_valu
// For every instruction class: a loop of 8 independent dependency chains x 8 repeats (64 instructions per iteration) per
// wave, run with W = 1, 2, 4 waves per SIMD on every CU; reports shader cycles (s_memtime) per wave-instruction seen by one
// wave and the aggregate per-SIMD issue interval (cycles per instruction per SIMD = wave cycles / W).  The roofline of
// mj_k_sp is priced with these numbers, not with the f32 FMA rate.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define REP8(S) S S S S S S S S
#define CHAINS(OP)                                                          \
    OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)

enum Kind { K_ADD, K_AND, K_BFE, K_CNDMASK, K_LSHR64, K_LSHL64, K_MUL_LO, K_MUL_U24, K_MAD_U24, K_BCNT, K_MIN3, K_LSHL_ADD_U64,
            K_MAD_U64_U32, K_FMA_F32, K_MUL_F32, K_ADD_F32, K_PK_FMA_F32, K_RCP_F32, K_CMP_U32, K_LSHL_ADD_U32, K_PERM, K_ALIGNBIT,
            K_FFBL, K_ADD_CO, K_SALU_ADD, K_SALU_AND64, K_DS_READ_B32, K_DS_READ_B64, K_DS_WRITE_B32, K_MOV_DPP, K_READLANE, K_N };
static const char* kind_name[K_N] = {
    "v_add_u32", "v_and_b32", "v_bfe_u32", "v_cndmask_b32", "v_lshrrev_b64", "v_lshlrev_b64", "v_mul_lo_u32", "v_mul_u32_u24",
    "v_mad_u32_u24", "v_bcnt_u32_b32", "v_min3_i32", "v_lshl_add_u64", "v_mad_u64_u32", "v_fma_f32", "v_mul_f32", "v_add_f32",
    "v_pk_fma_f32", "v_rcp_f32", "v_cmp_lt_u32(vcc)", "v_lshl_add_u32", "v_perm_b32", "v_alignbit_b32", "v_ffbl_b32", "v_add_co_u32",
    "s_add_u32", "s_and_b64", "ds_read_b32", "ds_read_b64", "ds_write_b32", "v_mov_b32 dpp row_shr", "v_readlane_b32"};

template <int KIND>
__global__ __launch_bounds__(256) void k_bench(int iters, uint32_t seed, unsigned long long* out_cycles, uint32_t* sink) {
    __shared__ uint32_t lds[2048];
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 11 + 4, a5 = a0 * 13 + 5,
             a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;
    uint64_t b0 = a0 | ((uint64_t)a1 << 32), b1 = a1 | ((uint64_t)a2 << 32), b2 = a2 | ((uint64_t)a3 << 32), b3 = a3 | ((uint64_t)a4 << 32),
             b4 = a4 | ((uint64_t)a5 << 32), b5 = a5 | ((uint64_t)a6 << 32), b6 = a6 | ((uint64_t)a7 << 32), b7 = a7 | ((uint64_t)a0 << 32);
    float f0 = a0 * 1e-9f, f1 = a1 * 1e-9f, f2 = a2 * 1e-9f, f3 = a3 * 1e-9f, f4 = a4 * 1e-9f, f5 = a5 * 1e-9f, f6 = a6 * 1e-9f, f7 = a7 * 1e-9f;
    uint32_t k = seed | 1, sh = (seed & 7) + 1;
    float fk = 1.0000001f;
    uint32_t s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3;
    uint64_t t0 = seed, t1 = seed + 5;
    lds[threadIdx.x] = a0; lds[threadIdx.x + 256] = a1;
    const uint32_t laddr = (threadIdx.x * 4) & 8191;
    __syncthreads();
    const long long c0 = clock64();
    for (int it = 0; it < iters; it++) {
        if constexpr (KIND == K_ADD) {
#define OP(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(k));
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_AND) {
#define OP(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(k));
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_BFE) {
#define OP(x) asm volatile("v_bfe_u32 %0, %0, %1, 31" : "+v"(x) : "v"(sh));
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_CNDMASK) {
#define OP(x) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(k), "s"(t1));  // the mask in an SGPR pair
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_LSHR64) {
#define OP(x) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(x) : "v"(sh));
            REP8(OP(b0) OP(b1) OP(b2) OP(b3) OP(b4) OP(b5) OP(b6) OP(b7))
#undef OP
        } else if constexpr (KIND == K_LSHL64) {
#define OP(x) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(x) : "v"(sh));
            REP8(OP(b0) OP(b1) OP(b2) OP(b3) OP(b4) OP(b5) OP(b6) OP(b7))
#undef OP
        } else if constexpr (KIND == K_MUL_LO) {
#define OP(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(k));
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_MUL_U24) {
#define OP(x) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(k));
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_MAD_U24) {
#define OP(x) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(k));
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_BCNT) {
#define OP(x) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(x) : "v"(k));
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_MIN3) {
#define OP(x) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(x) : "v"(k), "v"(sh));
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_LSHL_ADD_U64) {
#define OP(x) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(x) : "v"(b7));
            REP8(OP(b0) OP(b1) OP(b2) OP(b3) OP(b4) OP(b5) OP(b6) OP(b0))
#undef OP
        } else if constexpr (KIND == K_MAD_U64_U32) {
#define OP(x) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(k), "v"(sh) : "vcc");
            REP8(OP(b0) OP(b1) OP(b2) OP(b3) OP(b4) OP(b5) OP(b6) OP(b7))
#undef OP
        } else if constexpr (KIND == K_FMA_F32) {
#define OP(x) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(fk));
            REP8(OP(f0) OP(f1) OP(f2) OP(f3) OP(f4) OP(f5) OP(f6) OP(f7))
#undef OP
        } else if constexpr (KIND == K_MUL_F32) {
#define OP(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(fk));
            REP8(OP(f0) OP(f1) OP(f2) OP(f3) OP(f4) OP(f5) OP(f6) OP(f7))
#undef OP
        } else if constexpr (KIND == K_ADD_F32) {
#define OP(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(fk));
            REP8(OP(f0) OP(f1) OP(f2) OP(f3) OP(f4) OP(f5) OP(f6) OP(f7))
#undef OP
        } else if constexpr (KIND == K_PK_FMA_F32) {
#define OP(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(b7));
            REP8(OP(b0) OP(b1) OP(b2) OP(b3) OP(b4) OP(b5) OP(b6) OP(b0))
#undef OP
        } else if constexpr (KIND == K_RCP_F32) {
#define OP(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
            REP8(OP(f0) OP(f1) OP(f2) OP(f3) OP(f4) OP(f5) OP(f6) OP(f7))
#undef OP
        } else if constexpr (KIND == K_CMP_U32) {
#define OP(x) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(x), "v"(k) : "vcc");
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_LSHL_ADD_U32) {
#define OP(x) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x) : "v"(k));
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_PERM) {
#define OP(x) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(k), "v"(sh));
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_ALIGNBIT) {
#define OP(x) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(x) : "v"(k), "v"(sh));
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_FFBL) {
#define OP(x) asm volatile("v_ffbl_b32 %0, %0" : "+v"(x));
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_ADD_CO) {
#define OP(x) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(x) : "v"(k) : "vcc");
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_SALU_ADD) {
#define OP(x) asm volatile("s_add_u32 %0, %0, %1" : "+s"(x) : "s"(s3) : "scc");
            REP8(OP(s0) OP(s1) OP(s2) OP(s0) OP(s1) OP(s2) OP(s0) OP(s1))
#undef OP
        } else if constexpr (KIND == K_SALU_AND64) {
#define OP(x) asm volatile("s_and_b64 %0, %0, %1" : "+s"(x) : "s"(t1) : "scc");
            REP8(OP(t0) OP(t0) OP(t0) OP(t0) OP(t0) OP(t0) OP(t0) OP(t0))
#undef OP
        } else if constexpr (KIND == K_DS_READ_B32) {
#define OP(x) asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"(laddr) : "memory");
            REP8(CHAINS(OP))
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef OP
        } else if constexpr (KIND == K_DS_READ_B64) {
#define OP(x) asm volatile("ds_read_b64 %0, %1" : "=v"(x) : "v"(laddr * 2 & 8191) : "memory");
            REP8(OP(b0) OP(b1) OP(b2) OP(b3) OP(b4) OP(b5) OP(b6) OP(b7))
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef OP
        } else if constexpr (KIND == K_DS_WRITE_B32) {
#define OP(x) asm volatile("ds_write_b32 %1, %0" : : "v"(x), "v"(laddr) : "memory");
            REP8(CHAINS(OP))
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef OP
        } else if constexpr (KIND == K_MOV_DPP) {
#define OP(x) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x));
            REP8(CHAINS(OP))
#undef OP
        } else if constexpr (KIND == K_READLANE) {
#define OP(x) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s0) : "v"(x));
            REP8(CHAINS(OP))
#undef OP
        }
    }
    const long long c1 = clock64();
    uint32_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7) ^ s0 ^ s1 ^ s2 ^ (uint32_t)t0;
    r ^= __float_as_uint(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
    if (r == 0x12345678u) sink[0] = r;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out_cycles[threadIdx.x >> 6] = (unsigned long long)(c1 - c0);
}

template <int KIND>
int run_kind(int iters, unsigned long long* d_cyc, uint32_t* d_sink) {
    // W waves per SIMD on every CU: blocks of 256 threads (4 waves = one per SIMD), W blocks per CU
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
        // clock64() = s_memtime (shader clock); wall: ms * 2.4e6 cycles at the nominal clock
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
    return run_all<0>(20000, d_cyc, d_sink);
}
