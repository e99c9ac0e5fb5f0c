// Round 6 probe: does a CU-masked stream (hipExtStreamCreateWithCUMask) confine a kernel to a subset of the 256 CUs of an MI355X, and can two
// kernels on two disjointly masked streams run side by side?   hipcc --offload-arch=gfx950 -O2 -o tools/bin/cumask_probe tools/cumask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_where(uint32_t* out, long long spin) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
static int report(const char* name, const std::vector<uint32_t>& h) {
    std::set<uint32_t> cus;
    std::set<uint32_t> xccs;
    for (size_t i = 0; i < h.size() / 2; i++) {
        const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xF;
        const uint32_t cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        cus.insert(xcc << 16 | se << 8 | sh << 4 | cu);
        xccs.insert(xcc);
    }
    printf("%s: %zu workgroups on %zu distinct CUs, %zu XCCs\n", name, h.size() / 2, cus.size(), xccs.size());
    return (int)cus.size();
}
int main() {
    const int n = 4096;
    uint32_t *d0, *d1;
    CK(hipMalloc(&d0, 2 * n * 4)); CK(hipMalloc(&d1, 2 * n * 4));
    std::vector<uint32_t> h(2 * n);
    hipLaunchKernelGGL(k_where, dim3(n), dim3(256), 0, 0, d0, 2000LL);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d0, 2 * n * 4, hipMemcpyDeviceToHost));
    report("unmasked", h);
    // mask words: bit i of the flattened CU index
    for (int variant = 0; variant < 3; variant++) {
        std::vector<uint32_t> ma(8, 0), mb(8, 0);
        for (int i = 0; i < 256; i++) {
            bool a = variant == 0 ? i < 64 : variant == 1 ? (i % 4 == 0) : (i / 8) % 4 == 0;
            (a ? ma : mb)[i / 32] |= 1u << (i % 32);
        }
        hipStream_t sa, sb;
        hipError_t e = hipExtStreamCreateWithCUMask(&sa, 8, ma.data());
        if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask: %s\n", hipGetErrorString(e)); return 2; }
        CK(hipExtStreamCreateWithCUMask(&sb, 8, mb.data()));
        hipEvent_t e0, e1, e2, e3;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
        CK(hipEventRecord(e0, sa));
        hipLaunchKernelGGL(k_where, dim3(n), dim3(256), 0, sa, d0, 2000LL);
        CK(hipEventRecord(e1, sa));
        CK(hipEventRecord(e2, sb));
        hipLaunchKernelGGL(k_where, dim3(n), dim3(256), 0, sb, d1, 2000LL);
        CK(hipEventRecord(e3, sb));
        CK(hipDeviceSynchronize());
        float ta, tb;
        CK(hipEventElapsedTime(&ta, e0, e1)); CK(hipEventElapsedTime(&tb, e2, e3));
        CK(hipMemcpy(h.data(), d0, 2 * n * 4, hipMemcpyDeviceToHost));
        char nm[64];
        snprintf(nm, sizeof nm, "variant %d mask A (64 CUs)", variant);
        report(nm, h);
        CK(hipMemcpy(h.data(), d1, 2 * n * 4, hipMemcpyDeviceToHost));
        snprintf(nm, sizeof nm, "variant %d mask B (192 CUs)", variant);
        report(nm, h);
        printf("   kernel times: A %.3f ms, B %.3f ms\n", ta, tb);
        // a 1024-thread kernel on A (one workgroup per CU: 64 in flight)
        hipLaunchKernelGGL(k_where, dim3(64), dim3(1024), 0, sa, d0, 2000LL);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), d0, 2 * 64 * 4, hipMemcpyDeviceToHost));
        std::vector<uint32_t> h64(h.begin(), h.begin() + 128);
        report("   64 x 1024-thread workgroups on A", h64);
        CK(hipStreamDestroy(sa)); CK(hipStreamDestroy(sb));
    }
    return 0;
}
