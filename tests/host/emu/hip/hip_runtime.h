// TEST INFRASTRUCTURE ONLY — a minimal SIMT emulator that lets the library's HIP sources (mortal_amd/csrc/*.hip, unchanged)
// be compiled for the HOST and executed on CPU cores, so that the device code itself — not a restatement of it — can be
// checked against the oracle without a GPU (tests/test_emu_*.py, `-m "not gpu"`).  It shadows <hip/hip_runtime.h> for the
// emulator build only (tests/host/build_emu.py: clang++ -DMJ_EMU -I tests/host/emu ...); nothing under mortal_amd/ ever
// loads the resulting library.
//
// Execution model: one workgroup at a time; every work-item is a fiber (ucontext) that runs until it finishes or reaches a
// synchronisation point (__syncthreads, a wavefront collective such as __shfl / __ballot / a DPP move, or a team sync), where
// it yields to the scheduler.  Wavefront collectives rendezvous the live lanes of the addressed lane group (width 64, or the
// `width` argument / the 16-lane DPP row), so code whose lane groups diverge from each other (teams) runs as on hardware;
// a collective reached by only part of its group is reported as a deadlock instead of returning stale lanes.
// `__shared__` variables become function-local statics (one workgroup runs at a time); atomics are plain read-modify-writes
// (fibers are never pre-empted).  No timing model, no memory model: it finds logic errors, not races between wavefronts.
#pragma once
#define MJ_EMU 1

#include <sys/mman.h>
#include <ucontext.h>

#include <chrono>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__ static
#define __constant__
#define HIP_SYMBOL(x) x
#define __HIP_MEMORY_SCOPE_AGENT 4

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(16) float4 {
    float x, y, z, w;
};
inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }

typedef int hipError_t;
typedef void* hipStream_t;
struct EmuEvent {
    double t_ms;
};
typedef EmuEvent* hipEvent_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

#if defined(__x86_64__)
// minimal fiber switch (callee-saved registers + stack pointer); swapcontext costs two sigprocmask system calls per switch
extern "C" void emu_switch(void** save_sp, void* new_sp);
asm(R"(
.text
.weak emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_switch, .-emu_switch
)");
#define EMU_ASM_SWITCH 1
#endif

namespace emu {

constexpr int WAVE = 64;
constexpr size_t STACK = 512 * 1024;

struct Fiber {
    ucontext_t ctx;
    void* sp = nullptr;
    char* stack = nullptr;
    bool done = true;
    const volatile unsigned* wait_gen = nullptr;  // blocked until *wait_gen != wait_val (the scheduler skips it meanwhile)
    unsigned wait_val = 0;
};
struct GroupBar {  // rendezvous of the live lanes of one lane group
    int arrived = 0;
    unsigned gen = 0;
};
struct State {
    dim3 grid, block, bidx, tidx;
    int n_threads = 0, cur = 0, alive = 0;
    std::vector<Fiber> fibers;
    ucontext_t sched;
    void* sched_sp = nullptr;
    const std::function<void()>* body = nullptr;
    // workgroup barrier
    int bar_arrived = 0;
    unsigned bar_gen = 0;
    unsigned long long progress = 0;
    // wave collectives: [wave][log2 width] -> group barriers + exchange buffers (double buffered by generation parity)
    std::vector<GroupBar> gbar;           // [n_waves][width 1..64][group]
    std::vector<unsigned long long> xbuf;  // [n_waves][width][2][64]
    std::vector<char> dyn_shared;
};
inline State& S() {
    static State s;
    return s;
}
inline dim3& tid3() { return S().tidx; }
inline dim3& bid3() { return S().bidx; }

[[noreturn]] inline void die(const char* msg) {
    State& s = S();
    fprintf(stderr, "[emu] %s (block %u, thread %d of %d)\n", msg, s.bidx.x, s.cur, s.n_threads);
    abort();
}
inline void to_scheduler(Fiber& f) {
#ifdef EMU_ASM_SWITCH
    emu_switch(&f.sp, S().sched_sp);
#else
    swapcontext(&f.ctx, &S().sched);
#endif
}
inline void wait_for_gen(const unsigned* gen, unsigned val) {
    State& s = S();
    Fiber& f = s.fibers[s.cur];
    while (*(const volatile unsigned*)gen == val) {
        f.wait_gen = gen;
        f.wait_val = val;
        to_scheduler(f);
    }
    f.wait_gen = nullptr;
}
inline void release_block_barrier_if_complete() {
    State& s = S();
    if (s.bar_arrived > 0 && s.bar_arrived >= s.alive) {
        s.bar_arrived = 0;
        s.bar_gen++;
        s.progress++;
    }
}
inline void syncthreads() {
    State& s = S();
    const unsigned g = s.bar_gen;
    s.bar_arrived++;
    release_block_barrier_if_complete();
    wait_for_gen(&s.bar_gen, g);
    s.progress++;
}
inline int log2i(int w) { return w; }  // lane groups of ANY width (team k = lanes [k * width, (k + 1) * width)): indexed by the width itself
inline int group_alive(int wave, int width, int grp) {
    State& s = S();
    int n = 0;
    for (int l = grp * width; l < (grp + 1) * width; l++) {
        const int t = wave * WAVE + l;
        if (t < s.n_threads && !s.fibers[t].done) n++;
    }
    return n;
}
inline GroupBar& gb(int wave, int wl, int grp) { return S().gbar[((size_t)wave * 65 + wl) * WAVE + grp]; }
// rendezvous of the calling lane's `width`-lane group; returns the generation the group had on arrival
inline unsigned group_sync(int width) {
    State& s = S();
    const int lane = s.cur % WAVE, wave = s.cur / WAVE, wl = log2i(width), grp = lane / width;
    GroupBar& b = gb(wave, wl, grp);
    const unsigned g = b.gen;
    b.arrived++;
    if (b.arrived >= group_alive(wave, width, grp)) {
        b.arrived = 0;
        b.gen++;
        s.progress++;
    }
    wait_for_gen(&b.gen, g);
    return g;
}
// a lane that leaves the kernel must not be waited for
inline void on_exit_lane(int t) {
    State& s = S();
    const int lane = t % WAVE, wave = t / WAVE;
    for (int wl = 1; wl <= 64; wl++) {
        const int width = wl, grp = lane / width;
        GroupBar& b = gb(wave, wl, grp);
        if (b.arrived > 0 && b.arrived >= group_alive(wave, width, grp)) {
            b.arrived = 0;
            b.gen++;
            s.progress++;
        }
    }
    release_block_barrier_if_complete();
}
template <class T>
inline T exchange(T v, int width, const std::function<int(int lane)>& src_of) {
    static_assert(sizeof(T) <= 8, "exchange of <= 8-byte values");
    State& s = S();
    const int lane = s.cur % WAVE, wave = s.cur / WAVE, wl = log2i(width), grp = lane / width;
    const unsigned g = gb(wave, wl, grp).gen;
    unsigned long long* buf = &s.xbuf[(((size_t)wave * 65 + wl) * 2 + (g & 1)) * WAVE];
    unsigned long long raw = 0;
    memcpy(&raw, &v, sizeof(T));
    buf[lane] = raw;
    group_sync(width);
    int src = src_of(lane);
    T out;
    const int t = wave * WAVE + src;
    if (src < 0 || src >= WAVE || t >= s.n_threads) return v;
    memcpy(&out, &buf[src], sizeof(T));
    return out;
}

inline void fiber_main() {
    State& s = S();
    const int t = s.cur;
    (*s.body)();
    s.fibers[t].done = true;
    s.alive--;
    s.progress++;
    on_exit_lane(t);
    to_scheduler(s.fibers[t]);  // never resumed
}

inline void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    State& s = S();
    const int n = (int)(block.x * block.y * block.z);
    if (n > 1024) die("block too large");
    s.grid = grid;
    s.block = block;
    s.n_threads = n;
    s.body = &body;
    if ((int)s.fibers.size() < n) s.fibers.resize(n);
    for (int t = 0; t < n; t++)
        if (!s.fibers[t].stack) {
            void* p = mmap(nullptr, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (p == MAP_FAILED) die("mmap of a fiber stack failed");
            s.fibers[t].stack = (char*)p;
        }
    const int n_waves = (n + WAVE - 1) / WAVE;
    s.gbar.assign((size_t)n_waves * 65 * WAVE, GroupBar());
    s.xbuf.assign((size_t)n_waves * 65 * 2 * WAVE, 0ull);
    if (s.dyn_shared.size() < shmem + 64) s.dyn_shared.resize(shmem + 64);
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                s.bidx = dim3(bx, by, bz);
                s.alive = n;
                s.bar_arrived = 0;
                for (auto& b : s.gbar) b = GroupBar();
                for (int t = 0; t < n; t++) {
                    Fiber& f = s.fibers[t];
                    f.done = false;
                    f.wait_gen = nullptr;
#ifdef EMU_ASM_SWITCH
                    void** top = (void**)(((uintptr_t)f.stack + STACK) & ~(uintptr_t)15);
                    top[-1] = nullptr;                // return address of fiber_main (it never returns)
                    top[-2] = (void*)&fiber_main;     // popped by emu_switch's `ret`
                    for (int r = 3; r <= 8; r++) top[-r] = nullptr;  // rbp rbx r12 r13 r14 r15
                    f.sp = (void*)(top - 8);
#else
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack;
                    f.ctx.uc_stack.ss_size = STACK;
                    f.ctx.uc_link = nullptr;
                    makecontext(&f.ctx, (void (*)())fiber_main, 0);
#endif
                }
                while (s.alive > 0) {
                    const unsigned long long before = s.progress;
                    for (int t = 0; t < n; t++) {
                        if (s.fibers[t].done) continue;
                        if (s.fibers[t].wait_gen && *s.fibers[t].wait_gen == s.fibers[t].wait_val) continue;  // still blocked
                        s.cur = t;
                        s.tidx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
#ifdef EMU_ASM_SWITCH
                        emu_switch(&s.sched_sp, s.fibers[t].sp);
#else
                        swapcontext(&s.sched, &s.fibers[t].ctx);
#endif
                    }
                    if (s.progress == before && s.alive > 0)
                        die("deadlock: a barrier / wavefront collective was not reached by every live lane of its group");
                }
            }
    s.body = nullptr;
}
inline double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace emu

#define threadIdx (emu::tid3())
#define blockIdx (emu::bid3())
#define blockDim (emu::S().block)
#define gridDim (emu::S().grid)
#define warpSize 64

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu::launch(dim3(grid), dim3(block), (size_t)(shmem), [&]() { kernel(__VA_ARGS__); })
// dynamic LDS: `extern __shared__ T name[];` is spelled MJ_DYN_SHARED(T, name) in the sources
#define MJ_DYN_SHARED(T, name) T* name = reinterpret_cast<T*>((((uintptr_t)emu::S().dyn_shared.data()) + 63) & ~(uintptr_t)63)

// ---- synchronisation
inline void __syncthreads() { emu::syncthreads(); }
inline void __threadfence_block() {}
inline void __threadfence() {}
#define __builtin_amdgcn_wave_barrier() emu::die("bare wave_barrier: use mj_team_sync<W>() so the emulator knows the lane group")
#define __builtin_amdgcn_is_shared(p) true
#define __builtin_amdgcn_readfirstlane(x) (x)  /* callers pass wave-uniform values */

// ---- wavefront collectives
template <class T> inline T __shfl(T v, int src, int width = 64) {
    return emu::exchange<T>(v, width, [=](int lane) { return (lane & ~(width - 1)) | (src & (width - 1)); });
}
template <class T> inline T __shfl_down(T v, unsigned d, int width = 64) {
    return emu::exchange<T>(v, width, [=](int lane) {
        int s = lane + (int)d;
        return (s / width == lane / width) ? s : lane;
    });
}
template <class T> inline T __shfl_up(T v, unsigned d, int width = 64) {
    return emu::exchange<T>(v, width, [=](int lane) {
        int s = lane - (int)d;
        return (s >= 0 && s / width == lane / width) ? s : lane;
    });
}
template <class T> inline T __shfl_xor(T v, int m, int width = 64) {
    return emu::exchange<T>(v, width, [=](int lane) { return lane ^ m; });
}
inline unsigned long long __ballot(int pred) {
    emu::State& s = emu::S();
    const int lane = s.cur % emu::WAVE, wave = s.cur / emu::WAVE;
    const unsigned g = emu::gb(wave, 64, 0).gen;
    unsigned long long* buf = &s.xbuf[(((size_t)wave * 65 + 64) * 2 + (g & 1)) * emu::WAVE];
    buf[lane] = pred ? 1 : 0;
    emu::group_sync(64);
    unsigned long long m = 0;
    for (int l = 0; l < emu::WAVE; l++) {
        const int t = wave * emu::WAVE + l;
        if (t < s.n_threads && !s.fibers[t].done && buf[l]) m |= 1ull << l;
    }
    return m;
}
// DPP row_newbcast:N (ctrl 0x150 + N): lane N of every 16-lane row to the whole row
inline int emu_mov_dpp(int v, int ctrl) {
    if (ctrl < 0x150 || ctrl > 0x15F) emu::die("unsupported DPP control");
    const int n = ctrl - 0x150;
    return emu::exchange<int>(v, 16, [=](int lane) { return (lane & ~15) | n; });
}
#define __builtin_amdgcn_mov_dpp(v, ctrl, row_mask, bank_mask, bound_ctrl) emu_mov_dpp((v), (ctrl))

// ---- bit / conversion intrinsics
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __ffs(int x) { return __builtin_ffs(x); }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }
inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
inline long long wall_clock64() { return (long long)(emu::now_ms() * 1e5); }
using std::max;
using std::min;

// ---- atomics (fibers are never pre-empted)
template <class T, class U> inline T atomicAdd(T* p, U v) { T o = *p; *p = (T)(o + (T)v); return o; }
template <class T, class U> inline T atomicOr(T* p, U v) { T o = *p; *p = (T)(o | (T)v); return o; }
template <class T, class U> inline T atomicAnd(T* p, U v) { T o = *p; *p = (T)(o & (T)v); return o; }
template <class T, class U> inline T atomicMin(T* p, U v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <class T, class U> inline T atomicMax(T* p, U v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class U, class V> inline T atomicCAS(T* p, U cmp, V val) { T o = *p; if (o == (T)cmp) *p = (T)val; return o; }
template <class T, class U> inline T atomicExch(T* p, U v) { T o = *p; *p = (T)v; return o; }
template <class P, class T> inline bool emu_cas(P p, T* expected, T desired) {
    if (*p == *expected) { *p = desired; return true; }
    *expected = *p;
    return false;
}
#define __hip_atomic_compare_exchange_strong(p, expected, desired, so, fo, scope) emu_cas((p), (expected), (desired))
#define __hip_atomic_fetch_or(p, v, order, scope) atomicOr((p), (v))
#define __hip_atomic_fetch_add(p, v, order, scope) atomicAdd((p), (v))
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))

// ---- runtime API (host memory stands in for HBM)
inline hipError_t hipMalloc(void** p, size_t n) {
    *p = n ? aligned_alloc(256, (n + 255) & ~(size_t)255) : nullptr;
    return (*p || !n) ? 0 : 2;
}
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { free(p); return 0; }
template <class T> inline hipError_t hipHostMalloc(T** p, size_t n, unsigned = 0) { return hipMalloc((void**)p, n); }
inline hipError_t hipHostFree(void* p) { free(p); return 0; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memcpy(d, s, n); return 0; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) memcpy(d, s, n); return 0; }
inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return 0; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { if (n) memset(d, v, n); return 0; }
template <class T> inline hipError_t hipMemcpyToSymbol(T& sym, const void* src, size_t n) { memcpy((void*)&sym, src, n); return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
#define hipStreamNonBlocking 1u
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return 0; }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
inline hipError_t hipGetLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new EmuEvent{0}; return 0; }
#define hipEventDisableTiming 2u
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new EmuEvent{0}; return 0; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return 0; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t_ms = emu::now_ms(); return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }  // (the emulator runs every launch to completion)
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t_ms - a->t_ms); return 0; }
template <class F> inline hipError_t hipFuncSetAttribute(F, int, int) { return 0; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
