// TEST INFRASTRUCTURE ONLY -- a tiny single-process HIP kernel simulator.
//
// `tests/emu/build_emu.sh` compiles the product's .hip sources with g++ and THIS header shadowing
// <hip/hip_runtime.h>, producing tests/emu/libhmsg_emu.so.  It exists so kernel logic can be debugged
// in the GPU-less build container (wave64 collectives, block barriers, atomics, the MFMA fragment
// layout are simulated with ucontext fibers).  The product (holoagent_amd/) never loads it: the
// product loader only opens the gfx950 library and fails loudly without it.
#pragma once
#include <ucontext.h>
#include <immintrin.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <dlfcn.h>
#include <map>
#include <mutex>
#include <tuple>
#include <utility>
#include <vector>
using std::max;
using std::min;

#define HMSG_EMU_BUILD 1
#ifndef __HIP_MEMORY_SCOPE_WORKGROUP
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#endif
#ifndef __HIP_MEMORY_SCOPE_AGENT
#define __HIP_MEMORY_SCOPE_AGENT 4
#endif
#ifndef __HIP_MEMORY_SCOPE_SYSTEM
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#endif
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define __restrict__ __restrict

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3e { unsigned x, y, z; };
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
struct double2 { double x, y; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
static inline int2 make_int2(int a, int b) { return int2{a, b}; }

static inline double hipemu_now();
namespace hipemu {
inline thread_local uint3e t_idx, b_idx;
inline thread_local dim3 b_dim, g_dim;
// Fiber switch.  swapcontext() saves / restores the signal mask with two system calls per switch, and a wave
// collective is 2 x 64 switches: on x86-64 the switch is a dozen instructions instead (callee-saved registers +
// stack pointer, System V ABI); everything else falls back to ucontext.
#if defined(__x86_64__)
#define HIPEMU_FAST_SWITCH 1
__attribute__((naked, noinline)) static void fiber_switch(void** /*save_sp: rdi*/, void* /*load_sp: rsi*/) {
    __asm__ volatile(
        "pushq %rbp\n\t"
        "pushq %rbx\n\t"
        "pushq %r12\n\t"
        "pushq %r13\n\t"
        "pushq %r14\n\t"
        "pushq %r15\n\t"
        "movq %rsp, (%rdi)\n\t"
        "movq %rsi, %rsp\n\t"
        "popq %r15\n\t"
        "popq %r14\n\t"
        "popq %r13\n\t"
        "popq %r12\n\t"
        "popq %rbx\n\t"
        "popq %rbp\n\t"
        "ret\n\t");
}
#else
#define HIPEMU_FAST_SWITCH 0
#endif
struct Fiber {
    ucontext_t ctx;
    void* sp = nullptr;            // saved stack pointer (fast switch)
    char* stack = nullptr;
    bool done = true;
};
struct BlockState {
    std::vector<Fiber> fibers;
    ucontext_t main_ctx;
    void* main_sp = nullptr;
    int cur = -1;
    int n = 0;
    bool fiber_mode = false;
    bool used_collective = false;
    // block barrier
    int sync_arrived = 0;
    unsigned sync_gen = 0;
    // per-wave
    std::vector<int> w_arrived;
    std::vector<int> w_live;        // fibers of the wave that have not finished
    int b_live = 0;
    std::vector<unsigned> w_gen;
    std::vector<uint64_t> xbuf;     // n entries
    std::vector<uint64_t> xbuf2;
    std::vector<unsigned> xtag, xtag2;   // barrier count of the wave when the entry was written (2n entries, like xbuf)
    std::vector<char> dyn_shared;
};
inline thread_local BlockState* g_bs = nullptr;
inline BlockState& bs() { return *g_bs; }
inline int lin_tid() { return t_idx.x + b_dim.x * (t_idx.y + b_dim.y * t_idx.z); }
inline void to_scheduler() {
    BlockState& s = bs();
#if HIPEMU_FAST_SWITCH
    fiber_switch(&s.fibers[s.cur].sp, s.main_sp);
#else
    swapcontext(&s.fibers[s.cur].ctx, &s.main_ctx);
#endif
}
inline void yield() { to_scheduler(); }
inline void need_fiber(const char* what) {
    BlockState& s = bs();
    s.used_collective = true;
    if (!s.fiber_mode) {
        fprintf(stderr, "hipemu: collective %s called in direct mode (kernel mis-classified)\n", what);
        abort();
    }
}
// (live counts are kept incrementally: a Fiber holds a ucontext_t, so walking 64 `done` flags per barrier call touched
//  64 cache lines -- the simulated collectives were what made the CPU test suite slow)
inline int live_in_block() { return bs().b_live; }
inline int live_in_wave(int w) { return bs().w_live[(size_t)w]; }
inline void block_barrier() {
    need_fiber("__syncthreads");
    BlockState& s = bs();
    unsigned my = s.sync_gen;
    s.sync_arrived++;
    if (s.sync_arrived >= live_in_block()) { s.sync_arrived = 0; s.sync_gen++; return; }
    while (s.sync_gen == my) yield();
}
inline void wave_barrier() {
    need_fiber("wave collective");
    BlockState& s = bs();
    int w = lin_tid() / 64;
    unsigned my = s.w_gen[w];
    s.w_arrived[w]++;
    if (s.w_arrived[w] >= live_in_wave(w)) { s.w_arrived[w] = 0; s.w_gen[w]++; return; }
    while (s.w_gen[w] == my) yield();
}
// after a fiber exits the scheduler re-checks pending barriers
inline void recheck_barriers() {
    BlockState& s = bs();
    if (s.sync_arrived > 0 && s.sync_arrived >= live_in_block()) { s.sync_arrived = 0; s.sync_gen++; }
    for (size_t w = 0; w < s.w_arrived.size(); ++w)
        if (s.w_arrived[w] > 0 && s.w_arrived[w] >= live_in_wave((int)w)) { s.w_arrived[w] = 0; s.w_gen[w]++; }
}
// One barrier per collective: the values go through two buffers used in turn (parity of the wave's barrier count at
// entry -- every lane of a wave has passed the same number of barriers when it enters a collective, and a lane can be
// at most one collective ahead of the slowest one, which still reads the other buffer).
template <typename T>
inline T exchange(T v, int src_lane) {
    static_assert(sizeof(T) <= 8, "shuffle of >8 bytes");
    BlockState& s = bs();
    int tid = lin_tid();
    int base = (tid / 64) * 64;
    uint64_t raw = 0;
    memcpy(&raw, &v, sizeof(T));
    // (an entry counts only if its lane wrote it in THIS collective: lanes that sit at another call site, i.e. are
    //  inactive here, or have exited, are skipped -- reading an inactive lane is undefined on the hardware, here it yields
    //  the own value)
    const unsigned gen = s.w_gen[(size_t)(tid / 64)];
    const size_t par = (size_t)(gen & 1u) * (size_t)s.n;
    uint64_t* buf = s.xbuf.data() + par;
    unsigned* tag = s.xtag.data() + par;
    buf[tid] = raw;
    tag[tid] = gen;
    wave_barrier();
    int src = base + (src_lane & 63);
    uint64_t got = (src < s.n && tag[src] == gen) ? buf[src] : raw;   // (the source may have EXITED since: no `done` test)
    T out;
    memcpy(&out, &got, sizeof(T));
    return out;
}
struct KernelInfo { bool seen = false; bool needs_fiber = true; double ms = 0; double threads = 0; long launches = 0; };
inline std::map<const void*, KernelInfo>& kinfo() {
    static std::map<const void*, KernelInfo> m;
    return m;
}
inline bool emu_profile() {
    static const bool on = [] {
        const bool v = getenv("HIPEMU_PROFILE") != nullptr;
        if (v) atexit([] {
            std::vector<std::pair<double, const void*>> v2;
            for (auto& kv : kinfo()) v2.push_back({kv.second.ms, kv.first});
            std::sort(v2.begin(), v2.end());
            for (size_t i = v2.size(); i-- > 0 && v2.size() - i <= 25;) {
                const KernelInfo& k = kinfo()[v2[i].second];
                Dl_info di;
                const char* nm = dladdr(v2[i].second, &di) && di.dli_sname ? di.dli_sname : "?";
                fprintf(stderr, "[hipemu] %-48.48s launches %7ld  threads %12.0f  %9.1f ms\n", nm, k.launches, k.threads, k.ms);
            }
        });
        return v;
    }();
    return on;
}
template <typename F>
struct Thunk {
    static thread_local F* fn;
    static void entry() {
        (*fn)();
        BlockState& s = bs();
        s.fibers[s.cur].done = true;
        s.b_live--;
        s.w_live[(size_t)(s.cur / 64)]--;
        to_scheduler();               // never resumed
        abort();
    }
};
template <typename F>
thread_local F* Thunk<F>::fn = nullptr;

template <typename F>
void run_grid(const void* key, dim3 grid, dim3 block, size_t shmem, F body) {
    static thread_local BlockState state;
    g_bs = &state;
    BlockState& s = state;
    const double t_start = emu_profile() ? hipemu_now() : 0.0;
    KernelInfo* kip;
    {   // (host threads launch side by side: the fold worker of hmsg_merge.hip)
        static std::mutex kmu;
        std::lock_guard<std::mutex> lk(kmu);
        kip = &kinfo()[key];
    }
    KernelInfo& ki = *kip;
    // every launch runs its threads as fibers: a kernel without collectives finishes each thread on its first switch
    // (two register swaps per thread), and no kernel can be mis-classified because its first launch skipped a loop
    bool fiber_mode = true;
    int n = block.x * block.y * block.z;
    s.n = n;
    s.fiber_mode = fiber_mode;
    s.used_collective = false;
    s.dyn_shared.assign(shmem + 16, 0);
    b_dim = block;
    g_dim = grid;
    const size_t STK = 256 * 1024;
    if (fiber_mode) {
        if ((int)s.fibers.size() < n) {
            size_t old = s.fibers.size();
            s.fibers.resize(n);
            for (size_t i = old; i < (size_t)n; ++i) s.fibers[i].stack = (char*)malloc(STK);
        }
        s.w_arrived.assign((n + 63) / 64, 0);
        s.w_gen.assign((n + 63) / 64, 0);
        s.xbuf.assign(2 * (size_t)n, 0);
        s.xbuf2.assign(2 * (size_t)n, 0);
        s.xtag.assign(2 * (size_t)n, 0xffffffffu);
        s.xtag2.assign(2 * (size_t)n, 0xffffffffu);
    }
    Thunk<F>::fn = &body;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                b_idx = uint3e{bx, by, bz};
                if (!fiber_mode) {
                    for (int t = 0; t < n; ++t) {
                        t_idx = uint3e{(unsigned)(t % block.x), (unsigned)((t / block.x) % block.y),
                                       (unsigned)(t / (block.x * block.y))};
                        body();
                    }
                    continue;
                }
                s.sync_arrived = 0;
                for (auto& a : s.w_arrived) a = 0;
                for (int t = 0; t < n; ++t) {
                    Fiber& f = s.fibers[t];
                    f.done = false;
#if HIPEMU_FAST_SWITCH
                    {
                        // initial frame: six zeroed callee-saved registers, then the entry point as the return
                        // address of fiber_switch; entry sees rsp % 16 == 8 as after a call
                        void** sp = (void**)(((uintptr_t)f.stack + STK) & ~(uintptr_t)15);
                        sp -= 8;
                        for (int q = 0; q < 6; ++q) sp[q] = nullptr;
                        sp[6] = (void*)(void (*)())Thunk<F>::entry;
                        sp[7] = nullptr;
                        f.sp = sp;
                    }
#else
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack;
                    f.ctx.uc_stack.ss_size = STK;
                    f.ctx.uc_link = nullptr;
                    makecontext(&f.ctx, (void (*)())Thunk<F>::entry, 0);
#endif
                }
                for (int t = n; t < (int)s.fibers.size(); ++t) s.fibers[t].done = true;
                s.b_live = n;
                s.w_live.assign((size_t)(n + 63) / 64, 0);
                for (int w = 0; w < (n + 63) / 64; ++w) s.w_live[(size_t)w] = std::min(64, n - w * 64);
                int remaining = n;
                while (remaining > 0) {
                    for (int t = 0; t < n; ++t) {
                        if (s.fibers[t].done) continue;
                        s.cur = t;
                        t_idx = uint3e{(unsigned)(t % block.x), (unsigned)((t / block.x) % block.y),
                                       (unsigned)(t / (block.x * block.y))};
#if HIPEMU_FAST_SWITCH
                        fiber_switch(&s.main_sp, s.fibers[t].sp);
#else
                        swapcontext(&s.main_ctx, &s.fibers[t].ctx);
#endif
                        if (s.fibers[t].done) { remaining--; recheck_barriers(); }
                    }
                }
            }
    if (!ki.seen) { ki.seen = true; ki.needs_fiber = s.used_collective; }
    else if (s.used_collective) ki.needs_fiber = true;
    if (emu_profile()) {          // HIPEMU_PROFILE=1: where the simulated kernels spend the test suite's time
        ki.ms += hipemu_now() - t_start;
        ki.threads += (double)n * grid.x * grid.y * grid.z;
        ki.launches += 1;
    }
}
}  // namespace hipemu

#define threadIdx (hipemu::t_idx)
#define blockIdx (hipemu::b_idx)
#define blockDim (hipemu::b_dim)
#define gridDim (hipemu::g_dim)
#define warpSize 64
#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)(((uintptr_t)hipemu::bs().dyn_shared.data() + 15) & ~(uintptr_t)15);

typedef int hipError_t;
typedef void* hipStream_t;
struct hipEmuEvent { double t; };
typedef hipEmuEvent* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4, hipMemcpyHostToHost = 0 };
enum { hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2, hipMemoryTypeUnregistered = 0 };
struct hipPointerAttribute_t { int type; int device; };
struct hipDeviceProp_t { int multiProcessorCount; char name[64]; size_t totalGlobalMem; char gcnArchName[64]; };
// (256-byte alignment as the driver gives: kernels use alignas(32) records and 16-byte vector loads at buffer starts)
static inline hipError_t hipMalloc(void** p, size_t n) { *p = nullptr; return posix_memalign(p, 256, n ? n : 1) == 0 ? hipSuccess : 2; }
template <typename T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
#define hipHostMallocDefault 0
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = nullptr; return posix_memalign(p, 256, n ? n : 1) == 0 ? hipSuccess : 2; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = 0) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = 0) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
#define hipStreamNonBlocking 1
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = 0; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { memset(p, 0, sizeof(*p)); p->multiProcessorCount = 4; strcpy(p->name, "hipemu"); strcpy(p->gcnArchName, "emu"); p->totalGlobalMem = (size_t)8 << 30; return hipSuccess; }
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void*) { a->type = hipMemoryTypeHost; a->device = 0; return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = (size_t)8 << 30; *t = (size_t)8 << 30; return hipSuccess; }
#include <chrono>
static inline double hipemu_now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipEmuEvent{0}; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = 0) { e->t = hipemu_now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
#define hipErrorNotReady 600
#define hipEventDisableTiming 2
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new hipEmuEvent{0}; return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }

template <typename... KArgs, typename... Args>
static inline void hipLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t, Args... args) {
    if (grid.x == 0 || grid.y == 0 || grid.z == 0) return;
    std::tuple<KArgs...> targs(static_cast<KArgs>(args)...);
    hipemu::run_grid((const void*)kernel, grid, block, shmem, [&]() { std::apply(kernel, targs); });
}

// ---------------------------------------------------------------- device intrinsics
static inline void __syncthreads() { hipemu::block_barrier(); }
static inline void __threadfence() {}
static inline void __threadfence_system() {}
template <typename T> static inline T __shfl(T v, int src, int = 64) { return hipemu::exchange(v, src); }
template <typename T> static inline T __shfl_xor(T v, int m, int = 64) { return hipemu::exchange(v, (hipemu::lin_tid() & 63) ^ m); }
template <typename T> static inline T __shfl_down(T v, unsigned d, int = 64) { int l = hipemu::lin_tid() & 63; return hipemu::exchange(v, l + (int)d < 64 ? l + (int)d : l); }
template <typename T> static inline T __shfl_up(T v, unsigned d, int = 64) { int l = hipemu::lin_tid() & 63; return hipemu::exchange(v, l >= (int)d ? l - (int)d : l); }
static inline unsigned long long __ballot(int pred) {
    hipemu::BlockState& s = hipemu::bs();
    int tid = hipemu::lin_tid(), base = (tid / 64) * 64;
    const unsigned gen = s.w_gen[(size_t)(tid / 64)];
    const size_t par = (size_t)(gen & 1u) * (size_t)s.n;
    uint64_t* buf = s.xbuf2.data() + par;
    unsigned* tag = s.xtag2.data() + par;
    buf[tid] = pred ? 1 : 0;
    tag[tid] = gen;
    hipemu::wave_barrier();
    unsigned long long m = 0;
    for (int l = 0; l < 64 && base + l < s.n; ++l)
        if (tag[base + l] == gen && buf[base + l]) m |= 1ull << l;
    return m;
}
static inline int __builtin_amdgcn_readlane(int v, int src) { return hipemu::exchange(v, src); }
static inline void __builtin_amdgcn_wave_barrier() { hipemu::wave_barrier(); }   // (HW: a scheduling barrier; here the lanes of a wave meet)
static inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }
static inline unsigned long long wall_clock64() { return (unsigned long long)(hipemu_now() * 1e5); }   // 100 MHz ticks
static inline int __any(int p) { return __ballot(p) != 0; }
static inline int __all(int p) { return __ballot(!p) == 0; }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __ffs(unsigned x) { return __builtin_ffs((int)x); }
static inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }
static inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
static inline double __dmul_rn(double a, double b) { return a * b; }
static inline double __dadd_rn(double a, double b) { return a + b; }
static inline double __dsub_rn(double a, double b) { return a - b; }
static inline double __ddiv_rn(double a, double b) { return a / b; }
static inline float __fmul_rn(float a, float b) { return a * b; }
static inline float __fadd_rn(float a, float b) { return a + b; }
static inline float __fsub_rn(float a, float b) { return a - b; }
static inline float __fdiv_rn(float a, float b) { return a / b; }
static inline float __fsqrt_rn(float a) { return sqrtf(a); }
static inline double __dsqrt_rn(double a) { return sqrt(a); }
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline unsigned short hipemu_f2h(float f) { return (unsigned short)_cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC); }
static inline float hipemu_h2f(unsigned short h) { return _cvtsh_ss(h); }
struct __half { unsigned short x; };
static inline __half __float2half_rn(float f) { return __half{hipemu_f2h(f)}; }
static inline __half __float2half(float f) { return __half{hipemu_f2h(f)}; }
static inline float __half2float(__half h) { return hipemu_h2f(h.x); }
static inline long long __double_as_longlong(double d) { long long r; memcpy(&r, &d, 8); return r; }
static inline double __longlong_as_double(long long l) { double r; memcpy(&r, &l, 8); return r; }
static inline int __float_as_int(float f) { int r; memcpy(&r, &f, 4); return r; }
static inline float __int_as_float(int i) { float r; memcpy(&r, &i, 4); return r; }
static inline unsigned __float_as_uint(float f) { unsigned r; memcpy(&r, &f, 4); return r; }
static inline float __uint_as_float(unsigned i) { float r; memcpy(&r, &i, 4); return r; }

template <typename T> static inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline double atomicAdd(double* p, double v) { double o = *p; *p = o + v; return o; }
template <typename T> static inline T atomicSub(T* p, T v) { return __atomic_fetch_sub(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> static inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicAnd(T* p, T v) { return __atomic_fetch_and(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <typename T> static inline T atomicCAS(T* p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }

// f32-input MFMA (guide section 3): lane l holds A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
// C/D: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).  Result = k-ordered fmaf chain.
typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
static inline hipemu_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, hipemu_f32x16 c, int, int, int) {
    hipemu::BlockState& s = hipemu::bs();
    int tid = hipemu::lin_tid(), base = (tid / 64) * 64, lane = tid & 63;
    const size_t par = (size_t)(s.w_gen[(size_t)(tid / 64)] & 1u) * (size_t)s.n;   // (see hipemu::exchange)
    uint64_t* xb = s.xbuf.data() + par;
    uint64_t* xb2 = s.xbuf2.data() + par;
    (void)xb2;
    xb[tid] = (uint64_t)__float_as_uint(a) | ((uint64_t)__float_as_uint(b) << 32);
    hipemu::wave_barrier();
    hipemu_f32x16 d = c;
    int col = lane & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float av = __uint_as_float((unsigned)(xb[base + row + 32 * k] & 0xffffffffu));
            float bv = __uint_as_float((unsigned)(xb[base + col + 32 * k] >> 32));
            acc = fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    return d;
}
// f64 MFMA 16x16x4: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; C/D: col = lane&15, row = (lane>>4) + 4*reg
typedef double hipemu_f64x4 __attribute__((ext_vector_type(4)));
static inline hipemu_f64x4 __builtin_amdgcn_mfma_f64_16x16x4f64(double a, double b, hipemu_f64x4 c, int, int, int) {
    hipemu::BlockState& s = hipemu::bs();
    int tid = hipemu::lin_tid(), base = (tid / 64) * 64, lane = tid & 63;
    const size_t par = (size_t)(s.w_gen[(size_t)(tid / 64)] & 1u) * (size_t)s.n;   // (see hipemu::exchange)
    uint64_t* xb = s.xbuf.data() + par;
    uint64_t* xb2 = s.xbuf2.data() + par;
    (void)xb2;
    uint64_t ra, rb;
    memcpy(&ra, &a, 8);
    memcpy(&rb, &b, 8);
    xb[tid] = ra;
    xb2[tid] = rb;
    hipemu::wave_barrier();
    hipemu_f64x4 d = c;
    int col = lane & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (lane >> 4) + 4 * r;
        double acc = c[r];
        for (int k = 0; k < 4; ++k) {
            double av, bv;
            memcpy(&av, &xb[base + row + 16 * k], 8);
            memcpy(&bv, &xb2[base + col + 16 * k], 8);
            acc = fma(av, bv, acc);
        }
        d[r] = acc;
    }
    return d;
}
static inline hipemu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, hipemu_f32x4 c, int, int, int) {
    // A[l&15][k=l>>4], B[k=l>>4][l&15]; C/D: col = lane&15, row = (lane>>4)*4 + reg
    hipemu::BlockState& s = hipemu::bs();
    int tid = hipemu::lin_tid(), base = (tid / 64) * 64, lane = tid & 63;
    const size_t par = (size_t)(s.w_gen[(size_t)(tid / 64)] & 1u) * (size_t)s.n;   // (see hipemu::exchange)
    uint64_t* xb = s.xbuf.data() + par;
    uint64_t* xb2 = s.xbuf2.data() + par;
    (void)xb2;
    xb[tid] = (uint64_t)__float_as_uint(a) | ((uint64_t)__float_as_uint(b) << 32);
    hipemu::wave_barrier();
    hipemu_f32x4 d = c;
    int col = lane & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (lane >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            float av = __uint_as_float((unsigned)(xb[base + row + 16 * k] & 0xffffffffu));
            float bv = __uint_as_float((unsigned)(xb[base + col + 16 * k] >> 32));
            acc = fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    return d;
}
