// Shared state + device helpers of libhmsg (MI355X / gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/hmsg.h"
#include "../../include/hmsg_test.h"

#define HMSG_WAVE 64

// ------------------------------------------------------------------ error plumbing
struct hmsg_error {
    int code;
    std::string msg;
};
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) throw hmsg_error{HMSG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)}; \
    } while (0)
#define HMSG_REQUIRE(cond, code, text) \
    do {                               \
        if (!(cond)) throw hmsg_error{code, text}; \
    } while (0)
#define HMSG_CHECK_LAUNCH() HIP_TRY(hipGetLastError())

// np.sum of a short float64 array as numpy adds it (pairwise_sum below its 128-element block: eight running sums over
// the leading multiple of 8, combined as a tree, then the tail in order) -- kernel weights are normalised by such a sum
static inline double np_sum_f64(const double* a, size_t n) {
    if (n < 8) {
        double r = 0.0;          // (numpy starts from a[0]; adding it to 0.0 first is exact)
        for (size_t i = 0; i < n; ++i) r += a[i];
        return r;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    size_t i = 8;
    for (; i + 8 <= n; i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + (size_t)j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// ------------------------------------------------------------------ caching device allocator
// hipMalloc / hipFree of the GB-sized scratch of one build cost hundreds of ms and synchronise the device, and
// a service rebuilds scenes over and over: freed blocks are parked in a per-thread, per-device cache and handed
// out again (first block of at least the requested size and at most twice it).  Safe because every C-ABI entry
// point synchronises its stream before it returns and a handle is driven by one thread at a time.
struct DevCache {
    std::multimap<std::pair<int, size_t>, void*> free_;   // (device, bytes) -> block
    // Carving: a very large parked block (the frame store of a long episode, handed back before the merge) can be cut
    // into pieces for large requests instead of going back to the driver -- a fresh hipMalloc costs ~28 ms per GB on this
    // stack (49 GB: 1.4 s, measured), and the merge of a 10 000-frame episode asks for ~120 GB of arenas.  Only while
    // `allow_carve` is set (the caller guarantees that everything it carves is released again before the big block is
    // wanted back).  A piece can be parked and handed out like any block; it is never hipFree'd on its own: when every
    // piece of a root is parked again they are merged back into the root.
    struct Root {
        size_t bytes = 0;
        int dev = 0;
    };
    std::map<void*, Root> roots_;          // split blocks by base address
    std::map<void*, void*> piece_root_;    // piece base -> root base (parked or handed out)
    bool allow_carve = false;
    // merge the pieces of every root whose pieces are ALL parked back into one block
    void coalesce() {
        if (roots_.empty()) return;
        std::map<void*, size_t> parked;    // root -> parked bytes
        for (auto& kv : free_) {
            auto it = piece_root_.find(kv.second);
            if (it != piece_root_.end()) parked[it->second] += kv.first.second;
        }
        for (auto it = roots_.begin(); it != roots_.end();) {
            void* root = it->first;
            if (parked[root] != it->second.bytes) {
                ++it;
                continue;
            }
            for (auto f = free_.begin(); f != free_.end();) {
                auto pr = piece_root_.find(f->second);
                if (pr != piece_root_.end() && pr->second == root) {
                    piece_root_.erase(pr);
                    f = free_.erase(f);
                } else {
                    ++f;
                }
            }
            free_.insert({{it->second.dev, it->second.bytes}, root});
            it = roots_.erase(it);
        }
    }
    // (no hipFree in a destructor: it would run at thread exit, possibly after the HIP runtime is gone)
    void trim() {
        coalesce();
        for (auto it = free_.begin(); it != free_.end();) {
            if (piece_root_.count(it->second)) {       // a piece whose siblings are still in use: stays parked
                ++it;
                continue;
            }
            (void)hipFree(it->second);
            it = free_.erase(it);
        }
    }
    // HMSG_DEBUG_EXACT_ALLOC=1: every request is its own allocation of exactly the requested size and goes back to the
    // driver when released -- for runs of the kernel simulator under AddressSanitizer (scripts/emu_sanitize.sh), where the
    // cache's rounding and reuse would hide an access past the end of a buffer.
    static bool exact_mode() {
        static const bool on = getenv("HMSG_DEBUG_EXACT_ALLOC") != nullptr;
        return on;
    }
    void* get(int dev, size_t bytes, size_t* got) {
        if (exact_mode()) {
            void* q = nullptr;
            const size_t n = std::max<size_t>((bytes + 15) / 16 * 16, 16);
            if (hipMalloc(&q, n) != hipSuccess) throw hmsg_error{HMSG_ERR_NOMEM, "hipMalloc (exact mode)"};
            *got = n;
            return q;
        }
        size_t gran = bytes < ((size_t)1 << 20) ? ((size_t)1 << 12) : ((size_t)1 << 21);
        size_t want = (bytes + gran - 1) / gran * gran;
        if (!roots_.empty() && want >= ((size_t)1 << 28)) coalesce();     // (a cut-up block that is whole again)
        auto it = free_.lower_bound({dev, want});
        if (it != free_.end() && it->first.first == dev && it->first.second <= want * 2) {
            void* p = it->second;
            *got = it->first.second;
            free_.erase(it);
            return p;
        }
        if (allow_carve && want >= ((size_t)1 << 28)) {
            coalesce();
            // the largest parked block of this device
            auto big = free_.end();
            for (auto f = free_.begin(); f != free_.end(); ++f)
                if (f->first.first == dev && f->first.second >= want &&
                    (f->first.second >= ((size_t)8 << 30) || piece_root_.count(f->second)) &&
                    (big == free_.end() || f->first.second > big->first.second))
                    big = f;         // (only blocks of 8 GB and more are cut up -- and what is left of them)
            if (big != free_.end()) {
                void* base = big->second;
                const size_t total = big->first.second;
                free_.erase(big);
                void* root = base;
                auto pr = piece_root_.find(base);
                if (pr != piece_root_.end()) root = pr->second;
                else roots_[root] = Root{total, dev};
                piece_root_[base] = root;
                if (total > want) {
                    void* rest = (char*)base + want;
                    piece_root_[rest] = root;
                    free_.insert({{dev, total - want}, rest});
                }
                *got = want;
                return base;
            }
        }
        void* p = nullptr;
        const bool timing = want >= ((size_t)1 << 28) && getenv("HMSG_DEBUG_TIMING") != nullptr;
        const auto t_a = std::chrono::steady_clock::now();
        hipError_t e = hipMalloc(&p, want);
        if (timing)
            fprintf(stderr, "[hmsg alloc] hipMalloc of %.1f GB: %.1f ms\n", want / 1073741824.0,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_a).count());
        if (e != hipSuccess) {          // out of memory: drop the cache and retry once
            (void)hipGetLastError();    // (the failed call must not surface at the next launch check)
            size_t parked = 0, fr = 0, tot = 0;
            for (auto& kv : free_) parked += kv.first.second;
            (void)hipMemGetInfo(&fr, &tot);
            fprintf(stderr, "[hmsg alloc] %.1f MB did not fit (device free %.1f of %.1f GB): returning %.1f GB of parked blocks\n",
                    want / 1048576.0, fr / 1073741824.0, tot / 1073741824.0, parked / 1073741824.0);
            const auto t_b = std::chrono::steady_clock::now();
            trim();
            const auto t_c = std::chrono::steady_clock::now();
            e = hipMalloc(&p, want);
            if (timing)
                fprintf(stderr, "[hmsg alloc] hipFree of the parked blocks: %.1f ms, hipMalloc: %.1f ms\n",
                        std::chrono::duration<double, std::milli>(t_c - t_b).count(),
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_c).count());
            if (e != hipSuccess) (void)hipGetLastError();
        }
        if (e != hipSuccess) throw hmsg_error{HMSG_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e)};
        *got = want;
        return p;
    }
    void put(int dev, void* p, size_t bytes) {
        if (exact_mode()) {
            (void)hipFree(p);
            return;
        }
        free_.insert({{dev, bytes}, p});
    }
    void swap_state(DevCache& o) {
        free_.swap(o.free_);
        roots_.swap(o.roots_);
        piece_root_.swap(o.piece_root_);
    }
};
inline DevCache& dev_cache() {
    static thread_local DevCache c;
    return c;
}
struct CarveScope {          // large requests inside the scope may be cut out of a very large parked block
    bool prev;
    CarveScope() : prev(dev_cache().allow_carve) { dev_cache().allow_carve = true; }
    ~CarveScope() { dev_cache().allow_carve = prev; }
};

// ------------------------------------------------------------------ device buffer (owned)
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    size_t cap_bytes = 0;
    int dev = 0;
    bool is_view = false;        // points into a block somebody else owns (the frame arena of a long episode)
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p && !is_view) dev_cache().put(dev, p, cap_bytes);
        p = nullptr;
        n = 0;
        cap_bytes = 0;
        is_view = false;
    }
    void view(T* ptr, size_t count) {
        release();
        p = ptr;
        n = count;
        is_view = true;
    }
    void alloc(size_t count) {
        release();
        if (count == 0) count = 1;
        (void)hipGetDevice(&dev);
        p = (T*)dev_cache().get(dev, count * sizeof(T), &cap_bytes);
        n = count;
    }
    void swap(DevBuf& o) {
        std::swap(p, o.p);
        std::swap(n, o.n);
        std::swap(cap_bytes, o.cap_bytes);
        std::swap(dev, o.dev);
        std::swap(is_view, o.is_view);
    }
    void ensure(size_t count) {
        // scratch buffers: grow geometrically from a generous floor -- hipFree/hipMalloc synchronise the
        // device, and HBM is plentiful (a 25 % growth policy cost ~0.6 s of re-allocation per 1000-frame merge)
        if (count > n) alloc(DevCache::exact_mode() ? count : std::max<size_t>(count * 2, (size_t)1 << 16));
    }
    void zero(hipStream_t s) { HIP_TRY(hipMemsetAsync(p, 0, n * sizeof(T), s)); }
    size_t bytes() const { return n * sizeof(T); }
};

// ------------------------------------------------------------------ pinned host buffer + low-latency wait
// The merge fold reads a few hundred bytes back twice per step.  A pageable destination makes the copy a blocking
// staged transfer and hipStreamSynchronize parks the thread (tens of microseconds to wake up, a thousand times per
// scene): results land in pinned memory and the host spins on an event instead.
template <typename T>
struct PinnedBuf {
    T* p = nullptr;
    size_t n = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    ~PinnedBuf() {
        if (p) (void)hipHostFree(p);
    }
    void ensure(size_t count) {
        if (count <= n) return;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        n = std::max<size_t>(count * 2, 256);
        HIP_TRY(hipHostMalloc((void**)&p, n * sizeof(T), hipHostMallocDefault));
    }
};
// Large read-backs into CALLER memory (numpy arrays, std::vectors): a pageable destination makes hipMemcpy pin the
// pages on the fly (measured: 22 ms for the 4 MB map cloud); bounce through a persistent pinned buffer instead
// (device -> pinned at link speed, then a plain memcpy that takes the page faults at host speed).
inline void d2h_bounce(void* dst, const void* src, size_t bytes) {
    constexpr size_t CH = (size_t)16 << 20;
    if (bytes <= 4096) {
        HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
        return;
    }
    static std::mutex mu;
    static PinnedBuf<unsigned char> stage;
    std::lock_guard<std::mutex> lk(mu);
    stage.ensure(std::min(bytes, CH));
    for (size_t o = 0; o < bytes; o += CH) {
        const size_t n = std::min(CH, bytes - o);
        HIP_TRY(hipMemcpy(stage.p, (const unsigned char*)src + o, n, hipMemcpyDeviceToHost));
        memcpy((unsigned char*)dst + o, stage.p, n);
    }
}
// The other direction for large CALLER arrays (a numpy text table of 8 MB): hipMemcpyAsync from pageable memory above the runtime's
// staging threshold pins the caller's pages on the fly -- measured on the MI355X: 24 ms for the 8.2 MB query table of configs[2]
// against 0.2 ms below the threshold (D = 512: 4.1 MB).  Bounce through a persistent pinned buffer; returns when the bytes are on
// their way from pinned memory, i.e. the caller's array may change and later work on `s` sees the data.
inline void h2d_bounce(void* dst_dev, const void* src_host, size_t bytes, hipStream_t s) {
    constexpr size_t CH = (size_t)16 << 20;
    if (!bytes) return;
    if (bytes <= 65536) {                                   // (small tables: the runtime's own staged copy is as fast)
        HIP_TRY(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, s));
        return;
    }
    static std::mutex mu;
    static PinnedBuf<unsigned char> stage;
    std::lock_guard<std::mutex> lk(mu);
    stage.ensure(std::min(bytes, CH));
    for (size_t o = 0; o < bytes; o += CH) {
        const size_t n = std::min(CH, bytes - o);
        memcpy(stage.p, (const unsigned char*)src_host + o, n);
        HIP_TRY(hipMemcpyAsync((unsigned char*)dst_dev + o, stage.p, n, hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));                   // (the staging buffer is shared: it must be free before the lock goes)
    }
}
// A few KB of tables from PINNED host memory to the device inside the stream, by a small kernel that reads the host memory
// directly (as k_publish writes it): `hipMemcpyAsync` of such a table is a blit-kernel dispatch of ~6.5 us on this stack, three
// of them per merge-fold step.  src must be hipHostMalloc'ed memory (PinnedBuf) that stays untouched until the stream passes
// this point.
static __global__ void k_upload16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16, unsigned tail) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x < tail)
        ((unsigned char*)(dst + n16))[threadIdx.x] = ((const unsigned char*)(src + n16))[threadIdx.x];
}
inline void upload_pinned(void* dst_dev, const void* src_pinned, size_t bytes, hipStream_t s) {
    if (!bytes) return;
    if (bytes > ((size_t)1 << 18) || ((uintptr_t)dst_dev & 15) || ((uintptr_t)src_pinned & 15)) {
        HIP_TRY(hipMemcpyAsync(dst_dev, src_pinned, bytes, hipMemcpyHostToDevice, s));
        return;
    }
    const size_t n16 = bytes / 16;
    hipLaunchKernelGGL(k_upload16, dim3((unsigned)std::max<size_t>(1, std::min<size_t>((n16 + 255) / 256, 16))), dim3(256), 0, s, (const uint4*)src_pinned,
                       (uint4*)dst_dev, n16, (unsigned)(bytes % 16));
}
struct SpinWait {
    hipEvent_t ev = nullptr;
    ~SpinWait() {
        if (ev) (void)hipEventDestroy(ev);
    }
    void wait(hipStream_t s) {
        if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ev, s));
        for (;;) {
            const hipError_t e = hipEventQuery(ev);
            if (e == hipSuccess) return;
            if (e != hipErrorNotReady) HIP_TRY(e);
        }
    }
};

// ------------------------------------------------------------------ grid geometry (global voxel grid)
// Linearisation: lin = (ix * NY + iy) * NZP + iz, NZP = NZ rounded up to 64 so a z-column starts on a
// 64-bit word; slot of an occupied cell = rank[word] + popc(bits below) = its position in ascending
// (ix, iy, iz) order (the canonical order of the oracle's voxel_down_sample).
struct GridGeom {
    double ox, oy, oz;   // voxel_min_bound = min_bound - vs/2  (Open3D VoxelDownSample)
    double vs;
    int nx, ny, nz, nzp; // nzp % 64 == 0
    long long nwords;
};

struct CamK {
    double fx, fy, cx, cy;
};

// ------------------------------------------------------------------ device helpers
__device__ __forceinline__ unsigned long long enc_f64(double d) {
    unsigned long long b = (unsigned long long)__double_as_longlong(d);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
static inline double dec_f64(unsigned long long e) {   // host side inverse of enc_f64
    unsigned long long b = (e >> 63) ? (e & 0x7fffffffffffffffull) : ~e;
    double d;
    std::memcpy(&d, &b, 8);
    return d;
}

// generic.py:111-124 + Open3D transform (generic.py:137): z = f32(depth)/f32(scale); X = (x-cx)*z/fx in
// f64; world = ((X*T00 + Y*T01) + Z*T02) + T03, divided by the homogeneous w.  No FMA contraction.
__device__ __forceinline__ bool backproject(unsigned short d, int x, int y, const CamK& k, float scale,
                                            const double* __restrict__ T, double& wx, double& wy, double& wz) {
    float z = __fdiv_rn((float)d, scale);
    if (!(z > 0.0f)) return false;
    double Z = (double)z;
    double X = __ddiv_rn(__dmul_rn(__dsub_rn((double)x, k.cx), Z), k.fx);
    double Y = __ddiv_rn(__dmul_rn(__dsub_rn((double)y, k.cy), Z), k.fy);
    double a = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(X, T[0]), __dmul_rn(Y, T[1])), __dmul_rn(Z, T[2])), T[3]);
    double b = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(X, T[4]), __dmul_rn(Y, T[5])), __dmul_rn(Z, T[6])), T[7]);
    double c = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(X, T[8]), __dmul_rn(Y, T[9])), __dmul_rn(Z, T[10])), T[11]);
    // The last row of a rigid pose is (0, 0, 0, 1): w = X*0 + Y*0 + Z*0 + 1 is 1.0 exactly (X, Y, Z are finite) and a / 1.0 is a -- the
    // same bits without three of this function's five float64 divisions (the map and fusion passes are bound by them).
    if (T[12] == 0.0 && T[13] == 0.0 && T[14] == 0.0 && T[15] == 1.0) {
        wx = a;
        wy = b;
        wz = c;
        return true;
    }
    double w = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(X, T[12]), __dmul_rn(Y, T[13])), __dmul_rn(Z, T[14])), T[15]);
    wx = __ddiv_rn(a, w);
    wy = __ddiv_rn(b, w);
    wz = __ddiv_rn(c, w);
    return true;
}

__device__ __forceinline__ void cell_of(const GridGeom& g, double wx, double wy, double wz, int& ix, int& iy, int& iz) {
    ix = (int)floor(__ddiv_rn(__dsub_rn(wx, g.ox), g.vs));
    iy = (int)floor(__ddiv_rn(__dsub_rn(wy, g.oy), g.vs));
    iz = (int)floor(__ddiv_rn(__dsub_rn(wz, g.oz), g.vs));
}
__device__ __forceinline__ long long lin_of(const GridGeom& g, int ix, int iy, int iz) {
    return ((long long)ix * g.ny + iy) * g.nzp + iz;
}

__device__ __forceinline__ double wave_min_f64(double v) {
    for (int o = 32; o > 0; o >>= 1) {
        double t = __shfl_xor(v, o);
        v = t < v ? t : v;
    }
    return v;
}
__device__ __forceinline__ double wave_max_f64(double v) {
    for (int o = 32; o > 0; o >>= 1) {
        double t = __shfl_xor(v, o);
        v = t > v ? t : v;
    }
    return v;
}
__device__ __forceinline__ float wave_sum_f32(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// s <- fl(s + p), `len` times, in closed form.  While s stays inside one binade [2^e, 2^(e+1)) its ulp u is fixed,
// s is a multiple of u and p = q*u: every addition rounds S + q (S = s/u) to the nearest integer, i.e. adds the
// same integer d = round(q) -- unless q sits exactly on .5 (round-half-even then depends on S: stepped) or the sum
// leaves the binade (stepped across).  A mask voxel holds the SAME snapped map point hundreds of times
// (generic.py:181-188), and Open3D adds them one by one; this replays those additions bit for bit in O(binades).
__device__ __forceinline__ double repeat_add(double s, double p, int len) {
    while (len > 0) {
        const double as = fabs(s), ap = fabs(p);
        if (ap == 0.0) return __dadd_rn(s, p);
        const long long bs = __double_as_longlong(as);
        const int e = (int)(bs >> 52);                       // biased exponent (sign bit cleared)
        if (as >= ap && ((s < 0.0) == (p < 0.0)) && e >= 1 && e < 2046) {
            unsigned long long S = ((unsigned long long)bs & 0xfffffffffffffull) | (1ull << 52);
            const double q = ldexp(ap, 1075 - e);            // p in units of ulp(s): exact, < 2^53
            if (q < 0.5) return s;                           // p below half an ulp: s never moves
            const double qi = floor(q), qf = q - qi;
            if (qf != 0.5) {
                const unsigned long long iq = (unsigned long long)qi;
                const unsigned long long d = iq + (qf > 0.5 ? 1ull : 0ull);
                const unsigned long long top = (1ull << 53) - 1ull;
                if (S + iq <= top) {                         // (each step needs S + q < 2^53 before it)
                    // all `len` steps stay in the binade when S + (len-1)*d + iq <= top (the usual case: no division)
                    unsigned long long k = (unsigned long long)len;
                    const unsigned long long room = top - iq - S;
                    if (__umul64hi(k - 1ull, d) != 0ull || (k - 1ull) * d > room) k = room / d + 1ull;
                    S += k * d;
                    const double r = ldexp((double)S, e - 1075);
                    s = s < 0.0 ? -r : r;
                    len -= (int)k;
                    continue;
                }
            }
        }
        s = __dadd_rn(s, p);
        --len;
    }
    return s;
}

// value of lane `src` (wave-uniform) as a scalar broadcast: v_readlane, not an LDS permute
__device__ __forceinline__ double wave_bcast_f64(double v, int src) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)b >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// ------------------------------------------------------------------ scene state
struct MaskCloudSet {            // 3-D masks of all fused frames (generic.py:140-190 outputs)
    DevBuf<double> pts;          // [total][3]
    std::vector<long long> off;  // offset into pts per mask, masks of all frames in order (hmsg_ctx::mask_first), +1
    long long total = 0;
};

struct InstanceSet {             // merged instances (graph.py:425-448)
    DevBuf<double> pts;
    std::vector<long long> off;
    std::vector<double> box;     // [N][6] AABB (min xyz, max xyz)
    long long total = 0;
};

// live per-kernel timing with HIP events on the handle's own stream (bench.py roofline leg)
struct ProfEntry {
    std::string name;
    hipEvent_t a, b;
    double work;      // algorithmic bytes (or FLOP for the MFMA kernels) of this launch, DESIGN.md section 4
};
struct Prof {
    bool enabled = false;
    bool detail = false;        // also the FINE scopes (the DBSCAN batch's phases: a bracket costs the fold's dependent chain ~3 us each)
    std::vector<ProfEntry> ev;
    void clear() {
        for (auto& e : ev) {
            (void)hipEventDestroy(e.a);
            (void)hipEventDestroy(e.b);
        }
        ev.clear();
    }
};
struct ProfScope {
    Prof* p;
    hipStream_t s;
    size_t i = 0;
    ProfScope(Prof& pr, hipStream_t st, const char* name, double work = 0.0, bool fine = false) : ProfScope(&pr, st, name, work, fine) {}
    ProfScope(Prof* pr, hipStream_t st, const char* name, double work = 0.0, bool fine = false)
        : p(pr && pr->enabled && (!fine || pr->detail) ? pr : nullptr), s(st) {
        if (!p) return;
        ProfEntry e;
        e.name = name;
        e.work = work;
        (void)hipEventCreate(&e.a);
        (void)hipEventCreate(&e.b);
        (void)hipEventRecord(e.a, s);
        i = p->ev.size();
        p->ev.push_back(e);
    }
    ~ProfScope() {
        if (p) (void)hipEventRecord(p->ev[i].b, s);
    }
};

struct hmsg_ctx {
    Prof prof;
    hmsg_config cfg;
    hipStream_t stream = nullptr;
    std::string err;
    int n_frames = 0;       // frames with geometry
    long long n_offered = 0; // frames offered to hmsg_add_frames so far (cfg.skip_frames keeps every skip-th of them)
    int n_feat_frames = 0;  // frames with features handed over (prefix 0..n-1)
    int n_fused = 0;        // frames processed by hmsg_fuse_frames
    int MS = 0;             // mask slots per frame = NW * 64 >= cfg.max_masks (stride of the F_p tables)
    int NW = 0;             // 64-bit words of a pixel's mask-membership bitset
    std::vector<int> nmask;            // masks of every frame with features (SAM returns a different count per frame)
    std::vector<long long> mask_first; // first entry of frame f in masks3d.off (prefix sum of nmask over FUSED frames), +1
    CamK cam;
    double K[9];
    bool have_K = false;
    // resident frame store
    DevBuf<unsigned char> rgb;     // [F][H][W][3]
    DevBuf<unsigned short> depth;  // [F][H][W]
    DevBuf<double> pose;           // [F][16]
    DevBuf<unsigned long long> bits;  // [F][H][W][NW] mask-membership bitset per pixel
    DevBuf<float> fp;              // [F][MS][D]  F_p
    DevBuf<int> nn;                // [F][H][W]   NN index into the filtered cloud (-1 invalid)
    DevBuf<unsigned char> frame_arena;   // very large stores: rgb / depth / bits / nn are views into this ONE block (hmsg_api.hip)
    // global voxel map
    bool map_ready = false;
    GridGeom grid;
    long long V0 = 0;              // voxels before radius-outlier filtering
    long long V = 0;               // filtered cloud size
    DevBuf<unsigned long long> bitmap;  // occupancy of the FILTERED cloud
    DevBuf<unsigned> rank;              // exclusive popcount prefix per word
    DevBuf<double> pts;            // [V][3]
    DevBuf<double> cols;           // [V][3]
    // host copy of the cloud + restated scipy cKDTree over it (hmsg_ckdtree.h): built on a side thread when the map
    // is finalised, consulted only for bit-equal nearest-neighbour ties
    std::vector<double> host_pts;
    std::shared_ptr<struct CKDTree> kd;
    std::thread kd_thread;
    long long n_tie_queries = 0;   // (statistics)
    // NN candidate lists for the cells of voxels deleted by remove_radius_outlier (hmsg_nn.h)
    DevBuf<unsigned long long> bitmap_rm;
    DevBuf<unsigned> rank_rm, cand_off;
    DevBuf<int> cand;
    bool have_cand = false;
    DevBuf<float> sum;             // [V][D]
    DevBuf<unsigned> cnt;          // [V]
    bool feats_final = false;
    DevBuf<float> feats;           // [V][D] = sum / counter
    MaskCloudSet masks3d;
    InstanceSet inst;
    bool merged = false;
    int frame_window = 0;          // first frame with features (hmsg_set_frame_window: this handle owns a frame range)
    bool tree_partial = false;     // inst holds an unfinished list of the sharded hierarchical merge tree
    bool frames_released = false;  // the (very large) frame store was given back before the merge (hmsg_api.hip)
    DevBuf<float> inst_feats;      // [N][D]
    bool pooled = false;
    // the sequential fold running beside the fusion on a worker thread (hmsg_merge.hip: FoldPipe)
    std::shared_ptr<struct FoldPipe> fold_pipe;
    int fold_pipe_frames = 0;      // frames handed to it so far
    DevCache fold_cache;           // the worker's allocator cache between scenes (its thread-local one while it runs)
    bool inst_denoised = false;    // the per-object pcd_denoise_dbscan(0.05, 10) of graph.py:1589-1591 has run
    std::vector<hmsg_node> nodes;  // object nodes (hmsg_build_object_nodes)
    // room clouds of the last hmsg_room_clouds call, resident: per room the selected floor points (indices into the storey's
    // crop), room offsets, floor point -> map point
    DevBuf<int> room_sel, room_fmap;
    DevBuf<long long> room_off_dev;
    std::shared_ptr<void> room_scratch;   // hmsg_room_clouds' work buffers (hmsg_graph.hip: RoomScratch), kept from scene to scene
    int room_n = 0;
    long long room_total = 0;
    std::vector<int> node_label;   // per instance: arg-max label (-1 without a vocabulary)
    // scratch
    DevBuf<unsigned> scan_tmp;
};

// Decoupled look-back over a table of 64-bit status words {epoch:30 | flag:2 | value:32} (flag 1: the tile's aggregate, 2: its
// inclusive prefix).  Words carry the epoch of the call, so the table is never cleared (a stale word reads as "not ready").
// Tiles are numbered by blockIdx: workgroups are dispatched in index order, so every predecessor of a resident tile is
// resident or finished and the spin cannot starve it.  Called by ONE whole wave of the tile with the tile's aggregate:
// publishes it, looks back over the predecessors 64 at a time until it meets a published prefix, publishes the tile's
// inclusive prefix and returns the exclusive one (in every lane).
__device__ __forceinline__ unsigned long long scan_pack(unsigned epoch, unsigned flag, unsigned value) {
    return ((unsigned long long)((epoch << 2) | flag) << 32) | (unsigned long long)value;
}
__device__ __forceinline__ unsigned scan_lookback_prefix(unsigned long long* __restrict__ state, long long tile, unsigned epoch,
                                                         unsigned agg) {
    const int lane = threadIdx.x & 63;
    if (lane == 0)
        __hip_atomic_store(&state[tile], scan_pack(epoch, tile == 0 ? 2u : 1u, agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned prefix = 0;
    if (tile > 0) {
        long long hi = tile - 1;                       // the window is tiles hi, hi-1, ..., hi-63
        for (;;) {
            const long long idx = hi - lane;
            unsigned long long sw = scan_pack(epoch, 2u, 0u);        // before tile 0: prefix 0
            if (idx >= 0) sw = __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned tag = (unsigned)(sw >> 32);
            const unsigned flag = (tag >> 2) == (epoch & 0x3fffffffu) ? (tag & 3u) : 0u;
            const unsigned long long has_prefix = __ballot(flag == 2u), not_ready = __ballot(flag == 0u);
            unsigned long long take = ~0ull;                          // lanes whose value is added
            if (has_prefix) {
                const int first = __ffsll(has_prefix) - 1;            // nearest published prefix
                take = first == 63 ? ~0ull : ((1ull << (first + 1)) - 1ull);
            }
            if (not_ready & take) continue;                           // a needed predecessor is not there yet
            unsigned part = ((take >> lane) & 1ull) ? (unsigned)sw : 0u;
            for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
            prefix += part;
            if (has_prefix) break;
            hi -= 64;
        }
        if (lane == 0)
            __hip_atomic_store(&state[tile], scan_pack(epoch, 2u, prefix + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return prefix;
}
// a fresh epoch for a look-back over `ntiles` tiles of the status table `tmp` (grown, and cleared when it is fresh memory)
unsigned hmsg_scan_epoch(DevBuf<unsigned>& tmp, size_t ntiles, hipStream_t s);
// exclusive prefix sum of u32 (returns total via host sync when `total` != nullptr)
void hmsg_scan_u32(const unsigned* in, unsigned* out, size_t n, hipStream_t s, DevBuf<unsigned>& tmp,
                   unsigned long long* total);
// popcount-prefix of a bitmap: rank[w] = number of set bits in words < w; returns total set bits
unsigned long long hmsg_bitmap_rank(const unsigned long long* bitmap, unsigned* rank, size_t nwords, hipStream_t s,
                                    DevBuf<unsigned>& tmp);

// stable radix sort of (u32 key, u64 value) pairs by the low key_bits bits (hmsg_sort.hip).  Fill keys / vals,
// call hmsg_sort_pairs, read res_keys / res_vals (they point at whichever of the ping-pong buffers holds the result).
struct SortBufs {
    DevBuf<unsigned> keys, keys_alt, hist, scan_tmp;
    DevBuf<unsigned long long> vals, vals_alt;
    unsigned* res_keys = nullptr;
    unsigned long long* res_vals = nullptr;
};
void hmsg_sort_pairs(SortBufs& b, size_t n, int key_bits, hipStream_t s);
// off[k] = first position of key k in the sorted array (keys that occur), off[nkeys] = n
void hmsg_sort_segment_starts(const unsigned* sorted_keys, size_t n, unsigned* off, unsigned nkeys, hipStream_t s);
static inline int bits_for(unsigned long long n) {   // bits needed to hold values 0 .. n-1
    int b = 1;
    while (b < 63 && (1ull << b) < n) ++b;
    return b;
}

// development aid, HMSG_DEBUG_TIMING=1: host wall time of the phases of a call (every lap drains the stream first, so the phases do
// not overlap as they do in a normal run)
struct DbgLaps {
    const char* tag;
    hipStream_t s;
    bool on;
    std::chrono::steady_clock::time_point t;
    DbgLaps(const char* tag_, hipStream_t s_) : tag(tag_), s(s_), on(getenv("HMSG_DEBUG_TIMING") != nullptr) {
        if (on) {
            (void)hipStreamSynchronize(s);
            t = std::chrono::steady_clock::now();
        }
    }
    void lap(const char* what) {
        if (!on) return;
        (void)hipStreamSynchronize(s);
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[hmsg %s] %-26s %.3f ms\n", tag, what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

// development aid: when HMSG_DEBUG_DUMP=<dir> is set, write a device array to <dir>/<name>.bin
static inline void hmsg_dump(const char* name, const void* dev, size_t bytes, hipStream_t s) {
    const char* dir = getenv("HMSG_DEBUG_DUMP");
    if (!dir || !bytes) return;
    std::vector<char> host(bytes);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(host.data(), dev, bytes, hipMemcpyDeviceToHost);
    std::string path = std::string(dir) + "/" + name + ".bin";
    if (FILE* f = fopen(path.c_str(), "wb")) {
        fwrite(host.data(), 1, bytes, f);
        fclose(f);
    }
}

void hmsg_kd_start(hmsg_ctx* h);        // hmsg_api.hip: download the cloud, build the cKDTree restatement on a thread
void hmsg_kd_join(hmsg_ctx* h);
void hmsg_build_map(hmsg_ctx* h);       // hmsg_map.hip
void hmsg_fuse(hmsg_ctx* h);            // hmsg_fuse.hip
void hmsg_merge(hmsg_ctx* h);           // hmsg_merge.hip
void hmsg_fold_pipe_feed(hmsg_ctx* h, int f0, int nfr);   // hmsg_merge.hip: masks of frames [f0, f0 + nfr) are complete
void hmsg_fold_pipe_abort(hmsg_ctx* h);                  // hmsg_merge.hip
void hmsg_merge_tree_local_impl(hmsg_ctx* h, int total_frames, double* th_next, long long* lists_now, long long* my_index);
void hmsg_merge_tree_join_impl(hmsg_ctx* h, int n_ext, const long long* ext_sizes, const double* ext_pts, double th, int final_pass);
void hmsg_denoise_inst(hmsg_ctx* h, double eps, int min_points);   // hmsg_merge.hip
void hmsg_room_share(hmsg_ctx* h, int R, const long long* vert_off, const double* verts_xz, double radius,
                     double* share_out);                              // hmsg_merge.hip
void hmsg_pool(hmsg_ctx* h);            // hmsg_pool.hip
long long hmsg_voxel_ds(hmsg_ctx* h, const double* pts, long long n, double vs, double* out);   // hmsg_merge.hip
