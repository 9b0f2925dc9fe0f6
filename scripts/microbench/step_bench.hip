// Development microbenchmark (not part of the product), written at the end of round 3 for the FIRST GPU call of the next
// round: the numbers that decide how a merge-fold step with fewer phases has to be built.  The fold is a
// chain of ~25 dependent launches of 4-40 us on ~3x10^5 points (7.2 MB); launch_bench.hip measured 2.6 us per empty launch
// and 50 us per grid barrier with 1024 workgroups.  Open questions, one section each:
//
//   A. grid barrier vs. number of workgroups (8 .. 1024) for two barriers: every workgroup polling one counter, and a
//      counter + generation flag (the last arriver flips the flag, the others poll it with plain loads).
//   B. one streaming phase over 3x10^5 points (read 24 B, write 4 B per point) as a function of the grid: separate
//      launches vs. phases of one persistent kernel -- where is the floor of "a phase", and with how few workgroups?
//   C. device-scope atomic round trip (one lane, dependent fetch_adds on one word) and a flag ping-pong between two
//      workgroups (one hand-over = one release store + one acquire poll), i.e. what a dependency between workgroups costs.
//
//   hipcc --offload-arch=gfx950 -O3 -o step_bench step_bench.hip && ./step_bench
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                        \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// every spin is bounded (a workgroup that is not resident must not hang the box: the numbers are then wrong, not the GPU)
#define SPIN_CAP (1 << 22)

// ---- barriers (bar[0]: arrivals, bar[32]: generation -- on different cache lines)
__device__ __forceinline__ void barrier_counter(unsigned* bar, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        for (int it = 0; it < SPIN_CAP && __hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target; ++it) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}
__device__ __forceinline__ void barrier_flag(unsigned* bar, unsigned phase) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(bar, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1 == phase * gridDim.x) __hip_atomic_store(bar + 32, phase, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        else
            for (int it = 0; it < SPIN_CAP && __hip_atomic_load(bar + 32, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < phase; ++it) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

template <int KIND>
__global__ void k_barriers(unsigned* bar, int phases, unsigned* sink) {
    unsigned acc = 0;
    for (int p = 1; p <= phases; ++p) {
        acc += threadIdx.x * p;
        if (KIND == 0) barrier_counter(bar, (unsigned)p * gridDim.x);
        else barrier_flag(bar, (unsigned)p);
    }
    if (acc == 0xffffffffu) *sink = acc;
}

// ---- one streaming phase: cell id of every point (grid-stride)
__device__ __forceinline__ void phase_body(const double* pts, unsigned* out, long long n, int p) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const double x = pts[i * 3], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
        out[i] = (unsigned)(int)(x * 17.3) * 73856093u ^ (unsigned)(int)(y * 17.3) * 19349663u ^ (unsigned)(int)(z * 17.3 + p);
    }
}
__global__ void k_phase(const double* pts, unsigned* out, long long n, int p) { phase_body(pts, out, n, p); }
__global__ void k_phases_persistent(const double* pts, unsigned* out, long long n, int phases, unsigned* bar) {
    for (int p = 1; p <= phases; ++p) {
        phase_body(pts, out, n, p);
        barrier_flag(bar, (unsigned)p);
    }
}

// ---- C: atomics
__global__ void k_atomic_chain(unsigned* w, int hops, unsigned* out) {
    unsigned v = 0;
    for (int h = 0; h < hops; ++h) v = __hip_atomic_fetch_add(w + (v & 0), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *out = v;
}
__global__ void k_pingpong(unsigned* flag, int rounds, int far) {
    // workgroup 0 and workgroup `far` hand a token back and forth (the others exit at once)
    if (threadIdx.x != 0 || (blockIdx.x != 0 && (int)blockIdx.x != far)) return;
    const unsigned me = blockIdx.x == 0 ? 0u : 1u;
    for (int r = 0; r < rounds; ++r) {
        const unsigned want = (unsigned)r * 2 + me;
        for (int it = 0; it < SPIN_CAP && __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != want; ++it) {
        }
        __hip_atomic_store(flag, want + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    hipStream_t s;
    CK(hipStreamCreate(&s));
    unsigned* d;
    CK(hipMalloc(&d, 4096));
    CK(hipMemset(d, 0, 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    // A. barriers
    const int phases = 50;
    for (int threads : {256, 1024})
        for (int blocks : {8, 16, 32, 64, 128, 256, 512, 1024}) {
            if (blocks * threads > prop.multiProcessorCount * 2048) continue;       // all workgroups must be resident
            double us[2] = {0, 0}, base = 0;
            for (int kind = 0; kind < 3; ++kind) {
                const int n = 50;
                float ms = 0, tot = 0;
                for (int i = 0; i < n + 3; ++i) {
                    CK(hipMemsetAsync(d, 0, 256, s));
                    CK(hipEventRecord(e0, s));
                    if (kind == 0) hipLaunchKernelGGL(k_barriers<0>, dim3(blocks), dim3(threads), 0, s, d, phases, d + 100);
                    else if (kind == 1) hipLaunchKernelGGL(k_barriers<1>, dim3(blocks), dim3(threads), 0, s, d, phases, d + 100);
                    else hipLaunchKernelGGL(k_barriers<1>, dim3(blocks), dim3(threads), 0, s, d, 0, d + 100);
                    CK(hipEventRecord(e1, s));
                    CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (i >= 3) tot += ms;
                }
                if (kind < 2) us[kind] = tot / n * 1000.0;
                else base = tot / n * 1000.0;
            }
            printf("A. %4d workgroups x %4d threads: counter barrier %.2f us, counter+flag barrier %.2f us (kernel without barriers %.1f us)\n",
                   blocks, threads, (us[0] - base) / phases, (us[1] - base) / phases, base);
        }

    // B. a streaming phase over 3x10^5 points
    {
        const long long n = 300000;
        std::vector<double> h((size_t)n * 3);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 100000) * 1e-4;
        double* pts;
        unsigned* out;
        CK(hipMalloc(&pts, h.size() * 8));
        CK(hipMalloc(&out, (size_t)n * 4));
        CK(hipMemcpy(pts, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        const int P = 20;
        for (int threads : {256, 1024})
            for (int blocks : {16, 32, 64, 128, 256, 512, 1172}) {
                if (blocks * threads > prop.multiProcessorCount * 2048) continue;
                float ms = 0;
                double sep = 0, per = 0;
                const int reps = 20;
                for (int i = 0; i < reps + 2; ++i) {
                    CK(hipEventRecord(e0, s));
                    for (int p = 1; p <= P; ++p) hipLaunchKernelGGL(k_phase, dim3(blocks), dim3(threads), 0, s, pts, out, n, p);
                    CK(hipEventRecord(e1, s));
                    CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (i >= 2) sep += ms;
                    CK(hipMemsetAsync(d, 0, 256, s));
                    CK(hipEventRecord(e0, s));
                    hipLaunchKernelGGL(k_phases_persistent, dim3(blocks), dim3(threads), 0, s, pts, out, n, P, d);
                    CK(hipEventRecord(e1, s));
                    CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (i >= 2) per += ms;
                }
                printf("B. 3e5 points, %4d workgroups x %4d threads: %.2f us per phase as separate launches, %.2f us per phase of a persistent kernel\n",
                       blocks, threads, sep / reps / P * 1000.0, per / reps / P * 1000.0);
            }
        CK(hipFree(pts));
        CK(hipFree(out));
    }

    // C. atomics
    {
        const int hops = 20000;
        CK(hipMemsetAsync(d, 0, 256, s));
        hipLaunchKernelGGL(k_atomic_chain, dim3(1), dim3(1), 0, s, d, hops, d + 100);
        CK(hipStreamSynchronize(s));
        double t0 = now_us();
        hipLaunchKernelGGL(k_atomic_chain, dim3(1), dim3(1), 0, s, d, hops, d + 100);
        CK(hipStreamSynchronize(s));
        double t1 = now_us();
        printf("C. dependent device-scope fetch_add on one word: %.0f ns per round trip\n", (t1 - t0) * 1000.0 / hops);
        const int rounds = 5000;
        for (int far : {1, 2, 7, 8, 33, 255}) {
            CK(hipMemsetAsync(d, 0, 256, s));
            CK(hipStreamSynchronize(s));
            double a = now_us();
            hipLaunchKernelGGL(k_pingpong, dim3(256), dim3(64), 0, s, d, rounds, far);
            CK(hipStreamSynchronize(s));
            double b = now_us();
            printf("C. flag ping-pong between workgroups 0 and %3d: %.0f ns per hand-over\n", far, (b - a) * 1000.0 / (rounds * 2));
        }
    }
    CK(hipFree(d));
    return 0;
}
