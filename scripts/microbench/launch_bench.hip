// Development microbenchmark (not part of the product): what a short kernel costs on this GPU, and what the
// alternatives to "one launch per phase" would cost.  The merge fold runs ~25 launches of 5-60 us per frame
// (DESIGN.md 4b, 5); these numbers decide whether a captured graph or one persistent kernel with grid barriers
// is worth building.
//
//   hipcc --offload-arch=gfx950 -O3 -o launch_bench launch_bench.hip && ./launch_bench
//
// 1. stream launches of an empty kernel, back to back            -> us per launch (host rate + dispatch gap)
// 2. the same 20 kernels as one captured hipGraph, relaunched    -> us per kernel node
// 3. one persistent kernel, 20 phases separated by grid barriers -> us per barrier (n_cu x 4 workgroups)
// 4. dependent-load chain through a 64 MB table (one lane)        -> ns per L2 / HBM round trip
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

__global__ void k_empty(unsigned* sink) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && sink == (unsigned*)1) *sink = 0;
}

__global__ void k_phases(unsigned* bar, int phases, unsigned* sink) {
    // grid barrier: every workgroup bumps a counter, then waits until it reaches phase * gridDim.x
    unsigned acc = 0;
    for (int p = 1; p <= phases; ++p) {
        acc += threadIdx.x * p;
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)p * gridDim.x) {
            }
        }
        __syncthreads();
    }
    if (acc == 0xffffffffu) *sink = acc;
}

__global__ void k_chase(const unsigned* table, unsigned start, int hops, unsigned* out) {
    unsigned i = start;
    for (int h = 0; h < hops; ++h) i = table[i];
    *out = i;
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    hipStream_t s;
    CK(hipStreamCreate(&s));
    unsigned* d;
    CK(hipMalloc(&d, 64));
    CK(hipMemset(d, 0, 64));

    // 1. stream launches
    for (int grid : {1, 256, 2048}) {
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, s, d);
        CK(hipStreamSynchronize(s));
        const int n = 2000;
        double t0 = now_us();
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, s, d);
        double t1 = now_us();
        CK(hipStreamSynchronize(s));
        double t2 = now_us();
        printf("1. stream launches, grid %4d x 256: %.2f us/launch end to end (host enqueue alone %.2f us)\n", grid,
               (t2 - t0) / n, (t1 - t0) / n);
    }

    // 2. captured graph of 20 kernels
    {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, d);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        const int n = 200;
        double t0 = now_us();
        for (int i = 0; i < n; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        double t1 = now_us();
        printf("2. hipGraph of 20 empty kernels (grid 256): %.2f us per graph launch = %.2f us per kernel node\n",
               (t1 - t0) / n, (t1 - t0) / n / 20);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }

    // 3. persistent kernel with grid barriers (workgroups must all be resident: n_cu x 4 of 256 threads)
    {
        const int blocks = prop.multiProcessorCount * 4, phases = 20;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(d, 0, 4, s));
            hipLaunchKernelGGL(k_phases, dim3(blocks), dim3(256), 0, s, d, phases, d + 1);
        }
        CK(hipStreamSynchronize(s));
        const int n = 100;
        double t0 = now_us();
        for (int i = 0; i < n; ++i) {
            CK(hipMemsetAsync(d, 0, 4, s));
            hipLaunchKernelGGL(k_phases, dim3(blocks), dim3(256), 0, s, d, phases, d + 1);
        }
        CK(hipStreamSynchronize(s));
        double t1 = now_us();
        printf("3. persistent kernel, %d workgroups, %d grid barriers: %.2f us per launch = %.2f us per barrier (incl. launch + memset)\n",
               blocks, phases, (t1 - t0) / n, (t1 - t0) / n / phases);
    }

    // 4. dependent-load chain
    for (size_t mb : {1, 16, 512}) {
        const size_t n = mb * 1024 * 1024 / 4;
        std::vector<unsigned> h(n);
        // one cycle through the table with a large odd stride (defeats the prefetchers, stays a permutation)
        const size_t stride = (n / 2 + 12345) | 1;
        for (size_t i = 0; i < n; ++i) h[i] = (unsigned)((i + stride) % n);
        unsigned* t;
        CK(hipMalloc(&t, n * 4));
        CK(hipMemcpy(t, h.data(), n * 4, hipMemcpyHostToDevice));
        const int hops = 20000;
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, s, t, 0u, hops, d + 2);
        CK(hipStreamSynchronize(s));
        double t0 = now_us();
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, s, t, 1u, hops, d + 2);
        CK(hipStreamSynchronize(s));
        double t1 = now_us();
        printf("4. dependent loads through a %zu MB table: %.0f ns per round trip\n", mb, (t1 - t0) * 1000.0 / hops);
        CK(hipFree(t));
    }
    CK(hipFree(d));
    return 0;
}
