// Development microbenchmark (not part of the product): what a HOT WORD costs.  The fold's walk kernels push work items through
// global counters (one atomicAdd per wave) and read union-find roots with agent-scope atomic loads; round 3's per-kernel
// times (k_f_count 88 us for 7.7e3 waves, k_f_linkpre1 54 us for 2.4e4 threads) look like serialisation on one address.
//   hipcc --offload-arch=gfx950 -O3 -o atom_bench atom_bench.hip && ./atom_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// mode 0: lane 0 of every wave atomicAdd on ONE word; 1: on its own cache line; 2: every lane agent-scope atomic load of one word;
// 3: every lane plain load of one word; 4: every lane atomicAdd on one word (no wave aggregation); 5: lane 0 atomicMin+atomicAdd on one word pair
__global__ void k(unsigned* w, unsigned* out, int mode) {
    const unsigned gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    unsigned v = 0;
    if (mode == 0) { if (lane == 0) v = atomicAdd(w, 1u); }
    else if (mode == 1) { if (lane == 0) v = atomicAdd(w + 16 * (gw & 4095u), 1u); }
    else if (mode == 2) v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (mode == 3) v = *(volatile unsigned*)w;
    else if (mode == 4) v = atomicAdd(w, 1u);
    else if (mode == 5) { if (lane == 0) { atomicMin(w + 1, gw); v = atomicAdd(w, 1u); } }
    if (v == 0xfffffff0u) out[0] = v;
}
int main() {
    unsigned *w, *out;
    CK(hipMalloc(&w, 4096 * 64 + 64));
    CK(hipMalloc(&out, 64));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char* names[6] = {"lane0 atomicAdd, one word", "lane0 atomicAdd, own line", "all lanes atomic LOAD, one word", "all lanes plain load, one word",
                            "all lanes atomicAdd, one word", "lane0 atomicMin+atomicAdd, one pair"};
    for (int threads : {256, 1024})
        for (int waves : {1024, 8192, 32768, 131072}) {
            for (int mode = 0; mode < 6; ++mode) {
                float tot = 0;
                const int n = 20;
                for (int i = 0; i < n + 2; ++i) {
                    CK(hipMemsetAsync(w, 0, 4096 * 64, s));
                    CK(hipEventRecord(e0, s));
                    hipLaunchKernelGGL(k, dim3(waves * 64 / threads), dim3(threads), 0, s, w, out, mode);
                    CK(hipEventRecord(e1, s));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (i >= 2) tot += ms;
                }
                printf("%6d waves (block %4d): %-38s %8.2f us\n", waves, threads, names[mode], tot / n * 1000.0);
            }
        }
    return 0;
}
