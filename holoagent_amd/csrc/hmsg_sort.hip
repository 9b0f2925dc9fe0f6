// Stable LSD radix sort of (u32 key, u64 value) pairs + the "ordered segment" helpers built on it.
//
// Why the path needs a STABLE sort: every voxel mean of the reference is a sequential float64 sum in INPUT
// order (Open3D VoxelDownSample / AccumulatedPoint::AddPoint, restated in oracle/hmsg_oracle.py
// o3d_voxel_down_sample; call sites graph.py:348, generic.py:188, graph.py:456).  Floating-point addition is
// not associative, and the nearest-neighbour decisions downstream hinge on the last bit of those means (a merged
// instance point that averages two map voxels is equidistant from both up to rounding), so the sums are
// reproduced bit for bit: records are sorted by voxel slot with their input order kept, and one lane then walks
// each voxel's records in order.
//
// Kernel layout per pass (digit of RB <= 11 bits): tiles of 256 threads x SORT_IPT items; a wave owns a
// contiguous 64*SORT_IPT slice of the tile and walks it 64 items at a time, so "position in the tile" =
// (wave, iteration, lane).  k_sort_hist counts digits per tile; one look-back scan over the digit-major
// [digit][tile] table gives every (digit, tile) its global base; k_sort_scatter re-reads the tile, ranks equal
// digits inside a wave with RB ballots (multi-split), adds the running per-wave digit count kept in LDS, then the
// cross-wave prefix, and writes each pair to base + rank.
#include "hmsg_common.h"

#define SORT_IPT 16
#define SORT_TILE (256 * SORT_IPT)
#define SORT_MAXBINS 2048

__global__ void __launch_bounds__(256) k_sort_hist(const unsigned* __restrict__ keys, size_t n, int shift, int rb,
                                                   unsigned nblocks, unsigned* __restrict__ blockhist) {
    __shared__ unsigned hist[SORT_MAXBINS];
    const int nb = 1 << rb;
    for (int d = threadIdx.x; d < nb; d += 256) hist[d] = 0u;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * SORT_TILE;
    for (int j = 0; j < SORT_IPT; ++j) {
        const size_t i = base + (size_t)j * 256 + threadIdx.x;
        if (i < n) atomicAdd(&hist[(keys[i] >> shift) & (unsigned)(nb - 1)], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < nb; d += 256) blockhist[(size_t)d * nblocks + blockIdx.x] = hist[d];
}

__global__ void __launch_bounds__(256) k_sort_scatter(const unsigned* __restrict__ keys, const unsigned long long* __restrict__ vals,
                                                      size_t n, int shift, int rb, unsigned nblocks,
                                                      const unsigned* __restrict__ blockbase, unsigned* __restrict__ okeys,
                                                      unsigned long long* __restrict__ ovals) {
    __shared__ unsigned whist[4][SORT_MAXBINS];
    const int nb = 1 << rb;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int d = threadIdx.x; d < nb * 4; d += 256) whist[d / nb][d % nb] = 0u;
    __syncthreads();
    const size_t wbase = (size_t)blockIdx.x * SORT_TILE + (size_t)w * 64 * SORT_IPT;
    unsigned key[SORT_IPT], rnk[SORT_IPT];
    for (int j = 0; j < SORT_IPT; ++j) {
        const size_t i = wbase + (size_t)j * 64 + lane;
        const bool valid = i < n;
        key[j] = valid ? keys[i] : 0u;
        const unsigned d = (key[j] >> shift) & (unsigned)(nb - 1);
        unsigned long long mask = __ballot(valid);
        for (int b = 0; b < rb; ++b) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            mask &= ((d >> b) & 1u) ? m : ~m;
        }
        // (all lanes take part in the shuffles; invalid lanes form no group)
        const int leader = valid ? __ffsll(mask) - 1 : lane;
        unsigned old = 0u;
        if (valid && lane == leader) {
            old = whist[w][d];
            whist[w][d] = old + (unsigned)__popcll(mask);
        }
        old = __shfl(old, leader);
        rnk[j] = old + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    // cross-wave exclusive prefix per digit + the tile's global base
    for (int d = threadIdx.x; d < nb; d += 256) {
        unsigned run = blockbase[(size_t)d * nblocks + blockIdx.x];
        for (int q = 0; q < 4; ++q) {
            const unsigned c = whist[q][d];
            whist[q][d] = run;
            run += c;
        }
    }
    __syncthreads();
    for (int j = 0; j < SORT_IPT; ++j) {
        const size_t i = wbase + (size_t)j * 64 + lane;
        if (i >= n) continue;
        const unsigned d = (key[j] >> shift) & (unsigned)(nb - 1);
        const size_t pos = (size_t)whist[w][d] + rnk[j];
        okeys[pos] = key[j];
        ovals[pos] = vals[i];
    }
}

// segment starts of a sorted key array: off[k] = first position holding key k (every key in [0, nkeys) that
// occurs), off[nkeys] = n.  Keys that do not occur keep the value the caller initialised them with.
__global__ void k_sort_bounds(const unsigned* __restrict__ keys, size_t n, unsigned* __restrict__ off, unsigned nkeys) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) off[nkeys] = (unsigned)n;
    if (i >= n) return;
    const unsigned k = keys[i];
    if (i == 0 || keys[i - 1] != k) off[k] = (unsigned)i;
}

void hmsg_sort_pairs(SortBufs& b, size_t n, int key_bits, hipStream_t s) {
    b.res_keys = b.keys.p;
    b.res_vals = b.vals.p;
    if (n <= 1 || key_bits <= 0) return;
    HMSG_REQUIRE(n < ((size_t)1 << 32), HMSG_ERR_UNSUPPORTED, "sort: more than 2^32 records");
    key_bits = std::min(key_bits, 32);
    const int passes = (key_bits + 10) / 11;
    const int rb = (key_bits + passes - 1) / passes;
    const unsigned nblocks = cdiv(n, SORT_TILE);
    b.keys_alt.ensure(n);
    b.vals_alt.ensure(n);
    b.hist.ensure(((size_t)nblocks << rb) + 1);
    unsigned* ka = b.keys.p;
    unsigned long long* va = b.vals.p;
    unsigned* kb = b.keys_alt.p;
    unsigned long long* vb = b.vals_alt.p;
    for (int p = 0; p < passes; ++p) {
        const int shift = p * rb;
        hipLaunchKernelGGL(k_sort_hist, dim3(nblocks), dim3(256), 0, s, (const unsigned*)ka, n, shift, rb, nblocks, b.hist.p);
        hmsg_scan_u32(b.hist.p, b.hist.p, (size_t)nblocks << rb, s, b.scan_tmp, nullptr);
        hipLaunchKernelGGL(k_sort_scatter, dim3(nblocks), dim3(256), 0, s, (const unsigned*)ka, (const unsigned long long*)va, n,
                           shift, rb, nblocks, (const unsigned*)b.hist.p, kb, vb);
        HMSG_CHECK_LAUNCH();
        std::swap(ka, kb);
        std::swap(va, vb);
    }
    b.res_keys = ka;
    b.res_vals = va;
}

void hmsg_sort_segment_starts(const unsigned* sorted_keys, size_t n, unsigned* off, unsigned nkeys, hipStream_t s) {
    hipLaunchKernelGGL(k_sort_bounds, dim3(std::max(1u, cdiv(n, 256))), dim3(256), 0, s, sorted_keys, n, off, nkeys);
    HMSG_CHECK_LAUNCH();
}

// ---- test hook (include/hmsg.h: hmsg_test_sort_pairs) ------------------------------------------------------
extern "C" int hmsg_test_sort_pairs(uint32_t* keys, uint64_t* vals, int64_t n, int32_t key_bits) {
    try {
        hipStream_t s = nullptr;
        HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        {
            SortBufs b;
            b.keys.alloc((size_t)std::max<int64_t>(n, 1));
            b.vals.alloc((size_t)std::max<int64_t>(n, 1));
            HIP_TRY(hipMemcpyAsync(b.keys.p, keys, (size_t)n * 4, hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpyAsync(b.vals.p, vals, (size_t)n * 8, hipMemcpyHostToDevice, s));
            hmsg_sort_pairs(b, (size_t)n, key_bits, s);
            HIP_TRY(hipMemcpyAsync(keys, b.res_keys, (size_t)n * 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(vals, b.res_vals, (size_t)n * 8, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
        }
        (void)hipStreamDestroy(s);
        return HMSG_OK;
    } catch (const hmsg_error& e) {
        fprintf(stderr, "hmsg_test_sort_pairs: %s\n", e.msg.c_str());
        return e.code;
    }
}

// ---- test hook: repeat_add (hmsg_common.h) against the plain loop it replaces ------------------------------
__global__ void k_test_repeat_add(const double* __restrict__ s, const double* __restrict__ p, const int* __restrict__ len,
                                  double* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = repeat_add(s[i], p[i], len[i]);
}
extern "C" int hmsg_test_repeat_add(const double* s, const double* p, const int32_t* len, double* out, int64_t n) {
    try {
        if (n <= 0) return HMSG_OK;
        DevBuf<double> ds, dp, dout;
        DevBuf<int> dl;
        ds.alloc((size_t)n); dp.alloc((size_t)n); dout.alloc((size_t)n); dl.alloc((size_t)n);
        HIP_TRY(hipMemcpy(ds.p, s, (size_t)n * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dp.p, p, (size_t)n * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dl.p, len, (size_t)n * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_test_repeat_add, dim3(cdiv((size_t)n, 256)), dim3(256), 0, 0, (const double*)ds.p, (const double*)dp.p,
                           (const int*)dl.p, dout.p, (long long)n);
        HMSG_CHECK_LAUNCH();
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(out, dout.p, (size_t)n * 8, hipMemcpyDeviceToHost));
        return HMSG_OK;
    } catch (const hmsg_error& e) {
        fprintf(stderr, "hmsg_test_repeat_add: %s\n", e.msg.c_str());
        return e.code;
    }
}
