// A7: per-instance feature pooling (fsr_vln/memory/hmsg/graph/graph.py:450-491) with
// feats_denoise_dbscan (utils/graph_utils.py:682-728): for every merged instance
//   voxel_down_sample -> nearest map voxel (dist <= 0.8) -> gather the voxel features -> cosine DBSCAN
//   (eps 0.01, min_samples 100, sklearn semantics, SURVEY hazard 18) -> mean over the largest cluster.
//
// MI355X design: all instances are processed in ONE batch.  The n_i x n_i cosine matrices are never
// stored as floats: a float32 MFMA (v_mfma_f32_32x32x2_f32 -- an exact fmaf chain, so the result does not
// depend on tiling) kernel produces 32x32 tiles of X^.X^T, thresholds 1 - s <= eps in registers and writes
// one adjacency BIT per pair plus per-row neighbour counts.  Core components, border assignment and the
// largest-cluster choice then work on the bit matrix (n^2/8 bytes).  The final mean adds rows in index
// order in float32, which is what np.mean(axis=0) does on a C-contiguous [n, D] array.
#include "hmsg_cloudops.h"
#include "hmsg_nn.h"

#include <algorithm>
#include <cmath>

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct PoolSeg {            // one instance
    long long row_base;     // first row in the concatenated feature matrix
    long long bit_base;     // first u32 word of its adjacency bit matrix
    int n;                  // rows (valid points)
    int nw;                 // u32 words per row = ceil(n / 32)
    long long tile_base;    // first Gram tile id (upper triangle, super-tile order: gram_tile_of)
    int nt;                 // Gram tiles per side = ceil(n / GRAM_T)
    int pad;
};

// ---- NN of the down-sampled instance points; keep dist <= max_dist (graph.py:458-460)
__global__ void k_pool_nn(const double* __restrict__ q, long long N, NNIndex I, double max_dist, int* __restrict__ idx,
                          unsigned* __restrict__ valid, TieList ties) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double d2 = 0;
    int ntie = 0;
    int v = nn_search(I, q[i * 3], q[i * 3 + 1], q[i * 3 + 2], &d2, &ntie);
    if (ntie > 1) tie_push(ties, i, q[i * 3], q[i * 3 + 1], q[i * 3 + 2]);
    idx[i] = v;
    valid[i] = (v >= 0 && __dsqrt_rn(d2) <= max_dist) ? 1u : 0u;
}

// ---- gather + nan_to_num (graph.py:476-477) + row L2-normalised copy (sklearn cosine_distances)
__global__ void k_pool_gather(const int* __restrict__ idx, const unsigned* __restrict__ valid, const unsigned* __restrict__ pos,
                              long long N, const float* __restrict__ feats, int D, float* __restrict__ X, float* __restrict__ Xn) {
    const int lane = threadIdx.x & 63;
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= N || !valid[i]) return;
    const float* src = feats + (size_t)idx[i] * D;
    float* dx = X + (size_t)pos[i] * D;
    float* dn = Xn + (size_t)pos[i] * D;
    float n2 = 0.f;
    for (int e = lane; e < D; e += 64) {
        float v = src[e];
        if (v != v) v = 0.f;                                   // nan -> 0
        else if (v > 3.4028234663852886e38f) v = 3.4028234663852886e38f;    // +inf -> max
        else if (v < -3.4028234663852886e38f) v = -3.4028234663852886e38f;  // -inf -> min
        dx[e] = v;
        n2 += v * v;
    }
    float nrm = __fsqrt_rn(wave_sum_f32(n2));
    if (nrm == 0.f) nrm = 1.f;                                  // sklearn normalize: zero rows stay zero
    for (int e = lane; e < D; e += 64) dn[e] = __fdiv_rn(dx[e], nrm);
}

// ---- Gram tiles on the matrix cores -> adjacency bits + neighbour counts
// One 256-thread workgroup per 128x128 tile of X^ X^T of one instance, the four waves in a 2x2 arrangement, each
// wave a 64x64 quadrant = 2x2 accumulators of 32x32 (v_mfma_f32_32x32x2_f32: lane l feeds A[i = l&31][k = l>>5],
// B[k = l>>5][j = l&31]).  The A and B row panels (128 rows x 32 k) are staged once per workgroup in LDS with
// 16-byte global loads; rows are padded to 33 floats so the fragment reads are conflict free.  A 128x128 tile
// reads 2 x 128 x D floats for 128 x 128 x D multiply-adds -- half the L2/HBM traffic per FLOP of a 64x64 tile,
// which is what bounded the kernel -- and four MFMAs share four LDS fragment reads.
// Only tiles with tj >= ti are computed (the matrix is symmetric): the mirrored adjacency words are built
// from the same accumulators (a lane owns one column of a 32x32 block = one mirrored row half).
#define GRAM_T 128
#define GRAM_SUP 8          /* tiles per side of a super-tile */
// Tile order inside an instance: super-tiles of GRAM_SUP x GRAM_SUP tiles, super-rows top to bottom, the diagonal
// super-tile (a triangle) first, then the ones to its right, row-major inside.  Consecutive tile numbers therefore share
// row / column panels: the 64 tiles of one super-tile read 16 panels instead of 65.  (Row-major over the whole
// triangle re-fetched every column panel once per tile: 40.8 GB of L2 misses for 1.95 GB of input, PMC round 2.
// Measured on the MI355X: 110 -> 116 TFLOP/s; dealing every XCD a contiguous stretch of the sequence on top, or
// prefetching the next k slab into registers, changed nothing -- the kernel is not bound by its L2 misses.)
__device__ __forceinline__ void gram_tile_of(long long t, int nt, int& ti, int& tj) {
    int I = 0, r0 = 0, hgt = 0;
    for (;;) {
        r0 = I * GRAM_SUP;
        hgt = min(nt, r0 + GRAM_SUP) - r0;
        const long long cnt = (long long)hgt * (nt - r0) - (long long)hgt * (hgt - 1) / 2;
        if (t < cnt) break;
        t -= cnt;
        ++I;
    }
    const long long dcnt = (long long)hgt * (hgt + 1) / 2;
    if (t < dcnt) {
        int r = 0, len = hgt;
        while (t >= len) {
            t -= len;
            --len;
            ++r;
        }
        ti = r0 + r;
        tj = ti + (int)t;
        return;
    }
    t -= dcnt;
    for (int J = I + 1;; ++J) {
        const int w = min(GRAM_SUP, nt - J * GRAM_SUP);
        const long long c = (long long)hgt * w;
        if (t < c) {
            ti = r0 + (int)(t / w);
            tj = J * GRAM_SUP + (int)(t % w);
            return;
        }
        t -= c;
    }
}
__global__ void __launch_bounds__(256) k_pool_gram(const float* __restrict__ Xn, int D, const PoolSeg* __restrict__ segs, int K,
                                                   float eps, unsigned* __restrict__ adj, unsigned* __restrict__ ncount) {
    __shared__ float sa[GRAM_T][33];
    __shared__ float sb[GRAM_T][33];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wr = wv >> 1, wc = wv & 1;
    const long long tile = blockIdx.x;
    int lo = 0, hi = K - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (segs[mid].tile_base <= tile) lo = mid; else hi = mid - 1;
    }
    const PoolSeg sg = segs[lo];
    int ti, tj;
    gram_tile_of(tile - sg.tile_base, sg.nt, ti, tj);
    const int r0 = ti * GRAM_T, c0 = tj * GRAM_T;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float* base = Xn + (size_t)sg.row_base * D;
    const int lrow = tid >> 1, lk = (tid & 1) * 16;         // this thread stages 16 consecutive k of one row
    const bool vec_ok = (D & 3) == 0;                       // rows are 16-byte aligned
    // rows past the instance's end are CLAMPED to its last row instead of zero-filled: an output element depends on one
    // row of each panel only and the epilogue masks row / col >= n, so the loads stay unconditional 16-byte loads
    // (with a per-element select the compiler split them into dword loads: 22 -> 38 ms on the MI355X)
    const int ra = min(r0 + lrow, sg.n - 1), rb = min(c0 + lrow, sg.n - 1);
    for (int k0 = 0; k0 < D; k0 += 32) {
        if (vec_ok && k0 + lk + 16 <= D) {
            const float4* pa = reinterpret_cast<const float4*>(base + (size_t)ra * D + k0 + lk);
            const float4* pb = reinterpret_cast<const float4*>(base + (size_t)rb * D + k0 + lk);
            for (int u = 0; u < 4; ++u) {
                const float4 va = pa[u], vb = pb[u];
                sa[lrow][lk + 4 * u] = va.x; sa[lrow][lk + 4 * u + 1] = va.y; sa[lrow][lk + 4 * u + 2] = va.z; sa[lrow][lk + 4 * u + 3] = va.w;
                sb[lrow][lk + 4 * u] = vb.x; sb[lrow][lk + 4 * u + 1] = vb.y; sb[lrow][lk + 4 * u + 2] = vb.z; sb[lrow][lk + 4 * u + 3] = vb.w;
            }
        } else {
            for (int u = 0; u < 16; ++u) {
                int k = k0 + lk + u;
                sa[lrow][lk + u] = k < D ? base[(size_t)ra * D + k] : 0.f;
                sb[lrow][lk + u] = k < D ? base[(size_t)rb * D + k] : 0.f;
            }
        }
        __syncthreads();
        for (int k = 0; k < 32; k += 2) {
            const int kk = k + (lane >> 5);
            const float a0 = sa[wr * 64 + (lane & 31)][kk], a1 = sa[wr * 64 + 32 + (lane & 31)][kk];
            const float b0 = sb[wc * 64 + (lane & 31)][kk], b1 = sb[wc * 64 + 32 + (lane & 31)][kk];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    // epilogue per 32x32 block: d = 1 - s, clipped to [0, 2], diagonal forced to 0; neighbour iff d <= eps
    const bool diag_tile = (ti == tj);                      // computed in full: no mirrored writes
    for (int si = 0; si < 2; ++si)
        for (int sj = 0; sj < 2; ++sj) {
            const int rbase = r0 + wr * 64 + si * 32, cbase = c0 + wc * 64 + sj * 32;
            const int col = cbase + (lane & 31);
            unsigned mirror = 0u;                           // bits over rows rbase..rbase+31 for column `col`
            for (int r = 0; r < 16; ++r) {
                const int rloc = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int row = rbase + rloc;
                float d = __fadd_rn(-acc[si][sj][r], 1.0f);
                d = fminf(fmaxf(d, 0.f), 2.f);
                if (row == col) d = 0.f;
                const bool nb = row < sg.n && col < sg.n && d <= eps;
                mirror |= (nb ? 1u : 0u) << rloc;
                unsigned long long m = __ballot(nb);
                if ((lane & 31) == 0) {
                    unsigned word = (unsigned)(lane ? (m >> 32) : (m & 0xffffffffull));
                    if (row < sg.n && word) {
                        adj[sg.bit_base + (size_t)row * sg.nw + (cbase >> 5)] = word;
                        atomicAdd(&ncount[sg.row_base + row], (unsigned)__popc(word));
                    }
                }
            }
            if (!diag_tile) {
                // mirrored 32x32 block: row `col`, word index rbase/32; the two lane halves hold disjoint row bits
                unsigned other = __shfl_xor(mirror, 32);
                unsigned word = mirror | other;
                if (lane < 32 && col < sg.n && word) {
                    adj[sg.bit_base + (size_t)col * sg.nw + (rbase >> 5)] = word;
                    atomicAdd(&ncount[sg.row_base + col], (unsigned)__popc(word));
                }
            }
        }
}

// ---- label propagation over the adjacency bits (cores only): label = smallest core index reachable
__global__ void k_pool_init(const unsigned* __restrict__ ncount, long long N, int minpts, int* __restrict__ label,
                            const PoolSeg* __restrict__ segs, const int* __restrict__ seg_of_row, int* __restrict__ seg_first) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int k = seg_of_row[i];
    const PoolSeg sg = segs[k];
    const bool is_core = ncount[i] >= (unsigned)minpts;
    label[i] = is_core ? (int)(i - sg.row_base) : -1;
    // lowest core row of the instance: a label that has reached it cannot drop further
    if (is_core && (int)(i - sg.row_base) < seg_first[k]) atomicMin(&seg_first[k], (int)(i - sg.row_base));
}
__global__ void k_pool_prop(const unsigned* __restrict__ adj, const PoolSeg* __restrict__ segs, const int* __restrict__ seg_of_row,
                            const unsigned* __restrict__ ncount, int minpts, long long N, int* __restrict__ label,
                            int* __restrict__ changed, const int* __restrict__ seg_first, int first_round,
                            const unsigned* __restrict__ list /* rows still on their way (k_pool_list), or nullptr: every row */) {
    const int lane = threadIdx.x & 63;
    long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= N) return;                                       // (N: rows, or entries of the list)
    if (list) i = list[i];
    if (ncount[i] < (unsigned)minpts) return;                 // wave-uniform
    const int k = seg_of_row[i];
    int best = label[i];
    const int lowest = seg_first[k];
    if (best == lowest) return;                               // already at the instance's lowest core row (wave-uniform)
    const PoolSeg sg = segs[k];
    const unsigned* row = adj + sg.bit_base + (size_t)(i - sg.row_base) * sg.nw;
    if (first_round) {
        // Every core row still carries its own index (k_pool_init), and a lane meets its columns in rising order: the first core
        // neighbour it finds is its smallest label, and columns at or beyond the row's own label cannot lower it.
        for (int w = lane; w < sg.nw; w += 64) {
            unsigned bits = row[w];
            bool found = false;
            while (bits && !found) {
                const int b = __ffs(bits) - 1;
                bits &= bits - 1;
                const int jr = w * 32 + b;
                if (jr >= best) {
                    found = true;
                } else if (ncount[sg.row_base + jr] >= (unsigned)minpts) {
                    best = jr;
                    found = true;
                }
            }
            if (found) break;
        }
    } else {
        // later rounds: a row is done as soon as ONE neighbour carries the instance's lowest core row -- nothing is lower
        for (int w0 = 0; w0 < sg.nw; w0 += 64) {
            const int w = w0 + lane;
            unsigned bits = w < sg.nw ? row[w] : 0u;
            while (bits && best != lowest) {
                int b = __ffs(bits) - 1;
                bits &= bits - 1;
                long long j = sg.row_base + (long long)w * 32 + b;
                if (ncount[j] >= (unsigned)minpts) {
                    // plain (L1-cacheable) load: a stale label is an older, larger one -- it only delays the drop to
                    // a later launch, and the host loop runs until a whole round changes nothing
                    const int lj = label[j];
                    best = lj < best ? lj : best;
                }
            }
            if (__any(best == lowest)) break;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        int t = __shfl_xor(best, o);
        best = t < best ? t : best;
    }
    if (lane == 0 && best < label[i]) {
        atomicMin(&label[i], best);
        *changed = 1;
    }
}
// pointer jumping between propagation rounds: a core row's label is the (relative) index of a core row of the same
// component that is not larger than its own, so label[label[i]] is one too -- following the chain to its end makes
// the propagation converge in O(log diameter) rounds instead of O(diameter).  (Races only read older, larger labels.)
__global__ void k_pool_jump(const PoolSeg* __restrict__ segs, const int* __restrict__ seg_of_row, const unsigned* __restrict__ ncount,
                            int minpts, long long N, int* __restrict__ label, const unsigned* __restrict__ list) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    if (list) i = list[i];
    if (ncount[i] < (unsigned)minpts) return;
    const long long base = segs[seg_of_row[i]].row_base;
    int l = label[i];
    for (int hop = 0; hop < 64; ++hop) {
        const int p = label[base + l];
        if (p >= l) break;
        l = p;
    }
    if (l < label[i]) atomicMin(&label[i], l);
}
// The core rows that have not reached their instance's lowest core row yet.  A row that has is finished for good (nothing lower
// exists in its instance), and after the first two rounds that is most of them: the later rounds run over this list instead of
// starting a wave per row of the scene only to find it done (950 000 waves a round at configs[1], 1.1 ms).
__global__ void k_pool_list(const PoolSeg* __restrict__ segs, const int* __restrict__ seg_of_row, const unsigned* __restrict__ ncount,
                            int minpts, long long N, const int* __restrict__ label, const int* __restrict__ seg_first,
                            unsigned* __restrict__ list, unsigned* __restrict__ n_list) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool open = i < N && ncount[i] >= (unsigned)minpts && label[i] != seg_first[seg_of_row[i]];
    const unsigned long long m = __ballot(open);
    if (!m) return;
    const int lane = threadIdx.x & 63, leader = __ffsll(m) - 1;
    unsigned base = 0;
    if (lane == leader) base = atomicAdd(n_list, (unsigned)__popcll(m));
    base = __shfl(base, leader);
    if (open) list[base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = (unsigned)i;
}
// border points: smallest cluster label among adjacent cores; then sizes / first index per cluster
__global__ void k_pool_border(const unsigned* __restrict__ adj, const PoolSeg* __restrict__ segs, const int* __restrict__ seg_of_row,
                              const unsigned* __restrict__ ncount, int minpts, long long N, const int* __restrict__ label,
                              int* __restrict__ final_label, unsigned* __restrict__ csize, unsigned* __restrict__ cfirst) {
    const int lane = threadIdx.x & 63;
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= N) return;
    const PoolSeg sg = segs[seg_of_row[i]];
    int best;
    if (ncount[i] >= (unsigned)minpts) {
        best = label[i];
    } else {
        best = 0x7fffffff;
        const unsigned* row = adj + sg.bit_base + (size_t)(i - sg.row_base) * sg.nw;
        for (int w = lane; w < sg.nw; w += 64) {
            unsigned bits = row[w];
            while (bits) {
                int b = __ffs(bits) - 1;
                bits &= bits - 1;
                long long j = sg.row_base + (long long)w * 32 + b;
                if (ncount[j] >= (unsigned)minpts) best = label[j] < best ? label[j] : best;
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            int t = __shfl_xor(best, o);
            best = t < best ? t : best;
        }
        if (best == 0x7fffffff) best = -1;
    }
    if (lane == 0) {
        final_label[i] = best;
        if (best >= 0) {
            atomicAdd(&csize[sg.row_base + best], 1u);
            atomicMin(&cfirst[sg.row_base + best], (unsigned)(i - sg.row_base));
        }
    }
}
__global__ void k_pool_pick(const PoolSeg* __restrict__ segs, const int* __restrict__ seg_of_row, long long N,
                            const unsigned* __restrict__ csize, const unsigned* __restrict__ cfirst,
                            unsigned long long* __restrict__ best) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N || csize[i] == 0u) return;
    unsigned long long key = ((unsigned long long)csize[i] << 32) | (unsigned long long)(0xffffffffu - cfirst[i]);
    atomicMax(&best[seg_of_row[i]], key);
}
// mean over the selected rows in index order (np.mean axis 0 = sequential float32 adds, then / n)
#define POOL_MEAN_AHEAD 16
__global__ void k_pool_mean(const float* __restrict__ X, int D, const PoolSeg* __restrict__ segs, int K,
                            const int* __restrict__ final_label, const unsigned* __restrict__ csize,
                            const unsigned* __restrict__ cfirst, const unsigned long long* __restrict__ best,
                            float* __restrict__ out) {
    const int k = blockIdx.y;
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const PoolSeg sg = segs[k];
    float* o = out + (size_t)k * D;
    if (sg.n == 0) {                                          // graph.py:479-483 -> zeros(1, D)
        o[d] = 0.f;
        return;
    }
    int want = -2;                                            // -2: take all rows (no cluster)
    unsigned long long key = best[k];
    if (key) {
        unsigned first = 0xffffffffu - (unsigned)(key & 0xffffffffull);
        want = final_label[sg.row_base + first];
    }
    // The additions are one serial chain per feature, but the rows that feed it are not: POOL_MEAN_AHEAD rows (and their labels) are
    // fetched side by side, the next group while this one is being added -- a row per round trip (round 5) made the largest
    // instance's ~5 000 rows a 3.6 ms chain on the critical path behind the Gram.
    float acc = 0.f;
    unsigned cnt = 0;
    const float* const xb = X + (size_t)sg.row_base * D + d;
    const int* const lb = final_label + sg.row_base;
    float cur[POOL_MEAN_AHEAD], nxt[POOL_MEAN_AHEAD];
    int curl[POOL_MEAN_AHEAD], nxtl[POOL_MEAN_AHEAD];
#pragma unroll
    for (int j = 0; j < POOL_MEAN_AHEAD; ++j) {
        const int rr = min(j, sg.n - 1);
        cur[j] = xb[(size_t)rr * D];
        curl[j] = lb[rr];
    }
    for (int r = 0; r < sg.n; r += POOL_MEAN_AHEAD) {
        if (r + POOL_MEAN_AHEAD < sg.n) {
#pragma unroll
            for (int j = 0; j < POOL_MEAN_AHEAD; ++j) {
                const int rr = min(r + POOL_MEAN_AHEAD + j, sg.n - 1);
                nxt[j] = xb[(size_t)rr * D];
                nxtl[j] = lb[rr];
            }
        }
#pragma unroll
        for (int j = 0; j < POOL_MEAN_AHEAD; ++j) {
            if (r + j < sg.n && (want == -2 || curl[j] == want)) {
                acc = __fadd_rn(acc, cur[j]);
                ++cnt;
            }
        }
#pragma unroll
        for (int j = 0; j < POOL_MEAN_AHEAD; ++j) {
            cur[j] = nxt[j];
            curl[j] = nxtl[j];
        }
    }
    o[d] = cnt > 1 ? __fdiv_rn(acc, (float)cnt) : acc;
}

__global__ void k_pool_gather_u32(const unsigned* __restrict__ src, const long long* __restrict__ idx, int n,
                                  unsigned* __restrict__ dst) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) dst[t] = src[idx[t]];
}
__global__ void k_seg_rows(const PoolSeg* __restrict__ segs, int* __restrict__ seg_of_row) {
    const PoolSeg sg = segs[blockIdx.y];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < sg.n; i += gridDim.x * blockDim.x) seg_of_row[sg.row_base + i] = blockIdx.y;
}

void hmsg_pool(hmsg_ctx* h) {
    const hmsg_config& c = h->cfg;
    hipStream_t s = h->stream;
    HMSG_REQUIRE(h->merged, HMSG_ERR_INVALID, "hmsg_pool_instances: run hmsg_merge_instances first");
    const int K = (int)h->inst.off.size() - 1;
    const int D = c.feat_dim;
    h->inst_feats.alloc((size_t)std::max(K, 1) * D);
    if (K == 0) {
        h->pooled = true;
        return;
    }
    CloudOps ops;
    ops.s = s;
    DbgLaps laps("pool", s);
    // (a) voxel_down_sample(voxel_size) of every instance (graph.py:456)
    std::vector<SegDesc> segs(K);
    for (int k = 0; k < K; ++k) {
        segs[k].pt_base = h->inst.off[k];
        segs[k].n = (int)(h->inst.off[k + 1] - h->inst.off[k]);
    }
    ops.bounds(h->inst.pts.p, segs);
    DevBuf<double> ds;
    ds.alloc((size_t)std::max<long long>(h->inst.total, 1) * 3);
    std::vector<int> dn;
    const long long P = ops.voxel_down_sample(h->inst.pts.p, segs, c.voxel_size, ds.p, dn);
    laps.lap("voxel_down_sample");
    // (b) nearest map voxel, dist <= 0.8
    DevBuf<int> idx;
    DevBuf<unsigned> valid, pos;
    idx.alloc((size_t)std::max<long long>(P, 1));
    valid.alloc((size_t)std::max<long long>(P, 1));
    pos.alloc((size_t)std::max<long long>(P, 1));
    if (P) {
        TieBuf ties;
        for (;;) {
            ties.prepare(s);
            hipLaunchKernelGGL(k_pool_nn, dim3(cdiv((size_t)P, 256)), dim3(256), 0, s, (const double*)ds.p, P, hmsg_nn_index(h),
                               c.pool_max_dist, idx.p, valid.p, ties.list());
            HMSG_CHECK_LAUNCH();
            if (hmsg_resolve_ties(h, ties, idx.p)) break;     // bit-equal ties answered like cKDTree (hmsg_ckdtree.h)
        }
    }
    laps.lap("nearest map voxel");
    unsigned long long R = 0;   // total valid rows
    if (P) hmsg_scan_u32(valid.p, pos.p, (size_t)P, s, ops.scan_tmp, &R);
    // rows per instance = pos at the instance boundaries
    std::vector<long long> pstart(K + 1, 0);
    for (int k = 0; k < K; ++k) pstart[k + 1] = pstart[k] + dn[k];
    std::vector<unsigned> hpos(K + 1, (unsigned)R);
    {
        DevBuf<long long> d_i;
        DevBuf<unsigned> d_o;
        std::vector<long long> q;
        std::vector<int> which;
        for (int k = 0; k < K; ++k)
            if (pstart[k] < P) {
                q.push_back(pstart[k]);
                which.push_back(k);
            }
        if (!q.empty()) {
            std::vector<unsigned> tmp(q.size());
            d_i.alloc(q.size());
            d_o.alloc(q.size());
            HIP_TRY(hipMemcpyAsync(d_i.p, q.data(), q.size() * 8, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_pool_gather_u32, dim3(cdiv(q.size(), 256)), dim3(256), 0, s, (const unsigned*)pos.p,
                               (const long long*)d_i.p, (int)q.size(), d_o.p);
            HIP_TRY(hipMemcpyAsync(tmp.data(), d_o.p, q.size() * 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            for (size_t t = 0; t < q.size(); ++t) hpos[which[t]] = tmp[t];
        }
        for (int k = K - 1; k >= 0; --k)
            if (pstart[k] >= P) hpos[k] = hpos[k + 1];
    }
    std::vector<PoolSeg> ps(K);
    long long bitw = 0, tiles = 0;
    int maxn = 0;
    for (int k = 0; k < K; ++k) {
        PoolSeg& g = ps[k];
        g.row_base = hpos[k];
        g.n = (int)(hpos[k + 1] - hpos[k]);
        g.nw = (g.n + 31) / 32;
        g.nt = (g.n + GRAM_T - 1) / GRAM_T;
        g.bit_base = bitw;
        g.tile_base = tiles;
        g.pad = 0;
        bitw += (long long)g.n * g.nw;
        tiles += (long long)g.nt * (g.nt + 1) / 2;      // upper triangle of GRAM_T x GRAM_T tiles
        maxn = std::max(maxn, g.n);
    }
    DevBuf<PoolSeg> d_ps;
    d_ps.alloc(K);
    HIP_TRY(hipMemcpyAsync(d_ps.p, ps.data(), (size_t)K * sizeof(PoolSeg), hipMemcpyHostToDevice, s));
    DevBuf<float> X, Xn;
    DevBuf<unsigned> adj, ncount, csize, cfirst;
    DevBuf<int> label, flabel, seg_of_row, d_changed, seg_first;
    DevBuf<unsigned long long> best;
    const size_t Rn = (size_t)std::max<unsigned long long>(R, 1);
    X.alloc(Rn * D);
    Xn.alloc(Rn * D);
    adj.alloc((size_t)std::max<long long>(bitw, 1));
    ncount.alloc(Rn);
    csize.alloc(Rn);
    cfirst.alloc(Rn);
    label.alloc(Rn);
    flabel.alloc(Rn);
    seg_of_row.alloc(Rn);
    d_changed.alloc(1);
    seg_first.alloc((size_t)std::max(K, 1));
    HIP_TRY(hipMemsetAsync(seg_first.p, 0x7f, (size_t)std::max(K, 1) * 4, s));
    best.alloc(K);
    adj.zero(s);
    ncount.zero(s);
    csize.zero(s);
    HIP_TRY(hipMemsetAsync(cfirst.p, 0xff, Rn * 4, s));
    best.zero(s);
    laps.lap("row offsets + buffers");
    if (R) {
        hipLaunchKernelGGL(k_pool_gather, dim3(cdiv((size_t)P * 64, 256)), dim3(256), 0, s, (const int*)idx.p,
                           (const unsigned*)valid.p, (const unsigned*)pos.p, P, (const float*)h->feats.p, D, X.p, Xn.p);
        hipLaunchKernelGGL(k_seg_rows, dim3(std::max(1u, std::min(cdiv(maxn, 256), 256u)), K), dim3(256), 0, s,
                           (const PoolSeg*)d_ps.p, seg_of_row.p);
        laps.lap("gather + normalise");
        {
            double flop = 0;
            for (auto& g : ps) flop += (double)g.n * ((double)g.n + 1.0) * D;   // unique pairs x 2 FLOP x D
            ProfScope psc(h->prof, s, "k_pool_gram", flop);
            hipLaunchKernelGGL(k_pool_gram, dim3((unsigned)tiles), dim3(256), 0, s, (const float*)Xn.p, D, (const PoolSeg*)d_ps.p, K,
                               (float)c.feat_dbscan_eps, adj.p, ncount.p);
        }
        laps.lap("k_pool_gram");
        hipLaunchKernelGGL(k_pool_init, dim3(cdiv((size_t)R, 256)), dim3(256), 0, s, (const unsigned*)ncount.p, (long long)R,
                           c.feat_dbscan_min, label.p, (const PoolSeg*)d_ps.p, (const int*)seg_of_row.p, seg_first.p);
        HMSG_CHECK_LAUNCH();
        // (two rounds before the first look at the flag -- the first one is cheap and always changes something --, then one round
        //  per look: a round over rows that no longer change costs 1.2 ms, a look 20 us.  Round 6: behind those two rounds the rows
        //  that are still on their way are listed, and the later rounds run over the list.)
        DevBuf<unsigned> open_list;
        open_list.alloc(Rn + 1);                            // [0] the count, the rows behind it
        const unsigned* d_list = nullptr;
        unsigned n_open = 0;
        static const bool list_wanted = getenv("HMSG_DEBUG_POOL_ALL_ROWS") == nullptr;   // HMSG_DEBUG_POOL_ALL_ROWS=1: every round over every row (until round 5)
        for (int it = 0; it < 100000; ++it) {
            HIP_TRY(hipMemsetAsync(d_changed.p, 0, 4, s));
            const size_t rows = d_list ? (size_t)n_open : (size_t)R;
            for (int rep = 0; rep < (it == 0 ? 2 : 1) && rows; ++rep) {
                hipLaunchKernelGGL(k_pool_prop, dim3(cdiv(rows * 64, 256)), dim3(256), 0, s, (const unsigned*)adj.p,
                                   (const PoolSeg*)d_ps.p, (const int*)seg_of_row.p, (const unsigned*)ncount.p,
                                   c.feat_dbscan_min, (long long)rows, label.p, d_changed.p, (const int*)seg_first.p, (it == 0 && rep == 0) ? 1 : 0, d_list);
                hipLaunchKernelGGL(k_pool_jump, dim3(cdiv(rows, 256)), dim3(256), 0, s, (const PoolSeg*)d_ps.p,
                                   (const int*)seg_of_row.p, (const unsigned*)ncount.p, c.feat_dbscan_min, (long long)rows, label.p, d_list);
            }
            HMSG_CHECK_LAUNCH();
            if (it == 0 && list_wanted) {
                HIP_TRY(hipMemsetAsync(open_list.p, 0, 4, s));
                hipLaunchKernelGGL(k_pool_list, dim3(cdiv((size_t)R, 256)), dim3(256), 0, s, (const PoolSeg*)d_ps.p, (const int*)seg_of_row.p,
                                   (const unsigned*)ncount.p, c.feat_dbscan_min, (long long)R, (const int*)label.p, (const int*)seg_first.p,
                                   open_list.p + 1, open_list.p);
                HMSG_CHECK_LAUNCH();
                HIP_TRY(hipMemcpyAsync(&n_open, open_list.p, 4, hipMemcpyDeviceToHost, s));
            }
            int ch = 0;
            HIP_TRY(hipMemcpyAsync(&ch, d_changed.p, 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            if (it == 0 && list_wanted) d_list = open_list.p + 1;
            if (!ch) break;
        }
        laps.lap("label propagation");
        hipLaunchKernelGGL(k_pool_border, dim3(cdiv((size_t)R * 64, 256)), dim3(256), 0, s, (const unsigned*)adj.p,
                           (const PoolSeg*)d_ps.p, (const int*)seg_of_row.p, (const unsigned*)ncount.p, c.feat_dbscan_min,
                           (long long)R, (const int*)label.p, flabel.p, csize.p, cfirst.p);
        hipLaunchKernelGGL(k_pool_pick, dim3(cdiv((size_t)R, 256)), dim3(256), 0, s, (const PoolSeg*)d_ps.p,
                           (const int*)seg_of_row.p, (long long)R, (const unsigned*)csize.p, (const unsigned*)cfirst.p, best.p);
        HMSG_CHECK_LAUNCH();
    }
    laps.lap("border + pick");
    hmsg_dump("pool_ds", ds.p, (size_t)P * 24, s);
    hmsg_dump("pool_idx", idx.p, (size_t)P * 4, s);
    hmsg_dump("pool_valid", valid.p, (size_t)P * 4, s);
    hmsg_dump("pool_ncount", ncount.p, (size_t)R * 4, s);
    hmsg_dump("pool_flabel", flabel.p, (size_t)R * 4, s);
    hmsg_dump("pool_label", label.p, (size_t)R * 4, s);
    hmsg_dump("pool_segs", d_ps.p, (size_t)K * sizeof(PoolSeg), s);
    hipLaunchKernelGGL(k_pool_mean, dim3(cdiv(D, 64), K), dim3(64), 0, s, (const float*)X.p, D, (const PoolSeg*)d_ps.p, K,
                       (const int*)flabel.p, (const unsigned*)csize.p, (const unsigned*)cfirst.p,
                       (const unsigned long long*)best.p, h->inst_feats.p);
    HMSG_CHECK_LAUNCH();
    HIP_TRY(hipStreamSynchronize(s));
    laps.lap("k_pool_mean");
    h->pooled = true;
}
